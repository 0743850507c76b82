// General-structure batch QP solver for MI355X (gfx950): SURVEY 8 rows a21 / a22 / f-4.
//
//   min 1/2 x'Px + q'x   s.t.  l <= Ax <= u      P diagonal (on a subset of columns), A sparse (CSC)
//
// One GPU lane per robot, exactly the per-step call sequence of the reference's embedded OSQP 0.6.0
// (template/uprightmpc2/: scaling.c:44-156, auxil.c:103-145, kkt.c:184-222, qdldl.c:86-293,
// osqp.c:354-370, auxil.c:164-228, 243-362, 517-565, 684-789), in canonical-restart form (every call
// starts from the raw data; persistent state = x, y, z and the previous E used for row classification).
//
// Where umpc_step.h bakes the N = 3 uprightmpc2 structure into straight-line code + assembly, this kernel is
// TABLE-DRIVEN: the sparsity (A in CSC and CSR order, the permuted KKT's up-looking LDL' schedule, L in
// CSC and CSR order) is an int32 blob built on the host by robobee3d_amd/qpstruct.py, read with wave-uniform
// scalar loads, so one kernel serves any MPC structure (planar p5f N = 10: n = 87, m = 164; v1 template QP;
// uprightmpc2 at any horizon). Every per-robot vector and matrix value lives in a SoA workspace W[row][B]
// (row index wave-uniform -> SGPR base + 4*lane: every access is a coalesced 256 B wave transaction).
// The path is HBM/L2-stream bound by construction: per ADMM iteration a robot reads L twice and the solve
// vector ~2 nnz(L) times; DESIGN.md 10 has the byte count.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <limits>
#include <string>

#include "../../include/umpc_mi355x.h"
#include "umpc_bqp_common.h"
#include "umpc_bqp_registry.h"
#include "umpc_err.h"

namespace {

using namespace umpcqp;

template <typename T>
__global__ void __launch_bounds__(64) bqp_solve_kernel(const QPArgs<T> a) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= a.B) return;
  const int32_t *__restrict__ tab = a.tab;
  const size_t B = (size_t)a.B;
  const int n = tab[H_N], m = tab[H_M], nk = tab[H_NK], nnzP = tab[H_NNZP], nnzA = tab[H_NNZA];
  const int32_t *pinv = tab + tab[H_TAB0 + T_PINV], *pidx = tab + tab[H_TAB0 + T_PIDX];
  const int32_t *A_p = tab + tab[H_TAB0 + T_AP], *A_i = tab + tab[H_TAB0 + T_AI];
  const int32_t *Ar_p = tab + tab[H_TAB0 + T_ARP], *Ar_j = tab + tab[H_TAB0 + T_ARJ], *Ar_k = tab + tab[H_TAB0 + T_ARK];
  const int32_t *fi_p = tab + tab[H_TAB0 + T_FIP], *fi_b = tab + tab[H_TAB0 + T_FIB], *fi_s = tab + tab[H_TAB0 + T_FIS];
  const int32_t *fe_p = tab + tab[H_TAB0 + T_FEP], *fe_c = tab + tab[H_TAB0 + T_FEC], *fe_n = tab + tab[H_TAB0 + T_FEN];
  const int32_t *L_p = tab + tab[H_TAB0 + T_LP], *L_i = tab + tab[H_TAB0 + T_LI];
  const int32_t *Lr_p = tab + tab[H_TAB0 + T_LRP], *Lr_j = tab + tab[H_TAB0 + T_LRJ], *Lr_k = tab + tab[H_TAB0 + T_LRK];
  const int32_t *rows = tab + H_TAB0 + T_COUNT;
  T *__restrict__ W = a.W;
#define WR(base, i) W[(size_t)(rows[base] + (i)) * B + b]
#define WROW(r) W[(size_t)(r) * B + b]
#define IN(arr, i) (arr)[(size_t)(i) * B + b]
  const T sigma = a.sigma, alpha = a.alpha, oma = T(1.0) - a.alpha;

  // ---- raw data -> working copies; row classification with the previous E (auxil.c:103-145) ----
  for (int k = 0; k < nnzP; ++k) WR(R_PS, k) = IN(a.Pv, k);
  for (int k = 0; k < nnzA; ++k) WR(R_AS, k) = IN(a.Av, k);
  for (int j = 0; j < n; ++j) { WR(R_QS, j) = IN(a.q, j); WR(R_D, j) = T(1.0); }
  const T rho_eq = T(QP_RHO_EQ_OVER_RHO_INEQ * (double)a.rho);
  for (int i = 0; i < m; ++i) {
    const T e = IN(a.Eprev, i);
    T r, ri;
    qp_classify(IN(a.l, i) * e, IN(a.u, i) * e, a.rho, rho_eq, r, ri);
    WR(R_RHO, i) = r; WR(R_RINV, i) = ri; WR(R_E, i) = T(1.0);
  }

  // ---- scale_data: Ruiz equilibration, scaling.c:44-156 ----
  T c = T(1.0);
  for (int pass = 0; pass < a.scaling; ++pass) {
    // column norms of [P; A] (compute_inf_norm_cols_KKT), clamp, 1/sqrt
    for (int j = 0; j < n; ++j) {
      const int kp = pidx[j];
      T dP = T(0.0);
      if (kp >= 0) dP = qmax(qabs(WR(R_PS, kp)), dP);
      T dA = T(0.0);
      for (int p = A_p[j]; p < A_p[j + 1]; ++p) dA = qmax(qabs(WR(R_AS, p)), dA);
      T d = limit_scaling(qmax(dP, dA));
      WR(R_DT, j) = T(1.0) / qsqrt(d);
    }
    for (int i = 0; i < m; ++i) {
      T e = T(0.0);
      for (int p = Ar_p[i]; p < Ar_p[i + 1]; ++p) e = qmax(qabs(WR(R_AS, Ar_k[p])), e);
      WR(R_ET, i) = T(1.0) / qsqrt(limit_scaling(e));
    }
    // P <- D P D, A <- E A D, q <- D q, accumulate D, E
    T qn = T(0.0), csum = T(0.0);
    for (int j = 0; j < n; ++j) {
      const T d = WR(R_DT, j);
      const int kp = pidx[j];
      if (kp >= 0) {
        T pv = WR(R_PS, kp);
        pv *= d; pv *= d;
        WR(R_PS, kp) = pv;
        csum += qabs(pv);
      } else {
        csum += T(0.0);
      }
      for (int p = A_p[j]; p < A_p[j + 1]; ++p) {
        T v = WR(R_AS, p);
        v *= WR(R_ET, A_i[p]);
        v *= d;
        WR(R_AS, p) = v;
      }
      const T qv = WR(R_QS, j) * d;
      WR(R_QS, j) = qv;
      qn = qmax(qabs(qv), qn);   // vec_norm_inf: if (a > mx) mx = a
      WR(R_D, j) = d * WR(R_D, j);
    }
    for (int i = 0; i < m; ++i) WR(R_E, i) = WR(R_ET, i) * WR(R_E, i);
    // cost normalisation
    T ct = csum / T(n);
    qn = limit_scaling(qn);
    ct = qmax(ct, qn);
    ct = limit_scaling(ct);
    ct = T(1.0) / ct;
    for (int k = 0; k < nnzP; ++k) WR(R_PS, k) *= ct;
    for (int j = 0; j < n; ++j) WR(R_QS, j) *= ct;
    c *= ct;
  }
  const T cinv = T(1.0) / c;
  for (int i = 0; i < m; ++i) {
    const T e = WR(R_E, i);
    WR(R_LS, i) = IN(a.l, i) * e;
    WR(R_US, i) = IN(a.u, i) * e;
    // constr_type of the row (-1 loose, 0 inequality, 1 equality; auxil.c:103-145: decided on the bounds scaled by the
    // PREVIOUS E, like rho above) parked in the per-pass scaling row, which is dead after the passes: the adaptive-rho
    // update below drives rho_vec from the TYPE, as set_rho_vec / osqp_update_rho do, not from float comparisons of rho
    const T ep = IN(a.Eprev, i), lsp = IN(a.l, i) * ep, usp = IN(a.u, i) * ep;
    WR(R_ET, i) = (((double)lsp < -QP_INFTY * QP_MIN_SCALING) && ((double)usp > QP_INFTY * QP_MIN_SCALING)) ? T(-1.0)
                  : ((double)(usp - lsp) < QP_RHO_TOL)                                                      ? T(1.0)
                                                                                                             : T(0.0);
    IN(a.Eprev, i) = e;
  }

  int fail = 0;
  auto factor = [&]() {
    // ---- KKT diagonal (kkt.c:184-222) and the up-looking LDL' (qdldl.c:86-247) ----
    for (int j = 0; j < n; ++j) {
      const int kp = pidx[j];
      WR(R_KD, pinv[j]) = kp >= 0 ? WR(R_PS, kp) + sigma : sigma;
    }
    for (int i = 0; i < m; ++i) WR(R_KD, pinv[n + i]) = -WR(R_RINV, i);
    for (int k = 0; k < nk; ++k) WR(R_YV, k) = T(0.0);
    for (int k = 0; k < nk; ++k) {
      for (int p = fi_p[k]; p < fi_p[k + 1]; ++p) WR(R_YV, fi_b[p]) = WROW(fi_s[p]);
      T dk = WR(R_KD, k);
      for (int e = fe_p[k]; e < fe_p[k + 1]; ++e) {
        const int cidx = fe_c[e], lnew = fe_n[e];
        const T yv = WR(R_YV, cidx);
        for (int j = L_p[cidx]; j < lnew; ++j) WR(R_YV, L_i[j]) -= WR(R_LX, j) * yv;
        const T lv = yv * WR(R_DI, cidx);
        WR(R_LX, lnew) = lv;
        dk -= yv * lv;
        WR(R_YV, cidx) = T(0.0);
      }
      if (dk == T(0.0)) fail = 1;
      WR(R_DI, k) = T(1.0) / dk;
    }

  };
  factor();

  // ---- ADMM iterations, osqp.c:354-370 ----
  if (a.max_iter == 0) {
    for (int j = 0; j < n; ++j) WR(R_XP, j) = IN(a.x, j);
    for (int i = 0; i < m; ++i) WR(R_DY, i) = T(0.0);
  }
  auto admm_iteration = [&]() {
    // compute_rhs (auxil.c:164-178), permuted on the fly
    for (int j = 0; j < n; ++j) {
      const T xp = IN(a.x, j);
      WR(R_XP, j) = xp;
      WR(R_WV, pinv[j]) = sigma * xp - WR(R_QS, j);
    }
    for (int i = 0; i < m; ++i) {
      const T r = IN(a.z, i) - WR(R_RINV, i) * IN(a.y, i);
      WR(R_T3, i) = r;
      WR(R_WV, pinv[n + i]) = r;
    }
    // QDLDL_solve (qdldl.c:250-293): forward in row order (same per-element subtraction order as the reference's
    // column sweep), diagonal, backward
    for (int r = 0; r < nk; ++r) {
      T acc = WR(R_WV, r);
      for (int p = Lr_p[r]; p < Lr_p[r + 1]; ++p) acc -= WR(R_LX, Lr_k[p]) * WR(R_WV, Lr_j[p]);
      WR(R_WV, r) = acc;
    }
    for (int r = 0; r < nk; ++r) WR(R_WV, r) *= WR(R_DI, r);
    for (int r = nk - 1; r >= 0; --r) {
      T acc = WR(R_WV, r);
      for (int j = L_p[r]; j < L_p[r + 1]; ++j) acc -= WR(R_LX, j) * WR(R_WV, L_i[j]);
      WR(R_WV, r) = acc;
    }
    // update_x, update_z (+ projection), update_y (auxil.c:188-228)
    for (int j = 0; j < n; ++j) {
      const T xt = WR(R_WV, pinv[j]);
      IN(a.x, j) = alpha * xt + oma * WR(R_XP, j);
    }
    for (int i = 0; i < m; ++i) {
      const T ri = WR(R_RINV, i), yi = IN(a.y, i), zp = IN(a.z, i);
      const T zt = WR(R_T3, i) + ri * WR(R_WV, pinv[n + i]);
      T zn = alpha * zt + oma * zp + ri * yi;
      zn = qmin(qmax(zn, WR(R_LS, i)), WR(R_US, i));
      IN(a.z, i) = zn;
      const T dy = WR(R_RHO, i) * (alpha * zt + oma * zp - zn);
      WR(R_DY, i) = dy;
      IN(a.y, i) = yi + dy;
    }
  };

  // ---- update_info (auxil.c:243-307) + the tolerance-independent parts of the infeasibility certificates
  //      (auxil.c:362-512); decide(k) is check_termination (auxil.c:684-789) at k x the tolerances ----
  T pri_res = T(0.0), dua_res = T(0.0), dual_rel = T(0.0), prim_rel = T(0.0);
  T norm_dy = T(0.0), ineq_lhs = T(0.0), nAtdy = T(0.0), norm_dx = T(0.0), qdx = T(0.0), nPdx = T(0.0);
  T s_pri = T(0.0), s_nz = T(0.0), s_nAx = T(0.0), s_dua = T(0.0), s_nq = T(0.0), s_nAty = T(0.0), s_nPx = T(0.0);  // scaled
  auto update_info = [&]() {
    s_pri = s_nz = s_nAx = s_dua = s_nq = s_nAty = s_nPx = T(0.0);
    // T3 <- Ax (scaled), then primal residual and tolerance
    pri_res = T(0.0); T nz = T(0.0), nAx = T(0.0);
    for (int i = 0; i < m; ++i) {
      T acc = T(0.0);
      for (int p = Ar_p[i]; p < Ar_p[i + 1]; ++p) acc += WR(R_AS, Ar_k[p]) * IN(a.x, Ar_j[p]);
      WR(R_T3, i) = acc;
      const T einv = T(1.0) / WR(R_E, i), zi = IN(a.z, i);
      pri_res = qmax(pri_res, qabs(einv * (acc - zi)));
      nz = qmax(nz, qabs(einv * zi));
      nAx = qmax(nAx, qabs(einv * acc));
      s_pri = qmax(s_pri, qabs(acc - zi)); s_nz = qmax(s_nz, qabs(zi)); s_nAx = qmax(s_nAx, qabs(acc));
    }
    // T1 <- Px, T2 <- A'y, dual residual and tolerance
    dua_res = T(0.0); T nq = T(0.0), nAty = T(0.0), nPx = T(0.0);
    for (int j = 0; j < n; ++j) {
      const int kp = pidx[j];
      T px = T(0.0);
      if (kp >= 0) px += WR(R_PS, kp) * IN(a.x, j);
      T aty = T(0.0);
      for (int p = A_p[j]; p < A_p[j + 1]; ++p) aty += WR(R_AS, p) * IN(a.y, A_i[p]);
      WR(R_T1, j) = px; WR(R_T2, j) = aty;
      const T dinv = T(1.0) / WR(R_D, j), qj = WR(R_QS, j);
      dua_res = qmax(dua_res, qabs(dinv * ((qj + px) + aty)));
      nq = qmax(nq, qabs(dinv * qj)); nAty = qmax(nAty, qabs(dinv * aty)); nPx = qmax(nPx, qabs(dinv * px));
      s_dua = qmax(s_dua, qabs((qj + px) + aty)); s_nq = qmax(s_nq, qabs(qj)); s_nAty = qmax(s_nAty, qabs(aty));
      s_nPx = qmax(s_nPx, qabs(px));
    }
    dua_res = cinv * dua_res;
    dual_rel = qmax(qmax(nq, nAty), nPx) * cinv; prim_rel = qmax(nz, nAx);

    // infeasibility certificates: quantities shared by the exact and the 10x-relaxed check
    // is_primal_infeasible (auxil.c:362-424): project delta_y on the polar of the recession cone
    norm_dy = T(0.0); ineq_lhs = T(0.0);
    for (int i = 0; i < m; ++i) {
      const T us = WR(R_US, i), ls = WR(R_LS, i);
      T dy = WR(R_DY, i);
      const bool up = (double)us > QP_INFTY * QP_MIN_SCALING, lo = (double)ls < -QP_INFTY * QP_MIN_SCALING;
      if (up) dy = lo ? T(0.0) : qmin(dy, T(0.0));
      else if (lo) dy = qmax(dy, T(0.0));
      WR(R_DY, i) = dy;
      norm_dy = qmax(norm_dy, qabs(dy * WR(R_E, i)));
      ineq_lhs += us * qmax(dy, T(0.0)) + ls * qmin(dy, T(0.0));
    }
    nAtdy = T(0.0);
    for (int j = 0; j < n; ++j) {
      T acc = T(0.0);
      for (int p = A_p[j]; p < A_p[j + 1]; ++p) acc += WR(R_AS, p) * WR(R_DY, A_i[p]);
      nAtdy = qmax(nAtdy, qabs(acc * (T(1.0) / WR(R_D, j))));
    }
    // is_dual_infeasible (auxil.c:426-512)
    norm_dx = T(0.0); qdx = T(0.0); nPdx = T(0.0);
    for (int j = 0; j < n; ++j) {
      const T dx = IN(a.x, j) - WR(R_XP, j);
      WR(R_T1, j) = dx;
      norm_dx = qmax(norm_dx, qabs(WR(R_D, j) * dx));
      qdx += WR(R_QS, j) * dx;
      const int kp = pidx[j];
      T pdx = T(0.0);
      if (kp >= 0) pdx += WR(R_PS, kp) * dx;
      nPdx = qmax(nPdx, qabs(pdx * (T(1.0) / WR(R_D, j))));
    }
  };
  auto decide = [&](T k) -> int {
    if (((double)pri_res > QP_INFTY) || ((double)dua_res > QP_INFTY)) return -7;  // OSQP_NON_CVX
    const bool approx = k > T(1);
    const T eps_abs = a.eps_abs * k, eps_rel = a.eps_rel * k, eps_pinf = a.eps_pinf * k, eps_dinf = a.eps_dinf * k;
    const bool prim_ok = pri_res < eps_abs + eps_rel * prim_rel;
    const bool dual_ok = dua_res < eps_abs + eps_rel * dual_rel;
    bool pinf = false, dinf = false;
    if (!prim_ok && norm_dy > eps_pinf && ineq_lhs < -eps_pinf * norm_dy) pinf = nAtdy < eps_pinf * norm_dy;
    if (!dual_ok && norm_dx > eps_dinf && qdx < -c * eps_dinf * norm_dx && nPdx < c * eps_dinf * norm_dx) {
      dinf = true;
      const T thr = eps_dinf * norm_dx;
      for (int i = 0; i < m; ++i) {
        T acc = T(0.0);
        for (int p = Ar_p[i]; p < Ar_p[i + 1]; ++p) acc += WR(R_AS, Ar_k[p]) * WR(R_T1, Ar_j[p]);
        acc = acc * (T(1.0) / WR(R_E, i));
        if ((((double)WR(R_US, i) < QP_INFTY * QP_MIN_SCALING) && (acc > thr)) ||
            (((double)WR(R_LS, i) > -QP_INFTY * QP_MIN_SCALING) && (acc < -thr)))
          dinf = false;
      }
    }
    if (prim_ok && dual_ok) return approx ? 2 : 1;
    if (pinf) return approx ? 3 : -3;
    if (dinf) return approx ? 4 : -4;
    return -10;  // OSQP_UNSOLVED
  };

  // Main loop (osqp.c:354-450). check_termination == 0 (the embedded reference, uprightmpc2.c:116-117): exactly
  // max_iter iterations. check_termination = k > 0 (pip-osqp semantics of the reference's Python twin,
  // template_controllers.py:190-191,216-219): every k-th iteration update_info + check_termination(exact); a robot
  // that meets a criterion stops iterating (per-lane mask), the wave leaves the loop when all of its robots have.
  int status = -10, iters = 0, rho_updates = 0;
  bool info_fresh = false;
  T rho_cur = a.rho;
  for (int it = 1; it <= a.max_iter; ++it) {
    if (status == -10) {
      admm_iteration();
      iters = it;
      info_fresh = false;
      if (a.check_termination > 0 && it % a.check_termination == 0) {
        update_info();
        info_fresh = true;
        status = decide(T(1));
      }
      if (status == -10 && a.adaptive_rho_interval > 0 && it % a.adaptive_rho_interval == 0) {
        // adapt_rho / compute_rho_estimate (auxil.c:12-82) on the scaled residual vectors; osqp_update_rho
        // (osqp.c:1268-1330): rho_vec by constraint type, then the numeric refactorisation
        if (!info_fresh) { update_info(); info_fresh = true; }
        const T pr = s_pri / (qmax(s_nz, s_nAx) + T(1e-10));
        const T du = s_dua / (qmax(qmax(s_nq, s_nAty), s_nPx) + T(1e-10));
        T rho_new = rho_cur * qsqrt(pr / (du + T(1e-10)));
        rho_new = qmin(qmax(rho_new, T(QP_RHO_MIN)), T(1e6));
        if (rho_new > rho_cur * T(5.0) || rho_new < rho_cur / T(5.0)) {
          // by the stored constraint type (set_rho_vec, auxil.c:84-101): loose rows keep RHO_MIN, equality rows get
          // 1e3 rho, inequality rows rho -- also after rho has been clamped to RHO_MIN
          rho_cur = rho_new;
          const T req = T(QP_RHO_EQ_OVER_RHO_INEQ) * rho_cur;
          for (int i = 0; i < m; ++i) {
            const T ct = WR(R_ET, i);
            if (ct < T(0.0)) continue;
            const T r = ct > T(0.0) ? req : rho_cur;
            WR(R_RHO, i) = r; WR(R_RINV, i) = T(1.0) / r;
          }
          factor();
          ++rho_updates;
        }
      }
    }
    if (a.check_termination > 0 && __all(status != -10)) break;
  }
  // osqp.c:524-573: info + exact check if the last iteration did not do them, then the approximate check
  if (status == -10) {
    if (!info_fresh) { update_info(); status = decide(T(1)); }
    if (status == -10) status = decide(T(10));
  }
  if (status == -10) status = -2;  // OSQP_MAX_ITER_REACHED
  (void)fail;
  // ---- store_solution (auxil.c:517-565) ----
  const bool bad = status == -3 || status == 3 || status == -4 || status == 4 || status == -7;
  const T qnan = std::numeric_limits<T>::quiet_NaN();
  for (int j = 0; j < n; ++j) {
    if (a.sol_x) IN(a.sol_x, j) = bad ? qnan : IN(a.x, j) * WR(R_D, j);
    if (bad) IN(a.x, j) = T(0.0);
  }
  for (int i = 0; i < m; ++i) {
    if (a.sol_y) IN(a.sol_y, i) = bad ? qnan : (IN(a.y, i) * WR(R_E, i)) * cinv;
    if (bad) { IN(a.y, i) = T(0.0); IN(a.z, i) = T(0.0); }
  }
  if (a.status) a.status[b] = status;
  if (a.info) { IN(a.info, 0) = pri_res; IN(a.info, 1) = dua_res; IN(a.info, 2) = c; IN(a.info, 3) = fail ? T(1) : T(0); IN(a.info, 4) = T(iters); IN(a.info, 5) = T(rho_updates); }
#undef WR
#undef WROW
#undef IN
}

// ---------------------------------------------------------------------------------------------------------
// Wave-per-robot kernel: the same step with ONE WAVEFRONT per robot and the robot's whole working set in LDS.
// Lane-per-robot (above) is the right mapping while the batch fills the chip and the working set fits a lane's
// registers + LDS share (the N = 3 path). For a large structure at a moderate batch (planar p5f: 251 KKT unknowns,
// ~3 000 words per robot, B = 16 384 = one wave per CU) it is a chain of dependent L2 round trips. Here the
// element-wise phases run 64 unknowns per instruction and the triangular solves / the factorisation are
// level-scheduled over the elimination tree (qpstruct.py: level(c) = 1 + max level over the row pattern of c;
// 34 levels for p5f), one lane per unknown of a level, operands at LDS latency, 64x more waves in flight.
// Same operation order per unknown as QDLDL's sweeps in the solves; the factorisation is the right-looking
// dot-product form and the Ruiz cost sums are wave reductions, so results agree with the lane-per-robot kernels to
// rounding (1e-13 relative in fp64), not bit for bit.
// ---------------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T wave_sum(T v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <typename T> __device__ __forceinline__ T wave_max(T v) {
  for (int o = 32; o > 0; o >>= 1) { const T u = __shfl_xor(v, o, 64); v = qmax(u, v); }
  return v;
}
__device__ __forceinline__ int wave_or(int v) {
  for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
  return v;
}

__host__ __device__ inline size_t bqp_wave_lds_words(int n, int m, int nnzP, int nnzA, int nnzL) {
  // As Ps Lx | qs D Dt x xp t1 (6 n) | ls us E Et rho rinv y z dy t3 (10 m) | DI w (2 nk)
  return (size_t)nnzA + nnzP + nnzL + 6 * (size_t)n + 10 * (size_t)m + 2 * (size_t)(n + m);
}

template <typename T>
__global__ void __launch_bounds__(64) bqp_wave_kernel(const QPArgs<T> a) {
  extern __shared__ __align__(16) unsigned char bqp_smem[];
  T *S = reinterpret_cast<T *>(bqp_smem);
  const int lane = threadIdx.x, b = blockIdx.x;
  const int32_t *__restrict__ tab = a.tab;
  const size_t B = (size_t)a.B;
  const int n = tab[H_N], m = tab[H_M], nk = tab[H_NK], nnzP = tab[H_NNZP], nnzA = tab[H_NNZA], nnzL = tab[H_NNZL];
  const int nlev = tab[H_NLEV];
#define TB(t) (tab + tab[H_TAB0 + (t)])
  const int32_t *pinv = TB(T_PINV), *pidx = TB(T_PIDX), *perm = TB(T_PERM);
  const int32_t *A_p = TB(T_AP), *A_i = TB(T_AI), *A_j = TB(T_AJ), *Ar_p = TB(T_ARP), *Ar_j = TB(T_ARJ), *Ar_k = TB(T_ARK);
  const int32_t *L_p = TB(T_LP), *L_i = TB(T_LI), *Lr_p = TB(T_LRP), *Lr_j = TB(T_LRJ), *Lr_k = TB(T_LRK);
  const int32_t *lev_p = TB(T_LEVP), *lev_n = TB(T_LEVN), *elev_p = TB(T_ELEVP), *elev_e = TB(T_ELEVE);
  const int32_t *l_ksrc = TB(T_LKSRC), *l_col = TB(T_LCOL);
  const int32_t *ft_p = TB(T_FTP), *ft_a = TB(T_FTA), *ft_b = TB(T_FTB), *ft_j = TB(T_FTJ);
  const int32_t *fd_p = TB(T_FDP), *fd_a = TB(T_FDA), *fd_j = TB(T_FDJ);
#undef TB
  T *As = S, *Ps = As + nnzA, *qs = Ps + nnzP, *ls = qs + n, *us = ls + m, *D = us + m, *E = D + n, *Dt = E + m, *Et = Dt + n;
  T *rho = Et + m, *rinv = rho + m, *Lx = rinv + m, *DI = Lx + nnzL, *w = DI + nk, *x = w + nk, *y = x + n, *z = y + m;
  T *xp = z + m, *dy = xp + n, *t3 = dy + m, *t1 = t3 + m;
#define IN(arr, i) (arr)[(size_t)(i) * B + b]
#define FOR_LANES(i, cnt) for (int i = lane; i < (cnt); i += 64)
  const T sigma = a.sigma, alpha = a.alpha, oma = T(1.0) - a.alpha;

  FOR_LANES(k, nnzP) Ps[k] = IN(a.Pv, k);
  FOR_LANES(k, nnzA) As[k] = IN(a.Av, k);
  FOR_LANES(j, n) { qs[j] = IN(a.q, j); D[j] = T(1.0); x[j] = IN(a.x, j); }
  const T rho_eq = T(QP_RHO_EQ_OVER_RHO_INEQ * (double)a.rho);
  FOR_LANES(i, m) {
    const T e = IN(a.Eprev, i);
    T r, ri;
    qp_classify(IN(a.l, i) * e, IN(a.u, i) * e, a.rho, rho_eq, r, ri);
    rho[i] = r; rinv[i] = ri; E[i] = T(1.0); y[i] = IN(a.y, i); z[i] = IN(a.z, i);
  }
  __syncthreads();

  // ---- scale_data, scaling.c:44-156 ----
  T c = T(1.0);
  for (int pass = 0; pass < a.scaling; ++pass) {
    FOR_LANES(j, n) {
      const int kp = pidx[j];
      T dP = T(0.0);
      if (kp >= 0) dP = qmax(qabs(Ps[kp]), dP);
      T dA = T(0.0);
      for (int p = A_p[j]; p < A_p[j + 1]; ++p) dA = qmax(qabs(As[p]), dA);
      Dt[j] = T(1.0) / qsqrt(limit_scaling(qmax(dP, dA)));
    }
    FOR_LANES(i, m) {
      T e = T(0.0);
      for (int p = Ar_p[i]; p < Ar_p[i + 1]; ++p) e = qmax(qabs(As[Ar_k[p]]), e);
      Et[i] = T(1.0) / qsqrt(limit_scaling(e));
    }
    __syncthreads();
    FOR_LANES(k, nnzA) { T v = As[k]; v *= Et[A_i[k]]; v *= Dt[A_j[k]]; As[k] = v; }
    T csum = T(0.0), qn = T(0.0);
    FOR_LANES(j, n) {
      const T d = Dt[j];
      const int kp = pidx[j];
      if (kp >= 0) { T pv = Ps[kp]; pv *= d; pv *= d; Ps[kp] = pv; csum += qabs(pv); }
      const T qv = qs[j] * d;
      qs[j] = qv;
      qn = qmax(qabs(qv), qn);
      D[j] = d * D[j];
    }
    FOR_LANES(i, m) E[i] = Et[i] * E[i];
    csum = wave_sum(csum);
    qn = wave_max(qn);
    T ct = csum / T(n);
    qn = limit_scaling(qn);
    ct = qmax(ct, qn);
    ct = limit_scaling(ct);
    ct = T(1.0) / ct;
    __syncthreads();
    FOR_LANES(k, nnzP) Ps[k] *= ct;
    FOR_LANES(j, n) qs[j] *= ct;
    c *= ct;
    __syncthreads();
  }
  const T cinv = T(1.0) / c;
  FOR_LANES(i, m) { const T e = E[i]; ls[i] = IN(a.l, i) * e; us[i] = IN(a.u, i) * e; IN(a.Eprev, i) = e; }

  // ---- LDL', level by level: D[c] = K[c,c] - sum_j L[c,j]^2 D[j], L[i,c] = (K[i,c] - sum_j L[i,j] L[c,j] D[j]) / D[c]
  // (w holds D during the factorisation)
  int fail = 0;
  for (int l = 0; l < nlev; ++l) {
    for (int t = lane; t < lev_p[l + 1] - lev_p[l]; t += 64) {
      const int cc = lev_n[lev_p[l] + t], orig = perm[cc];
      T dk;
      if (orig < n) { const int kp = pidx[orig]; dk = kp >= 0 ? Ps[kp] + sigma : sigma; }
      else dk = -rinv[orig - n];
      for (int q = fd_p[cc]; q < fd_p[cc + 1]; ++q) { const T lv = Lx[fd_a[q]]; dk -= (lv * lv) * w[fd_j[q]]; }
      if (dk == T(0.0)) fail = 1;
      w[cc] = dk;
      DI[cc] = T(1.0) / dk;
    }
    __syncthreads();
    for (int t = lane; t < elev_p[l + 1] - elev_p[l]; t += 64) {
      const int e = elev_e[elev_p[l] + t], ks = l_ksrc[e];
      T v = ks >= 0 ? As[ks] : T(0.0);
      for (int q = ft_p[e]; q < ft_p[e + 1]; ++q) v -= (Lx[ft_a[q]] * Lx[ft_b[q]]) * w[ft_j[q]];
      Lx[e] = v * DI[l_col[e]];
    }
    __syncthreads();
  }

  // ---- ADMM iterations, osqp.c:354-370 ----
  FOR_LANES(j, n) xp[j] = x[j];
  FOR_LANES(i, m) dy[i] = T(0.0);
  __syncthreads();
  for (int it = 0; it < a.max_iter; ++it) {
    FOR_LANES(k, nk) {
      const int orig = perm[k];
      if (orig < n) {
        const T xv = x[orig];
        xp[orig] = xv;
        w[k] = sigma * xv - qs[orig];
      } else {
        const int i = orig - n;
        const T r = z[i] - rinv[i] * y[i];
        t3[i] = r;
        w[k] = r;
      }
    }
    __syncthreads();
    for (int l = 1; l < nlev; ++l) {      // forward: the entries of row r live in lower levels
      for (int t = lane; t < lev_p[l + 1] - lev_p[l]; t += 64) {
        const int r = lev_n[lev_p[l] + t];
        T acc = w[r];
        for (int p = Lr_p[r]; p < Lr_p[r + 1]; ++p) acc -= Lx[Lr_k[p]] * w[Lr_j[p]];
        w[r] = acc;
      }
      __syncthreads();
    }
    for (int l = nlev - 1; l >= 0; --l) {  // diagonal + backward: the entries of column r live in higher levels
      for (int t = lane; t < lev_p[l + 1] - lev_p[l]; t += 64) {
        const int r = lev_n[lev_p[l] + t];
        T acc = w[r] * DI[r];
        for (int j = L_p[r]; j < L_p[r + 1]; ++j) acc -= Lx[j] * w[L_i[j]];
        w[r] = acc;
      }
      __syncthreads();
    }
    FOR_LANES(j, n) x[j] = alpha * w[pinv[j]] + oma * xp[j];
    FOR_LANES(i, m) {
      const T ri = rinv[i], yi = y[i], zp = z[i];
      const T zt = t3[i] + ri * w[pinv[n + i]];
      T zn = alpha * zt + oma * zp + ri * yi;
      zn = qmin(qmax(zn, ls[i]), us[i]);
      z[i] = zn;
      const T d = rho[i] * (alpha * zt + oma * zp - zn);
      dy[i] = d;
      y[i] = yi + d;
    }
    __syncthreads();
  }

  // ---- update_info / check_termination (auxil.c:243-362, 684-789) ----
  T pri_res = T(0.0), nz = T(0.0), nAx = T(0.0);
  FOR_LANES(i, m) {
    T acc = T(0.0);
    for (int p = Ar_p[i]; p < Ar_p[i + 1]; ++p) acc += As[Ar_k[p]] * x[Ar_j[p]];
    const T einv = T(1.0) / E[i], zi = z[i];
    pri_res = qmax(pri_res, qabs(einv * (acc - zi)));
    nz = qmax(nz, qabs(einv * zi));
    nAx = qmax(nAx, qabs(einv * acc));
  }
  pri_res = wave_max(pri_res); nz = wave_max(nz); nAx = wave_max(nAx);
  T dua_res = T(0.0), nq = T(0.0), nAty = T(0.0), nPx = T(0.0);
  FOR_LANES(j, n) {
    const int kp = pidx[j];
    T px = T(0.0);
    if (kp >= 0) px += Ps[kp] * x[j];
    T aty = T(0.0);
    for (int p = A_p[j]; p < A_p[j + 1]; ++p) aty += As[p] * y[A_i[p]];
    const T dinv = T(1.0) / D[j], qj = qs[j];
    dua_res = qmax(dua_res, qabs(dinv * ((qj + px) + aty)));
    nq = qmax(nq, qabs(dinv * qj)); nAty = qmax(nAty, qabs(dinv * aty)); nPx = qmax(nPx, qabs(dinv * px));
  }
  dua_res = cinv * wave_max(dua_res); nq = wave_max(nq); nAty = wave_max(nAty); nPx = wave_max(nPx);
  const T dual_rel = qmax(qmax(nq, nAty), nPx) * cinv, prim_rel = qmax(nz, nAx);
  T norm_dy = T(0.0), ineq_lhs = T(0.0);
  FOR_LANES(i, m) {
    const T usi = us[i], lsi = ls[i];
    T d = dy[i];
    const bool up = (double)usi > QP_INFTY * QP_MIN_SCALING, lo = (double)lsi < -QP_INFTY * QP_MIN_SCALING;
    if (up) d = lo ? T(0.0) : qmin(d, T(0.0));
    else if (lo) d = qmax(d, T(0.0));
    dy[i] = d;
    norm_dy = qmax(norm_dy, qabs(d * E[i]));
    ineq_lhs += usi * qmax(d, T(0.0)) + lsi * qmin(d, T(0.0));
  }
  norm_dy = wave_max(norm_dy); ineq_lhs = wave_sum(ineq_lhs);
  __syncthreads();
  T nAtdy = T(0.0), norm_dx = T(0.0), qdx = T(0.0), nPdx = T(0.0);
  FOR_LANES(j, n) {
    T acc = T(0.0);
    for (int p = A_p[j]; p < A_p[j + 1]; ++p) acc += As[p] * dy[A_i[p]];
    const T dinv = T(1.0) / D[j];
    nAtdy = qmax(nAtdy, qabs(acc * dinv));
    const T dx = x[j] - xp[j];
    t1[j] = dx;
    norm_dx = qmax(norm_dx, qabs(D[j] * dx));
    qdx += qs[j] * dx;
    const int kp = pidx[j];
    T pdx = T(0.0);
    if (kp >= 0) pdx += Ps[kp] * dx;
    nPdx = qmax(nPdx, qabs(pdx * dinv));
  }
  nAtdy = wave_max(nAtdy); norm_dx = wave_max(norm_dx); qdx = wave_sum(qdx); nPdx = wave_max(nPdx);
  __syncthreads();
  int status = -10;
  if (((double)pri_res > QP_INFTY) || ((double)dua_res > QP_INFTY)) status = -7;
  for (int approx = 0; approx < 2 && status == -10; ++approx) {
    const T k = approx ? T(10) : T(1);
    const T eps_abs = a.eps_abs * k, eps_rel = a.eps_rel * k, eps_pinf = a.eps_pinf * k, eps_dinf = a.eps_dinf * k;
    const bool prim_ok = pri_res < eps_abs + eps_rel * prim_rel;
    const bool dual_ok = dua_res < eps_abs + eps_rel * dual_rel;
    bool pinf = false, dinf = false;
    if (!prim_ok && norm_dy > eps_pinf && ineq_lhs < -eps_pinf * norm_dy) pinf = nAtdy < eps_pinf * norm_dy;
    if (!dual_ok && norm_dx > eps_dinf && qdx < -c * eps_dinf * norm_dx && nPdx < c * eps_dinf * norm_dx) {
      const T thr = eps_dinf * norm_dx;
      int viol = 0;
      FOR_LANES(i, m) {
        T acc = T(0.0);
        for (int p = Ar_p[i]; p < Ar_p[i + 1]; ++p) acc += As[Ar_k[p]] * t1[Ar_j[p]];
        acc = acc * (T(1.0) / E[i]);
        if ((((double)us[i] < QP_INFTY * QP_MIN_SCALING) && (acc > thr)) ||
            (((double)ls[i] > -QP_INFTY * QP_MIN_SCALING) && (acc < -thr)))
          viol = 1;
      }
      dinf = wave_or(viol) == 0;
    }
    if (prim_ok && dual_ok) status = approx ? 2 : 1;
    else if (pinf) status = approx ? 3 : -3;
    else if (dinf) status = approx ? 4 : -4;
  }
  if (status == -10) status = -2;
  const bool bad = status == -3 || status == 3 || status == -4 || status == 4 || status == -7;
  const T qnan = std::numeric_limits<T>::quiet_NaN();
  FOR_LANES(j, n) {
    if (a.sol_x) IN(a.sol_x, j) = bad ? qnan : x[j] * D[j];
    IN(a.x, j) = bad ? T(0.0) : x[j];
  }
  FOR_LANES(i, m) {
    if (a.sol_y) IN(a.sol_y, i) = bad ? qnan : (y[i] * E[i]) * cinv;
    IN(a.y, i) = bad ? T(0.0) : y[i];
    IN(a.z, i) = bad ? T(0.0) : z[i];
  }
  fail = wave_or(fail);
  if (lane == 0) {
    if (a.status) a.status[b] = status;
    if (a.info) { IN(a.info, 0) = pri_res; IN(a.info, 1) = dua_res; IN(a.info, 2) = c; IN(a.info, 3) = fail ? T(1) : T(0); IN(a.info, 4) = T(a.max_iter); IN(a.info, 5) = T(0); }
  }
#undef IN
#undef FOR_LANES
}

// Av[k] = cst[k] if src[k] < 0 else par[src[k]][b] * cst[k]: assembles any per-robot value vector (A, P, q, l, u)
// whose entries are constants or scaled copies of a few per-robot parameters.
// One thread per (robot, kGatherRows entries): the grid covers B x nnz, so a small batch still fills the chip (round 4: one
// thread per robot walking all nnz entries kept 64 of the 256 CUs busy for 25 us per p5f tick; src / cst are wave-uniform
// scalar loads either way).
constexpr int kGatherRows = 4;
// DYN_ONLY (umpcQPGatherUpdate): the constant entries (src < 0) were written by an earlier umpcQPGather and are left alone.
template <typename T, bool DYN_ONLY>
__global__ void bqp_gather_kernel(int B, int nnz, const T *__restrict__ cst, const int32_t *__restrict__ src,
                                  const T *__restrict__ par, T *__restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int k0 = blockIdx.y * kGatherRows;
#pragma unroll
  for (int r = 0; r < kGatherRows; ++r) {
    const int k = k0 + r;
    if (k >= nnz) break;
    const int s = src[k];
    if (DYN_ONLY && s < 0) continue;
    const T cv = cst[k];
    out[(size_t)k * B + b] = s < 0 ? cv : par[(size_t)s * B + b] * cv;
  }
}

// getLin and the plant tick of planar/mpc_osqp_p5f.py: p5f_getlin / p5f_plant_tick in umpc_bqp_common.h (shared with the
// fused tick of the p5f10 assembly kernel)
// mode 0: lin[5][B] <- getLin(u[b], y[0][b], y[3][b]) (the reference linearises about the PREVIOUS state, :165-167)
// mode 1: the same, then the reference's plant tick y += (Ad y + Bd u) dt (:176)
template <typename T>
__global__ void p5f_kernel(int B, int mode, T dt, const T *__restrict__ u, T u_all, T *__restrict__ y, T *__restrict__ lin) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const size_t Bz = (size_t)B;
  T yy[7], o[5];
  for (int i = 0; i < 7; ++i) yy[i] = y[i * Bz + b];
  const T ub = u ? u[b] : u_all;   // (umpcP5fStepU: the reference's unom is ONE number for all robots, mpc_osqp_p5f.py:157)
  p5f_getlin(ub, yy[0], yy[3], o);
  if (lin) for (int i = 0; i < 5; ++i) lin[i * Bz + b] = o[i];
  if (mode == 1) {
    p5f_plant_tick(o, ub, dt, yy);
    for (int i = 0; i < 7; ++i) y[i * Bz + b] = yy[i];
  }
}

// getLin and the A update of one tick in ONE launch (umpcP5fLinearise = umpcP5fStep mode 0 + umpcQPGather[Update] with
// par = lin): blockIdx.y == 0 writes lin, the other block rows each kGatherRows entries of A, recomputing getLin from
// the same inputs (the same function of the same numbers: the values are the ones the two launches produce).
template <typename T>
__global__ void p5f_linearise_kernel(int B, const T *__restrict__ u, T u_all, const T *__restrict__ y, T *__restrict__ lin,
                                     int nnz, const T *__restrict__ cst, const int32_t *__restrict__ src, T *__restrict__ out,
                                     int update) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const size_t Bz = (size_t)B;
  const int k0 = ((int)blockIdx.y - 1) * kGatherRows;
  if (blockIdx.y > 0) {           // (wave-uniform: a block row without a state-dependent entry has nothing to compute)
    bool any = false;
    for (int r = 0; r < kGatherRows; ++r) any = any || (k0 + r < nnz && (src[k0 + r] >= 0 || !update));
    if (!any) return;
  }
  T o[5];
  p5f_getlin(u ? u[b] : u_all, y[b], y[3 * Bz + b], o);
  if (blockIdx.y == 0) {
    for (int i = 0; i < 5; ++i) lin[i * Bz + b] = o[i];
    return;
  }
#pragma unroll
  for (int r = 0; r < kGatherRows; ++r) {
    const int k = k0 + r;
    if (k >= nnz) break;
    const int s = src[k];
    if (update && s < 0) continue;
    const T cv = cst[k];
    out[(size_t)k * Bz + b] = s < 0 ? cv : o[s] * cv;
  }
}

// ---------------------------------------------------------------------------------------------------------
// uprightmpc2 at any horizon N: the assembly (template/template_controllers.py:65-143 = uprightmpc2.c:121-207)
// and the extraction (template_controllers.py:232-250 = uprightmpc2.c:253-269) around umpcQPSolve. Row layout
// of x: [y_1..y_N (6 each) | dy_1..dy_N | u_0..u_{N-1} (3 each)]; rows of A: 6N + 6N dynamics, N thrust rows.
// par rows (for umpcQPGather): 0 = dt*T0, 1..3 = dt*s0, 4..9 = dt*Btau (column-major 3x2).
// ---------------------------------------------------------------------------------------------------------
template <typename T>
struct NArgs {
  int B, N;
  T dt, g, Tmax, ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom, Ibi[3];
};

template <typename T>
__global__ void umpcn_assemble_kernel(NArgs<T> a, const T *__restrict__ state, const T *__restrict__ ref, T *T0io,
                                      const T *actualT0, T *Pv, T *q, T *l, T *u, T *par) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const size_t B = (size_t)a.B;
  const int N = a.N, NYc = 6, NUc = 3;
#define ST(i) state[(size_t)(i) * B + b]
#define O(arr, i) arr[(size_t)(i) * B + b]
  T T0 = T0io[b];
  if (actualT0) { const T t = actualT0[b]; if (t >= T(0)) T0 = t; }   // template_controllers.py:253-255
  T0io[b] = T0;
  T R0[9], dq0[6], p0[3];
  for (int i = 0; i < 3; ++i) p0[i] = ST(i);
  for (int i = 0; i < 9; ++i) R0[i] = ST(3 + i);
  for (int i = 0; i < 6; ++i) dq0[i] = ST(12 + i);
  const T dt = a.dt;
  T s0[3], ds0[3], Btau[6], y0[6], dy0[6], y1[6], ydes[6], dydes[6];
  for (int r = 0; r < 3; ++r) {
    s0[r] = R0[6 + r];
    ds0[r] = -(R0[r] * (-dq0[4]) + R0[r + 3] * dq0[3]);
    Btau[r] = -(R0[r + 3] * a.Ibi[0]);
    Btau[3 + r] = -(R0[r] * (-a.Ibi[1]));
  }
  for (int i = 0; i < 3; ++i) {
    y0[i] = p0[i]; y0[3 + i] = s0[i];
    dy0[i] = dq0[i]; dy0[3 + i] = ds0[i];
    ydes[i] = O(ref, i); ydes[3 + i] = O(ref, 6 + i);
    dydes[i] = O(ref, 3 + i); dydes[3 + i] = T(0);
  }
  const T c0[6] = {T(0), T(0), -a.g, T(0), T(0), T(0)};
  for (int i = 0; i < NYc; ++i) y1[i] = y0[i] + dt * dy0[i];
  const int neq = 2 * N * NYc;
  for (int i = 0; i < N * NYc; ++i) { const T v = i < NYc ? -y1[i] : T(0); O(l, i) = v; O(u, i) = v; }
  for (int k = 0; k < N; ++k)
    for (int i = 0; i < NYc; ++i) {
      T v;
      if (k == 0) v = -dy0[i] - dt * (i < 3 ? T0 * y0[i + 3] : T(0)) - dt * c0[i];
      else if (k == 1) v = -dt * (i < 3 ? T0 * y1[i + 3] : T(0)) - dt * c0[i];
      else v = -dt * c0[i];
      O(l, NYc * (N + k) + i) = v; O(u, NYc * (N + k) + i) = v;
    }
  for (int k = 0; k < N; ++k) { O(l, neq + k) = -T0; O(u, neq + k) = a.Tmax - T0; }
  O(par, 0) = dt * T0;
  for (int i = 0; i < 3; ++i) O(par, 1 + i) = dt * s0[i];
  for (int i = 0; i < 6; ++i) O(par, 4 + i) = dt * Btau[i];
  for (int k = 0; k < N; ++k) {
    for (int i = 0; i < NYc; ++i) {
      const T wy = i < 3 ? (k == N - 1 ? a.wpf : a.wpr) : a.ws;
      const T wd = i < 3 ? (k == N - 1 ? a.wvf : a.wvr) : a.wds;
      O(Pv, k * NYc + i) = wy; O(q, k * NYc + i) = -wy * ydes[i];
      O(Pv, N * NYc + k * NYc + i) = wd; O(q, N * NYc + k * NYc + i) = -wd * dydes[i];
    }
    for (int i = 0; i < NUc; ++i) {
      O(Pv, 2 * N * NYc + k * NUc + i) = i == 0 ? a.wthrust : a.wmom;
      O(q, 2 * N * NYc + k * NUc + i) = T(0);
    }
  }
}

template <typename T>
__global__ void umpcn_extract_kernel(int Bn, int N, T dt, const T *__restrict__ state, const T *__restrict__ sol_x,
                                     T *T0io, T *out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= Bn) return;
  const size_t B = (size_t)Bn;
  T R0[9], dq0[6], dy1[6];
  for (int i = 0; i < 9; ++i) R0[i] = ST(3 + i);
  for (int i = 0; i < 6; ++i) dq0[i] = ST(12 + i);
  for (int i = 0; i < 6; ++i) dy1[i] = O(sol_x, 6 * N + i);
  const T T0 = T0io[b] + O(sol_x, 12 * N);
  T0io[b] = T0;
  O(out, 0) = T0; O(out, 1) = O(sol_x, 12 * N + 1); O(out, 2) = O(sol_x, 12 * N + 2);
  const T rx = (R0[0] * dy1[3] + R0[1] * dy1[4]) + R0[2] * dy1[5];
  const T ry = (R0[3] * dy1[3] + R0[4] * dy1[4]) + R0[5] * dy1[5];
  const T dq1[6] = {dy1[0], dy1[1], dy1[2], -ry, rx, T(0)};
  for (int i = 0; i < 6; ++i) O(out, 3 + i) = (dq1[i] - dq0[i]) / dt;
#undef ST
#undef O
}


struct qp_batch {
  int B, dtype, n, m, nk, nnzP, nnzA, nnzL, nrows, fixed, use_tables, wave;   // wave: 1 = wave-per-robot kernel
  int no_asm;   // umpcQPSetKernel(h, 3): the lane specialisation without its assembly loop
  size_t asm_tail;   // elements of the stream buffer behind the workspace rows (0: none)
  size_t wave_lds;
  bool wave_ok;
  umpcQPSettings st;
  int32_t *tab;
  void *W;
};

// assembly specialisations (gen/bqp_*_asm.h, fp32) keep one iteration's read-only words in a [wave][item][lane] block
// (2048 items: the loop's stream, then the residual stream) that umpcQPCreate allocates behind the workspace rows, and hand the
// factor over through the first 1024 rows. UMPC_QP_NO_ASM=1 disables them.
size_t asm_tail_elems(int B) { return (size_t)((B + 63) / 64) * 64 * BQP_ASM_STREAM_ITEMS_PER_WAVE; }   // umpc_bqp_registry.h (codegen_qp.ASM_STREAM_ITEMS)
bool asm_room(const qp_batch *h) {
  static const bool no_asm = getenv("UMPC_QP_NO_ASM") != nullptr;
  return !no_asm && !h->no_asm && h->asm_tail > 0 && h->nrows >= 1024;
}
bool asm_active(const qp_batch *h) {
  const bool tables = h->use_tables || h->st.check_termination > 0 || h->st.adaptive_rho_interval > 0;
  return h->fixed >= 0 && !tables && !h->wave && h->dtype == UMPC_F32 && kFixedKernels[h->fixed].asm_f32 &&
         h->st.max_iter >= 3 && asm_room(h);
}

template <typename T>
int launch_solve(qp_batch *h, const void *Pv, const void *Av, const void *q, const void *l, const void *u, void *x,
                 void *y, void *z, void *Eprev, void *sol_x, void *sol_y, int32_t *status, void *info,
                 hipStream_t s, const QPArgs<T> *tick = nullptr) {
  QPArgs<T> a;
  if (tick) {
    a.tick_y = tick->tick_y; a.tick_lin = tick->tick_lin; a.tick_cst = tick->tick_cst; a.tick_src = tick->tick_src;
    a.tick_nnz = tick->tick_nnz; a.tick_u = tick->tick_u; a.tick_dt = tick->tick_dt;
  }
  a.tab = h->tab; a.B = h->B; a.W = (T *)h->W;
  a.Pv = (const T *)Pv; a.Av = (const T *)Av; a.q = (const T *)q; a.l = (const T *)l; a.u = (const T *)u;
  a.x = (T *)x; a.y = (T *)y; a.z = (T *)z; a.Eprev = (T *)Eprev; a.sol_x = (T *)sol_x; a.sol_y = (T *)sol_y;
  a.status = status; a.info = (T *)info;
  a.sigma = T(h->st.sigma); a.alpha = T(h->st.alpha); a.rho = T(h->st.rho);
  a.eps_abs = T(h->st.eps_abs); a.eps_rel = T(h->st.eps_rel);
  a.eps_pinf = T(h->st.eps_prim_inf); a.eps_dinf = T(h->st.eps_dual_inf);
  a.max_iter = h->st.max_iter; a.scaling = h->st.scaling;
  a.check_termination = h->st.check_termination;
  a.adaptive_rho_interval = h->st.adaptive_rho_interval;
  {
    a.asm_ok = asm_room(h) ? 1 : 0;
    a.S = h->asm_tail ? (T *)h->W + (size_t)h->nrows * (size_t)h->B : nullptr;
    a.oma = T(1.0) - a.alpha;
    a.rinv_eq = T(1. / (double)T(QP_RHO_EQ_OVER_RHO_INEQ * (double)a.rho));
    a.rinv0 = T(1. / (double)a.rho);
    a.rho_eq = T(QP_RHO_EQ_OVER_RHO_INEQ * (double)a.rho);
  }
  // early termination / adaptive rho live in the table kernel
  const bool tables = h->use_tables || h->st.check_termination > 0 || h->st.adaptive_rho_interval > 0;
  if (h->wave && !tables) {
    hipLaunchKernelGGL(bqp_wave_kernel<T>, dim3(h->B), dim3(64), h->wave_lds * sizeof(T), s, a);
  } else if (h->fixed >= 0 && !tables) {
    if constexpr (sizeof(T) == 4) kFixedKernels[h->fixed].f32(a, s);
    else kFixedKernels[h->fixed].f64(a, s);
  } else {
    hipLaunchKernelGGL(bqp_solve_kernel<T>, dim3((h->B + 63) / 64), dim3(64), 0, s, a);
  }
  return 0;
}

int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    umpc_set_error((std::string(what) + ": " + hipGetErrorString(e)).c_str());
    return -1;
  }
  return 0;
}

template <typename T>
NArgs<T> make_nargs(int B, int N, const umpcNParams *p) {
  NArgs<T> a;
  a.B = B; a.N = N;
  a.dt = T(p->dt); a.g = T(p->g); a.Tmax = T(p->TtoWmax * p->g);
  a.ws = T(p->ws); a.wds = T(p->wds); a.wpr = T(p->wpr); a.wpf = T(p->wpf); a.wvr = T(p->wvr); a.wvf = T(p->wvf);
  a.wthrust = T(p->wthrust); a.wmom = T(p->wmom);
  for (int i = 0; i < 3; ++i) a.Ibi[i] = T(1) / T(p->Ib[i]);
  return a;
}


}  // namespace

extern "C" {

void umpcQPDefaultSettings(umpcQPSettings *s) {
  // the generated workspace of the reference (template/uprightmpc2/workspace.c settings block) + umpcInit's
  // max_iter / check_termination = 0 override (uprightmpc2.c:116-117)
  s->rho = 0.1; s->sigma = 1e-6; s->alpha = 1.6;
  s->eps_abs = 1e-4; s->eps_rel = 1e-4; s->eps_prim_inf = 1e-4; s->eps_dual_inf = 1e-4;
  s->max_iter = 50; s->scaling = 10;
  s->check_termination = 0;
  s->adaptive_rho_interval = 0;
}

void *umpcQPCreate(const int32_t *blob, int nwords, int B, int dtype, const umpcQPSettings *st) {
  if (!blob || nwords < HEADER_WORDS || B <= 0 || (dtype != UMPC_F32 && dtype != UMPC_F64)) {
    umpc_set_error("umpcQPCreate: bad argument");
    return nullptr;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    umpc_set_error("umpcQPCreate: no HIP device (this library has no CPU path)");
    return nullptr;
  }
  // every table offset / row offset must lie inside the blob / workspace: the kernel trusts them
  const int n = blob[H_N], m = blob[H_M], nk = blob[H_NK], nrows = blob[H_NROWS];
  if (n <= 0 || m <= 0 || nk != n + m || nrows <= 0) { umpc_set_error("umpcQPCreate: bad header"); return nullptr; }
  for (int t = 0; t < T_COUNT; ++t)
    if (blob[H_TAB0 + t] < HEADER_WORDS || blob[H_TAB0 + t] > nwords) { umpc_set_error("umpcQPCreate: bad table offset"); return nullptr; }
  for (int r = 0; r < R_COUNT; ++r)
    if (blob[H_TAB0 + T_COUNT + r] < 0 || blob[H_TAB0 + T_COUNT + r] >= nrows) { umpc_set_error("umpcQPCreate: bad row offset"); return nullptr; }
  // index tables: every entry of a table that addresses rows / vectors must be in range
  {
    const int nnzA = blob[H_NNZA], nnzL = blob[H_NNZL], nnzP = blob[H_NNZP];
    auto tabp = [&](int t) { return blob + blob[H_TAB0 + t]; };
    auto in_range = [&](int t, int cnt, int lo, int hi) {
      if (blob[H_TAB0 + t] + cnt > nwords) return false;
      for (int k = 0; k < cnt; ++k) if (tabp(t)[k] < lo || tabp(t)[k] >= hi) return false;
      return true;
    };
    const int nfi = tabp(T_FIP)[nk], nfe = tabp(T_FEP)[nk];
    bool ok = in_range(T_PINV, nk, 0, nk) && in_range(T_PIDX, n, -1, nnzP) && in_range(T_AP, n + 1, 0, nnzA + 1) &&
              in_range(T_AI, nnzA, 0, m) && in_range(T_ARP, m + 1, 0, nnzA + 1) && in_range(T_ARJ, nnzA, 0, n) &&
              in_range(T_ARK, nnzA, 0, nnzA) && in_range(T_FIP, nk + 1, 0, nnzA + 1) && nfi >= 0 && nfi <= nnzA &&
              in_range(T_FIB, nfi, 0, nk) && in_range(T_FIS, nfi, 0, nrows) && in_range(T_FEP, nk + 1, 0, nnzL + 1) &&
              nfe == nnzL && in_range(T_FEC, nfe, 0, nk) && in_range(T_FEN, nfe, 0, nnzL) &&
              in_range(T_LP, nk + 1, 0, nnzL + 1) && in_range(T_LI, nnzL, 0, nk) && in_range(T_LRP, nk + 1, 0, nnzL + 1) &&
              in_range(T_LRJ, nnzL, 0, nk) && in_range(T_LRK, nnzL, 0, nnzL);
    // level schedules / right-looking factor terms of the wave-per-robot kernel
    const int nlev = blob[H_NLEV];
    ok = ok && nlev >= 1 && nlev <= nk && in_range(T_PERM, nk, 0, nk) && in_range(T_AJ, nnzA, 0, n) &&
         in_range(T_LEVP, nlev + 1, 0, nk + 1) && in_range(T_LEVN, nk, 0, nk) && in_range(T_ELEVP, nlev + 1, 0, nnzL + 1) &&
         in_range(T_ELEVE, nnzL, 0, nnzL) && in_range(T_LKSRC, nnzL, -1, nnzA) && in_range(T_LCOL, nnzL, 0, nk) &&
         in_range(T_FTP, nnzL + 1, 0, 1 << 30) && in_range(T_FDP, nk + 1, 0, 1 << 30);
    if (ok) {
      ok = tabp(T_LEVP)[nlev] == nk && tabp(T_ELEVP)[nlev] == nnzL;
      for (int k = 0; ok && k < nlev; ++k) ok = tabp(T_LEVP)[k] <= tabp(T_LEVP)[k + 1] && tabp(T_ELEVP)[k] <= tabp(T_ELEVP)[k + 1];
      for (int k = 0; ok && k < nnzL; ++k) ok = tabp(T_FTP)[k] <= tabp(T_FTP)[k + 1];
      for (int k = 0; ok && k < nk; ++k) ok = tabp(T_FDP)[k] <= tabp(T_FDP)[k + 1];
      const int nft = ok ? tabp(T_FTP)[nnzL] : 0, nfd = ok ? tabp(T_FDP)[nk] : 0;
      ok = ok && in_range(T_FTA, nft, 0, nnzL) && in_range(T_FTB, nft, 0, nnzL) && in_range(T_FTJ, nft, 0, nk) &&
           in_range(T_FDA, nfd, 0, nnzL) && in_range(T_FDJ, nfd, 0, nk);
    }
    if (!ok) { umpc_set_error("umpcQPCreate: table entry out of range"); return nullptr; }
  }
  qp_batch *h = new qp_batch();
  h->B = B; h->dtype = dtype; h->n = n; h->m = m; h->nk = nk;
  h->nnzP = blob[H_NNZP]; h->nnzA = blob[H_NNZA]; h->nnzL = blob[H_NNZL]; h->nrows = nrows;
  if (st) h->st = *st; else umpcQPDefaultSettings(&h->st);
  // a straight-line specialisation generated at build time for exactly this structure?
  h->fixed = -1; h->use_tables = 0; h->no_asm = 0;
  // wave-per-robot kernel: available (umpcQPSetKernel(h, 0)) whenever the robot's working set fits the LDS a
  // workgroup may own; the default is the lane-per-robot path, which is faster on most structures
  h->wave_lds = bqp_wave_lds_words(n, m, blob[H_NNZP], blob[H_NNZA], blob[H_NNZL]);
  h->wave = 0;
  {
    const size_t bytes = h->wave_lds * (dtype == UMPC_F32 ? 4 : 8);
    const void *kf = dtype == UMPC_F32 ? (const void *)bqp_wave_kernel<float> : (const void *)bqp_wave_kernel<double>;
    h->wave_ok = bytes <= 160 * 1024 &&
                 hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
  }
  {
    uint64_t hash = 0xcbf29ce484222325ull;
    for (int k = 0; k < nwords; ++k) {
      const uint32_t v = (uint32_t)blob[k];
      for (int q = 0; q < 4; ++q) { hash ^= (v >> (8 * q)) & 0xffu; hash *= 0x100000001b3ull; }
    }
    for (int k = 0; k < kNumFixedKernels; ++k) if (kFixedKernels[k].hash == hash) h->fixed = k;
  }
  const size_t esz = dtype == UMPC_F32 ? 4 : 8;
  h->asm_tail = (h->fixed >= 0 && kFixedKernels[h->fixed].asm_f32 && dtype == UMPC_F32) ? asm_tail_elems(B) : 0;
  if (hipMalloc((void **)&h->tab, (size_t)nwords * 4) != hipSuccess ||
      hipMalloc(&h->W, ((size_t)nrows * (size_t)B + h->asm_tail) * esz) != hipSuccess) {
    umpc_set_error("umpcQPCreate: hipMalloc failed");
    delete h;
    return nullptr;
  }
  if (hipMemcpy(h->tab, blob, (size_t)nwords * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemset(h->W, 0, (size_t)nrows * (size_t)B * esz) != hipSuccess) {
    umpc_set_error("umpcQPCreate: table upload failed");
    (void)hipFree(h->tab); (void)hipFree(h->W);
    delete h;
    return nullptr;
  }
  return h;
}

void umpcQPDestroy(void *hv) {
  qp_batch *h = (qp_batch *)hv;
  if (!h) return;
  (void)hipFree(h->tab);
  (void)hipFree(h->W);
  delete h;
}

int umpcQPUseTables(void *hv, int on) {
  qp_batch *h = (qp_batch *)hv;
  if (!h) { umpc_set_error("umpcQPUseTables: bad argument"); return -1; }
  h->use_tables = on ? 1 : 0;
  return h->fixed;
}

int umpcQPSetKernel(void *hv, int mode) {
  qp_batch *h = (qp_batch *)hv;
  if (!h || mode < 0 || mode > 3) { umpc_set_error("umpcQPSetKernel: bad argument"); return -1; }
  const size_t bytes = h->wave_lds * (h->dtype == UMPC_F32 ? 4 : 8);
  (void)bytes;
  if (mode == 0 && !h->wave_ok) { umpc_set_error("umpcQPSetKernel: working set exceeds the LDS of a CU"); return -1; }
  h->wave = mode == 0;
  h->use_tables = mode == 2;
  h->no_asm = mode == 3;
  return 0;
}

const char *umpcQPKernelName(void *hv) {
  qp_batch *h = (qp_batch *)hv;
  if (!h) return "";
  if (h->use_tables || h->st.check_termination > 0 || h->st.adaptive_rho_interval > 0) return "tables";
  if (h->wave) return "wave";
  if (asm_active(h)) {      // "<structure>+asm": the specialisation with its middle iterations in assembly
    static thread_local std::string nm;
    nm = std::string(kFixedKernels[h->fixed].name) + "+asm";
    return nm.c_str();
  }
  return h->fixed >= 0 ? kFixedKernels[h->fixed].name : "tables";
}

int umpcQPSetMaxIter(void *hv, int max_iter) {
  qp_batch *h = (qp_batch *)hv;
  if (!h || max_iter < 0) { umpc_set_error("umpcQPSetMaxIter: bad argument"); return -1; }
  h->st.max_iter = max_iter;
  return 0;
}

int umpcQPSetCheckTermination(void *hv, int every) {
  qp_batch *h = (qp_batch *)hv;
  if (!h || every < 0) { umpc_set_error("umpcQPSetCheckTermination: bad argument"); return -1; }
  h->st.check_termination = every;
  return 0;
}

int umpcQPSetAdaptiveRho(void *hv, int interval) {
  qp_batch *h = (qp_batch *)hv;
  if (!h || interval < 0) { umpc_set_error("umpcQPSetAdaptiveRho: bad argument"); return -1; }
  h->st.adaptive_rho_interval = interval;
  return 0;
}

int umpcQPSolve(void *hv, const void *Pv, const void *Av, const void *q, const void *l, const void *u, void *x,
                void *y, void *z, void *Eprev, void *sol_x, void *sol_y, int32_t *status, void *info, void *stream) {
  qp_batch *h = (qp_batch *)hv;
  if (!h || !Av || !q || !l || !u || !x || !y || !z || !Eprev || (h->nnzP > 0 && !Pv)) {
    umpc_set_error("umpcQPSolve: null array");
    return -1;
  }
  hipStream_t s = (hipStream_t)stream;
  if (h->dtype == UMPC_F32) launch_solve<float>(h, Pv, Av, q, l, u, x, y, z, Eprev, sol_x, sol_y, status, info, s);
  else launch_solve<double>(h, Pv, Av, q, l, u, x, y, z, Eprev, sol_x, sol_y, status, info, s);
  return check_launch("umpcQPSolve");
}

int umpcP5fTick(void *hv, const void *Pv, void *Av, const void *q, const void *l, const void *u, void *x, void *y, void *z,
                void *Eprev, void *sol_x, void *sol_y, int32_t *status, void *info, double unom, double dt, void *ystate,
                void *lin, int nnz, const void *cst, const int32_t *src, void *stream) {
  qp_batch *h = (qp_batch *)hv;
  if (!h || !Av || !q || !l || !u || !x || !y || !z || !Eprev || !ystate || !cst || !src || nnz <= 0 || (h->nnzP > 0 && !Pv)) {
    umpc_set_error("umpcP5fTick: bad argument");
    return -1;
  }
  // the fused tick is the prologue of ONE kernel: the fp32 p5f10 assembly specialisation on its all-assembly route
  if (!(asm_active(h) && std::string(kFixedKernels[h->fixed].name) == "p5f10" && nnz == h->nnzA)) {
    umpc_set_error("umpcP5fTick: this handle does not dispatch the p5f10 assembly kernel (use umpcP5fLinearise + umpcQPSolve + umpcP5fStepU)");
    return -2;
  }
  QPArgs<float> t;
  t.tick_y = (float *)ystate; t.tick_lin = (float *)lin; t.tick_cst = (const float *)cst; t.tick_src = src; t.tick_nnz = nnz;
  t.tick_u = (float)unom; t.tick_dt = (float)dt;
  launch_solve<float>(h, Pv, Av, q, l, u, x, y, z, Eprev, sol_x, sol_y, status, info, (hipStream_t)stream, &t);
  return check_launch("umpcP5fTick");
}

int umpcQPGather(int B, int dtype, int nnz, const void *cst, const int32_t *src, const void *par, void *out,
                 void *stream) {
  if (B <= 0 || nnz <= 0 || !cst || !src || !out) { umpc_set_error("umpcQPGather: bad argument"); return -1; }
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((B + 255) / 256, (nnz + kGatherRows - 1) / kGatherRows);
  if (dtype == UMPC_F32)
    hipLaunchKernelGGL((bqp_gather_kernel<float, false>), grid, dim3(256), 0, s, B, nnz, (const float *)cst, src,
                       (const float *)par, (float *)out);
  else
    hipLaunchKernelGGL((bqp_gather_kernel<double, false>), grid, dim3(256), 0, s, B, nnz, (const double *)cst, src,
                       (const double *)par, (double *)out);
  return check_launch("umpcQPGather");
}

int umpcQPGatherUpdate(int B, int dtype, int nnz, const void *cst, const int32_t *src, const void *par, void *out,
                       void *stream) {
  if (B <= 0 || nnz <= 0 || !cst || !src || !par || !out) { umpc_set_error("umpcQPGatherUpdate: bad argument"); return -1; }
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((B + 255) / 256, (nnz + kGatherRows - 1) / kGatherRows);
  if (dtype == UMPC_F32)
    hipLaunchKernelGGL((bqp_gather_kernel<float, true>), grid, dim3(256), 0, s, B, nnz, (const float *)cst, src,
                       (const float *)par, (float *)out);
  else
    hipLaunchKernelGGL((bqp_gather_kernel<double, true>), grid, dim3(256), 0, s, B, nnz, (const double *)cst, src,
                       (const double *)par, (double *)out);
  return check_launch("umpcQPGatherUpdate");
}

int umpcP5fStep(int B, int dtype, int mode, double dt, const void *u, void *y, void *lin, void *stream) {
  if (B <= 0 || !u || !y || (mode != 0 && mode != 1) || (mode == 0 && !lin)) {
    umpc_set_error("umpcP5fStep: bad argument");
    return -1;
  }
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UMPC_F32)
    hipLaunchKernelGGL(p5f_kernel<float>, dim3((B + 255) / 256), dim3(256), 0, s, B, mode, (float)dt, (const float *)u, 0.f,
                       (float *)y, (float *)lin);
  else
    hipLaunchKernelGGL(p5f_kernel<double>, dim3((B + 255) / 256), dim3(256), 0, s, B, mode, dt, (const double *)u, 0.0,
                       (double *)y, (double *)lin);
  return check_launch("umpcP5fStep");
}

int umpcP5fStepU(int B, int dtype, int mode, double dt, double u, void *y, void *lin, void *stream) {
  if (B <= 0 || !y || (mode != 0 && mode != 1) || (mode == 0 && !lin)) {
    umpc_set_error("umpcP5fStepU: bad argument");
    return -1;
  }
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UMPC_F32)
    hipLaunchKernelGGL(p5f_kernel<float>, dim3((B + 255) / 256), dim3(256), 0, s, B, mode, (float)dt, (const float *)nullptr,
                       (float)u, (float *)y, (float *)lin);
  else
    hipLaunchKernelGGL(p5f_kernel<double>, dim3((B + 255) / 256), dim3(256), 0, s, B, mode, dt, (const double *)nullptr, u,
                       (double *)y, (double *)lin);
  return check_launch("umpcP5fStepU");
}

int umpcP5fLinearise(int B, int dtype, const void *u, double u_all, const void *y, void *lin, int nnz, const void *cst,
                     const int32_t *src, void *Av, int update, void *stream) {
  if (B <= 0 || nnz <= 0 || !y || !lin || !cst || !src || !Av) { umpc_set_error("umpcP5fLinearise: bad argument"); return -1; }
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((B + 255) / 256, 1 + (nnz + kGatherRows - 1) / kGatherRows);
  if (dtype == UMPC_F32)
    hipLaunchKernelGGL(p5f_linearise_kernel<float>, grid, dim3(256), 0, s, B, (const float *)u, (float)u_all, (const float *)y,
                       (float *)lin, nnz, (const float *)cst, src, (float *)Av, update);
  else
    hipLaunchKernelGGL(p5f_linearise_kernel<double>, grid, dim3(256), 0, s, B, (const double *)u, u_all, (const double *)y,
                       (double *)lin, nnz, (const double *)cst, src, (double *)Av, update);
  return check_launch("umpcP5fLinearise");
}

int umpcNAssemble(int B, int dtype, int N, const umpcNParams *p, const void *state, const void *ref, void *T0,
                  const void *actualT0, void *Pv, void *q, void *l, void *u, void *par, void *stream) {
  if (B <= 0 || N < 2 || !p || !state || !ref || !T0 || !Pv || !q || !l || !u || !par) {
    umpc_set_error("umpcNAssemble: bad argument");
    return -1;
  }
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((B + 255) / 256), block(256);
  if (dtype == UMPC_F32)
    hipLaunchKernelGGL(umpcn_assemble_kernel<float>, grid, block, 0, s, make_nargs<float>(B, N, p), (const float *)state,
                       (const float *)ref, (float *)T0, (const float *)actualT0, (float *)Pv, (float *)q, (float *)l,
                       (float *)u, (float *)par);
  else
    hipLaunchKernelGGL(umpcn_assemble_kernel<double>, grid, block, 0, s, make_nargs<double>(B, N, p),
                       (const double *)state, (const double *)ref, (double *)T0, (const double *)actualT0, (double *)Pv,
                       (double *)q, (double *)l, (double *)u, (double *)par);
  return check_launch("umpcNAssemble");
}

int umpcNExtract(int B, int dtype, int N, double dt, const void *state, const void *sol_x, void *T0, void *out,
                 void *stream) {
  if (B <= 0 || N < 2 || !state || !sol_x || !T0 || !out) { umpc_set_error("umpcNExtract: bad argument"); return -1; }
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((B + 255) / 256), block(256);
  if (dtype == UMPC_F32)
    hipLaunchKernelGGL(umpcn_extract_kernel<float>, grid, block, 0, s, B, N, (float)dt, (const float *)state,
                       (const float *)sol_x, (float *)T0, (float *)out);
  else
    hipLaunchKernelGGL(umpcn_extract_kernel<double>, grid, block, 0, s, B, N, dt, (const double *)state,
                       (const double *)sol_x, (double *)T0, (double *)out);
  return check_launch("umpcNExtract");
}

}  // extern "C"
