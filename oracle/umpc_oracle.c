/*
 * TEST INFRASTRUCTURE — CPU oracle for the uprightmpc2 hot path (see
 * umpc_oracle.h). Plain C restatement of:
 *   QP assembly   template/uprightmpc2/uprightmpc2.c:19-272
 *   OSQP 0.6.0    template/uprightmpc2/{osqp,auxil,scaling,kkt,lin_alg,proj,
 *                 qdldl,qdldl_interface}.c (EMBEDDED==2 code paths, DFLOAT)
 *   symbolic KKT  kkt.c:6-177 (form_KKT), cs.c csc_symperm, qdldl.c:34-83 (etree)
 *   plant         template/genqp.py:24-41 (plant_impl.h)
 *
 * The operation ORDER and the float/double promotions of the reference are
 * mimicked expression by expression so that ORACLE_REAL=float is bit-identical
 * to the gcc -O2 build of the reference on x86-64 (no FMA contraction:
 * compile with -ffp-contract=off). Where the reference mixes a `double`
 * literal into a float expression the same literal type is used here and
 * marked  "dbl"  in a comment.
 *
 * Two modes:
 *   faithful  (default) the persistent scaled problem data is unscaled,
 *             patched and re-equilibrated on every call exactly like
 *             osqp_update_P_A does (osqp.c:1211-1248);
 *   canonical (o->canonical=1) every call starts from exact raw data (what the
 *             HIP kernel does); the only memory of the previous call is
 *             x, y, z, T0 and the thrust-row scaling E[36..38] that
 *             osqp_update_bounds uses to classify constraints.
 */
#include "umpc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- OSQP constants (template/uprightmpc2/constants.h:59-110) -------- */
#define O_RHO_MIN (1e-06)              /* dbl */
#define O_RHO_EQ_OVER_RHO_INEQ (1e03)  /* dbl */
#define O_RHO_TOL (1e-04)              /* dbl */
#define O_MIN_SCALING (1e-04)          /* dbl */
#define O_MAX_SCALING (1e+04)          /* dbl */
#define O_INFTY ((real)1e30)
#define O_NAN ((real)0x7fc00000UL) /* sic: the reference's "NaN" is the NUMBER 2143289344 (constants.h:96) */
#define O_SCALING_ITERS 10

enum { ST_SOLVED = 1, ST_SOLVED_INACC = 2, ST_PINF_INACC = 3, ST_DINF_INACC = 4,
       ST_MAX_ITER = -2, ST_PINF = -3, ST_DINF = -4, ST_NON_CVX = -7, ST_UNSOLVED = -10 };

#define NX UMPC_NX
#define NC UMPC_NC
#define NK UMPC_NK
#define NY UMPC_NY
#define NU UMPC_NU
#define NN UMPC_N

struct umpc_oracle {
  /* --- controller parameters (UprightMPC_t, uprightmpc2.h:27-43) --- */
  real dt, g, Tmax;
  real Qyr[6], Qyf[6], Qdyr[6], Qdyf[6], Rw[3];
  real e3h[9], e3hIbi[9];
  real l_new[NC], u_new[NC], q_new[NX];
  real Px_data[NX], Ax_data[UMPC_NADATA];
  int Ax_idx[UMPC_NADATA], nAxT0dt, nAxdt;
  real c0[NY];
  real T0;
  /* --- OSQP settings (workspace.c:561) --- */
  real rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
  int max_iter, canonical;
  real Eprev3[NN]; /* canonical mode: thrust-row E of the previous call */
  /* --- structure --- */
  int nnzA, nnzK, nnzL;
  int A_p[NX + 1], A_i[UMPC_NNZA_MAX];
  int perm[NK];
  int K_p[NK + 1], K_i[UMPC_NNZK_MAX];
  int PtoKKT[NX], AtoKKT[UMPC_NNZA_MAX], rhotoKKT[NC];
  int etree[NK], Lnz[NK], L_p[NK + 1], L_i[UMPC_NNZL_MAX];
  /* --- OSQP data (scaled) --- */
  real P_x[NX], A_x[UMPC_NNZA_MAX], q[NX], l[NC], u[NC];
  real A_x0[UMPC_NNZA_MAX]; /* setup values of A (the +-1 constants) */
  real c, cinv, D[NX], Dinv[NX], E[NC], Einv[NC];
  real rho_vec[NC], rho_inv_vec[NC];
  int constr_type[NC];
  real K_x[UMPC_NNZK_MAX], L_x[UMPC_NNZL_MAX], Dd[NK], Ddinv[NK];
  /* --- iterates --- */
  real xa[NX], xb[NX], za[NC], zb[NC];
  real *x, *x_prev, *z, *z_prev; /* swapped like osqp.c:356-357 */
  real y[NC], xz_tilde[NK];
  real Ax[NC], Px[NX], Aty[NX], delta_y[NC], Atdelta_y[NX], delta_x[NX],
      Pdelta_x[NX], Adelta_x[NC];
  real sol_x[NX], sol_y[NC];
  /* --- info --- */
  int iter, status_val, factor_ret;
  real pri_res, dua_res, obj_val;
};

size_t umpc_oracle_sizeof(void) { return sizeof(struct umpc_oracle); }
int umpc_oracle_real_size(void) { return (int)sizeof(real); }

static real r_sqrt(real v) { return sizeof(real) == 4 ? (real)sqrtf((float)v) : (real)sqrt((double)v); }
#define c_absval(x) (((x) < 0) ? -(x) : (x))
#define c_max(a, b) (((a) > (b)) ? (a) : (b))
#define c_min(a, b) (((a) < (b)) ? (a) : (b))

/* ====================================================================== */
/* Symbolic structure                                                      */
/* ====================================================================== */

/* A sparsity + setup values: template/template_controllers.py:28-63
 * (initConstraint with T0 = dt = 1, s0 = ones(3), Btau = ones((3,2))); the CSC
 * this produces is workspace.c:154-428. */
static void build_A(struct umpc_oracle *o) {
  static real dense[NC][NX];
  memset(dense, 0, sizeof(dense));
  const int n1 = NN * NY, n2 = 2 * NN * NY, nc1 = NN * NY, nc2 = 2 * NN * NY;
  for (int k = 0; k < NN; ++k) {
    for (int i = 0; i < NY; ++i) {
      dense[k * NY + i][k * NY + i] = -1;
      dense[k * NY + i][n1 + k * NY + i] = 1; /* dt */
      if (k > 0) dense[k * NY + i][(k - 1) * NY + i] = 1;
      dense[nc1 + k * NY + i][n1 + k * NY + i] = -1;
      if (k > 0) dense[nc1 + k * NY + i][n1 + (k - 1) * NY + i] = 1;
    }
    for (int i = 0; i < 3; ++i) {
      dense[nc1 + k * NY + i][n2 + k * NU] = 1; /* s0 */
      dense[nc1 + k * NY + 3 + i][n2 + k * NU + 1] = 1; /* Btau col 0 */
      dense[nc1 + k * NY + 3 + i][n2 + k * NU + 2] = 1; /* Btau col 1 */
      if (k > 1) dense[nc1 + k * NY + i][(k - 2) * NY + 3 + i] = 1; /* A0: T0*dt */
    }
    dense[nc2 + k][n2 + 3 * k] = 1;
  }
  int nz = 0;
  for (int j = 0; j < NX; ++j) {
    o->A_p[j] = nz;
    for (int i = 0; i < NC; ++i)
      if (dense[i][j] != 0) { o->A_i[nz] = i; o->A_x[nz] = dense[i][j]; ++nz; }
  }
  o->A_p[NX] = nz;
  o->nnzA = nz;
}

/* own fill-reducing ordering: greedy minimum degree on the KKT graph (ties ->
 * lowest index). Used only when no permutation is supplied. */
static void own_ordering(const struct umpc_oracle *o, int *perm) {
  static unsigned char adj[NK][NK];
  memset(adj, 0, sizeof(adj));
  for (int j = 0; j < NX; ++j)
    for (int p = o->A_p[j]; p < o->A_p[j + 1]; ++p) {
      int r = NX + o->A_i[p];
      adj[r][j] = adj[j][r] = 1;
    }
  int alive[NK];
  for (int i = 0; i < NK; ++i) alive[i] = 1;
  for (int k = 0; k < NK; ++k) {
    int best = -1, bestdeg = 1 << 30;
    for (int i = 0; i < NK; ++i) {
      if (!alive[i]) continue;
      int d = 0;
      for (int j = 0; j < NK; ++j) d += (alive[j] && adj[i][j]);
      if (d < bestdeg) { bestdeg = d; best = i; }
    }
    perm[k] = best;
    alive[best] = 0;
    for (int a = 0; a < NK; ++a)
      if (alive[a] && adj[best][a])
        for (int b = 0; b < NK; ++b)
          if (alive[b] && adj[best][b] && a != b) adj[a][b] = 1;
  }
}

/* form_KKT (kkt.c:6-177, upper-triangular CSC of [[P+sI, A'],[., -1/rho]])
 * followed by the symmetric permutation csc_symperm (cs.c; CSparse cs_symperm)
 * and QDLDL_etree (qdldl.c:34-83). */
static void build_KKT(struct umpc_oracle *o) {
  /* triplets in form_KKT insertion order */
  static int ti[UMPC_NNZK_MAX], tj[UMPC_NNZK_MAX];
  int nz = 0;
  int PtoT[NX], AtoT[UMPC_NNZA_MAX], RtoT[NC];
  for (int j = 0; j < NX; ++j) { ti[nz] = j; tj[nz] = j; PtoT[j] = nz++; }
  for (int j = 0; j < NX; ++j)
    for (int p = o->A_p[j]; p < o->A_p[j + 1]; ++p) { ti[nz] = j; tj[nz] = NX + o->A_i[p]; AtoT[p] = nz++; }
  for (int j = 0; j < NC; ++j) { ti[nz] = NX + j; tj[nz] = NX + j; RtoT[j] = nz++; }
  o->nnzK = nz;
  /* triplet_to_csc: bucket by column, stable */
  static int Cp[NK + 1], Ci[UMPC_NNZK_MAX], TtoC[UMPC_NNZK_MAX], w[NK];
  memset(w, 0, sizeof(w));
  for (int k = 0; k < nz; ++k) w[tj[k]]++;
  Cp[0] = 0;
  for (int j = 0; j < NK; ++j) { Cp[j + 1] = Cp[j] + w[j]; w[j] = Cp[j]; }
  for (int k = 0; k < nz; ++k) { int p = w[tj[k]]++; Ci[p] = ti[k]; TtoC[k] = p; }
  /* symmetric permutation C = P K P', upper part only */
  int pinv[NK];
  for (int k = 0; k < NK; ++k) pinv[o->perm[k]] = k;
  memset(w, 0, sizeof(w));
  for (int j = 0; j < NK; ++j) {
    int j2 = pinv[j];
    for (int p = Cp[j]; p < Cp[j + 1]; ++p) {
      int i = Ci[p];
      if (i > j) continue;
      int i2 = pinv[i];
      w[c_max(i2, j2)]++;
    }
  }
  o->K_p[0] = 0;
  for (int j = 0; j < NK; ++j) { o->K_p[j + 1] = o->K_p[j] + w[j]; w[j] = o->K_p[j]; }
  static int CtoK[UMPC_NNZK_MAX];
  for (int j = 0; j < NK; ++j) {
    int j2 = pinv[j];
    for (int p = Cp[j]; p < Cp[j + 1]; ++p) {
      int i = Ci[p];
      if (i > j) continue;
      int i2 = pinv[i];
      int q = w[c_max(i2, j2)]++;
      o->K_i[q] = c_min(i2, j2);
      CtoK[p] = q;
    }
  }
  for (int j = 0; j < NX; ++j) o->PtoKKT[j] = CtoK[TtoC[PtoT[j]]];
  for (int p = 0; p < o->nnzA; ++p) o->AtoKKT[p] = CtoK[TtoC[AtoT[p]]];
  for (int j = 0; j < NC; ++j) o->rhotoKKT[j] = CtoK[TtoC[RtoT[j]]];
  /* elimination tree + column counts */
  for (int i = 0; i < NK; ++i) { w[i] = 0; o->Lnz[i] = 0; o->etree[i] = -1; }
  for (int j = 0; j < NK; ++j) {
    w[j] = j;
    for (int p = o->K_p[j]; p < o->K_p[j + 1]; ++p) {
      int i = o->K_i[p];
      while (w[i] != j) {
        if (o->etree[i] == -1) o->etree[i] = j;
        o->Lnz[i]++;
        w[i] = j;
        i = o->etree[i];
      }
    }
  }
  o->L_p[0] = 0;
  for (int i = 0; i < NK; ++i) o->L_p[i + 1] = o->L_p[i] + o->Lnz[i];
  o->nnzL = o->L_p[NK];
}

/* ====================================================================== */
/* Numeric kernels of OSQP                                                 */
/* ====================================================================== */

/* QDLDL_factor, qdldl.c:86-247 */
static int ldl_factor(struct umpc_oracle *o) {
  const int n = NK;
  const int *Ap = o->K_p, *Ai = o->K_i, *Lp = o->L_p, *etree = o->etree;
  const real *Ax = o->K_x;
  int *Li = o->L_i;
  real *Lx = o->L_x, *D = o->Dd, *Dinv = o->Ddinv;
  int yMarkers[NK], yIdx[NK], elimBuffer[NK], LNext[NK];
  real yVals[NK];
  int positive = 0;
  for (int i = 0; i < n; ++i) { yMarkers[i] = 0; yVals[i] = 0.0; D[i] = 0.0; LNext[i] = Lp[i]; }
  D[0] = Ax[0];
  if (D[0] == 0.0) return -1;
  if (D[0] > 0.0) positive++;
  Dinv[0] = 1 / D[0];
  for (int k = 1; k < n; ++k) {
    int nnzY = 0;
    for (int i = Ap[k]; i < Ap[k + 1]; ++i) {
      int bidx = Ai[i];
      if (bidx == k) { D[k] = Ax[i]; continue; }
      yVals[bidx] = Ax[i];
      int next = bidx;
      if (yMarkers[next] == 0) {
        yMarkers[next] = 1;
        elimBuffer[0] = next;
        int nnzE = 1;
        next = etree[bidx];
        while (next != -1 && next < k) {
          if (yMarkers[next] == 1) break;
          yMarkers[next] = 1;
          elimBuffer[nnzE++] = next;
          next = etree[next];
        }
        while (nnzE) yIdx[nnzY++] = elimBuffer[--nnzE];
      }
    }
    for (int i = nnzY - 1; i >= 0; --i) {
      int cidx = yIdx[i];
      int tmpIdx = LNext[cidx];
      real yv = yVals[cidx];
      for (int j = Lp[cidx]; j < tmpIdx; ++j) yVals[Li[j]] -= Lx[j] * yv;
      Li[tmpIdx] = k;
      Lx[tmpIdx] = yv * Dinv[cidx];
      D[k] -= yv * Lx[tmpIdx];
      LNext[cidx]++;
      yVals[cidx] = 0.0;
      yMarkers[cidx] = 0;
    }
    if (D[k] == 0.0) return -1;
    if (D[k] > 0.0) positive++;
    Dinv[k] = 1 / D[k];
  }
  return positive;
}

/* solve_linsys_qdldl + LDLSolve + QDLDL_solve: qdldl_interface.c:322-369,
 * qdldl.c:250-293. b = xz_tilde (in/out). */
static void kkt_solve(struct umpc_oracle *o, real *b) {
  real bp[NK], sol[NK];
  const int *Lp = o->L_p, *Li = o->L_i;
  const real *Lx = o->L_x;
  for (int j = 0; j < NK; ++j) bp[j] = b[o->perm[j]];
  for (int i = 0; i < NK; ++i)
    for (int j = Lp[i]; j < Lp[i + 1]; ++j) bp[Li[j]] -= Lx[j] * bp[i];
  for (int i = 0; i < NK; ++i) bp[i] *= o->Ddinv[i];
  for (int i = NK - 1; i >= 0; --i)
    for (int j = Lp[i]; j < Lp[i + 1]; ++j) bp[i] -= Lx[j] * bp[Li[j]];
  for (int j = 0; j < NK; ++j) sol[o->perm[j]] = bp[j];
  for (int j = 0; j < NX; ++j) b[j] = sol[j];
  for (int j = 0; j < NC; ++j) b[j + NX] += o->rho_inv_vec[j] * sol[j + NX];
}

/* update_KKT_P / update_KKT_A / update_KKT_param2 (kkt.c:184-222) + factor */
static int kkt_update_PA_and_factor(struct umpc_oracle *o) {
  for (int i = 0; i < NX; ++i) o->K_x[o->PtoKKT[i]] = o->P_x[i];
  for (int i = 0; i < NX; ++i) o->K_x[o->PtoKKT[i]] += o->sigma;
  for (int i = 0; i < o->nnzA; ++i) o->K_x[o->AtoKKT[i]] = o->A_x[i];
  o->factor_ret = ldl_factor(o);
  return o->factor_ret < 0;
}
static int kkt_update_rho_and_factor(struct umpc_oracle *o) {
  /* update_linsys_solver_rho_vec_qdldl, qdldl_interface.c:389-403 */
  for (int i = 0; i < NC; ++i) o->K_x[o->rhotoKKT[i]] = -(real)(1. / o->rho_vec[i]); /* dbl */
  o->factor_ret = ldl_factor(o);
  return o->factor_ret < 0;
}

/* A*x, A'*y helpers: lin_alg.c:194-283 (same accumulation order) */
static void mat_vec_A(const struct umpc_oracle *o, const real *x, real *y) {
  for (int i = 0; i < NC; ++i) y[i] = 0;
  for (int j = 0; j < NX; ++j)
    for (int p = o->A_p[j]; p < o->A_p[j + 1]; ++p) y[o->A_i[p]] += o->A_x[p] * x[j];
}
static void mat_tpose_vec_A(const struct umpc_oracle *o, const real *x, real *y) {
  for (int j = 0; j < NX; ++j) y[j] = 0;
  for (int j = 0; j < NX; ++j)
    for (int p = o->A_p[j]; p < o->A_p[j + 1]; ++p) y[j] += o->A_x[p] * x[o->A_i[p]];
}
static real vec_norm_inf(const real *v, int n) {
  real mx = 0.0;
  for (int i = 0; i < n; ++i) { real a = c_absval(v[i]); if (a > mx) mx = a; }
  return mx;
}
static real vec_scaled_norm_inf(const real *S, const real *v, int n) {
  real mx = 0.0;
  for (int i = 0; i < n; ++i) { real a = c_absval(S[i] * v[i]); if (a > mx) mx = a; }
  return mx;
}

/* limit_scaling, scaling.c:7-14 (comparisons and the 1.0 / MAX literals are double) */
static void limit_scaling(real *D, int n) {
  for (int i = 0; i < n; ++i) {
    D[i] = D[i] < O_MIN_SCALING ? 1.0 : D[i];           /* dbl */
    D[i] = D[i] > O_MAX_SCALING ? O_MAX_SCALING : D[i]; /* dbl */
  }
}

/* scale_data, scaling.c:44-156 (P is diagonal here: Pdata_i[j] == j, workspace.c:11-57) */
static void scale_data(struct umpc_oracle *o) {
  real D_temp[NX], D_temp_A[NX], E_temp[NC];
  o->c = 1.0;
  for (int i = 0; i < NX; ++i) { o->D[i] = 1.; o->Dinv[i] = 1.; }
  for (int i = 0; i < NC; ++i) { o->E[i] = 1.; o->Einv[i] = 1.; }
  for (int it = 0; it < O_SCALING_ITERS; ++it) {
    /* compute_inf_norm_cols_KKT, scaling.c:29-42 */
    for (int j = 0; j < NX; ++j) { D_temp[j] = 0.; D_temp[j] = c_max(c_absval(o->P_x[j]), D_temp[j]); }
    for (int j = 0; j < NX; ++j) {
      D_temp_A[j] = 0.;
      for (int p = o->A_p[j]; p < o->A_p[j + 1]; ++p) D_temp_A[j] = c_max(c_absval(o->A_x[p]), D_temp_A[j]);
    }
    for (int j = 0; j < NX; ++j) D_temp[j] = c_max(D_temp[j], D_temp_A[j]);
    for (int i = 0; i < NC; ++i) E_temp[i] = 0.;
    for (int j = 0; j < NX; ++j)
      for (int p = o->A_p[j]; p < o->A_p[j + 1]; ++p) {
        int i = o->A_i[p];
        E_temp[i] = c_max(c_absval(o->A_x[p]), E_temp[i]);
      }
    limit_scaling(D_temp, NX);
    limit_scaling(E_temp, NC);
    for (int i = 0; i < NX; ++i) D_temp[i] = r_sqrt(D_temp[i]);
    for (int i = 0; i < NC; ++i) E_temp[i] = r_sqrt(E_temp[i]);
    for (int i = 0; i < NX; ++i) D_temp[i] = (real)1.0 / D_temp[i];
    for (int i = 0; i < NC; ++i) E_temp[i] = (real)1.0 / E_temp[i];
    /* P <- D P D (pre then post) */
    for (int j = 0; j < NX; ++j) o->P_x[j] *= D_temp[j];
    for (int j = 0; j < NX; ++j) o->P_x[j] *= D_temp[j];
    /* A <- E A D */
    for (int j = 0; j < NX; ++j)
      for (int p = o->A_p[j]; p < o->A_p[j + 1]; ++p) o->A_x[p] *= E_temp[o->A_i[p]];
    for (int j = 0; j < NX; ++j)
      for (int p = o->A_p[j]; p < o->A_p[j + 1]; ++p) o->A_x[p] *= D_temp[j];
    for (int i = 0; i < NX; ++i) o->q[i] = o->q[i] * D_temp[i];
    for (int i = 0; i < NX; ++i) o->D[i] = D_temp[i] * o->D[i];
    for (int i = 0; i < NC; ++i) o->E[i] = E_temp[i] * o->E[i];
    /* cost normalisation */
    for (int j = 0; j < NX; ++j) { D_temp[j] = 0.; D_temp[j] = c_max(c_absval(o->P_x[j]), D_temp[j]); }
    real c_temp = 0.0;
    for (int i = 0; i < NX; ++i) c_temp += D_temp[i];
    c_temp /= (real)NX;
    real inf_norm_q = vec_norm_inf(o->q, NX);
    limit_scaling(&inf_norm_q, 1);
    c_temp = c_max(c_temp, inf_norm_q);
    limit_scaling(&c_temp, 1);
    c_temp = 1. / c_temp; /* dbl */
    for (int i = 0; i < NX; ++i) o->P_x[i] *= c_temp;
    for (int i = 0; i < NX; ++i) o->q[i] *= c_temp;
    o->c *= c_temp;
  }
  o->cinv = 1. / o->c; /* dbl */
  for (int i = 0; i < NX; ++i) o->Dinv[i] = (real)1.0 / o->D[i];
  for (int i = 0; i < NC; ++i) o->Einv[i] = (real)1.0 / o->E[i];
  for (int i = 0; i < NC; ++i) o->l[i] = o->l[i] * o->E[i];
  for (int i = 0; i < NC; ++i) o->u[i] = o->u[i] * o->E[i];
}

/* unscale_data, scaling.c:160-175 */
static void unscale_data(struct umpc_oracle *o) {
  for (int i = 0; i < NX; ++i) o->P_x[i] *= o->cinv;
  for (int i = 0; i < NX; ++i) o->P_x[i] *= o->Dinv[i];
  for (int i = 0; i < NX; ++i) o->P_x[i] *= o->Dinv[i];
  for (int i = 0; i < NX; ++i) o->q[i] *= o->cinv;
  for (int i = 0; i < NX; ++i) o->q[i] = o->q[i] * o->Dinv[i];
  for (int j = 0; j < NX; ++j)
    for (int p = o->A_p[j]; p < o->A_p[j + 1]; ++p) o->A_x[p] *= o->Einv[o->A_i[p]];
  for (int j = 0; j < NX; ++j)
    for (int p = o->A_p[j]; p < o->A_p[j + 1]; ++p) o->A_x[p] *= o->Dinv[j];
  for (int i = 0; i < NC; ++i) o->l[i] = o->l[i] * o->Einv[i];
  for (int i = 0; i < NC; ++i) o->u[i] = o->u[i] * o->Einv[i];
}

/* update_rho_vec, auxil.c:103-145. ls/us are the E-scaled bounds. */
static int update_rho_vec(struct umpc_oracle *o, const real *ls, const real *us) {
  int changed = 0;
  for (int i = 0; i < NC; ++i) {
    if ((ls[i] < -O_INFTY * O_MIN_SCALING) && (us[i] > O_INFTY * O_MIN_SCALING)) { /* dbl */
      if (o->constr_type[i] != -1) {
        o->constr_type[i] = -1;
        o->rho_vec[i] = O_RHO_MIN;
        o->rho_inv_vec[i] = 1. / O_RHO_MIN;
        changed = 1;
      }
    } else if (us[i] - ls[i] < O_RHO_TOL) { /* dbl compare */
      if (o->constr_type[i] != 1) {
        o->constr_type[i] = 1;
        o->rho_vec[i] = O_RHO_EQ_OVER_RHO_INEQ * o->rho; /* dbl */
        o->rho_inv_vec[i] = 1. / o->rho_vec[i];          /* dbl */
        changed = 1;
      }
    } else {
      if (o->constr_type[i] != 0) {
        o->constr_type[i] = 0;
        o->rho_vec[i] = o->rho;
        o->rho_inv_vec[i] = 1. / o->rho; /* dbl */
        changed = 1;
      }
    }
  }
  return changed;
}

/* osqp_update_bounds, osqp.c:784-833 */
static int osqp_update_bounds_(struct umpc_oracle *o) {
  /* osqp.c:801-808: a crossed pair rejects the whole update (data untouched; umpcUpdate ignores the return,
   * uprightmpc2.c:247). Canonical mode states the HIP kernel's semantics instead: bounds are
   * applied as assembled -- only reachable with Tmax < 0, where the crossed thrust rows classify as equalities
   * (u - l < RHO_TOL, auxil.c:118) and project onto u (proj.c:4-14): the thrust is driven to Tmax. The reference
   * itself keeps solving its code-generated placeholder bounds (tests/golden/bounds_reject.npz). */
  if (!o->canonical)
    for (int i = 0; i < NC; ++i)
      if (o->l_new[i] > o->u_new[i]) return 1;
  if (!o->canonical) {
    for (int i = 0; i < NC; ++i) { o->l[i] = o->l_new[i]; o->u[i] = o->u_new[i]; }
    for (int i = 0; i < NC; ++i) o->l[i] = o->l[i] * o->E[i];
    for (int i = 0; i < NC; ++i) o->u[i] = o->u[i] * o->E[i];
    if (update_rho_vec(o, o->l, o->u)) return kkt_update_rho_and_factor(o);
  } else {
    real ls[NC], us[NC];
    for (int i = 0; i < NC; ++i) {
      real e = i >= 2 * NN * NY ? o->Eprev3[i - 2 * NN * NY] : (real)1;
      ls[i] = o->l_new[i] * e;
      us[i] = o->u_new[i] * e;
    }
    if (update_rho_vec(o, ls, us))
      for (int i = 0; i < NC; ++i) o->K_x[o->rhotoKKT[i]] = -(real)(1. / o->rho_vec[i]);
  }
  return 0;
}

/* osqp_update_lin_cost, osqp.c:752-782 */
static void osqp_update_lin_cost_(struct umpc_oracle *o) {
  for (int i = 0; i < NX; ++i) o->q[i] = o->q_new[i];
  if (!o->canonical) {
    for (int i = 0; i < NX; ++i) o->q[i] = o->q[i] * o->D[i];
    for (int i = 0; i < NX; ++i) o->q[i] *= o->c;
  }
}

/* osqp_update_P_A, osqp.c:1158-1266 (Px full, Ax by index) */
static int osqp_update_P_A_(struct umpc_oracle *o) {
  if (!o->canonical) {
    unscale_data(o);
  } else {
    /* raw data: constants exactly as assembled, bounds unscaled */
    memcpy(o->A_x, o->A_x0, sizeof(o->A_x));
    for (int i = 0; i < NC; ++i) { o->l[i] = o->l_new[i]; o->u[i] = o->u_new[i]; }
  }
  for (int i = 0; i < NX; ++i) o->P_x[i] = o->Px_data[i];
  for (int i = 0; i < UMPC_NADATA; ++i) o->A_x[o->Ax_idx[i]] = o->Ax_data[i];
  scale_data(o);
  int ret = kkt_update_PA_and_factor(o);
  if (o->canonical)
    for (int k = 0; k < NN; ++k) o->Eprev3[k] = o->E[2 * NN * NY + k];
  return ret;
}

/* compute_pri_res / compute_dua_res, auxil.c:243-307 (z_prev / x_prev are scratch) */
static real compute_pri_res(struct umpc_oracle *o) {
  mat_vec_A(o, o->x, o->Ax);
  for (int i = 0; i < NC; ++i) o->z_prev[i] = o->Ax[i] + (real)-1 * o->z[i];
  return vec_scaled_norm_inf(o->Einv, o->z_prev, NC);
}
static real compute_dua_res(struct umpc_oracle *o) {
  for (int i = 0; i < NX; ++i) o->x_prev[i] = o->q[i];
  for (int i = 0; i < NX; ++i) { o->Px[i] = 0; o->Px[i] += o->P_x[i] * o->x[i]; }
  /* mat_tpose_vec(P, x, Px, 1, skip_diag=1) adds 0 for a diagonal P */
  for (int i = 0; i < NX; ++i) o->Px[i] += 0;
  for (int i = 0; i < NX; ++i) o->x_prev[i] = o->x_prev[i] + (real)1 * o->Px[i];
  mat_tpose_vec_A(o, o->y, o->Aty);
  for (int i = 0; i < NX; ++i) o->x_prev[i] = o->x_prev[i] + (real)1 * o->Aty[i];
  return o->cinv * vec_scaled_norm_inf(o->Dinv, o->x_prev, NX);
}

/* is_primal_infeasible, auxil.c:362-424 */
static int is_primal_infeasible(struct umpc_oracle *o, real eps) {
  real ineq_lhs = 0.0;
  for (int i = 0; i < NC; ++i) {
    if (o->u[i] > O_INFTY * O_MIN_SCALING) {
      if (o->l[i] < -O_INFTY * O_MIN_SCALING) o->delta_y[i] = 0.0;
      else o->delta_y[i] = c_min(o->delta_y[i], 0.0);
    } else if (o->l[i] < -O_INFTY * O_MIN_SCALING) {
      o->delta_y[i] = c_max(o->delta_y[i], 0.0);
    }
  }
  for (int i = 0; i < NC; ++i) o->Adelta_x[i] = o->delta_y[i] * o->E[i];
  real norm_dy = vec_norm_inf(o->Adelta_x, NC);
  if (norm_dy > eps) {
    for (int i = 0; i < NC; ++i)
      ineq_lhs += o->u[i] * c_max(o->delta_y[i], 0) + o->l[i] * c_min(o->delta_y[i], 0);
    if (ineq_lhs < -eps * norm_dy) {
      mat_tpose_vec_A(o, o->delta_y, o->Atdelta_y);
      for (int i = 0; i < NX; ++i) o->Atdelta_y[i] = o->Atdelta_y[i] * o->Dinv[i];
      return vec_norm_inf(o->Atdelta_y, NX) < eps * norm_dy;
    }
  }
  return 0;
}

/* is_dual_infeasible, auxil.c:426-512 */
static int is_dual_infeasible(struct umpc_oracle *o, real eps) {
  real norm_dx = vec_scaled_norm_inf(o->D, o->delta_x, NX);
  real cost_scaling = o->c;
  if (norm_dx > eps) {
    real qdx = 0.0;
    for (int i = 0; i < NX; ++i) qdx += o->q[i] * o->delta_x[i];
    if (qdx < -cost_scaling * eps * norm_dx) {
      for (int i = 0; i < NX; ++i) { o->Pdelta_x[i] = 0; o->Pdelta_x[i] += o->P_x[i] * o->delta_x[i]; }
      for (int i = 0; i < NX; ++i) o->Pdelta_x[i] = o->Pdelta_x[i] * o->Dinv[i];
      if (vec_norm_inf(o->Pdelta_x, NX) < cost_scaling * eps * norm_dx) {
        mat_vec_A(o, o->delta_x, o->Adelta_x);
        for (int i = 0; i < NC; ++i) o->Adelta_x[i] = o->Adelta_x[i] * o->Einv[i];
        for (int i = 0; i < NC; ++i) {
          if (((o->u[i] < O_INFTY * O_MIN_SCALING) && (o->Adelta_x[i] > eps * norm_dx)) ||
              ((o->l[i] > -O_INFTY * O_MIN_SCALING) && (o->Adelta_x[i] < -eps * norm_dx)))
            return 0;
        }
        return 1;
      }
    }
  }
  return 0;
}

/* check_termination, auxil.c:684-789 */
static int check_termination(struct umpc_oracle *o, int approximate) {
  real eps_abs = o->eps_abs, eps_rel = o->eps_rel;
  real eps_prim_inf = o->eps_prim_inf, eps_dual_inf = o->eps_dual_inf;
  int prim_res_check = 0, dual_res_check = 0, prim_inf_check = 0, dual_inf_check = 0;
  if ((o->pri_res > O_INFTY) || (o->dua_res > O_INFTY)) {
    o->status_val = ST_NON_CVX;
    o->obj_val = O_NAN;
    return 1;
  }
  if (approximate) { eps_abs *= 10; eps_rel *= 10; eps_prim_inf *= 10; eps_dual_inf *= 10; }
  /* compute_pri_tol, auxil.c:262-288 */
  real max_rel = vec_scaled_norm_inf(o->Einv, o->z, NC);
  real tmp = vec_scaled_norm_inf(o->Einv, o->Ax, NC);
  max_rel = c_max(max_rel, tmp);
  real eps_prim = eps_abs + eps_rel * max_rel;
  if (o->pri_res < eps_prim) prim_res_check = 1;
  else prim_inf_check = is_primal_infeasible(o, eps_prim_inf);
  /* compute_dua_tol, auxil.c:309-349 */
  max_rel = vec_scaled_norm_inf(o->Dinv, o->q, NX);
  tmp = vec_scaled_norm_inf(o->Dinv, o->Aty, NX);
  max_rel = c_max(max_rel, tmp);
  tmp = vec_scaled_norm_inf(o->Dinv, o->Px, NX);
  max_rel = c_max(max_rel, tmp);
  max_rel *= o->cinv;
  real eps_dual = eps_abs + eps_rel * max_rel;
  if (o->dua_res < eps_dual) dual_res_check = 1;
  else dual_inf_check = is_dual_infeasible(o, eps_dual_inf);
  if (prim_res_check && dual_res_check) {
    o->status_val = approximate ? ST_SOLVED_INACC : ST_SOLVED;
    return 1;
  } else if (prim_inf_check) {
    o->status_val = approximate ? ST_PINF_INACC : ST_PINF;
    for (int i = 0; i < NC; ++i) o->delta_y[i] = o->delta_y[i] * o->E[i];
    o->obj_val = O_INFTY;
    return 1;
  } else if (dual_inf_check) {
    o->status_val = approximate ? ST_DINF_INACC : ST_DINF;
    for (int i = 0; i < NX; ++i) o->delta_x[i] = o->delta_x[i] * o->D[i];
    o->obj_val = -O_INFTY;
    return 1;
  }
  return 0;
}

static int has_solution(int s) {
  return s != ST_PINF && s != ST_PINF_INACC && s != ST_DINF && s != ST_DINF_INACC && s != ST_NON_CVX;
}

/* osqp_solve, osqp.c:288-641 with check_termination = 0, adaptive_rho_interval
 * = 0 (never adapts in EMBEDDED 2 without PROFILING: osqp.c:489-491), warm start. */
static int osqp_solve_(struct umpc_oracle *o) {
  int iter;
  for (iter = 1; iter <= o->max_iter; ++iter) {
    real *t;
    t = o->x_prev; o->x_prev = o->x; o->x = t;
    t = o->z_prev; o->z_prev = o->z; o->z = t;
    /* compute_rhs, auxil.c:164-178 */
    for (int i = 0; i < NX; ++i) o->xz_tilde[i] = o->sigma * o->x_prev[i] - o->q[i];
    for (int i = 0; i < NC; ++i) o->xz_tilde[i + NX] = o->z_prev[i] - o->rho_inv_vec[i] * o->y[i];
    kkt_solve(o, o->xz_tilde);
    /* update_x, auxil.c:188-201 */
    for (int i = 0; i < NX; ++i)
      o->x[i] = o->alpha * o->xz_tilde[i] + ((real)1.0 - o->alpha) * o->x_prev[i];
    for (int i = 0; i < NX; ++i) o->delta_x[i] = o->x[i] - o->x_prev[i];
    /* update_z, auxil.c:203-215 + project, proj.c:4-14 */
    for (int i = 0; i < NC; ++i)
      o->z[i] = o->alpha * o->xz_tilde[i + NX] + ((real)1.0 - o->alpha) * o->z_prev[i] +
                o->rho_inv_vec[i] * o->y[i];
    for (int i = 0; i < NC; ++i) o->z[i] = c_min(c_max(o->z[i], o->l[i]), o->u[i]);
    /* update_y, auxil.c:217-228 */
    for (int i = 0; i < NC; ++i) {
      o->delta_y[i] = o->rho_vec[i] * (o->alpha * o->xz_tilde[i + NX] +
                                       ((real)1.0 - o->alpha) * o->z_prev[i] - o->z[i]);
      o->y[i] += o->delta_y[i];
    }
  }
  /* update_info (auxil.c:567-632), check_termination twice (osqp.c:524-573) */
  o->iter = iter - 1;
  o->pri_res = compute_pri_res(o);
  o->dua_res = compute_dua_res(o);
  check_termination(o, 0);
  if (has_solution(o->status_val)) {
    /* compute_obj_val, auxil.c:230-241 (diagonal P) */
    real qf = 0.;
    for (int i = 0; i < NX; ++i) qf += (real).5 * o->P_x[i] * o->x[i] * o->x[i];
    real lin = 0.0;
    for (int i = 0; i < NX; ++i) lin += o->q[i] * o->x[i];
    o->obj_val = (qf + lin) * o->cinv;
  }
  if (o->status_val == ST_UNSOLVED)
    if (!check_termination(o, 1)) o->status_val = ST_MAX_ITER;
  /* store_solution, auxil.c:527-565 */
  if (has_solution(o->status_val)) {
    for (int i = 0; i < NX; ++i) o->sol_x[i] = o->x[i];
    for (int i = 0; i < NC; ++i) o->sol_y[i] = o->y[i];
    for (int i = 0; i < NX; ++i) o->sol_x[i] = o->sol_x[i] * o->D[i];
    for (int i = 0; i < NC; ++i) o->sol_y[i] = o->sol_y[i] * o->E[i];
    for (int i = 0; i < NC; ++i) o->sol_y[i] *= o->cinv;
  } else {
    for (int i = 0; i < NX; ++i) o->sol_x[i] = O_NAN;
    for (int i = 0; i < NC; ++i) o->sol_y[i] = O_NAN;
    for (int i = 0; i < NX; ++i) o->x[i] = 0.;
    for (int i = 0; i < NC; ++i) { o->z[i] = 0.; o->y[i] = 0.; }
  }
  return 0;
}

/* ====================================================================== */
/* uprightmpc2.c                                                           */
/* ====================================================================== */

/* host matMult, column-major (contract: matmult.h:22-35) */
static void mm(real *C, const real *A, const real *B, int m, int n, int k, real alpha, int AT, int BT) {
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < m; ++i) {
      real acc = 0;
      for (int l = 0; l < k; ++l) {
        real a = AT ? A[l + i * k] : A[i + l * m];
        real b = BT ? B[j + l * n] : B[l + j * k];
        acc += a * b;
      }
      C[i + j * m] = alpha * acc;
    }
}

void umpc_oracle_init(umpc_oracle_t *o, real dt, real g, real TtoWmax, real ws,
                      real wds, real wpr, real wpf, real wvr, real wvf,
                      real wthrust, real wmom, const real Ib[3], int maxIter,
                      const int *perm) {
  memset(o, 0, sizeof(*o));
  /* umpcInit, uprightmpc2.c:19-118 */
  real Ibi[9];
  o->dt = dt; o->g = g; o->Tmax = TtoWmax * g;
  for (int i = 0; i < 3; ++i) {
    o->Qyr[i] = wpr; o->Qyf[i] = wpf;
    o->Qyr[3 + i] = o->Qyf[3 + i] = ws;
    o->Qdyr[i] = wvr; o->Qdyf[i] = wvf;
    o->Qdyr[3 + i] = o->Qdyf[3 + i] = wds;
  }
  o->Rw[0] = wthrust; o->Rw[1] = o->Rw[2] = wmom;
  for (int i = 0; i < NY; ++i) o->c0[i] = i == 2 ? -o->g : 0;
  o->T0 = 0;
  o->e3h[1] = 1; o->e3h[3] = -1;
  for (int i = 0; i < 9; ++i) Ibi[i] = 0;
  Ibi[0] = (real)1.0 / Ib[0]; Ibi[4] = (real)1.0 / Ib[1]; Ibi[8] = (real)1.0 / Ib[2];
  mm(o->e3hIbi, o->e3h, Ibi, 3, 3, 3, (real)1.0, 0, 0);
  /* Ax_idx table, uprightmpc2.c:65-113 */
  int offs = 0, n2 = 2 * NY + 3, n1;
  for (int k = 0; k < NN - 2; ++k) {
    o->Ax_idx[offs + 0] = n2 * k + 8; o->Ax_idx[offs + 1] = n2 * k + 11; o->Ax_idx[offs + 2] = n2 * k + 14;
    offs += 3;
  }
  o->nAxT0dt = offs;
  n1 = (2 * NN - 1) * NY + (NN - 2) * 3;
  n2 = 3 * NY;
  for (int k = 0; k < NN; ++k) {
    o->Ax_idx[offs + 0] = n1 + n2 * k + 0;
    for (int i = 1; i < 6; ++i)
      o->Ax_idx[offs + i] = n1 + n2 * k + (k < NN - 1 ? 3 * i : 2 * i);
    offs += 6;
  }
  o->nAxdt = offs;
  n1 += 3 * NY * (NN - 1) + 2 * NY;
  n2 = 10;
  for (int k = 0; k < NN; ++k) { for (int i = 0; i < 3; ++i) o->Ax_idx[offs + i] = n1 + n2 * k + i; offs += 3; }
  for (int k = 0; k < NN; ++k) { for (int i = 0; i < 6; ++i) o->Ax_idx[offs + i] = n1 + n2 * k + 4 + i; offs += 6; }

  /* pristine OSQP workspace: what template_controllers.py:190-191 set up and
   * codegen froze in workspace.c (P = I, A = initConstraint ones, q = 0, l = 0,
   * u = +inf -> 1e30; Ruiz on that data leaves D = E = c = 1; rho = 0.1,
   * all rows "inequality"; settings workspace.c:561; max_iter / no
   * termination check from uprightmpc2.c:116-117). */
  o->rho = (real)0.1; o->sigma = (real)1e-6; o->alpha = (real)1.6;
  o->eps_abs = o->eps_rel = (real)1e-4; o->eps_prim_inf = o->eps_dual_inf = (real)1e-4;
  o->max_iter = maxIter;
  build_A(o);
  memcpy(o->A_x0, o->A_x, sizeof(o->A_x));
  if (perm) memcpy(o->perm, perm, sizeof(o->perm)); else own_ordering(o, o->perm);
  build_KKT(o);
  for (int i = 0; i < NX; ++i) { o->P_x[i] = 1; o->q[i] = 0; o->D[i] = o->Dinv[i] = 1; }
  for (int i = 0; i < NC; ++i) {
    o->l[i] = 0; o->u[i] = O_INFTY; o->E[i] = o->Einv[i] = 1;
    o->rho_vec[i] = o->rho; o->rho_inv_vec[i] = 1. / o->rho; o->constr_type[i] = 0;
  }
  for (int k = 0; k < NN; ++k) o->Eprev3[k] = 1;
  o->c = o->cinv = 1;
  for (int i = 0; i < NX; ++i) { o->K_x[o->PtoKKT[i]] = o->P_x[i]; o->K_x[o->PtoKKT[i]] += o->sigma; }
  for (int i = 0; i < o->nnzA; ++i) o->K_x[o->AtoKKT[i]] = o->A_x[i];
  for (int i = 0; i < NC; ++i) o->K_x[o->rhotoKKT[i]] = -o->rho_inv_vec[i];
  o->x = o->xa; o->x_prev = o->xb; o->z = o->za; o->z_prev = o->zb;
  o->status_val = ST_UNSOLVED;
}

static real A0_times_i(const struct umpc_oracle *o, const real *y, int i) {
  return (i < 3) ? o->T0 * y[i + 3] : 0;
}

int umpc_oracle_update(umpc_oracle_t *o, real uquad[3], real accdes[6],
                       const real p0[3], const real R0[9], const real dq0[6],
                       const real pdes[3], const real dpdes[3],
                       const real sdes[3], real actualT0) {
  real s0[3], ds0[3], y0[NY], dy0[NY], ydes[NY], dydes[NY], dummy[9], Btau[9];
  real dy1des[NY], dq1des[NY], e3hR0T[9], y1[NY];
  /* umpcUpdate, uprightmpc2.c:209-272 */
  if (actualT0 >= 0) o->T0 = actualT0;
  memcpy(s0, &R0[6], 3 * sizeof(real));
  mm(dummy, o->e3h, &dq0[3], 3, 1, 3, (real)1.0, 0, 0);
  mm(ds0, R0, dummy, 3, 1, 3, (real)-1.0, 0, 0);
  mm(Btau, R0, o->e3hIbi, 3, 3, 3, (real)-1.0, 0, 0);
  for (int i = 0; i < NY; ++i) {
    y0[i] = i < 3 ? p0[i] : s0[i - 3];
    dy0[i] = i < 3 ? dq0[i] : ds0[i - 3];
    if (i < 3) { ydes[i] = pdes[i]; dydes[i] = dpdes[i]; }
  }
  ydes[3] = sdes[0]; ydes[4] = sdes[1]; ydes[5] = sdes[2];
  dydes[3] = dydes[4] = dydes[5] = 0;

  /* umpcUpdateConstraint, uprightmpc2.c:126-180 */
  for (int i = 0; i < NN * NY; ++i) o->l_new[i] = 0;
  for (int i = 0; i < NY; ++i) { y1[i] = y0[i] + o->dt * dy0[i]; o->l_new[i] = -y1[i]; }
  for (int k = 0; k < NN; ++k) {
    real *pk = &o->l_new[NY * (NN + k)];
    for (int i = 0; i < NY; ++i) {
      if (k == 0) pk[i] = -dy0[i] - o->dt * A0_times_i(o, y0, i) - o->dt * o->c0[i];
      else if (k == 1) pk[i] = -o->dt * A0_times_i(o, y1, i) - o->dt * o->c0[i];
      else pk[i] = -o->dt * o->c0[i];
    }
  }
  for (int i = 0; i < 2 * NN * NY; ++i) o->u_new[i] = o->l_new[i];
  for (int k = 0; k < NN; ++k) {
    o->l_new[2 * NN * NY + k] = -o->T0;
    o->u_new[2 * NN * NY + k] = o->Tmax - o->T0;
  }
  for (int i = 0; i < o->nAxdt; ++i) o->Ax_data[i] = i < o->nAxT0dt ? o->dt * o->T0 : o->dt;
  int offs = o->nAxdt;
  for (int k = 0; k < NN; ++k) { for (int i = 0; i < 3; ++i) o->Ax_data[offs + i] = o->dt * s0[i]; offs += 3; }
  for (int k = 0; k < NN; ++k) { for (int i = 0; i < 6; ++i) o->Ax_data[offs + i] = o->dt * Btau[i]; offs += 6; }

  /* updateObjective, uprightmpc2.c:182-207 */
  offs = 0;
  for (int k = 0; k < NN; ++k) {
    for (int i = 0; i < NY; ++i) {
      o->Px_data[offs + i] = k == NN - 1 ? o->Qyf[i] : o->Qyr[i];
      o->q_new[offs + i] = -o->Px_data[offs + i] * ydes[i];
    }
    offs += NY;
  }
  for (int k = 0; k < NN; ++k) {
    for (int i = 0; i < NY; ++i) {
      o->Px_data[offs + i] = k == NN - 1 ? o->Qdyf[i] : o->Qdyr[i];
      o->q_new[offs + i] = -o->Px_data[offs + i] * dydes[i];
    }
    offs += NY;
  }
  for (int k = 0; k < NN; ++k) { for (int i = 0; i < NU; ++i) o->Px_data[offs + i] = o->Rw[i]; offs += NU; }

  /* the four OSQP calls, uprightmpc2.c:247-250 (every call resets the status:
   * reset_info, auxil.c:634-652) */
  osqp_update_bounds_(o); /* NB: on l>u it returns early WITHOUT touching data; the reference ignores the code */
  o->status_val = ST_UNSOLVED;
  osqp_update_lin_cost_(o);
  osqp_update_P_A_(o);
  int ret = osqp_solve_(o);

  for (int i = 0; i < NU; ++i) uquad[i] = o->sol_x[2 * NY * NN + i];
  o->T0 += uquad[0];
  uquad[0] = o->T0;
  for (int i = 0; i < NY; ++i) {
    dy1des[i] = o->sol_x[NY * NN + i];
    if (i < 3) dq1des[i] = dy1des[i];
  }
  mm(e3hR0T, o->e3h, R0, 3, 3, 3, (real)1.0, 0, 1);
  mm(&dq1des[3], e3hR0T, &dy1des[3], 3, 1, 3, (real)1.0, 0, 0);
  for (int i = 0; i < NY; ++i) accdes[i] = (dq1des[i] - dq0[i]) / o->dt;
  return ret;
}

/* ====================================================================== */
/* Introspection                                                           */
/* ====================================================================== */
const void *umpc_oracle_get(const umpc_oracle_t *o, const char *name, int *n, int *is_int) {
#define RET_R(nm, ptr, cnt) if (!strcmp(name, nm)) { *n = (cnt); *is_int = 0; return (ptr); }
#define RET_I(nm, ptr, cnt) if (!strcmp(name, nm)) { *n = (cnt); *is_int = 1; return (ptr); }
  RET_R("l_new", o->l_new, NC) RET_R("u_new", o->u_new, NC) RET_R("q_new", o->q_new, NX)
  RET_R("Px_data", o->Px_data, NX) RET_R("Ax_data", o->Ax_data, UMPC_NADATA)
  RET_I("Ax_idx", o->Ax_idx, UMPC_NADATA) RET_R("T0", &o->T0, 1)
  RET_I("A_p", o->A_p, NX + 1) RET_I("A_i", o->A_i, o->nnzA) RET_I("perm", o->perm, NK)
  RET_I("K_p", o->K_p, NK + 1) RET_I("K_i", o->K_i, o->nnzK)
  RET_I("PtoKKT", o->PtoKKT, NX) RET_I("AtoKKT", o->AtoKKT, o->nnzA) RET_I("rhotoKKT", o->rhotoKKT, NC)
  RET_I("etree", o->etree, NK) RET_I("Lnz", o->Lnz, NK) RET_I("L_p", o->L_p, NK + 1) RET_I("L_i", o->L_i, o->nnzL)
  RET_R("P_x", o->P_x, NX) RET_R("A_x", o->A_x, o->nnzA) RET_R("q", o->q, NX) RET_R("l", o->l, NC) RET_R("u", o->u, NC)
  RET_R("c", &o->c, 1) RET_R("cinv", &o->cinv, 1) RET_R("D", o->D, NX) RET_R("Dinv", o->Dinv, NX)
  RET_R("E", o->E, NC) RET_R("Einv", o->Einv, NC)
  RET_R("rho_vec", o->rho_vec, NC) RET_R("rho_inv_vec", o->rho_inv_vec, NC) RET_I("constr_type", o->constr_type, NC)
  RET_R("K_x", o->K_x, o->nnzK) RET_R("L_x", o->L_x, o->nnzL) RET_R("Dd", o->Dd, NK) RET_R("Ddinv", o->Ddinv, NK)
  RET_R("x", o->x, NX) RET_R("y", o->y, NC) RET_R("z", o->z, NC)
  RET_R("sol_x", o->sol_x, NX) RET_R("sol_y", o->sol_y, NC)
  RET_I("iter", &o->iter, 1) RET_I("status_val", &o->status_val, 1) RET_I("factor_ret", &o->factor_ret, 1)
  RET_R("pri_res", &o->pri_res, 1) RET_R("dua_res", &o->dua_res, 1) RET_R("obj_val", &o->obj_val, 1)
  RET_R("Eprev3", o->Eprev3, NN) RET_I("canonical", &o->canonical, 1)
  *n = 0; *is_int = 0;
  return NULL;
}
void umpc_oracle_set_iterates(umpc_oracle_t *o, const real *x, const real *y, const real *z) {
  memcpy(o->x, x, NX * sizeof(real)); memcpy(o->y, y, NC * sizeof(real)); memcpy(o->z, z, NC * sizeof(real));
}
void umpc_oracle_set_T0(umpc_oracle_t *o, real T0) { o->T0 = T0; }
void umpc_oracle_set_max_iter(umpc_oracle_t *o, int k) { o->max_iter = k; }
void umpc_oracle_set_canonical(umpc_oracle_t *o, int on, const real *Eprev3) {
  o->canonical = on;
  if (Eprev3) for (int k = 0; k < NN; ++k) o->Eprev3[k] = Eprev3[k];
}

/* wrench-linearisation controller state (functions further down; the coupled rollout needs the type) */
#define NDELU 4
#define NW 6
struct umpc_oracle_wl {
  real u0[NDELU], umin[NDELU], umax[NDELU], dumax[NDELU];
  real Qw[NW * NW];
  struct { real a0, a1[NDELU], A2[NDELU * NDELU]; } fa[NW];
};

/* ====================================================================== */
/* Plant + batched closed loop                                             */
/* ====================================================================== */
#define PR real
#define PFN(n) n##_r
#include "plant_impl.h"
#undef PR
#undef PFN
#define PR double
#define PFN(n) n##_d
#include "plant_impl.h"
#undef PR
#undef PFN

void umpc_oracle_plant_step(real p[3], real R[9], real dq[6], const real u[3],
                            real dt, const real Ib[3], real thrust_gain, int mode) {
  plant_step_r(p, R, dq, u, dt, Ib, thrust_gain, mode);
}
void umpc_oracle_plant_step_d(double p[3], double R[9], double dq[6],
                              const double u[3], double dt, const double Ib[3],
                              double thrust_gain, int mode) {
  plant_step_d(p, R, dq, u, dt, Ib, thrust_gain, mode);
}

/* Reference generators, template/flight_tasks.py:6-49 (same task ids / parameter order as
 * include/umpc_mi355x.h). r: (initialPos, -, -) in, (pdes, dpdes, sdes) out. */
void umpc_oracle_task_reference(int task, const real tp[4], real t, real r[9]) {
  if (task == 0) return;
  const real ip[3] = {r[0], r[1], r[2]};
  const double PI = 3.14159265358979323846;
  for (int i = 0; i < 3; ++i) { r[3 + i] = 0; r[6 + i] = i == 2 ? 1 : 0; }
  if (task == 1) {
    const real amp = tp[0], omg = (real)(2 * PI) * tp[1] * (real)1e-3;
    r[0] = ip[0] + amp * (real)sin((double)(omg * t));
    r[3] = amp * omg * (real)cos((double)(omg * t));
    if (tp[3] != 0) {
      r[1] = ip[1] + amp * ((real)1 - (real)cos((double)(omg * t)));
      r[4] = amp * omg * (real)sin((double)(omg * t));
    }
    if (amp > (real)1e-3) { r[2] = ip[2] + tp[2] * t; r[5] = tp[2]; }
  } else if (task == 2) {
    r[3] = t < tp[0] ? tp[1] : 0;
    r[0] = ip[0] + tp[1] * c_min(c_max(t, 0), tp[0]);
  } else if (task == 3) {
    const real ph = c_min(c_max((t - tp[0]) / tp[1], 0), 1);
    r[6] = -(real)sin((double)ph * 2 * PI); r[7] = 0; r[8] = (real)cos((double)ph * 2 * PI);
  } else if (task == 4) {
    r[0] = ip[0] + tp[3] * c_min(c_max(t, 0), tp[0]);
    r[3] = t < tp[0] ? tp[3] : 0;
    if (t < tp[0]) {
      const real ph = c_min(c_max((t - tp[2]) / tp[1], 0), 1);
      r[6] = -(real)sin((double)ph * PI); r[7] = 0; r[8] = (real)cos((double)ph * PI);
    } else { r[6] = -1; r[7] = 0; r[8] = 0; }
  }
}

void umpc_oracle_batch_rollout(const umpc_oracle_params_t *prm, const int *perm,
                               int B, int K, real *state, real *ctrl,
                               const real *ref, const real *Ib,
                               const real *thrust_gain, real *out, real *stats,
                               int *status, int nthreads) {
  umpc_oracle_batch_rollout2(prm, perm, B, K, state, ctrl, ref, Ib, thrust_gain, NULL, 0, NULL, 0, out, stats,
                             status, nthreads);
}

/* + per-robot weights [8][B] (ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom) and an on-line task */
void umpc_oracle_batch_rollout2(const umpc_oracle_params_t *prm, const int *perm,
                                int B, int K, real *state, real *ctrl,
                                const real *ref, const real *Ib,
                                const real *thrust_gain, const real *weights, int task,
                                const real *task_p, real t0, real *out, real *stats,
                                int *status, int nthreads) {
  umpc_oracle_batch_rollout3(prm, perm, B, K, state, ctrl, ref, Ib, thrust_gain, weights, task, task_p, t0, out,
                             stats, status, nthreads, NULL, NULL, NULL, NULL);
}

/* + the MPC -> WL -> actualT0 coupling (template/robobee_test_controllers.py:162-171, conn_MPC_WL.m:2-10):
 * wl = an initialised WL controller (parameters; its u0 is replaced per robot by wlu [4][B], in/out),
 * Mdiag = diag of M0 (template/ca6dynamics.py:5-10), wlw [6][B] or NULL = w0 of the last step. */
void umpc_oracle_batch_rollout3(const umpc_oracle_params_t *prm, const int *perm,
                                int B, int K, real *state, real *ctrl,
                                const real *ref, const real *Ib,
                                const real *thrust_gain, const real *weights, int task,
                                const real *task_p, real t0, real *out, real *stats,
                                int *status, int nthreads, const umpc_oracle_wl_t *wl, const real *Mdiag,
                                real *wlu, real *wlw) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  /* symbolic analysis once; cloned per thread */
  struct umpc_oracle *proto = (struct umpc_oracle *)malloc(sizeof(struct umpc_oracle));
  umpc_oracle_init(proto, prm->dt, prm->g, prm->TtoWmax, prm->ws, prm->wds, prm->wpr,
                   prm->wpf, prm->wvr, prm->wvf, prm->wthrust, prm->wmom, prm->Ib,
                   prm->maxIter, perm);
#pragma omp parallel
  {
    struct umpc_oracle *o = (struct umpc_oracle *)malloc(sizeof(struct umpc_oracle));
#pragma omp for schedule(static)
    for (int b = 0; b < B; ++b) {
      real ib[3] = {prm->Ib[0], prm->Ib[1], prm->Ib[2]};
      if (Ib) for (int i = 0; i < 3; ++i) ib[i] = Ib[i * (size_t)B + b];
      const real gain = thrust_gain ? thrust_gain[b] : (real)1;
      memcpy(o, proto, sizeof(*o));
      o->x = o->xa; o->x_prev = o->xb; o->z = o->za; o->z_prev = o->zb;
      if (Ib) { /* per-robot e3hIbi (uprightmpc2.c:50-56) */
        real Ibi[9] = {0};
        Ibi[0] = (real)1.0 / ib[0]; Ibi[4] = (real)1.0 / ib[1]; Ibi[8] = (real)1.0 / ib[2];
        mm(o->e3hIbi, o->e3h, Ibi, 3, 3, 3, (real)1.0, 0, 0);
      }
      o->canonical = 1;
      real p[3], R[9], dq[6], rf[9], uq[3] = {0, 0, 0}, acc[6] = {0};
      for (int i = 0; i < 3; ++i) p[i] = state[(size_t)i * B + b];
      for (int i = 0; i < 9; ++i) R[i] = state[(size_t)(3 + i) * B + b];
      for (int i = 0; i < 6; ++i) dq[i] = state[(size_t)(12 + i) * B + b];
      for (int i = 0; i < NX; ++i) o->x[i] = ctrl[(size_t)i * B + b];
      for (int i = 0; i < NC; ++i) o->y[i] = ctrl[(size_t)(NX + i) * B + b];
      for (int i = 0; i < NC; ++i) o->z[i] = ctrl[(size_t)(NX + NC + i) * B + b];
      o->T0 = ctrl[(size_t)(NX + 2 * NC) * B + b];
      for (int k = 0; k < NN; ++k) o->Eprev3[k] = ctrl[(size_t)(NX + 2 * NC + 1 + k) * B + b];
      if (weights) { /* umpcInit weight vectors, uprightmpc2.c:27-36 */
        const real ws = weights[b], wds = weights[(size_t)B + b], wpr = weights[(size_t)2 * B + b],
                   wpf = weights[(size_t)3 * B + b], wvr = weights[(size_t)4 * B + b],
                   wvf = weights[(size_t)5 * B + b];
        for (int i = 0; i < 3; ++i) {
          o->Qyr[i] = wpr; o->Qyf[i] = wpf; o->Qyr[3 + i] = o->Qyf[3 + i] = ws;
          o->Qdyr[i] = wvr; o->Qdyf[i] = wvf; o->Qdyr[3 + i] = o->Qdyf[3 + i] = wds;
        }
        o->Rw[0] = weights[(size_t)6 * B + b]; o->Rw[1] = o->Rw[2] = weights[(size_t)7 * B + b];
      }
      real s_err = stats ? stats[b] : 0, s_eff = stats ? stats[(size_t)B + b] : 0;
      struct umpc_oracle_wl wlr;
      real w0[6] = {0};
      if (wl) {
        memcpy(&wlr, wl, sizeof(wlr));
        for (int j = 0; j < NDELU; ++j) wlr.u0[j] = wlu[(size_t)j * B + b];
      }
      for (int k = 0; k < K; ++k) {
        for (int i = 0; i < 9; ++i) rf[i] = ref[(size_t)i * B + b];
        if (task) umpc_oracle_task_reference(task, task_p, t0 + (real)k * ((real)prm->nsub * prm->dtsim), rf);
        umpc_oracle_update(o, uq, acc, p, R, dq, &rf[0], &rf[3], &rf[6], (real)-1);
        if (wl) {
          /* accController, robobee_test_controllers.py:162-171: the NEXT update gets actualT0 = w0[2] / M0[2,2],
           * which replaces the accumulator when >= 0 (uprightmpc2.c:215-216) */
          real h0[6] = {0}, pd[6], u1[4];
          const real mbg = Mdiag[2] * prm->g;
          for (int c = 0; c < 3; ++c) h0[c] = R[2 + 3 * c] * mbg;
          for (int i = 0; i < 6; ++i) pd[i] = Mdiag[i] * acc[i];
          umpc_oracle_wl_update(&wlr, u1, w0, h0, pd);
          const real aT0 = w0[2] / Mdiag[2];
          if (aT0 >= 0) o->T0 = aT0;
        }
        real uc[3] = {uq[0], c_min(c_max(uq[1], -prm->taulim), prm->taulim),
                      c_min(c_max(uq[2], -prm->taulim), prm->taulim)};
        for (int s = 0; s < prm->nsub; ++s) {
          plant_step_r(p, R, dq, uc, prm->dtsim, ib, gain, prm->plant_mode);
          s_err += p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
          s_eff += uc[1] * uc[1] + uc[2] * uc[2];
        }
      }
      for (int i = 0; i < 3; ++i) state[(size_t)i * B + b] = p[i];
      for (int i = 0; i < 9; ++i) state[(size_t)(3 + i) * B + b] = R[i];
      for (int i = 0; i < 6; ++i) state[(size_t)(12 + i) * B + b] = dq[i];
      for (int i = 0; i < NX; ++i) ctrl[(size_t)i * B + b] = o->x[i];
      for (int i = 0; i < NC; ++i) ctrl[(size_t)(NX + i) * B + b] = o->y[i];
      for (int i = 0; i < NC; ++i) ctrl[(size_t)(NX + NC + i) * B + b] = o->z[i];
      ctrl[(size_t)(NX + 2 * NC) * B + b] = o->T0;
      for (int k = 0; k < NN; ++k) ctrl[(size_t)(NX + 2 * NC + 1 + k) * B + b] = o->Eprev3[k];
      if (out) {
        for (int i = 0; i < 3; ++i) out[(size_t)i * B + b] = uq[i];
        for (int i = 0; i < 6; ++i) out[(size_t)(3 + i) * B + b] = acc[i];
      }
      if (stats) { stats[b] = s_err; stats[(size_t)B + b] = s_eff; }
      if (status) status[b] = o->status_val;
      if (wl) {
        for (int j = 0; j < NDELU; ++j) wlu[(size_t)j * B + b] = wlr.u0[j];
        if (wlw) for (int i = 0; i < 6; ++i) wlw[(size_t)i * B + b] = w0[i];
      }
    }
    free(o);
  }
  free(proto);
}

/* ====================================================================== */
/* Reactive baseline                                         */
/* ====================================================================== */
/* reactiveController, template/template_controllers.py:282-296. k = (kpos0, kpos1, kz0, kz1, ks0, ks1). */
void umpc_oracle_reactive(const real p[3], const real R[9], const real dq[6], const real pdes[3], const real k[6],
                          real u[3]) {
  real sdes[3], ds[3], fT[3];
  for (int i = 0; i < 3; ++i) sdes[i] = c_min(c_max(k[0] * (pdes[i] - p[i]) - k[1] * dq[i], (real)-0.5), (real)0.5);
  sdes[2] = 1;
  for (int r = 0; r < 3; ++r) {
    ds[r] = -(R[r] * (-dq[4]) + R[r + 3] * dq[3]);
    fT[r] = k[4] * (R[6 + r] - sdes[r]) + k[5] * ds[r];
  }
  fT[2] = 0;
  const real wx = (R[0] * fT[0] + R[1] * fT[1]) + R[2] * fT[2];
  const real wy = (R[3] * fT[0] + R[4] * fT[1]) + R[5] * fT[2];
  u[0] = k[2] * (pdes[2] - p[2]) - k[3] * dq[2];
  u[1] = wy;
  u[2] = -wx;
}

/* controlTest(useMPC=False), template/uprightmpc2.py:121-151, for B robots: nsteps substeps, controller every
 * `every` substeps. log (or NULL) receives [nsteps][15] rows (p, s, dq, u) of robot `log_robot`. */
void umpc_oracle_reactive_rollout(const umpc_oracle_params_t *prm, int B, int nsteps, int every, real *state,
                                  const real *ref, const real *gains, const real *Ib, const real *thrust_gain,
                                  int task, const real *task_p, real t0, real *out, real *stats, real *log,
                                  int log_robot) {
  for (int b = 0; b < B; ++b) {
    real ib[3] = {prm->Ib[0], prm->Ib[1], prm->Ib[2]};
    if (Ib) for (int i = 0; i < 3; ++i) ib[i] = Ib[i * (size_t)B + b];
    const real gain = thrust_gain ? thrust_gain[b] : (real)1;
    real k[6] = {(real)5e-3, (real)5e-1, (real)1e-1, (real)1e0, (real)10e0, (real)1e2};
    if (gains) for (int i = 0; i < 6; ++i) k[i] = gains[(size_t)i * B + b];
    real p[3], R[9], dq[6], rf[9], u[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i) p[i] = state[(size_t)i * B + b];
    for (int i = 0; i < 9; ++i) R[i] = state[(size_t)(3 + i) * B + b];
    for (int i = 0; i < 6; ++i) dq[i] = state[(size_t)(12 + i) * B + b];
    real s_err = stats ? stats[b] : 0, s_eff = stats ? stats[(size_t)B + b] : 0;
    for (int ti = 0; ti < nsteps; ++ti) {
      if (ti % every == 0) {
        for (int i = 0; i < 9; ++i) rf[i] = ref[(size_t)i * B + b];
        if (task) umpc_oracle_task_reference(task, task_p, t0 + (real)ti * prm->dtsim, rf);
        umpc_oracle_reactive(p, R, dq, rf, k, u);
        u[1] = c_min(c_max(u[1], -prm->taulim), prm->taulim);
        u[2] = c_min(c_max(u[2], -prm->taulim), prm->taulim);
      }
      plant_step_r(p, R, dq, u, prm->dtsim, ib, gain, prm->plant_mode);
      s_err += p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
      s_eff += u[1] * u[1] + u[2] * u[2];
      if (log && b == log_robot) {
        real *row = log + (size_t)ti * 15;
        for (int i = 0; i < 3; ++i) { row[i] = p[i]; row[3 + i] = R[6 + i]; row[12 + i] = u[i]; }
        for (int i = 0; i < 6; ++i) row[6 + i] = dq[i];
      }
    }
    for (int i = 0; i < 3; ++i) state[(size_t)i * B + b] = p[i];
    for (int i = 0; i < 9; ++i) state[(size_t)(3 + i) * B + b] = R[i];
    for (int i = 0; i < 6; ++i) state[(size_t)(12 + i) * B + b] = dq[i];
    if (out) for (int i = 0; i < 3; ++i) out[(size_t)i * B + b] = u[i];
    if (stats) { stats[b] = s_err; stats[(size_t)B + b] = s_eff; }
  }
}

/* ====================================================================== */
/* Wrench-linearisation step: funapprox.c                    */
/* ====================================================================== */
size_t umpc_oracle_wl_sizeof(void) { return sizeof(struct umpc_oracle_wl); }

/* wlConInit + funApproxInit, funapprox.c:35-51, 102-116 */
void umpc_oracle_wl_init(umpc_oracle_wl_t *wl, const real u0[4], const real umin[4], const real umax[4],
                         const real dumax[4], const real Qw[6], real controlRate, const real popts[90]) {
  memset(wl, 0, sizeof(*wl));
  for (int i = 0; i < NDELU; ++i) {
    wl->u0[i] = u0[i]; wl->umin[i] = umin[i]; wl->umax[i] = umax[i];
    wl->dumax[i] = dumax[i] / controlRate;
  }
  for (int i = 0; i < NW; ++i) {
    const real *p = &popts[15 * i];
    wl->fa[i].a0 = p[0];
    memcpy(wl->fa[i].a1, &p[1], NDELU * sizeof(real));
    int kk = 0;
    for (int r = 0; r < NDELU; ++r)
      for (int c = r; c < NDELU; ++c) {
        wl->fa[i].A2[r + NDELU * c] = wl->fa[i].A2[c + NDELU * r] = p[NDELU + 1 + kk];
        kk++;
      }
    wl->Qw[i + NW * i] = Qw[i];
  }
}

/* wlConUpdate, funapprox.c:118-165 (funApproxF :53-65, funApproxDf :67-76) */
void umpc_oracle_wl_update(umpc_oracle_wl_t *wl, real u1[4], real w0[6], const real h0[6], const real pdotdes[6]) {
  real A1[NW * NDELU], a0[NW], delu[NDELU], dum[NW * NDELU], L[NDELU], U[NDELU], vout[NDELU], fout;
  for (int i = 0; i < NW; ++i) { /* wrenchMap */
    real res = wl->fa[i].a0;
    mm(vout, wl->u0, wl->fa[i].a1, 1, 1, NDELU, (real)1.0, 0, 0);
    res += vout[0];
    mm(vout, wl->fa[i].A2, wl->u0, NDELU, 1, NDELU, (real)1.0, 0, 0);
    mm(&fout, wl->u0, vout, 1, 1, NDELU, (real)0.5, 0, 0);
    w0[i] = res + fout;
  }
  for (int i = 0; i < NW; ++i) { /* wrenchJacMap */
    real Df[NDELU];
    memcpy(Df, wl->fa[i].a1, NDELU * sizeof(real));
    mm(vout, wl->fa[i].A2, wl->u0, NDELU, 1, NDELU, (real)1.0, 0, 0);
    for (int j = 0; j < NDELU; ++j) { Df[j] += vout[j]; A1[i + NW * j] = Df[j]; }
  }
  for (int i = 0; i < NW; ++i) a0[i] = w0[i] - h0[i] - pdotdes[i];
  for (int i = 0; i < NDELU; ++i) { L[i] = -wl->dumax[i]; U[i] = wl->dumax[i]; }
  for (int i = 0; i < NDELU; ++i) {
    if (wl->u0[i] < wl->umin[i]) L[i] = 0;
    else if (wl->u0[i] > wl->umax[i]) U[i] = 0;
  }
  mm(dum, wl->Qw, a0, NW, 1, NW, (real)1.0, 0, 0);
  mm(delu, A1, dum, NDELU, 1, NW, (real)-1e3, 1, 0);
  for (int i = 0; i < NDELU; ++i) {
    if (delu[i] < L[i]) delu[i] = L[i];
    else if (delu[i] > U[i]) delu[i] = U[i];
  }
  for (int i = 0; i < NDELU; ++i) { wl->u0[i] += delu[i]; u1[i] = wl->u0[i]; }
}
void umpc_oracle_wl_set_u0(umpc_oracle_wl_t *wl, const real u0[4]) { memcpy(wl->u0, u0, sizeof(wl->u0)); }

/* ====================================================================== */
/* a19 / a20: the reference's other vector fields + build-defined RK4      */
/* ====================================================================== */
static real r_sin(real v) { return (real)sin((double)v); }
static real r_cos(real v) { return (real)cos((double)v); }

/* template/ca6dynamics.py:35-50; y = (p, R col-major, dq=(v_world, omega_body)); aux = ydot[18] w[6] h[6] */
static void ca6_vf(const real *y, const real *u, real *yd, real *w, real *h) {
  const real ycp = 10, mb = 100, g = (real)9.81e-3, I[3] = {3333, 3333, 1000};
  const real *R = &y[3];
  w[0] = u[2] + u[5]; w[1] = 0; w[2] = u[0] + u[3];
  w[3] = (u[0] - u[3]) * ycp; w[4] = -u[0] * u[1] - u[3] * u[4]; w[5] = (-u[2] + u[5]) * ycp;
  for (int c = 0; c < 3; ++c) { h[c] = R[2 + 3 * c] * (mb * g); h[3 + c] = 0; }
  real ab[3];
  for (int c = 0; c < 3; ++c) ab[c] = (w[c] - h[c]) / mb;
  for (int i = 0; i < 3; ++i) yd[i] = y[12 + i];
  const real wx = y[15], wy = y[16], wz = y[17];
  for (int r = 0; r < 3; ++r) {
    yd[3 + r + 0] = R[r + 3] * wz - R[r + 6] * wy;
    yd[3 + r + 3] = -R[r + 0] * wz + R[r + 6] * wx;
    yd[3 + r + 6] = R[r + 0] * wy - R[r + 3] * wx;
    yd[12 + r] = (R[r] * ab[0] + R[r + 3] * ab[1]) + R[r + 6] * ab[2];
    yd[15 + r] = w[3 + r] / I[r];
  }
}

/* ThrustStrokeDev.dynamics, template/FlappingModels3D.py:19-38 (as written) */
static void tsd_vf(const real *y, const real *u, real *yd) {
  const real m = (real)0.5, g = (real)9.81, ycp = (real)0.5, Ib[3] = {(real)5e-4, (real)5e-4, (real)1e-3};
  const real ax = y[3], ay = y[4], az = y[5], t = ax * ax + ay * ay + az * az;
  real a, b, e[3][3];
  if (t < (real)1e-2) {
    a = (real)1 - t * ((real)1 / 6 - t * ((real)1 / 120 - t * ((real)1 / 5040 - t * ((real)1 / 362880))));
    b = (real)0.5 - t * ((real)1 / 24 - t * ((real)1 / 720 - t * ((real)1 / 40320 - t * ((real)1 / 3628800))));
  } else {
    const real th = r_sqrt(t);
    a = r_sin(th) / th; b = ((real)1 - r_cos(th)) / t;
  }
  e[0][0] = (real)1 - b * (ay * ay + az * az); e[1][1] = (real)1 - b * (ax * ax + az * az);
  e[2][2] = (real)1 - b * (ax * ax + ay * ay);
  e[0][1] = -a * az + b * ax * ay; e[1][0] = a * az + b * ax * ay;
  e[0][2] = a * ay + b * ax * az;  e[2][0] = -a * ay + b * ax * az;
  e[1][2] = -a * ax + b * ay * az; e[2][1] = a * ax + b * ay * az;
  const real om[3] = {y[9], y[10], y[11]}, Fz = u[0] + u[2];
  const real tq[3] = {ycp * u[0] + (-ycp) * u[2], -(u[1] * u[0]) - (u[3] * u[2]), 0};
  const real Iw[3] = {Ib[0] * om[0], Ib[1] * om[1], Ib[2] * om[2]};
  const real cr[3] = {om[1] * Iw[2] - om[2] * Iw[1], om[2] * Iw[0] - om[0] * Iw[2], om[0] * Iw[1] - om[1] * Iw[0]};
  real ob[3];
  for (int i = 0; i < 3; ++i) ob[i] = (tq[i] - cr[i]) / Ib[i];
  for (int i = 0; i < 6; ++i) yd[i] = y[6 + i];
  for (int i = 0; i < 3; ++i) {
    yd[6 + i] = ((i == 2 ? -m * g : (real)0) + e[i][2] * Fz) / m;
    yd[9 + i] = (e[0][i] * ob[0] + e[1][i] * ob[1]) + e[2][i] * ob[2];
  }
}

static void model_vf(int model, const real *y, const real *u, real *yd) {
  real w[6], h[6];
  if (model == 0) ca6_vf(y, u, yd, w, h); else tsd_vf(y, u, yd);
}

/* one robot: nsub == 0 -> aux = ydot (ca6: + w, h), else y advanced by nsub RK4 steps */
void umpc_oracle_model(int model, int nsub, real dt, real *y, const real *u, real *aux) {
  const int ny = model == 0 ? 18 : 12;
  if (nsub == 0) {
    if (model == 0) ca6_vf(y, u, aux, aux + 18, aux + 24); else tsd_vf(y, u, aux);
    return;
  }
  for (int s = 0; s < nsub; ++s) {
    real y0[18], ys[18], acc[18], k[18];
    const real cs[4] = {0, (real)0.5, (real)0.5, 1}, wt[4] = {1, 2, 2, 1};
    for (int i = 0; i < ny; ++i) y0[i] = y[i];
    for (int st = 0; st < 4; ++st) {
      for (int i = 0; i < ny; ++i) ys[i] = st ? y0[i] + cs[st] * dt * k[i] : y0[i];
      model_vf(model, ys, u, k);
      for (int i = 0; i < ny; ++i) acc[i] = st ? acc[i] + wt[st] * k[i] : k[i];
    }
    for (int i = 0; i < ny; ++i) y[i] = y0[i] + dt * acc[i] / (real)6;
  }
}
