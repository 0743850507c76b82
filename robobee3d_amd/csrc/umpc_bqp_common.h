// Shared by the translation units of the general-structure batch QP solver (umpc_bqp.hip and the generated
// straight-line specialisations csrc/gen/bqp_*.hip): constants of the embedded OSQP step, the kernel argument block
// and the scalar helpers. Internal header, not part of the C ABI.
#ifndef UMPC_BQP_COMMON_H
#define UMPC_BQP_COMMON_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <limits>

namespace umpcqp {

constexpr double QP_INFTY = 1e30, QP_MIN_SCALING = 1e-4, QP_MAX_SCALING = 1e4;
constexpr double QP_RHO_MIN = 1e-6, QP_RHO_TOL = 1e-4, QP_RHO_EQ_OVER_RHO_INEQ = 1e3;
constexpr int HEADER_WORDS = 64;
// header word indices (qpstruct.py)
enum { H_N, H_M, H_NK, H_NNZP, H_NNZA, H_NNZL, H_NROWS, H_NLEV, H_TAB0 };
enum { T_PINV, T_PIDX, T_AP, T_AI, T_ARP, T_ARJ, T_ARK, T_FIP, T_FIB, T_FIS, T_FEP, T_FEC, T_FEN, T_LP, T_LI, T_LRP,
       T_LRJ, T_LRK,
       // wave-per-robot kernel: permutation, column of every A entry, level schedules, right-looking factor terms
       T_PERM, T_AJ, T_LEVP, T_LEVN, T_ELEVP, T_ELEVE, T_LKSRC, T_LCOL, T_FTP, T_FTA, T_FTB, T_FTJ, T_FDP, T_FDA, T_FDJ,
       T_COUNT };
enum { R_PS, R_AS, R_QS, R_LS, R_US, R_D, R_E, R_DT, R_ET, R_RHO, R_RINV, R_KD, R_LX, R_DI, R_YV, R_WV, R_XP, R_DY,
       R_T1, R_T2, R_T3, R_SC, R_COUNT };

template <typename T>
struct QPArgs {
  const int32_t *tab;
  int B;
  T *W;
  const T *Pv, *Av, *q, *l, *u;
  T *x, *y, *z, *Eprev, *sol_x, *sol_y;
  int32_t *status;
  T *info;
  T sigma, alpha, rho, eps_abs, eps_rel, eps_pinf, eps_dinf;
  int max_iter, scaling;
  int adaptive_rho_interval;  // 0: fixed rho; k > 0: adapt_rho every k iterations (table kernel)
  int check_termination;  // 0: exactly max_iter iterations; k > 0: exact termination test every k iterations (table kernel)
  int asm_ok;  // an assembly specialisation (gen/bqp_*_asm.h) may run: S is allocated
  T *S;        // its stream buffer: [wave][item][lane], BQP_ASM_STREAM_ITEMS_PER_WAVE (umpc_bqp_registry.h) items per wave: the
               // loop's stream first, the residual stream behind it; allocated behind the workspace rows
  T oma, rinv_eq;  // 1 - alpha and 1 / rho_eq as the kernels compute them, evaluated on the host (assembly operands)
  T rinv0, rho_eq;  // 1 / rho and rho_eq likewise
  // Planar-p5f tick fused into the p5f10 assembly kernel (umpcP5fTick; null tick_y = a plain QP solve): before the QP blocks
  // the workgroup evaluates getLin of its 64 robots at (tick_u, y[0], y[3]), writes lin, rewrites the state-dependent entries
  // of A (Av[k] = tick_cst[k] * lin[tick_src[k]] where tick_src[k] >= 0) and advances the plant, y += (Ad y + Bd u) dt
  T *tick_y = nullptr, *tick_lin = nullptr;
  const T *tick_cst = nullptr;
  const int32_t *tick_src = nullptr;
  int tick_nnz = 0;
  T tick_u = T(0), tick_dt = T(0);
};

template <typename T> __device__ __forceinline__ T qabs(T v) { return v < T(0) ? -v : v; }
template <> __device__ __forceinline__ float qabs<float>(float v) { return __builtin_fabsf(v); }
template <> __device__ __forceinline__ double qabs<double>(double v) { return __builtin_fabs(v); }
// c_max / c_min of the reference: (a > b) ? a : b
template <typename T> __device__ __forceinline__ T qmax(T a, T b) { return a > b ? a : b; }
template <typename T> __device__ __forceinline__ T qmin(T a, T b) { return a < b ? a : b; }
__device__ __forceinline__ float qsqrt(float v) { return __fsqrt_rn(v); }
__device__ __forceinline__ double qsqrt(double v) { return __dsqrt_rn(v); }
// limit_scaling, scaling.c:7-14 (comparisons in double)
template <typename T> __device__ __forceinline__ T limit_scaling(T v) {
  v = (double)v < QP_MIN_SCALING ? T(1.0) : v;
  v = (double)v > QP_MAX_SCALING ? T(QP_MAX_SCALING) : v;
  return v;
}

// update_rho_vec, auxil.c:103-145 (ls, us: bounds scaled by the previous E; comparisons in double)
template <typename T>
__device__ __forceinline__ void qp_classify(T ls, T us, T rho0, T rho_eq, T &r, T &ri) {
  if (((double)ls < -QP_INFTY * QP_MIN_SCALING) && ((double)us > QP_INFTY * QP_MIN_SCALING)) {
    r = T(QP_RHO_MIN); ri = T(1. / QP_RHO_MIN);
  } else if ((double)(us - ls) < QP_RHO_TOL) {
    r = rho_eq; ri = T(1. / (double)rho_eq);
  } else {
    r = rho0; ri = T(1. / (double)rho0);
  }
}

// getLin, planar/mpc_osqp_p5f.py:45-85: (u, sigma, phi) -> the state-dependent entries of (Ad, Bd):
// out rows = Ad[4][3], Ad[5][3], Bd[4], Bd[5], Bd[6]. Constants of the module (:33-43).
template <typename T>
__device__ __forceinline__ void p5f_getlin(T u, T sigma, T phi, T o[5]) {
  const T CDmax = T(3.4), CLmax = T(1.8), CD0 = T(0.4), khinge0 = T(0.1), mb = T(100), kaero = T(1), d = T(5);
  const T ib = T(1.) / T(12.) * mb * T(144);
  const T kh = u < T(0) ? -khinge0 : khinge0;
  const T uu = u * u;
  const T ang = T(2) * kh * uu;
  const T s2 = std::sin(ang), c2 = std::cos(ang), sp = std::sin(phi), cp = std::cos(phi);
  const T CDs = CD0 + CDmax, CDd = CD0 - CDmax;
  o[0] = -(kaero * u * (T(2) * CLmax * cp * s2 + (CDs + CDd * c2) * sp)) / (T(2.) * mb);
  o[1] = (kaero * u * ((CDs + CDd * c2) * cp - T(2) * CLmax * s2 * sp)) / (T(2.) * mb);
  o[2] = (kaero * (cp * (CDs + CDd * c2 - T(4) * CDd * kh * uu * s2) -
                   T(2) * CLmax * (T(4) * kh * uu * c2 + s2) * sp)) / (T(2.) * mb);
  o[3] = (kaero * (T(2) * CLmax * cp * s2 + (CDs - T(4) * CDd * kh * uu * s2) * sp +
                   c2 * (T(8) * CLmax * kh * uu * cp + CDd * sp))) / (T(2.) * mb);
  o[4] = (kaero * (-(d * (CDs + CDd * c2 - T(4) * CDd * kh * uu * s2)) +
                   T(2) * CLmax * (T(4) * kh * uu * c2 + s2) * sigma)) / (T(2.) * ib);
}

// the reference's plant tick (:176) from the getLin values: y += (Ad y + Bd u) dt with
// Ad rows 1: y4, 2: y5, 3: y6, 4: Ad43 y3, 5: Ad53 y3; Bd = (1, 0, 0, 0, Bd4, Bd5, Bd6)
template <typename T>
__device__ __forceinline__ void p5f_plant_tick(const T o[5], T ub, T dt, T yy[7]) {
  const T dy[7] = {ub, yy[4], yy[5], yy[6], o[0] * yy[3] + o[2] * ub, o[1] * yy[3] + o[3] * ub, o[4] * ub};
  for (int i = 0; i < 7; ++i) yy[i] = yy[i] + dy[i] * dt;
}

}  // namespace umpcqp
#endif
