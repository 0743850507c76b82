// MI355X (gfx950) device code for one uprightmpc2 closed-loop step.
//
// Mapping: ONE LANE = ONE ROBOT. A 64-lane wavefront carries 64 independent
// robots; there is no cross-lane communication anywhere on the path, every
// sparse index is a literal baked in by codegen.py (umpc_gen.h), and all HBM
// traffic is SoA [row][B] so that a wave load/store of one row is one
// contiguous 256-byte (fp32) segment.
//
// What one step computes (reference file:line, template/uprightmpc2/...):
//   assembly                  uprightmpc2.c:209-245, 126-207
//   constraint classification osqp.c:784-833 -> auxil.c:103-145 (update_rho_vec)
//   Ruiz equilibration x10    scaling.c:44-156
//   KKT fill + LDL'           kkt.c:184-222, qdldl.c:86-247
//   ADMM x maxIter            osqp.c:354-370, auxil.c:164-228, qdldl_interface.c:322-369
//   residuals + status        auxil.c:243-362, 517-565, 684-789; osqp.c:524-573
//   extraction                uprightmpc2.c:253-269
//   plant substeps            template/genqp.py:24-41, template/uprightmpc2.py:148-151
//
// Numerics: every scalar is T (float for the reference's DFLOAT build). FMAs
// are used where the reference has a multiply followed by an add on the sparse
// kernels (explicit UMPC_FMA); everything else is compiled with
// -ffp-contract=off. The problem data is rebuilt from raw values on every step
// (the reference un-scales and re-scales its persistent copy,
// osqp.c:1211-1248, which is the same up to round-off).
#pragma once
#include <hip/hip_runtime.h>

#include "umpc_gen.h"
#include "umpc_admm_asm.h"
#include "umpc_admm_asm64.h"
#include "umpc_admm_asm64_quad.h"   // the fp64 ADMM phase with one robot per lane quad (asmquad64.py)
#include "umpc_quad64_tab.h"

namespace umpc {

using namespace umpcgen;

template <typename T>
struct DevParams {
  T dt, g, Tmax;
  T wpr, wpf, ws, wvr, wvf, wds, wthrust, wmom;
  T Ib[3];
  T dtsim, taulim;
  int maxIter, nsub, plant_mode;
  int task;       // 0: ref rows are (pdes, dpdes, sdes); 1 helix, 2 straightAcc, 3 flip, 4 perch (flight_tasks.py)
  T task_p[4];    // task parameters, see task_reference()
};

// objective weights of one robot (createMPC arguments, template_controllers.py:260); batch-constant unless
// the caller supplies a per-robot table (gain-tuning sweeps, uprightmpc2.py:272-303)
template <typename T>
struct Weights {
  T ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom;
};

__device__ __forceinline__ float umpc_sin(float v);
__device__ __forceinline__ double umpc_sin(double v);
__device__ __forceinline__ float umpc_cos(float v);
__device__ __forceinline__ double umpc_cos(double v);
__device__ __forceinline__ float umpc_max(float a, float b);
__device__ __forceinline__ float umpc_min(float a, float b);
__device__ __forceinline__ double umpc_max(double a, double b);
__device__ __forceinline__ double umpc_min(double a, double b);

// Reference generators of template/flight_tasks.py:6-49, evaluated on device at the MPC fire time t (ms).
// `r` holds (initialPos, -, -) on entry for task != 0 and (pdes, dpdes, sdes) on exit.
//   1 helix(trajAmp, trajFreq[Hz], dz, useY)   :6-20     2 straightAcc(tduration, vdes)            :22-29
//   3 flip(tstart, tend)                       :31-36    4 perch(tend, trotstart, trotend, vdes)   :38-49
template <typename T>
__device__ __forceinline__ void task_reference(int task, const T (&tp)[4], T t, T (&r)[9]) {
  if (task == 0) return;
  const T ip[3] = {r[0], r[1], r[2]};
  const T twopi = T(6.283185307179586476925286766559);
  const T pi = T(3.1415926535897932384626433832795);
#pragma unroll
  for (int i = 0; i < 3; ++i) { r[3 + i] = T(0); r[6 + i] = i == 2 ? T(1) : T(0); }
  if (task == 1) {
    const T amp = tp[0], omg = twopi * tp[1] * T(1e-3);
    r[0] = ip[0] + amp * umpc_sin(omg * t);
    r[3] = amp * omg * umpc_cos(omg * t);
    if (tp[3] != T(0)) {
      r[1] = ip[1] + amp * (T(1) - umpc_cos(omg * t));
      r[4] = amp * omg * umpc_sin(omg * t);
    }
    if (amp > T(1e-3)) { r[2] = ip[2] + tp[2] * t; r[5] = tp[2]; }
  } else if (task == 2) {
    const T tdur = tp[0], vdes = tp[1];
    r[3] = t < tdur ? vdes : T(0);
    r[0] = ip[0] + vdes * umpc_min(umpc_max(t, T(0)), tdur);
  } else if (task == 3) {
    const T ph = umpc_min(umpc_max((t - tp[0]) / tp[1], T(0)), T(1));
    r[6] = -umpc_sin(ph * twopi); r[7] = T(0); r[8] = umpc_cos(ph * twopi);
  } else if (task == 4) {
    const T tend = tp[0], trotstart = tp[1], trotend = tp[2], vdes = tp[3];
    r[0] = ip[0] + vdes * umpc_min(umpc_max(t, T(0)), tend);
    r[3] = t < tend ? vdes : T(0);
    if (t < tend) {
      const T ph = umpc_min(umpc_max((t - trotend) / trotstart, T(0)), T(1));
      r[6] = -umpc_sin(ph * pi); r[7] = T(0); r[8] = umpc_cos(ph * pi);
    } else {
      r[6] = T(-1); r[7] = T(0); r[8] = T(0);
    }
  }
}

// OSQP constants (template/uprightmpc2/constants.h:59-110, workspace.c:561)
#define RHO_EQ 100.0   /* RHO_EQ_OVER_RHO_INEQ * rho, rounded to T */
#define RINV_EQ 0.01   /* 1 / RHO_EQ */
#define UMPC_RHO 0.1
#define UMPC_RHO_MIN 1e-6
#define UMPC_RHO_TOL 1e-4
#define UMPC_MIN_SCALING 1e-4
#define UMPC_MAX_SCALING 1e4
#define UMPC_INFTY 1e30
#ifndef UMPC_SCALING_ITERS   /* tools/build_variant.py overrides it for phase timing only */
#define UMPC_SCALING_ITERS 10
#endif

enum { ST_SOLVED = 1, ST_SOLVED_INACC = 2, ST_PINF_INACC = 3, ST_DINF_INACC = 4,
       ST_MAX_ITER = -2, ST_PINF = -3, ST_DINF = -4, ST_NON_CVX = -7, ST_UNSOLVED = -10 };

// |v| as a source modifier (c_absval of glob_opts.h:91; identical for every non-NaN value)
__device__ __forceinline__ float umpc_abs(float v) { return __builtin_fabsf(v); }
__device__ __forceinline__ double umpc_abs(double v) { return __builtin_fabs(v); }
// max/min of non-NaN values (c_max / c_min of glob_opts.h:95-99): one v_max / v_min
__device__ __forceinline__ float umpc_max(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float umpc_min(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double umpc_max(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ double umpc_min(double a, double b) { return __builtin_fmin(a, b); }
// 1/sqrt(v): the reference does sqrtf then 1.0f/ (two roundings, scaling.c:98-103). fp32 uses the
// hardware v_rsq_f32 (1 ulp, one quarter-rate instruction instead of ~20). fp64: v_rsq_f64 (~2^-24) + two Newton
// steps (error squares each step -> rounding level, within ~2 ulp of the reference's two-rounding value) in 9
// instructions where the correctly rounded sqrt followed by the IEEE divide expands to ~35; the Ruiz passes take
// 840 of them per step. Arguments are limit_scaling()'d: finite, in [1e-4, 1e4].
__device__ __forceinline__ float umpc_rsqrt(float v) { return __builtin_amdgcn_rsqf(v); }
__device__ __forceinline__ double umpc_rsqrt(double v) {
  double y = __builtin_amdgcn_rsq(v);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const double e = __builtin_fma(-(v * y), y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
  }
  return y;
}
// 1/v for finite, normal, nonzero v: fp32 IEEE divide; fp64 v_rcp_f64 + two Newton steps (5 instructions against ~25)
__device__ __forceinline__ float umpc_recip(float v) { return 1.0f / v; }
__device__ __forceinline__ double umpc_recip(double v) {
  double y = __builtin_amdgcn_rcp(v);
#pragma unroll
  for (int k = 0; k < 2; ++k) y = __builtin_fma(y, __builtin_fma(-v, y, 1.0), y);
  return y;
}
// 1/v where only residual NORMS consume the result
__device__ __forceinline__ float umpc_rcp_fast(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ double umpc_rcp_fast(double v) { return umpc_recip(v); }
__device__ __forceinline__ float umpc_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double umpc_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float umpc_sqrt(float v) { return __fsqrt_rn(v); }
__device__ __forceinline__ float umpc_sqrt_fast(float v) { return __builtin_amdgcn_sqrtf(v); }   // v_sqrt_f32, 1 ulp
__device__ __forceinline__ double umpc_sqrt_fast(double v) { return __dsqrt_rn(v); }
__device__ __forceinline__ double umpc_sqrt(double v) { return __dsqrt_rn(v); }
__device__ __forceinline__ float umpc_sin(float v) { return sinf(v); }
__device__ __forceinline__ double umpc_sin(double v) { return sin(v); }
__device__ __forceinline__ float umpc_cos(float v) { return cosf(v); }
__device__ __forceinline__ double umpc_cos(double v) { return cos(v); }

// limit_scaling, scaling.c:7-14
template <typename T> __device__ __forceinline__ T limit_scaling(T v) {
  return v < T(UMPC_MIN_SCALING) ? T(1) : umpc_min(v, T(UMPC_MAX_SCALING));
}

// ---------------------------------------------------------------------------
// plant: template/genqp.py:24-41 (R column-major, dq = (v_world, omega_body))
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void plant_vf(const T (&R)[9], const T (&dq)[6], const T (&u)[3],
                                         const T (&Ib)[3], const T (&Ibinv)[3], T gain, T (&ddq)[6]) {
  // the library is built with -ffp-contract=off (the QP phases place their FMAs explicitly); the plant is plain
  // arithmetic where a fused multiply-add only removes a rounding
#pragma clang fp contract(fast)
  const T wx = dq[3], wy = dq[4], wz = dq[5];
  const T Th = gain * u[0];
  ddq[0] = Th * R[6];
  ddq[1] = Th * R[7];
  ddq[2] = Th * R[8] - T(9.81e-3);
  const T hx = Ib[0] * wx, hy = Ib[1] * wy, hz = Ib[2] * wz;
  const T cx = wy * hz - wz * hy;
  const T cy = wz * hx - wx * hz;
  const T cz = wx * hy - wy * hx;
  // np.linalg.inv(Ib) @ (...), genqp.py:28: the reference multiplies by the inverse too
  ddq[3] = (-cx + u[1]) * Ibinv[0];
  ddq[4] = (-cy + u[2]) * Ibinv[1];
  ddq[5] = (-cz) * Ibinv[2];
}

// R <- R expm(skew(w) h): Rodrigues form of scipy.linalg.expm(skew(w) dt), genqp.py:39
template <typename T>
__device__ __forceinline__ void plant_rot(T (&R)[9], T w0, T w1, T w2, T h) {
  // the library is built with -ffp-contract=off (the QP phases place their FMAs explicitly); the plant is plain
  // arithmetic where a fused multiply-add only removes a rounding
#pragma clang fp contract(fast)
  const T ax = w0 * h, ay = w1 * h, az = w2 * h;
  const T t = ax * ax + ay * ay + az * az;
  T a, b;
  if (t < T(1e-2)) {
    a = T(1) - t * (T(1) / 6 - t * (T(1) / 120 - t * (T(1) / 5040 - t * (T(1) / 362880))));
    b = T(0.5) - t * (T(1) / 24 - t * (T(1) / 720 - t * (T(1) / 40320 - t * (T(1) / 3628800))));
  } else {
    const T th = umpc_sqrt(t);
    a = umpc_sin(th) / th;
    b = (T(1) - umpc_cos(th)) / t;
  }
  T e[3][3];
  e[0][0] = T(1) - b * (ay * ay + az * az);
  e[1][1] = T(1) - b * (ax * ax + az * az);
  e[2][2] = T(1) - b * (ax * ax + ay * ay);
  e[0][1] = -a * az + b * ax * ay;
  e[1][0] = a * az + b * ax * ay;
  e[0][2] = a * ay + b * ax * az;
  e[2][0] = -a * ay + b * ax * az;
  e[1][2] = -a * ax + b * ay * az;
  e[2][1] = a * ax + b * ay * az;
  T Rn[9];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) Rn[r + 3 * c] = R[r] * e[0][c] + R[r + 3] * e[1][c] + R[r + 6] * e[2][c];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = Rn[i];
}

template <typename T>
__device__ __forceinline__ void plant_step(T (&p)[3], T (&R)[9], T (&dq)[6], const T (&u)[3], T dt,
                                           const T (&Ib)[3], const T (&Ibinv)[3], T gain, int mode) {
  // the library is built with -ffp-contract=off (the QP phases place their FMAs explicitly); the plant is plain
  // arithmetic where a fused multiply-add only removes a rounding
#pragma clang fp contract(fast)
  if (mode == 0) {
    T ddq[6];
    plant_vf(R, dq, u, Ib, Ibinv, gain, ddq);
#pragma unroll
    for (int i = 0; i < 3; ++i) p[i] = p[i] + dt * dq[i];
    plant_rot(R, dq[3], dq[4], dq[5], dt);
#pragma unroll
    for (int i = 0; i < 6; ++i) dq[i] = dq[i] + dt * ddq[i];
  } else {
    // build-defined classical RK4 on y = (p, R, dq), dR/dt = R skew(w)
    T yv0[18], ys[18], acc[18], k[18];
#pragma unroll
    for (int i = 0; i < 3; ++i) yv0[i] = p[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) yv0[3 + i] = R[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) yv0[12 + i] = dq[i];
    const T cs[4] = {T(0), T(0.5), T(0.5), T(1)};
    const T wt[4] = {T(1), T(2), T(2), T(1)};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < 18; ++i) ys[i] = s ? yv0[i] + cs[s] * dt * k[i] : yv0[i];
      T Rs[9], dqs[6], dd[6];
#pragma unroll
      for (int i = 0; i < 9; ++i) Rs[i] = ys[3 + i];
#pragma unroll
      for (int i = 0; i < 6; ++i) dqs[i] = ys[12 + i];
#pragma unroll
      for (int i = 0; i < 3; ++i) k[i] = dqs[i];
      const T wx = dqs[3], wy = dqs[4], wz = dqs[5];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        k[3 + r + 0] = Rs[r + 3] * wz - Rs[r + 6] * wy;
        k[3 + r + 3] = -Rs[r + 0] * wz + Rs[r + 6] * wx;
        k[3 + r + 6] = Rs[r + 0] * wy - Rs[r + 3] * wx;
      }
      plant_vf(Rs, dqs, u, Ib, Ibinv, gain, dd);
#pragma unroll
      for (int i = 0; i < 6; ++i) k[12 + i] = dd[i];
#pragma unroll
      for (int i = 0; i < 18; ++i) acc[i] = s ? acc[i] + wt[s] * k[i] : k[i];
    }
    // (k1 + 2 k2 + 2 k3 + k4) associated as in the oracle; the final dt/6 is one multiply (no per-word division)
    const T dt6 = dt / T(6);
#pragma unroll
    for (int i = 0; i < 18; ++i) ys[i] = yv0[i] + dt6 * acc[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) p[i] = ys[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = ys[3 + i];
#pragma unroll
    for (int i = 0; i < 6; ++i) dq[i] = ys[12 + i];
  }
}

// ---------------------------------------------------------------------------
// QP assembly: uprightmpc2.c:209-245 (umpcUpdate head), 126-180, 182-207
// ---------------------------------------------------------------------------
template <typename T>
struct RawQP {
  T l[NC], u3[N];  // u == l on the NEQ dynamics rows
  T q[NX], Px[NX];
  T dtT0, s0dt[3], Btaudt[6];
};

template <typename T>
__device__ __forceinline__ void assemble(const DevParams<T> &prm, const Weights<T> &wt, const T (&Ibi)[3], T T0,
                                         const T (&p0)[3], const T (&R0)[9], const T (&dq0)[6], const T (&ref)[9],
                                         RawQP<T> &qp) {
  const T dt = prm.dt;
  T s0[3], ds0[3], Btau[6], yv0[NY], dy0[NY], yv1[NY], ydes[NY], dydes[NY];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    s0[r] = R0[6 + r];
    // ds0 = -R0 e3h w,  e3h w = (-wy, wx, 0)
    ds0[r] = -(R0[r] * (-dq0[4]) + R0[r + 3] * dq0[3]);
    // Btau = (-R0 e3h Ib^-1)[:, :2]
    Btau[r] = -(R0[r + 3] * Ibi[0]);
    Btau[3 + r] = -(R0[r] * (-Ibi[1]));
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    yv0[i] = p0[i]; yv0[3 + i] = s0[i];
    dy0[i] = dq0[i]; dy0[3 + i] = ds0[i];
    ydes[i] = ref[i]; ydes[3 + i] = ref[6 + i];
    dydes[i] = ref[3 + i]; dydes[3 + i] = T(0);
  }
  T c0[NY] = {T(0), T(0), -prm.g, T(0), T(0), T(0)};
#pragma unroll
  for (int i = 0; i < NY; ++i) { yv1[i] = yv0[i] + dt * dy0[i]; }
#pragma unroll
  for (int i = 0; i < N * NY; ++i) qp.l[i] = T(0);
#pragma unroll
  for (int i = 0; i < NY; ++i) qp.l[i] = -yv1[i];
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      T v;
      if (k == 0) v = -dy0[i] - dt * (i < 3 ? T0 * yv0[i + 3] : T(0)) - dt * c0[i];
      else if (k == 1) v = -dt * (i < 3 ? T0 * yv1[i + 3] : T(0)) - dt * c0[i];
      else v = -dt * c0[i];
      qp.l[NY * (N + k) + i] = v;
    }
  }
#pragma unroll
  for (int k = 0; k < N; ++k) { qp.l[NEQ + k] = -T0; qp.u3[k] = prm.Tmax - T0; }
  qp.dtT0 = dt * T0;
#pragma unroll
  for (int i = 0; i < 3; ++i) qp.s0dt[i] = dt * s0[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) qp.Btaudt[i] = dt * Btau[i];
  // objective
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const T wy = i < 3 ? (k == N - 1 ? wt.wpf : wt.wpr) : wt.ws;
      const T wd = i < 3 ? (k == N - 1 ? wt.wvf : wt.wvr) : wt.wds;
      qp.Px[k * NY + i] = wy;
      qp.q[k * NY + i] = -wy * ydes[i];
      qp.Px[N * NY + k * NY + i] = wd;
      qp.q[N * NY + k * NY + i] = -wd * dydes[i];
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      qp.Px[2 * N * NY + k * NU + i] = i == 0 ? wt.wthrust : wt.wmom;
      qp.q[2 * N * NY + k * NU + i] = T(0);
    }
  }
}

// ---------------------------------------------------------------------------
// Where the per-robot working set lives (one lane = one robot, one wave per SIMD).
//
// The ADMM loop needs, per robot: L 213 + 1/D 84 + q 45 + W 84 + x,y,z 123 + 12 thrust-row
// words = 561 words. A CU can give one lane 256 VGPRs + 256 AGPRs + 160 LDS words at one wave
// per SIMD, so the loop only fits when every word has a fixed home in one of the three. hipcc
// cannot do that (it allocates from 256 VGPRs and spills the factor to scratch: measured 2-3 GB
// of spill traffic per launch), so for fp32 the loop is the generated assembly of
// umpc_admm_asm.h (asmgen.py): W, x, y, z in VGPRs, L[160..], 1/D and q in AGPRs, L[0..160)
// in LDS. The C++ phases around it (assembly, Ruiz, LDL' before; residuals, status, extraction,
// plant after) talk to it through the HBM workspace rows below; that traffic stays in L2 /
// Infinity Cache. fp64 has no such budget (2x words): it runs the C++ loop (UMPC_GEN_ADMM_ITER)
// and lets the compiler spill.
//
// workspace rows [WS_ROWS][B]:
//   FAC_L 213 | FAC_DI 84 | FAC_Q 45 | FAC_LOEQ 36 | FAC_M 12 (lo3 up3 rho3 rinv3)   phase A -> loop
//   WS_DS 45 | WS_ES 39 | WS_C 1                                                       phase A -> phase C
//   WS_XPREV 45 | WS_DY 39   x_prev / delta_y of the last iteration                    loop -> phase C
// ---------------------------------------------------------------------------
using namespace umpcasm;

// ---------------------------------------------------------------------------
// Wrench-linearisation step, template/uprightmpc2/funapprox.c:118-165: w0 = w(u0), A1 = dw/du(u0),
// one clipped gradient step on |A1 du + (w0 - h0 - pdotdes)|^2_Qw. Parameters are batch-constant.
// ---------------------------------------------------------------------------
struct WLDev {
  float umin[4], umax[4], dumax[4], Qw[6];
  float a0[6], a1[6][4], A2[6][16];
  float Md[6];   // M0 = diag(mb, mb, mb, ixx, iyy, izz) of dynamicsTerms (template/ca6dynamics.py:5-10, 44-50)
};

// P may live in kernarg (by value) or in global memory (wave-uniform loads): u0 is updated in place.
template <typename T>
__device__ __forceinline__ void wl_step(const WLDev &P, T (&u0)[4], const T (&h0)[6], const T (&pd)[6], T (&w0)[6]) {
  T A1[6][4], a0v[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    // funApproxF: a0 + u.a1 + 0.5 u'(A2 u), accumulated like the reference's matMult (funapprox.c:53-65)
    T dot = T(0), vout[4], quad = T(0);
#pragma unroll
    for (int l = 0; l < 4; ++l) dot += u0[l] * T(P.a1[i][l]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      T acc = T(0);
#pragma unroll
      for (int l = 0; l < 4; ++l) acc += T(P.A2[i][r + 4 * l]) * u0[l];
      vout[r] = acc;
    }
#pragma unroll
    for (int l = 0; l < 4; ++l) quad += u0[l] * vout[l];
    const T w = (T(P.a0[i]) + dot) + T(0.5) * quad;
    w0[i] = w;
    a0v[i] = w - h0[i] - pd[i];
    // funApproxDf: a1 + A2 u  (funapprox.c:67-76)
#pragma unroll
    for (int j = 0; j < 4; ++j) A1[i][j] = T(P.a1[i][j]) + vout[j];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    T Lb = -T(P.dumax[j]), Ub = T(P.dumax[j]);
    if (u0[j] < T(P.umin[j])) Lb = T(0);
    else if (u0[j] > T(P.umax[j])) Ub = T(0);
    T acc = T(0);
#pragma unroll
    for (int i = 0; i < 6; ++i) acc += A1[i][j] * (T(P.Qw[i]) * a0v[i]);
    T d = T(-1e3) * acc;
    if (d < Lb) d = Lb;
    else if (d > Ub) d = Ub;
    u0[j] = u0[j] + d;
  }
}

template <typename T>
struct StepIO {
  DevParams<T> prm;
  int B;
  T *state;        // [18][B]
  T *ctrl;         // [127][B]
  const T *ref;    // [9][B]
  const T *Ib;     // [3][B] or null
  const T *gain;   // [B] or null
  const T *weights;  // [8][B] per-robot (ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom) or null
  T t0;            // time (ms) of the first step of this launch, for the task generators
  T *ws;           // [WS_ROWS][B] workspace
  T *out;          // [9][B]
  T *stats;        // [2][B] or null
  int32_t *status; // [B] or null
  T *info;         // [2][B] or null
  // MPC -> WL -> actualT0 coupling (robobee_test_controllers.py:162-171, conn_MPC_WL.m:2-10), off when wl == null
  const WLDev *wl; // device copy of the WL parameters
  T *wlu;          // [4][B] WL input state u4, in/out
  T *wlw;          // [6][B] w0 of the last step, or null
};

// One closed-loop step of robot b: controller step (= umpcUpdate) + nsub plant substeps.
// Everything persistent round-trips through the SoA arrays, so K steps in one launch and K
// launches of one step are the same computation.
// LDSF (fp64, small batches): the factor L and 1/D live in LDS ([word][lane], one wave per CU owns 152 kB) instead
// of local arrays the compiler spills to scratch; same arithmetic, same results.
// QUAD (fp64 + ASM + LDSF): the four lanes of a quad were given the SAME robot b; they run every phase redundantly, each
// on its own LDS slice, and the ADMM phase splits iterations 2..maxIter over lanes 0..2 (umpc_admm_asm64_quad.h).
template <typename T, bool ASM, bool LDSF = false, bool QUAD = false>
__device__ __forceinline__ void closed_loop_step(const StepIO<T> &a, const int b, const unsigned ldsaddr, T *ldsw,
                                                 const int step, const T *actualT0) {
// word w of this lane's LDS slot: float4-interleaved, ldsw = (T *)lds + 4 * lane
#define LDSW(w) ldsw[((w) >> 2) * 256 + ((w) & 3)]
  const bool first_step = step == 0;
  // ASM: the ADMM phase is generated assembly -- fp32: umpc_admm_asm.h (asmgen.py); fp64: umpc_admm_asm64.h (asmgen64.py),
  // which needs LDSF (L and 1/D handed over in LDS) and maxIter >= 1 (the host dispatches accordingly)
  constexpr bool ASM32 = ASM && sizeof(T) == 4, ASM64 = ASM && sizeof(T) == 8;
  static_assert(!ASM64 || LDSF, "the fp64 assembly loop takes the factor from LDS");
  const size_t B = (size_t)a.B;
  const DevParams<T> &prm = a.prm;
  const T sigma = T(1e-6), alpha = T(1.6), oma = T(1.0) - T(1.6);
  const T eps_abs0 = T(1e-4), eps_rel0 = T(1e-4), eps_pinf0 = T(1e-4), eps_dinf0 = T(1e-4);
  // Row pointers are wave-uniform (SGPR) and the robot index is a 32-bit lane offset. `bb` is
  // laundered through an empty asm at every phase boundary so that the compiler re-derives the
  // (cheap) addresses instead of keeping ~400 precomputed 64-bit pointers alive across the loop.
  unsigned bb = (unsigned)b;
// scheduling fence: nothing (in particular no LDS read) is moved across it
#define UMPC_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define UMPC_PHASE_FENCE() asm volatile("" : "+v"(bb)::"memory")
#define GLD(arr, row) ((arr) + (size_t)(row) * B)[bb]
#define Q_(j) qv[j]
// LDSF word w of this lane: 16-byte quads of two words, quads interleaved over the lanes (ldsw = (T *)lds + 2 * lane),
// so that the assembly loop fetches two words per ds_read_b128
#define LDSF_W(w) ldsw[((w) >> 1) * 128 + ((w) & 1)]
#define DI_(k) (*(LDSF ? &LDSF_W(NNZL + (k)) : &Di[LDSF ? 0 : (k)]))
#define LX_(e) (*(LDSF ? &LDSF_W(e) : &Lx[LDSF ? 0 : (e)]))
#define RINV3_(k) rinv3[k]
#define RHO3_(k) rho3[k]
#define LO3_(k) lo3[k]
#define UP3_(k) up3[k]
#define W_(k) W[k]
#define X_(j) x[j]
#define Y_(i) y[i]
#define Z_(i) z[i]

  T Lx[LDSF ? 1 : NNZL], Di[LDSF ? 1 : NK], qv[NX];
  T lo[NEQ];  // scaled bounds of the dynamics rows; consumed by the first ADMM iteration
  T lo3[N], up3[N], rho3[N], rinv3[N], Eprev3[N];
  T T0 = GLD(a.ctrl, NX + 2 * NC);
  if (first_step && actualT0) {
    const T t = actualT0[bb];
    if (t >= T(0)) T0 = t;  // uprightmpc2.c:215-216
  }
  T Ibi[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) Ibi[i] = T(1) / (a.Ib ? GLD(a.Ib, i) : prm.Ib[i]);  // uprightmpc2.c:50-52
  // MPC fire time of this step (the harness evaluates the task at the fire substep, uprightmpc2.py:124-139)
  const T tnow = a.t0 + T(step) * (T(prm.nsub) * prm.dtsim);
  Weights<T> wt = {prm.ws, prm.wds, prm.wpr, prm.wpf, prm.wvr, prm.wvf, prm.wthrust, prm.wmom};
  if (a.weights) {
    wt.ws = GLD(a.weights, 0); wt.wds = GLD(a.weights, 1); wt.wpr = GLD(a.weights, 2); wt.wpf = GLD(a.weights, 3);
    wt.wvr = GLD(a.weights, 4); wt.wvf = GLD(a.weights, 5); wt.wthrust = GLD(a.weights, 6); wt.wmom = GLD(a.weights, 7);
  }

// raw diagonal of P for column j (umpcInit weight layout, uprightmpc2.c:27-36), from a Weights object
#define PXRAW_OF(wt, j) ((j) < 2 * N * NY ? ((j) % NY < 3 ? ((j) < N * NY ? ((j) / NY == N - 1 ? wt.wpf : wt.wpr)                    \
                                                                     : (((j) - N * NY) / NY == N - 1 ? wt.wvf : wt.wvr))        \
                                                       : ((j) < N * NY ? wt.ws : wt.wds))                                       \
                                   : (((j) - 2 * N * NY) % NU == 0 ? wt.wthrust : wt.wmom))
#define PXRAW(j) PXRAW_OF(wt, j)
  // reciprocals of the eight weights: D is recovered after Ruiz as sqrt(P_scaled / (P_raw c))
  const Weights<T> wti = {T(1) / wt.ws, T(1) / wt.wds, T(1) / wt.wpr, T(1) / wt.wpf, T(1) / wt.wvr, T(1) / wt.wvf,
                          T(1) / wt.wthrust, T(1) / wt.wmom};
// -DUMPC_PHASE_TIMING (tools/build_variant.py, diagnostics only): 100 MHz timestamps at the phase boundaries; the
// six intervals (assemble, Ruiz, D/E + factor, ADMM, residuals/extraction, plant) replace accdes in out rows 3..8
#ifdef UMPC_PHASE_TIMING
  long long tmark[7];
  long long asm64_stamps[6] = {0, 0, 0, 0, 0, 0};
#define UMPC_TMARK(k) tmark[k] = __builtin_amdgcn_s_memrealtime()
#else
#define UMPC_TMARK(k)
#endif
  // =========================== phase A: assemble, equilibrate, factor ===========================
  UMPC_PHASE_FENCE();
  UMPC_TMARK(0);
  {
    T p0[3], R0[9], dq0[6], ref[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) p0[i] = GLD(a.state, i);
#pragma unroll
    for (int i = 0; i < 9; ++i) R0[i] = GLD(a.state, 3 + i);
#pragma unroll
    for (int i = 0; i < 6; ++i) dq0[i] = GLD(a.state, 12 + i);
#pragma unroll
    for (int i = 0; i < 9; ++i) ref[i] = GLD(a.ref, i);
    task_reference(prm.task, prm.task_p, tnow, ref);
#pragma unroll
    for (int k = 0; k < N; ++k) Eprev3[k] = GLD(a.ctrl, NX + 2 * NC + 1 + k);

    // LDSF: q and the raw lower bounds wait in LDS words 84..167 (free until the factor is written) instead of
    // occupying 84 live fp64 words through the ten Ruiz passes
    T P[NX], A[NNZA], q[LDSF ? 1 : NX], Ds[NX], Es[NC], lraw[LDSF ? 1 : NC];
    // LDSF: the matrix itself stays in LDS words 168..278 through the passes (read twice, written once per pass; LDS
    // round trips overlap, scratch ones did not) and comes back to registers once, for the factorisation
#define QA_(j) (*(LDSF ? &LDSF_W(NX + NC + (j)) : &q[LDSF ? 0 : (j)]))
#define LRAW_(i) (*(LDSF ? &LDSF_W(2 * NX + NC + (i)) : &lraw[LDSF ? 0 : (i)]))
    {
      RawQP<T> qp;
      assemble(prm, wt, Ibi, T0, p0, R0, dq0, ref, qp);
#define A_(p) (*(LDSF ? &LDSF_W(2 * NX + 2 * NC + (p)) : &A[LDSF ? 0 : (p)]))
      UMPC_GEN_ASSEMBLE_A(prm.dt, qp.dtT0, qp.s0dt, qp.Btaudt);
#pragma unroll
      for (int j = 0; j < NX; ++j) { P[j] = qp.Px[j]; QA_(j) = qp.q[j]; }
#pragma unroll
      for (int i = 0; i < NC; ++i) LRAW_(i) = qp.l[i];
#pragma unroll
      for (int k = 0; k < N; ++k) up3[k] = qp.u3[k];
    }
    // constraint classification with the PREVIOUS call's E (osqp.c:812-820 -> auxil.c:103-145).
    // Rows < NEQ have l == u bit-for-bit: always "equality".
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const T ls = LRAW_(NEQ + k) * Eprev3[k], us = up3[k] * Eprev3[k];
      if ((ls < -T(UMPC_INFTY) * T(UMPC_MIN_SCALING)) && (us > T(UMPC_INFTY) * T(UMPC_MIN_SCALING))) {
        rho3[k] = T(UMPC_RHO_MIN); rinv3[k] = T(1) / T(UMPC_RHO_MIN);
      } else if (us - ls < T(UMPC_RHO_TOL)) {
        rho3[k] = T(RHO_EQ); rinv3[k] = T(RINV_EQ);
      } else {
        rho3[k] = T(UMPC_RHO); rinv3[k] = T(1) / T(UMPC_RHO);
      }
    }
    UMPC_TMARK(1);
    // ---- Ruiz equilibration, scaling.c:44-156 ----
    // The accumulated D and E are not carried through the passes (84 live words and 84 multiplies per pass):
    // they are recovered afterwards from the equilibrated data, D_j = sqrt(P_jj / (P_raw_jj c)) and
    // E_i = |A_ip| / D_p on the +-1 entry of row i -- the same numbers up to rounding.
    T cscale = T(1);
#define P_(j) P[j]
// LDSF: the per-pass scalings live in LDS words 0..83 (the factor is written there only after the last pass): 84 fewer
// live fp64 words in the most register-starved part of the kernel
#define DT_(j) (*(LDSF ? &LDSF_W(j) : &Dt[LDSF ? 0 : (j)]))
#define ET_(i) (*(LDSF ? &LDSF_W(NX + (i)) : &Et[LDSF ? 0 : (i)]))
    if constexpr (ASM64) {
      // the passes as generated fp64 assembly (asmgen64.ruiz_program): the matrix in VGPRs, Et / P / q in AGPRs, nothing
      // leaves the register files for the ten passes; P joins q and A in LDS for the hand-over
      static_assert(umpcasm64::RZ_Q == NX + NC && umpcasm64::RZ_A == 2 * NX + 2 * NC && umpcasm64::RZ_P + NX <= umpcasm64::RZ_C,
                    "LDS staging words of the Ruiz block");
#pragma unroll
      for (int j = 0; j < NX; ++j) LDSF_W(umpcasm64::RZ_P + j) = P[j];
      UMPC_PHASE_FENCE();
      if constexpr (QUAD) {
        UMPC_RUIZ_ASM64_QUAD(ldsaddr, UMPC_SCALING_ITERS);      // the passes on the lane quad (asmquad64.ruiz_program)
      } else {
        UMPC_RUIZ_ASM64(ldsaddr, UMPC_SCALING_ITERS);
      }
      UMPC_PHASE_FENCE();
#pragma unroll
      for (int j = 0; j < NX; ++j) P[j] = LDSF_W(umpcasm64::RZ_P + j);
      cscale = LDSF_W(umpcasm64::RZ_C);
    } else
#pragma nounroll
    for (int it = 0; it < UMPC_SCALING_ITERS; ++it) {
      T Dt[LDSF ? 1 : NX], Et[LDSF ? 1 : NC];
      UMPC_GEN_RUIZ_NORMS();
#pragma unroll
      for (int j = 0; j < NX; ++j) DT_(j) = umpc_rsqrt(limit_scaling(DT_(j)));
#pragma unroll
      for (int i = 0; i < NC; ++i) ET_(i) = umpc_rsqrt(limit_scaling(ET_(i)));
      // LDSF: without this fence the compiler forwards the 84 scalings from the stores above to the loads below in
      // registers, and spills the matrix to scratch instead (147 exposed scratch round trips per pass)
      if constexpr (LDSF) asm volatile("" ::: "memory");
#pragma unroll
      for (int j = 0; j < NX; ++j) P[j] = (P[j] * DT_(j)) * DT_(j);
      UMPC_GEN_RUIZ_APPLY_A();
      T pmean = T(0), qn = T(0);
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        const T qj = QA_(j) * DT_(j);
        QA_(j) = qj;
        pmean += umpc_abs(P[j]);
        qn = umpc_max(qn, umpc_abs(qj));
      }
      pmean /= T(NX);
      qn = limit_scaling(qn);
      T ct = limit_scaling(umpc_max(pmean, qn));
      ct = T(1) / ct;
#pragma unroll
      for (int j = 0; j < NX; ++j) { P[j] *= ct; QA_(j) = QA_(j) * ct; }
      cscale *= ct;
    }
#undef DT_
#undef ET_
    UMPC_TMARK(2);
    {
      const T cinv_ = T(1) / cscale;
#pragma unroll
      for (int j = 0; j < NX; ++j) Ds[j] = umpc_sqrt_fast((P[j] * cinv_) * PXRAW_OF(wti, j));
#define DS_(j) Ds[j]
#define ES_(i) Es[i]
      UMPC_GEN_E_FROM_A(umpc_rcp_fast);
#undef DS_
#undef ES_
    }
    // hand-off rows: what phase C needs again, and what the loop consumes
#pragma unroll
    for (int j = 0; j < NX; ++j) { GLD(a.ws, WS_DS + j) = Ds[j]; const T qj = QA_(j); GLD(a.ws, FAC_Q + j) = qj; Q_(j) = qj; }
#pragma unroll
    for (int i = 0; i < NC; ++i) GLD(a.ws, WS_ES + i) = Es[i];
    GLD(a.ws, WS_C) = cscale;
#pragma unroll
    for (int i = 0; i < NEQ; ++i) lo[i] = LRAW_(i) * Es[i];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      lo3[k] = LRAW_(NEQ + k) * Es[NEQ + k];
      up3[k] = up3[k] * Es[NEQ + k];
      // the thrust-row E of THIS call classifies the NEXT call's thrust rows (osqp.c:812-820): stored here, where it is
      // formed, so that it does not live (in scratch, around the assembly blocks) until the end of the step
      GLD(a.ctrl, NX + 2 * NC + 1 + k) = Es[NEQ + k];
      GLD(a.ws, FAC_M + k) = lo3[k];
      GLD(a.ws, FAC_M + N + k) = up3[k];
      GLD(a.ws, FAC_M + 2 * N + k) = rho3[k];
      GLD(a.ws, FAC_M + 3 * N + k) = rinv3[k];
    }
    // ---- KKT fill + LDL', kkt.c:184-222 + qdldl.c:86-247 ----
    if constexpr (LDSF) {   // the factor is written over the staging words: the matrix moves to registers first
#pragma unroll
      for (int p = 0; p < NNZA; ++p) A[p] = A_(p);
      asm volatile("" ::: "memory");
    }
#undef A_
#define A_(p) A[p]
    int npos = 0;
    UMPC_GEN_KKT_FACTOR(npos);
    (void)npos;
#undef A_
#undef P_
    if constexpr (ASM64) {
      // L and 1/D are in LDS already (LX_ / DI_); the loop streams l of the dynamics rows from its rows
#pragma unroll
      for (int i = 0; i < NEQ; ++i) GLD(a.ws, FAC_LOEQ + i) = lo[i];
    }
    if constexpr (ASM32) {
      // the first 160 storage positions of L go straight to their loop home in LDS (float4-interleaved per lane),
      // the rest through rows; the storage order (paired entries adjacent) is the generator's
#define UMPC_ROW_(r) GLD(a.ws, r)
#define UMPC_LX_(j) Lx[j]
      UMPC_ASM_STORE_L(LDSW, UMPC_ROW_, UMPC_LX_);
#undef UMPC_ROW_
#undef UMPC_LX_
#pragma unroll
      for (int k = 0; k < NK; ++k) GLD(a.ws, FAC_DI + k) = Di[k];
#pragma unroll
      for (int i = 0; i < NEQ; ++i) GLD(a.ws, FAC_LOEQ + i) = lo[i];
    }
  }

  // =========================== phase B: ADMM, osqp.c:354-370 ===========================
  UMPC_PHASE_FENCE();
  UMPC_TMARK(3);
  if constexpr (ASM64) {
    const unsigned voff = bb * 8u, stride = (unsigned)a.B * 8u;
    const int iters = prm.maxIter;
    if constexpr (QUAD) {
      const unsigned *tab = &umpcquad64::kTab[0][0];
      UMPC_ADMM_ASM64_QUAD(voff, ldsaddr, a.ws, a.ctrl, tab, stride, iters);
    } else {
      UMPC_ADMM_ASM64(voff, ldsaddr, a.ws, a.ctrl, stride, iters);
    }
#ifdef UMPC_ASM64_TIMING   /* header generated with UMPC_ASM64_TIMING=1: the block's own stamps, LDS words 297..302 (the residual block reuses them) */
#pragma unroll
    for (int k = 0; k < 6; ++k) asm64_stamps[k] = __builtin_bit_cast(long long, LDSF_W(297 + k));
#endif
  } else if constexpr (ASM32) {
    // generated gfx950 assembly (umpc_admm_asm.h): reads FAC_* and x,y,z, runs maxIter iterations with
    // a static VGPR/AGPR/LDS placement, writes x,y,z back and x_prev / delta_y of the last iteration
    const unsigned voff = bb * 4u, stride = (unsigned)a.B * 4u;
    const int iters = prm.maxIter;
    UMPC_ADMM_ASM(voff, ldsaddr, a.ws, a.ctrl, stride, iters);
  } else {
    T x[NX], y[NC], z[NC];
#pragma unroll
    for (int j = 0; j < NX; ++j) x[j] = GLD(a.ctrl, j);
#pragma unroll
    for (int i = 0; i < NC; ++i) y[i] = GLD(a.ctrl, NX + i);
#pragma unroll
    for (int i = 0; i < NC; ++i) z[i] = GLD(a.ctrl, NX + NC + i);
    T W[NK];
    // x_prev / delta_y of the LAST iteration feed the infeasibility tests (auxil.c:362-512)
#define UMPC_CAPTURE_X() _Pragma("unroll") for (int j = 0; j < NX; ++j) GLD(a.ws, WS_XPREV + j) = x[j]
    if (prm.maxIter >= 1) {
      // first iteration: z_prev is the warm start; afterwards z == l == u on the dynamics rows
      UMPC_CAPTURE_X();
#define LOEQ_(i) lo[i]
#define UMPC_ADMM_DY(i, v) GLD(a.ws, WS_DY + (i)) = (v)
      UMPC_GEN_ADMM_ITER();
#undef LOEQ_
#undef UMPC_ADMM_DY
    } else {
      UMPC_CAPTURE_X();
#pragma unroll
      for (int i = 0; i < NC; ++i) GLD(a.ws, WS_DY + i) = T(0);
    }
#define LOEQ_(i) z[i]
#define UMPC_ADMM_DY(i, v)
#pragma nounroll
    for (int it = 2; it < prm.maxIter; ++it) {
      UMPC_GEN_ADMM_ITER();
    }
#undef UMPC_ADMM_DY
#define UMPC_ADMM_DY(i, v) GLD(a.ws, WS_DY + (i)) = (v)
    if (prm.maxIter >= 2) {
      UMPC_CAPTURE_X();
      UMPC_GEN_ADMM_ITER();
    }
#undef UMPC_ADMM_DY
#undef LOEQ_
#pragma unroll
    for (int j = 0; j < NX; ++j) GLD(a.ctrl, j) = x[j];
#pragma unroll
    for (int i = 0; i < NC; ++i) GLD(a.ctrl, NX + i) = y[i];
#pragma unroll
    for (int i = 0; i < NC; ++i) GLD(a.ctrl, NX + NC + i) = z[i];
  }

  // =========================== phase C: residuals, status, extraction, plant ===========================
  UMPC_PHASE_FENCE();
  UMPC_TMARK(4);
  // x, y, z are consumed where they lie: LDS words 0..122 after the assembly loop (fp32), the ctrl rows after
  // the C++ loop (fp64). No arrays: this phase must stay under 256 live VGPRs (the assembly block clobbers
  // every AGPR, so the compiler has nowhere cheap to spill).
// (fp64 assembly loop: x and y are still in their LDS words, and also in the ctrl rows; z only in the rows)
#define XV(j) (ASM32 ? LDSW(j) : ASM64 ? LDSF_W(umpcasm64::LW_X + (j)) : GLD(a.ctrl, j))
#define YV(i) (ASM32 ? LDSW(NX + (i)) : ASM64 ? LDSF_W(umpcasm64::LW_Y + (i)) : GLD(a.ctrl, NX + (i)))
#define ZV(i) (ASM32 ? LDSW(NX + NC + (i)) : ASM64 ? LDSF_W(umpcasm64::PC_Z + (i)) : GLD(a.ctrl, NX + NC + (i)))
// fp64 assembly loop: its epilogue staged everything else this phase reads in LDS as well (hipcc fetches the rows one
// exposed global load at a time, ~2 us each for a lone wave): D, E, x_prev, delta_y, the thrust-row bounds
#define PC_DS_(j) (ASM64 ? LDSF_W(umpcasm64::PC_DS + (j)) : GLD(a.ws, WS_DS + (j)))
#define PC_ES_(i) (ASM64 ? LDSF_W(umpcasm64::PC_ES + (i)) : GLD(a.ws, WS_ES + (i)))
#define PC_XP_(j) (*(ASM64 ? &LDSF_W(umpcasm64::PC_XP + (j)) : &GLD(a.ws, WS_XPREV + (j))))
#define PC_DY_(i) (*(ASM64 ? &LDSF_W(umpcasm64::PC_DY + (i)) : &GLD(a.ws, WS_DY + (i))))
#pragma unroll
  for (int k = 0; k < N; ++k) {
    lo3[k] = ASM64 ? LDSF_W(umpcasm64::PC_LO3 + k) : GLD(a.ws, FAC_M + k);
    up3[k] = ASM64 ? LDSF_W(umpcasm64::PC_UP3 + k) : GLD(a.ws, FAC_M + N + k);
  }
  if (prm.maxIter < 1) {  // no iteration ran: nothing was captured
#pragma unroll
    for (int j = 0; j < NX; ++j) PC_XP_(j) = XV(j);
#pragma unroll
    for (int i = 0; i < NC; ++i) PC_DY_(i) = T(0);
  }
  T p0[3], R0[9], dq0[6];
#pragma unroll
  for (int i = 0; i < 9; ++i) R0[i] = GLD(a.state, 3 + i);
  // (fp64 assembly route: p and dq are fetched after the residual block, which clobbers the register files)
#define UMPC_PC_LOAD_PDQ() do { \
  _Pragma("unroll") for (int i = 0; i < 3; ++i) p0[i] = GLD(a.state, i); \
  _Pragma("unroll") for (int i = 0; i < 6; ++i) dq0[i] = GLD(a.state, 12 + i); } while (0)
  if constexpr (!ASM64) UMPC_PC_LOAD_PDQ();
  int status = ST_UNSOLVED;
  T pri_res = T(0), dua_res = T(0);
  T u0 = T(0), u1 = T(0), u2 = T(0), dy1[NY];
  {
    T Ds[NX], Es[NC];
    const T cscale = GLD(a.ws, WS_C);
    const T cinv = T(1) / cscale;
    // (fp64 assembly route: D and E are fetched AFTER the residual block, which clobbers the register files)
#define UMPC_PC_LOAD_DE() do { \
    _Pragma("unroll") for (int j = 0; j < NX; ++j) Ds[j] = PC_DS_(j); \
    _Pragma("unroll") for (int i = 0; i < NC; ++i) Es[i] = PC_ES_(i); } while (0)
    if constexpr (!ASM64) UMPC_PC_LOAD_DE();
    // raw entries of A: the scaled matrix is re-derived entry by entry as (raw * E_i) * D_j wherever it is
    // needed (the factorisation consumed the equilibrated copy); nothing of size 111 is ever live here
    T dtT0, s0dt[3], Btaudt[6];
    {
      const T dt = prm.dt;
      dtT0 = dt * T0;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        s0dt[r] = dt * R0[6 + r];
        Btaudt[r] = dt * (-(R0[r + 3] * Ibi[0]));
        Btaudt[3 + r] = dt * (-(R0[r] * (-Ibi[1])));
      }
    }
#define DT_(j) Ds[j]
#define ET_(i) Es[i]
    // ---- update_info: residuals (auxil.c:243-307) ----
    T nz = T(0), nAx = T(0), nq = T(0), nAty = T(0), nPx = T(0);
    // The reference's inf-norm skips NaN entries (`if (abs > max)`, lin_alg.c:300-311) exactly like v_max; a NaN
    // residual entry means an overflow (inf - inf) on finite inputs, where whether some OTHER entry is still
    // > OSQP_INFTY is an accident of the evaluation order. `nanacc` turns every such step into OSQP_NON_CVX (what
    // the reference reports on tests/golden/nan_branch.npz): NaN-propagating sum of the residual entries.
    T nanacc = T(0);
    if constexpr (ASM64) {
      // the norms as generated fp64 assembly (asmgen64.resid_program): its inputs next to x, y, z, D, E in LDS
      namespace r64 = umpcasm64;
      LDSF_W(r64::RS_PAR) = dtT0;
#pragma unroll
      for (int r = 0; r < 3; ++r) LDSF_W(r64::RS_PAR + 1 + r) = s0dt[r];
#pragma unroll
      for (int r = 0; r < 6; ++r) LDSF_W(r64::RS_PAR + 4 + r) = Btaudt[r];
      LDSF_W(r64::RS_C) = cscale;
      LDSF_W(r64::RS_DT) = prm.dt;
      const T w8[8] = {wt.ws, wt.wds, wt.wpr, wt.wpf, wt.wvr, wt.wvf, wt.wthrust, wt.wmom};
#pragma unroll
      for (int k = 0; k < 8; ++k) LDSF_W(r64::RS_W + k) = w8[k];
      const unsigned voff = bb * 8u, stride = (unsigned)a.B * 8u;
      UMPC_RESID_ASM64(voff, ldsaddr, a.ws, stride);
      pri_res = LDSF_W(r64::RS_OUT + 0); dua_res = LDSF_W(r64::RS_OUT + 1);
      nz = LDSF_W(r64::RS_OUT + 2); nAx = LDSF_W(r64::RS_OUT + 3); nq = LDSF_W(r64::RS_OUT + 4);
      nAty = LDSF_W(r64::RS_OUT + 5); nPx = LDSF_W(r64::RS_OUT + 6); nanacc = LDSF_W(r64::RS_OUT + 7);
      UMPC_PC_LOAD_DE();
      UMPC_PC_LOAD_PDQ();
    } else {
    {
      T Ax[NC];
#pragma unroll
      for (int i = 0; i < NC; ++i) Ax[i] = T(0);
#define OUT_AX(i) Ax[i]
      UMPC_GEN_A_MUL_SCALED(prm.dt, dtT0, s0dt, Btaudt, XV, OUT_AX);
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const T einv = umpc_rcp_fast(Es[i]);
        const T zi = ZV(i);
        pri_res = umpc_max(pri_res, umpc_abs(einv * (Ax[i] - zi)));
        nanacc = umpc_fma(T(0), Ax[i] - zi, nanacc);
        nz = umpc_max(nz, umpc_abs(einv * zi));
        nAx = umpc_max(nAx, umpc_abs(einv * Ax[i]));
      }
    }
    {
      T Aty[NX];
#define OUT_ATY(j) Aty[j]
      UMPC_GEN_AT_MUL_SCALED(prm.dt, dtT0, s0dt, Btaudt, YV, OUT_ATY);
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        const T dinv = umpc_rcp_fast(Ds[j]);
        const T qj = GLD(a.ws, FAC_Q + j);
        const T Pxj = (((PXRAW(j) * Ds[j]) * Ds[j]) * cscale) * XV(j);
        dua_res = umpc_max(dua_res, umpc_abs(dinv * ((qj + Pxj) + Aty[j])));
        nanacc = umpc_fma(T(0), (qj + Pxj) + Aty[j], nanacc);
        nq = umpc_max(nq, umpc_abs(dinv * qj));
        nAty = umpc_max(nAty, umpc_abs(dinv * Aty[j]));
        nPx = umpc_max(nPx, umpc_abs(dinv * Pxj));
      }
    }
    }
    dua_res = cinv * dua_res;

    // ---- check_termination (auxil.c:684-789), exact then approximate (osqp.c:524-573) ----
    // The infeasibility certificates are seven scalars that do not depend on the tolerance (all bounds are
    // finite, so is_primal_infeasible projects nothing): they are computed once, straight-line, and compared
    // against the exact and then the 10x tolerances. (A two-trip loop here made the compiler hoist ~500
    // loop-invariant words into scratch.)
    if ((pri_res > T(UMPC_INFTY)) || (dua_res > T(UMPC_INFTY)) || nanacc != nanacc) {
      status = ST_NON_CVX;
    } else {
      const T rel_p = umpc_max(nz, nAx), rel_d = umpc_max(umpc_max(nq, nAty), nPx) * cinv;
      const bool prim_ok0 = pri_res < eps_abs0 + eps_rel0 * rel_p, dual_ok0 = dua_res < eps_abs0 + eps_rel0 * rel_d;
      const bool prim_ok1 = pri_res < T(10) * eps_abs0 + (T(10) * eps_rel0) * rel_p;
      const bool dual_ok1 = dua_res < T(10) * eps_abs0 + (T(10) * eps_rel0) * rel_d;
      T ndy = T(0), lhs = T(0), nrm = T(UMPC_INFTY);                 // is_primal_infeasible, auxil.c:362-424
      T ndx = T(0), qdx = T(0), nP = T(UMPC_INFTY), nAdx = T(UMPC_INFTY);  // is_dual_infeasible, auxil.c:426-512
      if (!prim_ok0) {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
          const T dyi = PC_DY_(i);
          const T li = i < NEQ ? ZV(i) : lo3[i < NEQ ? 0 : i - NEQ];  // z == l == u on the dynamics rows
          const T ui = i < NEQ ? li : up3[i < NEQ ? 0 : i - NEQ];
          ndy = umpc_max(ndy, umpc_abs(dyi * Es[i]));
          lhs += ui * umpc_max(dyi, T(0)) + li * umpc_min(dyi, T(0));
        }
        if (ndy > eps_pinf0 && lhs < -eps_pinf0 * ndy) {  // the 10x test implies this one
          T Atdy[NX];
#define IN_DY(i) PC_DY_(i)
#define OUT_ATDY(j) Atdy[j]
          UMPC_GEN_AT_MUL_SCALED(prm.dt, dtT0, s0dt, Btaudt, IN_DY, OUT_ATDY);
          nrm = T(0);
#pragma unroll
          for (int j = 0; j < NX; ++j) nrm = umpc_max(nrm, umpc_abs(Atdy[j] * umpc_rcp_fast(Ds[j])));
        }
      }
      if (!dual_ok0) {
        T dxv[NX];
#pragma unroll
        for (int j = 0; j < NX; ++j) {
          dxv[j] = XV(j) - PC_XP_(j);
          ndx = umpc_max(ndx, umpc_abs(Ds[j] * dxv[j]));
          qdx += GLD(a.ws, FAC_Q + j) * dxv[j];
        }
        if (ndx > eps_dinf0 && qdx < -cscale * eps_dinf0 * ndx) {
          nP = T(0);
#pragma unroll
          for (int j = 0; j < NX; ++j)
            nP = umpc_max(nP, umpc_abs(((((PXRAW(j) * Ds[j]) * Ds[j]) * cscale) * dxv[j]) * umpc_rcp_fast(Ds[j])));
          if (nP < cscale * (T(10) * eps_dinf0) * ndx) {
            T Adx[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i) Adx[i] = T(0);
#define IN_DX(j) dxv[j]
#define OUT_ADX(i) Adx[i]
            UMPC_GEN_A_MUL_SCALED(prm.dt, dtT0, s0dt, Btaudt, IN_DX, OUT_ADX);
            nAdx = T(0);
#pragma unroll
            for (int i = 0; i < NC; ++i) nAdx = umpc_max(nAdx, umpc_abs(Adx[i] * umpc_rcp_fast(Es[i])));
          }
        }
      }
#define UMPC_PINF(e) (ndy > (e) && lhs < -(e) * ndy && nrm < (e) * ndy)
#define UMPC_DINF(e) (ndx > (e) && qdx < -cscale * (e) * ndx && nP < cscale * (e) * ndx && !(nAdx > (e) * ndx))
      if (prim_ok0 && dual_ok0) status = ST_SOLVED;
      else if (!prim_ok0 && UMPC_PINF(eps_pinf0)) status = ST_PINF;
      else if (!dual_ok0 && UMPC_DINF(eps_dinf0)) status = ST_DINF;
      else if (prim_ok1 && dual_ok1) status = ST_SOLVED_INACC;
      else if (!prim_ok1 && UMPC_PINF(T(10) * eps_pinf0)) status = ST_PINF_INACC;
      else if (!dual_ok1 && UMPC_DINF(T(10) * eps_dinf0)) status = ST_DINF_INACC;
      else status = ST_MAX_ITER;
#undef UMPC_PINF
#undef UMPC_DINF
    }
    // ---- store_solution (auxil.c:527-565): unscale what the extraction needs ----
    u0 = XV(2 * NY * N + 0) * Ds[2 * NY * N + 0];
    u1 = XV(2 * NY * N + 1) * Ds[2 * NY * N + 1];
    u2 = XV(2 * NY * N + 2) * Ds[2 * NY * N + 2];
#pragma unroll
    for (int i = 0; i < NY; ++i) dy1[i] = XV(NY * N + i) * Ds[NY * N + i];
#undef DT_
#undef ET_
#undef UMPC_PC_LOAD_DE
#undef UMPC_PC_LOAD_PDQ
  }

  // ---- extraction (uprightmpc2.c:253-269) ----
  const bool has_sol = status != ST_PINF && status != ST_PINF_INACC && status != ST_DINF &&
                       status != ST_DINF_INACC && status != ST_NON_CVX;
  if (!has_sol) {
    // the reference's OSQP_NAN is the NUMBER (c_float)0x7fc00000 = 2143289344 (constants.h:96),
    // and the iterates are cold-started (auxil.c:563)
    const T nanv = T(2143289344.0);
    u0 = u1 = u2 = nanv;
#pragma unroll
    for (int i = 0; i < NY; ++i) dy1[i] = nanv;
  }
  T0 += u0;
  T uq[3] = {T0, u1, u2}, acc[NY];
  {
    // dq1des = (dy1des[0:3], e3h R0' dy1des[3:6]); e3h R0' v = (-(R0' v)_y, (R0' v)_x, 0)
    const T rx = (R0[0] * dy1[3] + R0[1] * dy1[4]) + R0[2] * dy1[5];
    const T ry = (R0[3] * dy1[3] + R0[4] * dy1[4]) + R0[5] * dy1[5];
    const T dq1[NY] = {dy1[0], dy1[1], dy1[2], -ry, rx, T(0)};
#pragma unroll
    for (int i = 0; i < NY; ++i) acc[i] = (dq1[i] - dq0[i]) / prm.dt;
  }
  // ---- MPC -> WL -> actualT0 (robobee_test_controllers.py:162-171): h0 = (Rb' (0,0,mb g), 0), pdotdes = M0 accdes,
  // (u4, w0) = wlConUpdate(h0, pdotdes), and the NEXT umpcUpdate gets actualT0 = w0[2] / M0[2,2], which overrides
  // the accumulator when >= 0 (uprightmpc2.c:215-216). The command of THIS step (uq) is not touched.
  T T0next = T0;
  if (a.wl) {
    const WLDev &P = *a.wl;
    T u4[4], h0[6], pd[6], w0[6];
#pragma unroll
    for (int j = 0; j < 4; ++j) u4[j] = GLD(a.wlu, j);
    const T mbg = T(P.Md[2]) * prm.g;
#pragma unroll
    for (int c = 0; c < 3; ++c) { h0[c] = R0[2 + 3 * c] * mbg; h0[3 + c] = T(0); }
#pragma unroll
    for (int i = 0; i < NY; ++i) pd[i] = T(P.Md[i]) * acc[i];
    wl_step(P, u4, h0, pd, w0);
#pragma unroll
    for (int j = 0; j < 4; ++j) GLD(a.wlu, j) = u4[j];
    if (a.wlw) {
#pragma unroll
      for (int i = 0; i < 6; ++i) GLD(a.wlw, i) = w0[i];
    }
    const T aT0 = w0[2] / T(P.Md[2]);
    if (aT0 >= T(0)) T0next = aT0;
  }
  // controller record + outputs back to HBM (fp64: x, y, z are already in their rows unless cold-started)
  if (ASM32 || !has_sol) {
#pragma unroll
    for (int j = 0; j < NX; ++j) { const T v = has_sol ? XV(j) : T(0); GLD(a.ctrl, j) = v; }
#pragma unroll
    for (int i = 0; i < NC; ++i) { const T v = has_sol ? YV(i) : T(0); GLD(a.ctrl, NX + i) = v; }
#pragma unroll
    for (int i = 0; i < NC; ++i) { const T v = has_sol ? ZV(i) : T(0); GLD(a.ctrl, NX + NC + i) = v; }
  }
  GLD(a.ctrl, NX + 2 * NC) = T0next;
#pragma unroll
  for (int i = 0; i < 3; ++i) GLD(a.out, i) = uq[i];
#pragma unroll
  for (int i = 0; i < NY; ++i) GLD(a.out, 3 + i) = acc[i];
  if (a.status) a.status[bb] = status;
  if (a.info) { GLD(a.info, 0) = pri_res; GLD(a.info, 1) = dua_res; }

  UMPC_TMARK(5);
  // ---- plant: template/uprightmpc2.py:148-151 ----
  if (prm.nsub > 0) {
    T Ib[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) Ib[i] = a.Ib ? GLD(a.Ib, i) : prm.Ib[i];
    const T gain = a.gain ? a.gain[bb] : T(1);
    // the harness clips what the PLANT sees; the controller's return value stays unclipped
    const T uc[3] = {uq[0], umpc_min(umpc_max(uq[1], -prm.taulim), prm.taulim),
                     umpc_min(umpc_max(uq[2], -prm.taulim), prm.taulim)};
    T s_err = a.stats ? GLD(a.stats, 0) : T(0), s_eff = a.stats ? GLD(a.stats, 1) : T(0);
    const T Ibinv[3] = {T(1) / Ib[0], T(1) / Ib[1], T(1) / Ib[2]};
#pragma nounroll
    for (int s = 0; s < prm.nsub; ++s) {
      plant_step(p0, R0, dq0, uc, prm.dtsim, Ib, Ibinv, gain, prm.plant_mode);
      s_err += p0[0] * p0[0] + p0[1] * p0[1] + p0[2] * p0[2];
      s_eff += uc[1] * uc[1] + uc[2] * uc[2];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) GLD(a.state, i) = p0[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) GLD(a.state, 3 + i) = R0[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) GLD(a.state, 12 + i) = dq0[i];
    if (a.stats) { GLD(a.stats, 0) = s_err; GLD(a.stats, 1) = s_eff; }
  }
#ifdef UMPC_PHASE_TIMING
  UMPC_TMARK(6);
#pragma unroll
  for (int i = 0; i < 6; ++i) GLD(a.out, 3 + i) = T(tmark[i + 1] - tmark[i]);
#ifdef UMPC_ASM64_TIMING   /* header generated with UMPC_ASM64_TIMING=1: the block's own stamps, LDS words 297..302 */
  if constexpr (ASM64) {
    const long long *st = asm64_stamps;
    // out rows 0..2: block prologue, (first iteration .. loop), epilogue; rows 3.. keep the phase intervals
    GLD(a.out, 0) = T(st[1] - st[0]); GLD(a.out, 1) = T(st[4] - st[1]); GLD(a.out, 2) = T(st[5] - st[4]);
  }
#endif
#endif
#undef UMPC_TMARK
#undef GLD
#undef UMPC_PHASE_FENCE
#undef UMPC_SCHED_FENCE
#undef Q_
#undef DI_
#undef LX_
#undef RINV3_
#undef RHO3_
#undef LO3_
#undef UP3_
#undef W_
#undef X_
#undef Y_
#undef Z_
#undef UMPC_CAPTURE_X
#undef LDSW
#undef XV
#undef YV
#undef ZV
#undef PC_DS_
#undef PC_ES_
#undef PC_XP_
#undef PC_DY_
#undef PXRAW
#undef PXRAW_OF
}

}  // namespace umpc
