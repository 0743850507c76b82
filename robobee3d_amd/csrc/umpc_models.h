// Vector fields of the reference's other rigid-body models and a build-defined
// classical RK4 step over them (the reference has NO integrator for either: SURVEY finding 2).
// One lane = one robot, SoA [rows][B] like everything else.
//
//  model 0  "ca6"  template/ca6dynamics.py:5-10, 35-50
//      wrenchMap(u6) = (u3L+u3R, 0, u1L+u1R, (u1L-u1R) ycp, -u1L u2L - u1R u2R, (-u3L+u3R) ycp),  ycp = 10
//      dynamicsTerms: M = diag(mb,mb,mb,ixx,iyy,izz) = diag(100,100,100,3333,3333,1000), h = (R' (0,0,mb g), 0)
//      build-defined forward dynamics M a = w(u) - h in the BODY frame; state = the controller's (p, R, dq) with
//      dq = (v_world, omega_body): dv = R (w_lin - h_lin)/mb, domega = w_ang / I (the reference's h has no
//      gyroscopic term), dR = R skew(omega), dp = v.
//  model 1  "ThrustStrokeDev"  template/FlappingModels3D.py:11-38
//      y = (p, rotvec, v, omega) [12], u = (FzL, dxL, FzR, dxR) [4]; restated exactly as written, including
//      d(rotvec)/dt = omega and omegadot = R' omegadot_b (sic); m = 0.5, Ib = diag(5e-4,5e-4,1e-3), ycp = 0.5,
//      g = 9.81.
#pragma once
#include "umpc_step.h"

namespace umpc {

template <typename T>
__device__ __forceinline__ void ca6_wrench(const T (&u)[6], T (&w)[6]) {
  const T ycp = T(10);
  const T u1L = u[0], u2L = u[1], u3L = u[2], u1R = u[3], u2R = u[4], u3R = u[5];
  w[0] = u3L + u3R;
  w[1] = T(0);
  w[2] = u1L + u1R;
  w[3] = (u1L - u1R) * ycp;
  w[4] = -u1L * u2L - u1R * u2R;
  w[5] = (-u3L + u3R) * ycp;
}

// ydot of y = (p[3], R[9] column-major, dq[6]) ; also returns h (body-frame bias) and the wrench
template <typename T>
__device__ __forceinline__ void ca6_vf(const T (&y)[18], const T (&u)[6], T (&yd)[18], T (&w)[6], T (&h)[6]) {
  const T mb = T(100), g = T(9.81e-3), I[3] = {T(3333), T(3333), T(1000)};
  const T *R = &y[3];
  ca6_wrench(u, w);
  // h = (R' (0,0,mb g), 0): third ROW of R times mb g
#pragma unroll
  for (int c = 0; c < 3; ++c) { h[c] = R[2 + 3 * c] * (mb * g); h[3 + c] = T(0); }
  T ab[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) ab[c] = (w[c] - h[c]) / mb;
#pragma unroll
  for (int i = 0; i < 3; ++i) yd[i] = y[12 + i];
  const T wx = y[15], wy = y[16], wz = y[17];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    yd[3 + r + 0] = R[r + 3] * wz - R[r + 6] * wy;
    yd[3 + r + 3] = -R[r + 0] * wz + R[r + 6] * wx;
    yd[3 + r + 6] = R[r + 0] * wy - R[r + 3] * wx;
    yd[12 + r] = (R[r] * ab[0] + R[r + 3] * ab[1]) + R[r + 6] * ab[2];
    yd[15 + r] = w[3 + r] / I[r];
  }
}

// Rodrigues rotation matrix (row-major e[r][c]) of a rotation vector (scipy Rotation.from_rotvec)
template <typename T>
__device__ __forceinline__ void rotvec_matrix(T ax, T ay, T az, T (&e)[3][3]) {
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
  e[0][0] = T(1) - b * (ay * ay + az * az);
  e[1][1] = T(1) - b * (ax * ax + az * az);
  e[2][2] = T(1) - b * (ax * ax + ay * ay);
  e[0][1] = -a * az + b * ax * ay;
  e[1][0] = a * az + b * ax * ay;
  e[0][2] = a * ay + b * ax * az;
  e[2][0] = -a * ay + b * ax * az;
  e[1][2] = -a * ax + b * ay * az;
  e[2][1] = a * ax + b * ay * az;
}

template <typename T>
__device__ __forceinline__ void tsd_vf(const T (&y)[12], const T (&u)[4], T (&yd)[12]) {
  const T m = T(0.5), g = T(9.81), ycp = T(0.5), Ib[3] = {T(5e-4), T(5e-4), T(1e-3)};
  T Rm[3][3];
  rotvec_matrix(y[3], y[4], y[5], Rm);
  const T om[3] = {y[9], y[10], y[11]};
  const T Fz = u[0] + u[2];
  // rL x FL + rR x FR with rL = (u1, ycp, 0), FL = (0,0,u0), rR = (u3, -ycp, 0), FR = (0,0,u2)
  const T tq[3] = {ycp * u[0] + (-ycp) * u[2], -(u[1] * u[0]) - (u[3] * u[2]), T(0)};
  const T Iw[3] = {Ib[0] * om[0], Ib[1] * om[1], Ib[2] * om[2]};
  const T cr[3] = {om[1] * Iw[2] - om[2] * Iw[1], om[2] * Iw[0] - om[0] * Iw[2], om[0] * Iw[1] - om[1] * Iw[0]};
  T ob[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) ob[i] = (tq[i] - cr[i]) / Ib[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) yd[i] = y[6 + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    yd[6 + i] = ((i == 2 ? -m * g : T(0)) + Rm[i][2] * Fz) / m;
    yd[9 + i] = (Rm[0][i] * ob[0] + Rm[1][i] * ob[1]) + Rm[2][i] * ob[2];  // wRotb.T @ omegadotb
  }
}

// classical RK4 over either vector field (NY = 18 / 12), inputs held over the step
template <typename T, int NYV, typename F>
__device__ __forceinline__ void rk4_step(T (&y)[NYV], T dt, F &&vf) {
  T y0[NYV], ys[NYV], acc[NYV], k[NYV];
#pragma unroll
  for (int i = 0; i < NYV; ++i) y0[i] = y[i];
  const T cs[4] = {T(0), T(0.5), T(0.5), T(1)}, wt[4] = {T(1), T(2), T(2), T(1)};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int i = 0; i < NYV; ++i) ys[i] = s ? y0[i] + cs[s] * dt * k[i] : y0[i];
    vf(ys, k);
#pragma unroll
    for (int i = 0; i < NYV; ++i) acc[i] = s ? acc[i] + wt[s] * k[i] : k[i];
  }
#pragma unroll
  for (int i = 0; i < NYV; ++i) y[i] = y0[i] + dt * acc[i] / T(6);
}

}  // namespace umpc
