/*
 * TEST INFRASTRUCTURE (part of the CPU oracle) — plant step, included twice by
 * umpc_oracle.c with PR = real and PR = double.
 *
 * Restates template/genqp.py:24-41:
 *   quadrotorNLVF (24-30): dv = u0 * Rb e3 - (0,0,9.81e-3);
 *                          dw = Ib^-1 (-w x (Ib w) + (u1,u2,0))
 *   quadrotorNLDyn (32-41): p2 = p + dt v; Rb2 = Rb expm(skew(w) dt); dq2 = dq + dt ddq
 * scipy's Pade expm of a 3x3 skew matrix is the Rodrigues rotation; it is
 * evaluated here in closed form (agreement with scipy to fp64 round-off is
 * pinned by tests/golden/plant.npz).
 * The gravity constant is hard-coded 9.81e-3 in the reference plant (not the
 * controller's g parameter); kept so.
 */
#ifndef PR
#error "define PR and PFN before including plant_impl.h"
#endif

/* vector field: ddq[6] from (R col-major, dq, u) */
static void PFN(plant_vf)(const PR R[9], const PR dq[6], const PR u[3],
                          const PR Ib[3], PR gain, PR ddq[6]) {
  const PR wx = dq[3], wy = dq[4], wz = dq[5];
  const PR T = gain * u[0];
  /* Rb @ e3 = third column of Rb */
  ddq[0] = T * R[6] - (PR)0;
  ddq[1] = T * R[7] - (PR)0;
  ddq[2] = T * R[8] - (PR)9.81e-3;
  /* w x (Ib w) */
  const PR hx = Ib[0] * wx, hy = Ib[1] * wy, hz = Ib[2] * wz;
  const PR cx = wy * hz - wz * hy;
  const PR cy = wz * hx - wx * hz;
  const PR cz = wx * hy - wy * hx;
  ddq[3] = (-cx + u[1]) / Ib[0];
  ddq[4] = (-cy + u[2]) / Ib[1];
  ddq[5] = (-cz + (PR)0) / Ib[2];
}

/* R <- R * expm(skew(w) * h) */
static void PFN(plant_rot)(PR R[9], const PR w[3], PR h) {
  const PR ax = w[0] * h, ay = w[1] * h, az = w[2] * h;
  const PR t = ax * ax + ay * ay + az * az; /* theta^2 */
  PR a, b; /* sin(th)/th, (1-cos(th))/th^2 */
  if (t < (PR)1e-2) {
    a = (PR)1 - t * ((PR)1 / 6 - t * ((PR)1 / 120 - t * ((PR)1 / 5040 - t * ((PR)1 / 362880))));
    b = (PR)0.5 - t * ((PR)1 / 24 - t * ((PR)1 / 720 - t * ((PR)1 / 40320 - t * ((PR)1 / 3628800))));
  } else {
    const PR th = (sizeof(PR) == 4) ? (PR)sqrtf((float)t) : (PR)sqrt((double)t);
    a = (sizeof(PR) == 4) ? (PR)sinf((float)th) / th : (PR)sin((double)th) / th;
    b = (sizeof(PR) == 4) ? ((PR)1 - (PR)cosf((float)th)) / t : ((PR)1 - (PR)cos((double)th)) / t;
  }
  /* E = I + a K + b K^2, K = skew(ax,ay,az); row-major e[r][c] */
  PR e[3][3];
  e[0][0] = (PR)1 - b * (ay * ay + az * az);
  e[1][1] = (PR)1 - b * (ax * ax + az * az);
  e[2][2] = (PR)1 - b * (ax * ax + ay * ay);
  e[0][1] = -a * az + b * ax * ay;
  e[1][0] = a * az + b * ax * ay;
  e[0][2] = a * ay + b * ax * az;
  e[2][0] = -a * ay + b * ax * az;
  e[1][2] = -a * ax + b * ay * az;
  e[2][1] = a * ax + b * ay * az;
  PR Rn[9];
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r)
      Rn[r + 3 * c] = R[r + 0] * e[0][c] + R[r + 3] * e[1][c] + R[r + 6] * e[2][c];
  for (int i = 0; i < 9; ++i) R[i] = Rn[i];
}

static void PFN(plant_step)(PR p[3], PR R[9], PR dq[6], const PR u[3], PR dt,
                            const PR Ib[3], PR gain, int mode) {
  if (mode == 0) {
    /* reference step: template/genqp.py:32-41 */
    PR ddq[6];
    PFN(plant_vf)(R, dq, u, Ib, gain, ddq);
    for (int i = 0; i < 3; ++i) p[i] = p[i] + dt * dq[i];
    PFN(plant_rot)(R, &dq[3], dt);
    for (int i = 0; i < 6; ++i) dq[i] = dq[i] + dt * ddq[i];
  } else {
    /* build-defined classical RK4 on y = (p, R, dq); dR/dt = R skew(w) */
    PR k[4][18], y0[18], ys[18];
    for (int i = 0; i < 3; ++i) y0[i] = p[i];
    for (int i = 0; i < 9; ++i) y0[3 + i] = R[i];
    for (int i = 0; i < 6; ++i) y0[12 + i] = dq[i];
    const PR cs[4] = {(PR)0, (PR)0.5, (PR)0.5, (PR)1};
    for (int s = 0; s < 4; ++s) {
      for (int i = 0; i < 18; ++i) ys[i] = s ? y0[i] + cs[s] * dt * k[s - 1][i] : y0[i];
      const PR *Rs = &ys[3], *dqs = &ys[12];
      for (int i = 0; i < 3; ++i) k[s][i] = dqs[i];
      const PR wx = dqs[3], wy = dqs[4], wz = dqs[5];
      for (int r = 0; r < 3; ++r) { /* (R K)[r][c], K = skew(w) */
        k[s][3 + r + 0] = Rs[r + 3] * wz - Rs[r + 6] * wy;
        k[s][3 + r + 3] = -Rs[r + 0] * wz + Rs[r + 6] * wx;
        k[s][3 + r + 6] = Rs[r + 0] * wy - Rs[r + 3] * wx;
      }
      PFN(plant_vf)(Rs, dqs, u, Ib, gain, &k[s][12]);
    }
    for (int i = 0; i < 18; ++i)
      ys[i] = y0[i] + dt * (k[0][i] + (PR)2 * k[1][i] + (PR)2 * k[2][i] + k[3][i]) / (PR)6;
    for (int i = 0; i < 3; ++i) p[i] = ys[i];
    for (int i = 0; i < 9; ++i) R[i] = ys[3 + i];
    for (int i = 0; i < 6; ++i) dq[i] = ys[12 + i];
  }
}
