/*
 * TEST INFRASTRUCTURE -- host side of the reference's generated planar controller (planar/code: emosqp, OSQP 0.5.0,
 * EMBEDDED 1, DFLOAT; n = 46, m = 82, planar/code/include/workspace.h:1068), the role planar/mcuqp/main.cpp plays
 * on the STM32F4: it owns the translation unit that includes the generated workspace (main.cpp:8-13) and calls
 * osqp_solve on it (main.cpp:131); the vector updates used here are that library's API (osqp.c:756, 785, 1346, 1468).
 * This file is OUR host: getters
 * for the generated problem data and one solve entry. It contains no reference code; oracle/Makefile compiles the
 * reference's sources where they lie (target _ref/libplanar_ref.so).
 */
#include <string.h>
/* the generated workspace is C++ as committed ("(linsys_solver_type)0", workspace.h:1071) and main.cpp:8-11 includes
 * it under the same pragma */
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wnarrowing"
#include "workspace.h"
#pragma GCC diagnostic pop
#include "osqp.h"

extern "C" {

int planar_dims(int *out) { /* n, m, nnz(P), nnz(A), n+m, nnz(L), max_iter, check_termination, scaling, warm_start */
  out[0] = (int)data.n; out[1] = (int)data.m; out[2] = (int)Pdata.p[data.n]; out[3] = (int)Adata.p[data.n];
  out[4] = (int)(data.n + data.m); out[5] = (int)linsys_solver_L.p[data.n + data.m];
  out[6] = (int)settings.max_iter; out[7] = (int)settings.check_termination; out[8] = (int)settings.scaling;
  out[9] = (int)settings.warm_start; out[10] = (int)settings.scaled_termination;
  return 11;
}
void planar_settings(float *out) { /* rho sigma eps_abs eps_rel eps_prim_inf eps_dual_inf alpha */
  out[0] = settings.rho; out[1] = settings.sigma; out[2] = settings.eps_abs; out[3] = settings.eps_rel;
  out[4] = settings.eps_prim_inf; out[5] = settings.eps_dual_inf; out[6] = settings.alpha;
}
}
static void cpi(int *d, const c_int *s, int n) { for (int k = 0; k < n; ++k) d[k] = (int)s[k]; }
extern "C" {
void planar_pattern(int *P_p, int *P_i, int *A_p, int *A_i, int *perm, int *L_p, int *L_i) {
  const int n = (int)data.n, nk = (int)(data.n + data.m);
  cpi(P_p, Pdata.p, n + 1); cpi(P_i, Pdata.i, (int)Pdata.p[n]);
  cpi(A_p, Adata.p, n + 1); cpi(A_i, Adata.i, (int)Adata.p[n]);
  cpi(perm, linsys_solver_P, nk); cpi(L_p, linsys_solver_L.p, nk + 1); cpi(L_i, linsys_solver_L.i, (int)linsys_solver_L.p[nk]);
}
void planar_values(float *P_x, float *A_x, float *q, float *l, float *u, float *rho_vec, float *L_x, float *Dinv) {
  const int n = (int)data.n, m = (int)data.m, nk = n + m;
  memcpy(P_x, Pdata.x, sizeof(float) * Pdata.p[n]); memcpy(A_x, Adata.x, sizeof(float) * Adata.p[n]);
  memcpy(q, data.q, sizeof(float) * n); memcpy(l, data.l, sizeof(float) * m); memcpy(u, data.u, sizeof(float) * m);
  memcpy(rho_vec, workspace.rho_vec, sizeof(float) * m);
  memcpy(L_x, linsys_solver_L.x, sizeof(float) * linsys_solver_L.p[nk]); memcpy(Dinv, linsys_solver_Dinv, sizeof(float) * nk);
}
/* One controller call as main.cpp makes it: new q, l, u (NULL keeps the generated vectors), the iterates x, y, z
 * handed in as the warm start (the workspace's own carry over otherwise), osqp_solve. Outputs: the solution, the
 * iterates left in the workspace, info = (pri_res, dua_res, obj_val), st = (status_val, iter, return codes). */
int planar_solve(const float *q, const float *l, const float *u, const float *x0, const float *y0, const float *z0,
                 float *sol_x, float *sol_y, float *wx, float *wy, float *wz, float *inf, int *st) {
  const int n = (int)data.n, m = (int)data.m;
  int rc = 0;
  if (q) rc |= (int)osqp_update_lin_cost(&workspace, q);
  if (l && u) rc |= (int)osqp_update_bounds(&workspace, l, u) << 4;
  if (x0) memcpy(workspace.x, x0, sizeof(float) * n);
  if (y0) memcpy(workspace.y, y0, sizeof(float) * m);
  if (z0) memcpy(workspace.z, z0, sizeof(float) * m);
  st[2] = rc;
  st[3] = (int)osqp_solve(&workspace);
  st[0] = (int)workspace.info->status_val; st[1] = (int)workspace.info->iter;
  inf[0] = workspace.info->pri_res; inf[1] = workspace.info->dua_res; inf[2] = workspace.info->obj_val;
  memcpy(sol_x, workspace.solution->x, sizeof(float) * n); memcpy(sol_y, workspace.solution->y, sizeof(float) * m);
  memcpy(wx, workspace.x, sizeof(float) * n); memcpy(wy, workspace.y, sizeof(float) * m); memcpy(wz, workspace.z, sizeof(float) * m);
  return rc;
}
int planar_set(int max_iter, int check_termination) {
  int rc = (int)osqp_update_max_iter(&workspace, max_iter);
  rc |= (int)osqp_update_check_termination(&workspace, check_termination);
  return rc;
}
}
