/*
 * libumpc_mi355x.so -- C ABI of the MI355X-native uprightmpc2 path.
 *
 * Part 1 re-exports the reference's own three entry points with identical
 * signatures and semantics (avikde/robobee3d,
 * template/uprightmpc2/uprightmpc2.h:27-49), so every existing host of the
 * reference (pybind module py/uprightmpc2py.cpp:30-52, Simulink S-function
 * legacy_code_gen.m:6, MCU loop g4bee/app/loop_update.cpp:36,54) links
 * unchanged. Unlike the reference this library does NOT import `matMult`
 * (matmult.h:35) and keeps no global solver workspace: any number of
 * UprightMPC_t controllers may coexist (the reference allows one per process,
 * workspace.c:2620).
 *
 * Part 2 is additive: batched controllers (one GPU lane per robot) that run
 * the same step for B independent robots, optionally fused with the
 * rigid-body plant (template/genqp.py:24-41) into a closed loop
 * (template/uprightmpc2.py:120-154).
 *
 * Plain C types only; device buffers are raw HIP device pointers; `stream` is a
 * hipStream_t passed as void* (NULL = default stream). No call allocates or
 * synchronises except where stated.
 */
#ifndef UMPC_MI355X_H
#define UMPC_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* Part 1: the reference ABI (template/uprightmpc2/uprightmpc2.h)      */
/* ------------------------------------------------------------------ */
#define UMPC_N 3
#define UMPC_NY 6
#define UMPC_NU 3
#define UMPC_NX (UMPC_N * (2 * UMPC_NY + UMPC_NU))
#define UMPC_NC (2 * UMPC_N * UMPC_NY + UMPC_N)
#define UMPC_nAdata 48

/* Caller-allocated POD, same field layout as uprightmpc2.h:27-43 (1308 B):
 * the reference's pybind `vectors()/matrices()` read l,u,q,Px_data,Ax_data,
 * Ax_idx straight out of it (py/uprightmpc2py.cpp:46-51), so umpcUpdate keeps
 * them filled. The two words of `smin` -- which no reference source reads or
 * writes -- carry an opaque controller id set by umpcInit, so the POD may be
 * copied or moved by its host (the pybind class holds it by value) without
 * losing the warm start kept on the device. `T0` is the thrust accumulator of
 * record, as in the reference: a host edit between calls is honoured. */
typedef struct {
  float dt, g, Tmax;
  float Qyr[6], Qyf[6], Qdyr[6], Qdyf[6], R[3];
  float smin[3], smax[3];
  float e3h[3 * 3];
  float e3hIbi[3 * 3];
  float l[UMPC_NC], u[UMPC_NC], q[UMPC_NX];
  float Px_data[UMPC_NX];
  float Ax_data[UMPC_nAdata];
  int Ax_idx[UMPC_nAdata], nAxT0dt, nAxdt;
  float c0[UMPC_NY];
  float T0;
} UprightMPC_t;

/* uprightmpc2.h:45 / uprightmpc2.c:19-118 */
void umpcInit(UprightMPC_t *up, float dt, float g, float TtoWmax, float ws, float wds,
              float wpr, float wpf, float wvr, float wvf, float wthrust, float wmom,
              const float Ib[/* 3 */], int maxIter);

/* uprightmpc2.h:47 / uprightmpc2.c:209-272. R0 column-major. actualT0 >= 0
 * overrides the internal thrust accumulator. Returns 0 (the reference returns
 * osqp_solve's exit flag, which is 0 on every reachable path). Synchronous. */
int umpcUpdate(UprightMPC_t *up, float uquad[/* 3 */], float accdes[/* 6 */],
               const float p0[/* 3 */], const float R0[/* 9 */], const float dq0[/* 6 */],
               const float pdes[/* 3 */], const float dpdes[/* 3 */],
               const float sdes[/* 3 */], float actualT0);

/* uprightmpc2.h:49 / uprightmpc2.c:275-284: lazily initialised singleton. */
void umpcS(float uquad_y1[/* 3 */], float accdes_y2[/* 6 */], const float p0_u1[/* 3 */],
           const float R0_u2[/* 9 */], const float dq0_u3[/* 6 */], const float pdes_u4[/* 3 */],
           const float dpdes_u5[/* 3 */], const float sdes_u6[/* 3 */], float dt_u7, float g_u8,
           float TtoWmax_u9, float ws_u10, float wds_u11, float wpr_u12, float wpf_u13,
           float wvr_u14, float wvf_u15, float wthrust_u16, float wmom_u17,
           const float Ib_u18[/* 3 */], int maxIter_u19, float actualT0_u20);

/* OSQP status of the most recent umpcUpdate on `up` (the reference leaves it
 * in its global workspace.info->status_val; constants.h:18-30). */
int umpcLastStatus(const UprightMPC_t *up);
/* Releases the device-side state attached to `up` (optional: umpcInit on a POD that already carries a live
 * controller releases the previous one first, so re-initialising in a gain sweep or at a Simulink / MCU restart does
 * not accumulate device memory or streams). */
void umpcRelease(UprightMPC_t *up);
/* Number of drop-in controllers currently holding device state (diagnostics / tests). */
int umpcLiveControllers(void);
/* Opt-in reference compatibility switches of ONE controller (flags OR-ed; returns the previous flags, -1 when `up`
 * carries no live controller). A controller starts with the flags of the environment variable UMPC_COMPAT (decimal)
 * at umpcInit time, 0 when it is unset.
 *   UMPC_COMPAT_BOUNDS_REJECT  reproduce osqp_update_bounds' early return (template/uprightmpc2/osqp.c:801-808) whose
 *     value umpcUpdate drops (uprightmpc2.c:246): when ANY assembled l[i] > u[i] (reachable with TtoWmax < 0) NO bound
 *     of that call is applied and the step solves with the bounds the reference's workspace still holds -- Tmax is fixed
 *     by umpcInit in this library, so every call of such a controller is rejected and those are the generated
 *     workspace's placeholder l = 0, u = 1e30 on all 39 rows (workspace.c:476-557), every row an inequality at
 *     rho = 0.1 -- while q, P and A are the new ones; up->l / up->u still show the assembled
 *     (crossed) pair, as in the reference. Such a call runs on the general-structure solver (Part 5), not on the
 *     specialised stream (its dynamics rows are hard-wired equalities): milliseconds, not 0.1 ms. Without the flag the
 *     crossed pair is applied as assembled. tests/golden/bounds_reject.npz pins both behaviours. */
#define UMPC_COMPAT_BOUNDS_REJECT 1
int umpcSetCompat(UprightMPC_t *up, int flags);

/* ------------------------------------------------------------------ */
/* Part 2: batched controllers                                         */
/* ------------------------------------------------------------------ */
#define UMPC_F32 0
#define UMPC_F64 1

/* rows of the SoA device arrays (every array is [rows][B], robot index fastest) */
#define UMPC_STATE_ROWS 18 /* p(3), R column-major (9), dq = (v_world, omega_body) (6) */
#define UMPC_CTRL_ROWS 127 /* x(45) y(39) z(39) T0(1) Eprev(3): warm start of the solver */
#define UMPC_REF_ROWS 9    /* pdes(3) dpdes(3) sdes(3) */
#define UMPC_OUT_ROWS 9    /* uquad(3) = (specific thrust, 2 moments), accdes(6) */
#define UMPC_STAT_ROWS 2   /* sum |p|^2, sum |tau|^2 over plant substeps (logMetric, uprightmpc2.py:161-175) */

typedef struct {
  /* createMPC / umpcInit arguments (template_controllers.py:260-279) */
  double dt, g, TtoWmax, ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom;
  double Ib[3];
  int maxIter; /* fixed ADMM iteration count (50 in the reference) */
  /* closed loop (template/uprightmpc2.py:87, 148-151) */
  double dtsim;   /* plant substep (0.2) */
  double taulim;  /* moment clip (100) */
  int nsub;       /* plant substeps per MPC step (25); 0 = controller only */
  int plant_mode; /* 0 = reference Euler + expm step, 1 = RK4 on the same vector field */
} umpc_batch_params_t;

typedef struct umpc_batch umpc_batch_t;

/* Fills `prm` with the reference defaults (createMPC + controlTest). */
void umpcBatchDefaultParams(umpc_batch_params_t *prm);

/* Creates a batched controller for B robots of scalar type dtype on the
 * current HIP device. Returns NULL on error (umpcLastError()). The handle owns
 * one scratch workspace of 559 rows x B scalars (hipMalloc here, hipFree in
 * umpcBatchDestroy); every other array is caller-provided. */
umpc_batch_t *umpcBatchCreate(const umpc_batch_params_t *prm, int B, int dtype);
void umpcBatchDestroy(umpc_batch_t *h);

/* Writes the cold-start controller record (x=y=z=0, T0=0, Eprev=1) for B robots. */
int umpcBatchInitCtrl(umpc_batch_t *h, void *ctrl, void *stream);

/* K closed-loop MPC steps for every robot in ONE launch. Each step = QP
 * assembly + 10 Ruiz passes + LDL' + maxIter ADMM iterations + status +
 * extraction (= one umpcUpdate), then nsub plant substeps with the moments
 * clipped. Device pointers, SoA [rows][B]:
 *   state   in/out (not written when nsub == 0)
 *   ctrl    in/out
 *   ref     in
 *   actualT0 in, [B] or NULL (values >= 0 override T0 before the first step)
 *   Ib      in, [3][B] or NULL (per-robot inertia for controller and plant)
 *   gain    in, [B] or NULL (per-robot plant thrust gain, Monte-Carlo mass sweep)
 *   out     out
 *   stats   in/out or NULL (accumulated)
 *   status  out int32 [B] or NULL (OSQP status of the last step)
 *   info    out [2][B] or NULL (pri_res, dua_res of the last step)
 * Asynchronous on `stream`. Returns 0 or a hipError_t. */
int umpcBatchRollout(umpc_batch_t *h, int K, void *state, void *ctrl, const void *ref,
                     const void *actualT0, const void *Ib, const void *gain, void *out,
                     void *stats, int32_t *status, void *info, void *stream);

/* One controller step without plant (= umpcUpdate for B robots): `state` is
 * only read. */
int umpcBatchUpdate(umpc_batch_t *h, const void *state, void *ctrl, const void *ref,
                    const void *actualT0, const void *Ib, void *out, int32_t *status,
                    void *info, void *stream);

/* Plant only: nsub substeps of the rigid-body model under inputs u[3][B]. */
int umpcBatchPlant(umpc_batch_t *h, int nsub, void *state, const void *u, const void *Ib,
                   const void *gain, void *stream);

/* Debug/parity: QP assembly for B robots, raw l[39],u[39],q[45],Px[45],Ax[48] rows. */
int umpcBatchAssemble(umpc_batch_t *h, const void *state, const void *ctrl, const void *ref,
                      const void *Ib, void *l, void *u, void *q, void *Px, void *Ax, void *stream);

/* Reference generators evaluated on device at every MPC fire (template/flight_tasks.py:6-49,
 * called at template/uprightmpc2.py:124-131). task 0 (default): `ref` rows are (pdes, dpdes, sdes).
 * task != 0: `ref` rows 0..2 hold initialPos and the reference is generated at time
 *   t = t_ms + step * nsub * dtsim   (umpcBatchTime() advances with every rollout):
 *   1 helix       params (trajAmp, trajFreq [Hz], dz, useY)
 *   2 straightAcc params (tduration, vdes)
 *   3 flip        params (tstart, tend)
 *   4 perch       params (tend, trotstart, trotend, vdes) */
#define UMPC_TASK_REF 0
#define UMPC_TASK_HELIX 1
#define UMPC_TASK_STRAIGHTACC 2
#define UMPC_TASK_FLIP 3
#define UMPC_TASK_PERCH 4
int umpcBatchSetTask(umpc_batch_t *h, int task, const double params[/* 4 */], double t_ms);
double umpcBatchTime(const umpc_batch_t *h);
/* Per-robot objective weights for gain sweeps (template/uprightmpc2.py:272-303): device table
 * [8][B] = (ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom) in the handle's dtype, or NULL for the
 * batch-constant createMPC weights. The pointer is kept, not copied: the table must stay allocated (and may be
 * rewritten between launches) until it is replaced or the handle destroyed. Every weight must be > 0 (the step
 * recovers the Ruiz scaling from the equilibrated diagonal of P); the table is checked once here (synchronous
 * copy), umpcBatchCreate checks the batch-constant ones. Returns 0, -1 (bad weights) or a hipError_t. */
int umpcBatchSetWeights(umpc_batch_t *h, const void *weights);

/* Step-kernel choice. 0 (default): automatic. fp32: the all-assembly kernel (robobee3d_amd/asmstep.py: phase A, ADMM
 * loop, phase C and the plant as one generated gfx950 stream) whenever the call is inside its scope (no task
 * generator, batch-constant weights, no WL coupling, maxIter >= 1), else the C++ kernel with the assembly ADMM loop.
 * fp64: the C++ kernel with L and 1/D in LDS and the Ruiz passes and the ADMM phase as generated fp64 assembly
 * (robobee3d_amd/asmgen64.py; one workgroup per CU at a time, maxIter >= 1, batches up to ~4.8e5 robots), else the all-C++
 * kernel.
 * In its scope the fp32 all-assembly stream exists in two forms with equal results up to rounding: one LANE per robot
 * (64 robots per wavefront: throughput; B = 65 536 is one wave per SIMD) and one lane QUAD per robot (robobee3d_amd/
 * asmquad.py: 16 robots per wavefront, ADMM iterations 2.. split over three lanes, ~0.6x the instructions per step:
 * latency). Automatic = the quad form for B <= 16 384 (and for the B = 1 drop-in of Part 1), the lane form above.
 * 1: always the C++ kernel -- fp32 around the assembly ADMM loop, fp64 with the C++ loop (ablation / cross-check).
 * 2 / 3: the assembly path in its lane / quad form whatever the batch size (a run that must equal another batch size's
 * run bit for bit -- a shard against the whole -- pins the form). fp64 has the same two forms of its assembly ADMM phase
 * (robobee3d_amd/asmquad64.py; automatic = the quad form for B <= 4 096, one workgroup per CU in one round). */
int umpcBatchSetStepKernel(umpc_batch_t *h, int mode);
/* The size of the WHOLE job this handle's batch is a block of (contiguous blocks of robots per rank; default
 * = the handle's own B). The automatic choice between the lane and the quad form is made from THIS number, so that a
 * shard runs the instruction stream the undivided batch would run and a sharded job equals the single-GPU job bit for
 * bit whatever the partition (65 536 robots as 8 blocks of 8 192 stay on the lane form). global_B < B is refused. */
int umpcBatchSetGlobalBatch(umpc_batch_t *h, long long global_B);
long long umpcBatchGlobalBatch(const umpc_batch_t *h);

/* Static facts */
int umpcBatchSize(const umpc_batch_t *h);
int umpcBatchDtype(const umpc_batch_t *h);
const int *umpcAxIdx(void);    /* 48 entries, uprightmpc2.c:65-113 */
const int *umpcKKTPerm(void);  /* 84 entries, the build's own elimination order */
int umpcNnzL(void);
const char *umpcLastError(void);
/* name of the kernel a DEFAULT rollout of this dtype dispatches (environment overrides included), and of the kernel
 * the handle's last umpcBatchRollout / umpcBatchUpdate actually dispatched ("" before the first launch): bench.py and
 * the tests attribute timings and profiles through the second one */
const char *umpcKernelName(int dtype, int plant_mode);
const char *umpcBatchKernelName(const umpc_batch_t *h);

/* The reactive baseline of the reference's gain sweeps: reactiveController (template/template_controllers.py:
 * 282-296) inside controlTest(useMPC=False) (template/uprightmpc2.py:121-151): `nsteps` plant substeps of dtsim,
 * the controller evaluated every `every` substeps (reference: 1), moments clipped at +-taulim, the handle's task
 * generator / plant mode / time. gains [6][B] = (kpos0, kpos1, kz0, kz1, ks0, ks1) or NULL (the reference's
 * defaults); out [3][B] last command or NULL; stats as in umpcBatchRollout. */
int umpcBatchReactive(umpc_batch_t *h, int nsteps, int every, void *state, const void *ref, const void *gains,
                      const void *Ib, const void *thrust_gain, void *out, void *stats, void *stream);
/* out [9][B] = (pdes, dpdes, sdes) of the handle's task (template/flight_tasks.py:6-49) at time t_ms, what the
 * step kernel evaluates at an MPC fire; ref as in umpcBatchRollout (rows 0..2 = initialPos for a task). */
int umpcBatchTaskReference(umpc_batch_t *h, double t_ms, const void *ref, void *out, void *stream);

/* ------------------------------------------------------------------ */
/* Part 3: wrench-linearisation step (the consumer of accdes)          */
/* template/uprightmpc2/funapprox.h:18-54, funapprox.c:102-176          */
/* ------------------------------------------------------------------ */
#define NDELU 4
typedef struct {
  int k;
  float a0;
  float a1[NDELU];
  float A2[NDELU * NDELU];
} FunApprox_t; /* funapprox.h:18-23 */

typedef struct {
  float u0[NDELU], umin[NDELU], umax[NDELU], dumax[NDELU];
  float Qw[6 * 6];
  FunApprox_t fa[6];
} WLCon_t; /* funapprox.h:37-41; caller-allocated, holds ALL state (u0) like the reference */

/* funapprox.h:43 / funapprox.c:102-116. popts: 6 x (a0, a1[4], upper-triangular A2 row-major [10]). */
void wlConInit(WLCon_t *wl, const float u0[/* 4 */], const float umin[/* 4 */], const float umax[/* 4 */],
               const float dumax[/* 4 */], const float Qw[/* 6 */], float controlRate,
               const float popts[/* 90 */]);
/* funapprox.h:45 / funapprox.c:118-165: w0 = w(u0); one projected-gradient step of
 * |w(u) - h0 - pdotdes|^2_Qw with step 1e3, clipped to the rate limit and frozen at the box. */
void wlConUpdate(WLCon_t *wl, float u1[/* 4 */], float w0[/* 6 */], const float h0[/* 6 */],
                 const float pdotdes[/* 6 */]);
/* funapprox.h:48 / funapprox.c:171-176 */
void wlconS(float u1_y1[/* 4 */], float w0_y2[/* 6 */], const float u0init_u1[/* 4 */],
            const float umin_u2[/* 4 */], const float umax_u3[/* 4 */], const float dumax_u4[/* 4 */],
            const float Qw_u5[/* 6 */], float controlRate_u6, const float popts_u7[/* 90 */],
            const float h0_u8[/* 6 */], const float pdotdes_u9[/* 6 */]);

/* Batched: `wl` supplies limits, weights and wrench-map coefficients for every robot (its u0 is
 * ignored); u [4][B] is the per-robot input state (in: u0, out: u1), h0 / pdotdes [6][B] in,
 * w0 [6][B] out. fp32 (UMPC_F32) or fp64 device arrays. Asynchronous on `stream`. */
int umpcBatchWLUpdate(const WLCon_t *wl, int B, int dtype, void *u, const void *h0, const void *pdotdes,
                      void *w0, void *stream);

/* The coupling of the two steps as every real caller of the reference wires them
 * (template/robobee_test_controllers.py:162-171, template/uprightmpc2/conn_MPC_WL.m:2-10), FUSED into the step
 * kernel: after each MPC step of umpcBatchRollout / umpcBatchUpdate
 *     h0 = (Rb' (0, 0, mb g), 0, 0, 0),  pdotdes = M0 accdes,  (u4, w0) = wlConUpdate(h0, pdotdes),
 *     actualT0 = w0[2] / M0[2,2]  -> overrides the thrust accumulator of the NEXT step when >= 0
 * with M0 = diag(Mdiag) (dynamicsTerms, template/ca6dynamics.py:5-10,44-50: (100,100,100,3333,3333,1000)) and g
 * the handle's g. wl: parameters (umin, umax, dumax, Qw, fa; its u0 is ignored), copied to the device here
 * (synchronous); NULL switches the coupling off. u4 [4][B]: per-robot WL input state, in/out, kept by pointer;
 * w0 [6][B] or NULL: wrench w(u4) evaluated by the last step. */
int umpcBatchSetWL(umpc_batch_t *h, const WLCon_t *wl, const double Mdiag[/* 6 */], void *u4, void *w0);

/* ------------------------------------------------------------------ */
/* Part 4: the reference's other rigid-body vector fields */
/* ------------------------------------------------------------------ */
/* UMPC_MODEL_CA6: template/ca6dynamics.py:35-50. y [18][B] = (p, R column-major, dq = (v_world, omega_body)),
 *   u [6][B] = (u1L,u2L,u3L,u1R,u2R,u3R); wrenchMap + M ddq = w - h (h = (R'(0,0,mb g), 0), body frame).
 * UMPC_MODEL_TSD: ThrustStrokeDev.dynamics, template/FlappingModels3D.py:19-38. y [12][B] = (p, rotvec, v, omega),
 *   u [4][B] = (FzL, dxL, FzR, dxR), restated as written.
 * nsub == 0: aux receives ydot ([18] or [12] rows; CA6 appends wrench [6] and h [6] -> 30 rows), y unchanged.
 * nsub  > 0: y advances by nsub classical RK4 steps of dt with u held (build-defined integrator: the
 *            reference has none for these models); aux unused. */
#define UMPC_MODEL_CA6 0
#define UMPC_MODEL_TSD 1
int umpcBatchModel(int model, int B, int dtype, int nsub, double dt, void *y, const void *u, void *aux,
                   void *stream);

/* ------------------------------------------------------------------ */
/* Part 5: general-structure batch QP            */
/* ------------------------------------------------------------------ */
/* The embedded-OSQP step of Part 1/2 for an ARBITRARY structure: min 1/2 x'Px + q'x s.t. l <= Ax <= u with
 * P diagonal on a subset of the columns and A sparse. This is what the reference's other MPC formulations hand
 * to `osqp.OSQP().setup / update / solve`: planar/mpc_osqp_p5f.py:87-147,172 (n = 87, m = 164 at N = 10),
 * template/genqp.py:43-168 (v1 UprightMPC, n = m = 9N), template/template_controllers.py:170-258 (UprightMPC2 at
 * any horizon N). The sparsity is analysed on the host (robobee3d_amd/qpstruct.py: KKT ordering, elimination
 * tree, LDL' schedule = what OSQP's code generator bakes into workspace.c:743-2467) and handed over as one int32
 * table blob. Per call and per robot: 10 Ruiz passes, row classification, numeric LDL', max_iter ADMM
 * iterations (no early exit), residuals / status / solution -- the call sequence of
 * template/uprightmpc2/osqp.c:288-641,752-833,1158-1266 in canonical-restart form. */
typedef struct {
  double rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
  int max_iter, scaling;
  /* 0 (default, the embedded reference: uprightmpc2.c:116-117): exactly max_iter iterations, no early exit.
   * k > 0: the pip-osqp semantics of the reference's Python twin (template_controllers.py:190-191, 216-219;
   * osqp.c:411-450): every k-th iteration update_info + check_termination at the exact tolerances, a robot that
   * meets a criterion stops iterating (per-lane mask; osqp's default k = 25, max_iter = 4000). Runs on the
   * table-driven kernel. */
  int check_termination;
  /* 0 (default; the embedded reference has none): fixed rho. k > 0: adapt_rho (auxil.c:12-82, osqp.c:482-520,
   * 1268-1330) every k iterations: rho <- rho sqrt(rel. primal / rel. dual residual) when it changes by more than
   * 5x, rho_vec by constraint type, numeric refactorisation. pip osqp derives its interval from wall-clock timings
   * (osqp.c:455-480: a multiple of check_termination), so any multiple of 25 is a behaviour the reference's
   * Python twin can show; the twin here uses 25. */
  int adaptive_rho_interval;
} umpcQPSettings;
/* the reference's generated settings (workspace.c) with umpcInit's max_iter = 50 (uprightmpc2.c:116-117) */
void umpcQPDefaultSettings(umpcQPSettings *s);
/* blob: robobee3d_amd/qpstruct.py layout (64-word header, 18 index tables); validated here. NULL on error. */
void *umpcQPCreate(const int32_t *blob, int nwords, int B, int dtype, const umpcQPSettings *settings);
void umpcQPDestroy(void *h);
int umpcQPSetMaxIter(void *h, int max_iter);
int umpcQPSetCheckTermination(void *h, int every);
int umpcQPSetAdaptiveRho(void *h, int interval);
/* For the structures known at build time (robobee3d_amd/codegen_qp.py: planar p5f N = 10, v1 N = 3, UprightMPC2
 * N = 5) umpcQPCreate selects a generated straight-line kernel (same arithmetic, literal indices). UseTables(1)
 * forces the table-driven kernel; returns the index of the specialisation or -1. KernelName: its name or "tables". */
int umpcQPUseTables(void *h, int on);
/* Kernel choice. 1 (default): one LANE per robot (the build-time specialisation if there is one, else the
 * table-driven kernel). 2: lane per robot, tables. 0: one WAVEFRONT per robot, working set in LDS, level-scheduled
 * solves (needs the working set to fit a CU's LDS; faster for the planar p5f structure at B = 16 384, slower on
 * the others measured -- DESIGN.md 10). 3: as 1 but without the assembly loop some specialisations carry (fp32 planar
 * p5f: the middle ADMM iterations as generated gfx950 assembly, robobee3d_amd/asmqp.py; results equal to rounding). */
int umpcQPSetKernel(void *h, int mode);
/* "wave", the specialisation's name ("+asm" appended when its assembly loop will run), or "tables" */
const char *umpcQPKernelName(void *h);
/* All arrays are DEVICE pointers, SoA [rows][B] of the handle's dtype:
 *   Pv [nnzP], Av [nnzA] (CSC order), q [n], l, u [m]   raw problem data                     in
 *   x [n], y [m], z [m]   OSQP's (scaled) iterates, warm start                                in/out
 *   Eprev [m]             E of the previous call (1 before the first), osqp.c:812-820          in/out
 *   sol_x [n], sol_y [m]  unscaled solution (NaN on an infeasibility status) or NULL           out
 *   status [B] int32 (OSQP codes) or NULL; info [6][B] = pri_res, dua_res, c, zero-pivot flag, iterations run,
 *                         rho updates, or NULL
 * Asynchronous on `stream`. */
int umpcQPSolve(void *h, const void *Pv, const void *Av, const void *q, const void *l, const void *u, void *x,
                void *y, void *z, void *Eprev, void *sol_x, void *sol_y, int32_t *status, void *info, void *stream);
/* out[k][b] = src[k] < 0 ? cst[k] : cst[k] * par[src[k]][b]  (k < nnz): fills a value array whose entries are
 * constants or scaled per-robot parameters, e.g. A <- kron(I,-I) + kron(eye(k=-1), Ad) | kron(.., Bd) of
 * planar/mpc_osqp_p5f.py:168-170. cst, src are device arrays of length nnz. */
int umpcQPGather(int B, int dtype, int nnz, const void *cst, const int32_t *src, const void *par, void *out,
                 void *stream);
/* The same for the entries with src[k] >= 0 only: `out` already holds the constant entries from an earlier umpcQPGather
 * with the same cst / src (the reference rewrites only the Ad / Bd blocks of its A every tick, mpc_osqp_p5f.py:168-170). */
int umpcQPGatherUpdate(int B, int dtype, int nnz, const void *cst, const int32_t *src, const void *par, void *out,
                       void *stream);
/* planar/mpc_osqp_p5f.py: getLin (:45-85) at (u[b], sigma = y[0][b], phi = y[3][b]) -> lin [5][B] =
 * (Ad[4][3], Ad[5][3], Bd[4], Bd[5], Bd[6]); mode 1 additionally applies the reference's plant tick
 * y <- y + (Ad y + Bd u) dt (:176). y [7][B], u [B]; lin may be NULL in mode 1. */
int umpcP5fStep(int B, int dtype, int mode, double dt, const void *u, void *y, void *lin, void *stream);
/* The same with ONE nominal input for the whole batch, as the reference's loop has it (unom = 15 sin(2 pi 170 t) is a
 * scalar, planar/mpc_osqp_p5f.py:157): no [B] array to fill per tick. */
int umpcP5fStepU(int B, int dtype, int mode, double dt, double u, void *y, void *lin, void *stream);
/* getLin and the A update of one tick (mpc_osqp_p5f.py:165-170) in ONE launch: umpcP5fStep[U] mode 0 followed by
 * umpcQPGather (update = 0) or umpcQPGatherUpdate (update != 0) with par = lin; the entries of the structure refer to lin
 * rows (src[k] in 0..4). u [B] or NULL (then u_all is every robot's input); lin [5][B], Av [nnz][B] as above. */
int umpcP5fLinearise(int B, int dtype, const void *u, double u_all, const void *y, void *lin, int nnz, const void *cst,
                     const int32_t *src, void *Av, int update, void *stream);
/* One tick of planar/mpc_osqp_p5f.py:157-176 in ONE launch (round 5): umpcP5fLinearise (getLin at (unom, y[0], y[3]) -> lin and
 * the state-dependent entries of Av) + umpcQPSolve + umpcP5fStepU(mode 1) (the plant tick y += (Ad y + Bd unom) dt), with the
 * first and the last folded into the prologue of the QP kernel (they depend on the previous state only). Same arguments as
 * umpcQPSolve, then the tick's: Av is rewritten in place where src[k] >= 0 (its constant entries must be there already: a
 * first tick goes through umpcP5fLinearise), ystate [7][B] is advanced, lin [5][B] may be null. Only for a handle that
 * dispatches the fp32 p5f10 assembly kernel; returns -2 (and does nothing) otherwise, so that a caller can fall back to the
 * three calls. Results are the three calls' bit for bit. */
int umpcP5fTick(void *h, const void *Pv, void *Av, const void *q, const void *l, const void *u, void *x, void *y, void *z,
                void *Eprev, void *sol_x, void *sol_y, int32_t *status, void *info, double unom, double dt, void *ystate,
                void *lin, int nnz, const void *cst, const int32_t *src, void *stream);
/* UprightMPC2 at any horizon N (template/template_controllers.py:170-258; N = 3 is Parts 1-2's specialised path):
 * assembly (updateConstraint :65-125, updateObjective :127-143 = uprightmpc2.c:121-207) and extraction (update2 /
 * getAccDes :232-250 = uprightmpc2.c:253-269) around umpcQPSolve on the structure of initConstraint (:28-63).
 *   state [18][B] (p, R column-major, dq), ref [9][B] (pdes, dpdes, sdes), T0 [B] in/out, actualT0 [B] or NULL
 *   Pv [nx], q [nx], l, u [nc]  (nx = 15 N, nc = 13 N);  par [10][B] = (dt T0, dt s0[3], dt Btau[6]) for umpcQPGather
 *   out [9][B] = (uquad[3], accdes[6]); T0 <- T0 + x[12 N]. */
typedef struct {
  double dt, g, TtoWmax, ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom, Ib[3];
} umpcNParams;
int umpcNAssemble(int B, int dtype, int N, const umpcNParams *p, const void *state, const void *ref, void *T0,
                  const void *actualT0, void *Pv, void *q, void *l, void *u, void *par, void *stream);
int umpcNExtract(int B, int dtype, int N, double dt, const void *state, const void *sol_x, void *T0, void *out,
                 void *stream);

#ifdef __cplusplus
}
#endif
#endif /* UMPC_MI355X_H */
