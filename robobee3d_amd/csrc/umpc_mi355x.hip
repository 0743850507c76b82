// libumpc_mi355x.so: HIP kernels (gfx950) + the C ABI declared in include/umpc_mi355x.h.
// No torch, no reference code, no CPU fallback: every entry point launches a kernel.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

#include "../../include/umpc_mi355x.h"
#include "umpc_step.h"
#ifndef UMPC_STEP_ASM_HEADER   // tools/build_variant.py points this at another generated stream (A/B timing)
#define UMPC_STEP_ASM_HEADER "umpc_step_asm.h"
#endif
#include UMPC_STEP_ASM_HEADER
#include "umpc_step_asm_quad.h"   // the same stream with one robot per lane QUAD (asmquad.py): the latency-bound shapes
#include "umpc_models.h"
#include "umpc_err.h"
#include "umpc_n3_general.h"   // the N = 3 QP as data for the general-structure solver (compat bounds-reject path only)

namespace {

using umpc::DevParams;
using namespace umpcgen;

constexpr int kBlock = 64;  // one wavefront per workgroup: 64 robots, no barriers anywhere

thread_local std::string g_err;
int fail(hipError_t e, const char *what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return (int)e;
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
// K closed-loop steps per robot; lane b of the grid owns robot b. One wavefront per workgroup
// (no barriers, no cross-lane traffic). fp32: the ADMM phase is the generated assembly with
// L[0..160) in LDS (40 float4 per lane = 40,960 B per workgroup -> 4 workgroups = 4 waves per CU,
// one per SIMD, which is also what 512 registers per lane allow).
// fp64: the C++ loop; LDSF (a batch of at most one wave per CU) keeps L and 1/D in LDS; LDSF + ASM64 additionally runs
// the ADMM phase as the generated fp64 assembly (umpc_admm_asm64.h), which owns the whole 160 KiB of the CU.
template <typename T, bool LDSF = false, bool ASM64 = false, bool QUAD = false>
__global__ __launch_bounds__(kBlock) void umpc_rollout_kernel(umpc::StepIO<T> a, int K, const T *actualT0, int skew_ticks) {
  constexpr bool kAsm = sizeof(T) == 4;
  static_assert(!(kAsm && LDSF) && (!ASM64 || LDSF) && (!QUAD || ASM64), "LDSF / ASM64 / QUAD are the fp64 variants");
  __shared__ float4 lds[kAsm ? (umpcasm::LDS_BYTES_PER_LANE / 16) * kBlock
                             : ASM64 ? (umpcasm64::LDS_BYTES_PER_LANE / 16) * kBlock
                             : LDSF ? ((umpcgen::NNZL + umpcgen::NK + 1) / 2) * kBlock : 1];
  // QUAD: 16 robots per wavefront, the four lanes of a quad own one robot (umpc::closed_loop_step)
  const int b = QUAD ? blockIdx.x * (kBlock / 4) + (int)(threadIdx.x >> 2) : blockIdx.x * kBlock + threadIdx.x;
  if (b >= a.B) return;
  // low 32 bits of a flat LDS pointer = the LDS byte address
  const unsigned ldsaddr = (kAsm || ASM64) ? (unsigned)(size_t)(&lds[threadIdx.x]) : 0u;
  // Every wave runs the same phases (memory-heavy hand-offs, then the ALU-only ADMM loop). Started
  // together, all 1024 resident waves hit HBM at the same instants and idle the ALUs meanwhile. For
  // multi-step launches the waves are started in `skew_groups` staggered groups so that the memory
  // phase of one group overlaps the ADMM loop of the others (costs one partial step per launch).
  if (skew_ticks > 0) {
    const unsigned g = (blockIdx.x / 8u) % 4u;  // blocks b and b+8 share an XCD: spread groups inside each
    const long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < (long long)g * skew_ticks) __builtin_amdgcn_s_sleep(64);
  }
  T *ldsw = kAsm ? reinterpret_cast<T *>(lds) + 4 * threadIdx.x : LDSF ? reinterpret_cast<T *>(lds) + 2 * threadIdx.x : nullptr;
#pragma nounroll
  for (int k = 0; k < K; ++k) umpc::closed_loop_step<T, kAsm || ASM64, LDSF, QUAD>(a, b, ldsaddr, ldsw, k, actualT0);
}

// The all-assembly fp32 fast path (asmstep.py -> umpc_step_asm.h): the whole K-step loop of one wavefront is ONE
// generated instruction stream; C++ only hands over the lane's offsets and the parameter block (kernarg).
__global__ __launch_bounds__(kBlock) void umpc_rollout_asm_kernel(const umpcasm::StepParams prm, int B, int skew_ticks,
                                                                  int skew_groups) {
  __shared__ float4 lds[(umpcasm::STEP_LDS_BYTES_PER_LANE / 16) * kBlock];
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= B) return;
  // All 1024 resident waves run the same phases; started together they hit HBM at the same instants (phase A's loads,
  // phase C's loads and stores) and idle the ALUs meanwhile, until they have drifted apart. Starting them in
  // `skew_groups` groups `skew_ticks` (100 MHz) apart desynchronises them from the first step on; the last group ends
  // (groups - 1) x skew later, which is why the skew is a fraction of ONE memory phase, not of a step.
  if (skew_ticks > 0) {
    const unsigned g = (blockIdx.x / 8u) % (unsigned)skew_groups;  // blocks b and b+8 share an XCD: spread groups inside each
    const long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < (long long)g * skew_ticks) __builtin_amdgcn_s_sleep(8);
  }
  const unsigned ldsaddr = (unsigned)(size_t)(&lds[threadIdx.x]);
  const unsigned voff = (unsigned)b * 4u;
  // the parameter block is read where it lies, in the kernarg segment (first argument, offset 0): taking &prm would
  // copy it to private memory, which scalar loads cannot reach
  (void)prm;
  const void *pp = (const void *)__builtin_amdgcn_kernarg_segment_ptr();
  UMPC_STEP_ASM(voff, ldsaddr, pp);
}

// One robot per lane QUAD (asmquad.py): 16 robots per wavefront. The four lanes of a quad get the SAME robot offset, run
// every phase redundantly (each on its own LDS slice; their global stores write identical words to identical addresses)
// and split the unknowns of ADMM iterations 2..maxIter over lanes 0..2 -- ~340 instructions per iteration instead of
// 804. For batches that cannot give every SIMD a wave of its own anyway (B <= kQuadMaxB) and for the B = 1 drop-in.
constexpr int kQuadMaxB = 16384;      // 1024 waves of 16 robots: one per SIMD
__global__ __launch_bounds__(kBlock) void umpc_rollout_asm_quad_kernel(const umpcasm::StepParams prm, int B) {
  __shared__ float4 lds[(umpcasm::STEP_LDS_BYTES_PER_LANE / 16) * kBlock];
  // XCD-aware block -> robot-group mapping: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), and a
  // wave of 16 robots touches 64 B of every SoA row -- half a 128-B line. Consecutive groups therefore go to workgroups
  // of ONE XCD (ids congruent mod 8), so that the two halves of a line meet in one L2 instead of being fetched by two
  // (measured before: 1.85 kB read per robot-step against 0.97 kB for the 64-robot waves of the lane form).
  const unsigned nb = gridDim.x, xcd = blockIdx.x % 8u, q8 = nb / 8u, r8 = nb % 8u;
  const unsigned grp = xcd * q8 + (xcd < r8 ? xcd : r8) + blockIdx.x / 8u;
  const int b = (int)grp * (kBlock / 4) + (int)(threadIdx.x >> 2);
  if (b >= B) return;                 // (whole quads: the generated stream keeps EXEC as it finds it)
  const unsigned ldsaddr = (unsigned)(size_t)(&lds[threadIdx.x]);
  const unsigned voff = (unsigned)b * 4u;
  (void)prm;
  const void *pp = (const void *)__builtin_amdgcn_kernarg_segment_ptr();
  UMPC_STEP_ASM_QUAD(voff, ldsaddr, pp);
}
// UMPC_QUAD=0 keeps every batch on the one-lane kernel, UMPC_QUAD=<n> moves the switch-over batch size (A/B timing)
static int quad_max_b() {
  static const int v = [] { const char *e_ = getenv("UMPC_QUAD"); return e_ ? atoi(e_) : kQuadMaxB; }();
  return v;
}

template <typename T>
__device__ __forceinline__ void assemble_rows(const DevParams<T> &prm, int B_, int b, const T *state, const T *ctrl,
                                              const T *refA, const T *IbA, const T *actualT0, T *l, T *u, T *q, T *Px, T *Ax);

// B = 1 drop-in (umpcUpdate): the caller-visible debug fields of UprightMPC_t (l, u, q, Px_data, Ax_data: what vectors() /
// matrices() of the reference's pybind class read, py/uprightmpc2py.cpp:46-51) are the raw assembly of this call. They do not
// depend on the step, so this one-lane kernel runs on a SECOND stream beside the step kernel; both write into the pinned,
// mapped host buffer and end with a system-scope release of a completion word that the host polls (no stream
// synchronisation on the critical path). `t0dbg` is the thrust accumulator this call assembles with (host-known: the POD's
// T0 or actualT0), so this kernel never reads the controller record the step kernel is updating.
__global__ void umpc_dropin_debug_kernel(DevParams<float> dprm, const float *state, const float *ref, const float *t0dbg,
                                         float *l, float *u, float *q, float *Px, float *Ax, unsigned *done1, unsigned seq) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  assemble_rows<float>(dprm, 1, 0, state, nullptr, ref, nullptr, t0dbg, l, u, q, Px, Ax);
  __threadfence_system();
  __hip_atomic_store(done1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Round 5: the same in ONE launch. The drop-in's step kernel is the quad stream on robot 0 = lanes 0..3 of one wavefront;
// lane 4 -- outside the quad, it never enters the stream -- assembles the debug fields FIRST (the barrier is a convergence
// point: its stores are issued before the stream starts), and the stream's own epilogue (s_waitcnt vmcnt(0), L2 write-back at
// system scope, completion word) then releases them with the results: one launch, one completion word to poll.
__global__ __launch_bounds__(kBlock) void umpc_dropin_quad_kernel(const umpcasm::StepParams prm, DevParams<float> dprm,
                                                                  const float *state, const float *ref, const float *t0dbg,
                                                                  float *l, float *u, float *q, float *Px, float *Ax) {
  __shared__ float4 lds[(umpcasm::STEP_LDS_BYTES_PER_LANE / 16) * kBlock];
  if (threadIdx.x == 4) assemble_rows<float>(dprm, 1, 0, state, nullptr, ref, nullptr, t0dbg, l, u, q, Px, Ax);
  __syncthreads();
  if (threadIdx.x >= 4) return;
  const unsigned ldsaddr = (unsigned)(size_t)(&lds[threadIdx.x]);
  const unsigned voff = 0u;           // robot 0 in all four lanes of the quad
  (void)prm;
  const void *pp = (const void *)__builtin_amdgcn_kernarg_segment_ptr();
  UMPC_STEP_ASM_QUAD(voff, ldsaddr, pp);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void umpc_plant_kernel(DevParams<T> prm, int B_, int nsub, T *state,
                                                            const T *u, const T *IbA, const T *gainA) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= B_) return;
  const size_t B = (size_t)B_;
  T p[3], R[9], dq[6], uq[3], Ib[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { p[i] = state[(size_t)i * B + b]; uq[i] = u[(size_t)i * B + b]; }
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = state[(size_t)(3 + i) * B + b];
#pragma unroll
  for (int i = 0; i < 6; ++i) dq[i] = state[(size_t)(12 + i) * B + b];
#pragma unroll
  for (int i = 0; i < 3; ++i) Ib[i] = IbA ? IbA[(size_t)i * B + b] : prm.Ib[i];
  const T gain = gainA ? gainA[b] : T(1);
  const T Ibinv[3] = {T(1) / Ib[0], T(1) / Ib[1], T(1) / Ib[2]};
#pragma nounroll
  for (int s = 0; s < nsub; ++s) umpc::plant_step(p, R, dq, uq, prm.dtsim, Ib, Ibinv, gain, prm.plant_mode);
#pragma unroll
  for (int i = 0; i < 3; ++i) state[(size_t)i * B + b] = p[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) state[(size_t)(3 + i) * B + b] = R[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) state[(size_t)(12 + i) * B + b] = dq[i];
}

template <typename T>
__global__ __launch_bounds__(kBlock) void umpc_assemble_kernel(DevParams<T> prm, int B_, const T *state,
                                                               const T *ctrl, const T *refA, const T *IbA,
                                                               const T *actualT0, T *l, T *u, T *q, T *Px, T *Ax) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= B_) return;
  assemble_rows(prm, B_, b, state, ctrl, refA, IbA, actualT0, l, u, q, Px, Ax);
}

template <typename T>
__device__ __forceinline__ void assemble_rows(const DevParams<T> &prm, int B_, int b, const T *state, const T *ctrl,
                                              const T *refA, const T *IbA, const T *actualT0, T *l, T *u, T *q, T *Px, T *Ax) {
  const size_t B = (size_t)B_;
  T p[3], R[9], dq[6], ref[9], Ibi[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) p[i] = state[(size_t)i * B + b];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = state[(size_t)(3 + i) * B + b];
#pragma unroll
  for (int i = 0; i < 6; ++i) dq[i] = state[(size_t)(12 + i) * B + b];
#pragma unroll
  for (int i = 0; i < 9; ++i) ref[i] = refA[(size_t)i * B + b];
#pragma unroll
  for (int i = 0; i < 3; ++i) Ibi[i] = T(1) / (IbA ? IbA[(size_t)i * B + b] : prm.Ib[i]);
  // ctrl == null (the drop-in's debug assembly): actualT0 IS the accumulator this call assembles with
  T T0 = ctrl ? ctrl[(size_t)(NX + 2 * NC) * B + b] : T(0);
  if (actualT0 && (!ctrl || actualT0[b] >= T(0))) T0 = actualT0[b];  // uprightmpc2.c:215-216
  umpc::RawQP<T> qp;
  const umpc::Weights<T> wt = {prm.ws, prm.wds, prm.wpr, prm.wpf, prm.wvr, prm.wvf, prm.wthrust, prm.wmom};
  umpc::assemble(prm, wt, Ibi, T0, p, R, dq, ref, qp);
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    l[(size_t)i * B + b] = qp.l[i];
    u[(size_t)i * B + b] = i < NEQ ? qp.l[i] : qp.u3[i - NEQ];
  }
#pragma unroll
  for (int j = 0; j < NX; ++j) { q[(size_t)j * B + b] = qp.q[j]; Px[(size_t)j * B + b] = qp.Px[j]; }
  // Ax_data order: [T0dt x3(N-2) | dt x6N | s0 x3N | Btau x6N], uprightmpc2.c:161-179
  int o = 0;
#pragma unroll
  for (int k = 0; k < 3 * (N - 2); ++k) Ax[(size_t)(o++) * B + b] = qp.dtT0;
#pragma unroll
  for (int k = 0; k < 6 * N; ++k) Ax[(size_t)(o++) * B + b] = prm.dt;
#pragma unroll
  for (int k = 0; k < N; ++k)
#pragma unroll
    for (int i = 0; i < 3; ++i) Ax[(size_t)(o++) * B + b] = qp.s0dt[i];
#pragma unroll
  for (int k = 0; k < N; ++k)
#pragma unroll
    for (int i = 0; i < 6; ++i) Ax[(size_t)(o++) * B + b] = qp.Btaudt[i];
}

// reactiveController (template/template_controllers.py:282-296) in the closed loop of controlTest with
// useMPC=False (template/uprightmpc2.py:121-151): the reference evaluates it at every plant substep
// (hlInterval=None); `every` > 1 holds the command for that many substeps (fixed schedule).
// gains rows: kpos[2], kz[2], ks[2] (defaults 5e-3, 5e-1 | 1e-1, 1 | 10, 1e2).
template <typename T>
__device__ __forceinline__ void reactive_controller(const T (&p)[3], const T (&R)[9], const T (&dq)[6],
                                                    const T (&pdes)[3], const T (&k)[6], T (&u)[3]) {
  T sdes[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    sdes[i] = umpc::umpc_min(umpc::umpc_max(k[0] * (pdes[i] - p[i]) - k[1] * dq[i], T(-0.5)), T(0.5));
  sdes[2] = T(1);
  T ds[3], fT[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    ds[r] = -(R[r] * (-dq[4]) + R[r + 3] * dq[3]);      // -Rb e3h omega
    fT[r] = k[4] * (R[6 + r] - sdes[r]) + k[5] * ds[r];
  }
  fT[2] = T(0);
  // fAorn = -e3h Rb' fTorn = ((Rb' f)_y, -(Rb' f)_x, 0)
  const T wx = (R[0] * fT[0] + R[1] * fT[1]) + R[2] * fT[2];
  const T wy = (R[3] * fT[0] + R[4] * fT[1]) + R[5] * fT[2];
  u[0] = k[2] * (pdes[2] - p[2]) - k[3] * dq[2];
  u[1] = wy;
  u[2] = -wx;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void umpc_reactive_kernel(DevParams<T> prm, int B_, int nsteps, int every, T t0,
                                                              T *state, const T *ref, const T *gains, const T *IbA,
                                                              const T *gainA, T *out, T *stats) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= B_) return;
  const size_t B = (size_t)B_;
  T p[3], R[9], dq[6], Ib[3], rf0[9], k[6] = {T(5e-3), T(5e-1), T(1e-1), T(1e0), T(10e0), T(1e2)}, u[3] = {T(0), T(0), T(0)};
#pragma unroll
  for (int i = 0; i < 3; ++i) p[i] = state[(size_t)i * B + b];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = state[(size_t)(3 + i) * B + b];
#pragma unroll
  for (int i = 0; i < 6; ++i) dq[i] = state[(size_t)(12 + i) * B + b];
#pragma unroll
  for (int i = 0; i < 9; ++i) rf0[i] = ref[(size_t)i * B + b];
#pragma unroll
  for (int i = 0; i < 3; ++i) Ib[i] = IbA ? IbA[(size_t)i * B + b] : prm.Ib[i];
  if (gains) {
#pragma unroll
    for (int i = 0; i < 6; ++i) k[i] = gains[(size_t)i * B + b];
  }
  const T gain = gainA ? gainA[b] : T(1);
  const T Ibinv[3] = {T(1) / Ib[0], T(1) / Ib[1], T(1) / Ib[2]};
  T s_err = stats ? stats[b] : T(0), s_eff = stats ? stats[B + b] : T(0);
#pragma nounroll
  for (int ti = 0; ti < nsteps; ++ti) {
    if (ti % every == 0) {
      T rf[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) rf[i] = rf0[i];
      umpc::task_reference(prm.task, prm.task_p, t0 + T(ti) * prm.dtsim, rf);
      const T pdes[3] = {rf[0], rf[1], rf[2]};
      reactive_controller(p, R, dq, pdes, k, u);
      u[1] = umpc::umpc_min(umpc::umpc_max(u[1], -prm.taulim), prm.taulim);
      u[2] = umpc::umpc_min(umpc::umpc_max(u[2], -prm.taulim), prm.taulim);
    }
    umpc::plant_step(p, R, dq, u, prm.dtsim, Ib, Ibinv, gain, prm.plant_mode);
    s_err += p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    s_eff += u[1] * u[1] + u[2] * u[2];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) state[(size_t)i * B + b] = p[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) state[(size_t)(3 + i) * B + b] = R[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) state[(size_t)(12 + i) * B + b] = dq[i];
  if (out) {
#pragma unroll
    for (int i = 0; i < 3; ++i) out[(size_t)i * B + b] = u[i];
  }
  if (stats) { stats[b] = s_err; stats[B + b] = s_eff; }
}

// (pdes, dpdes, sdes) of the handle's task at time t for every robot (what the step kernel evaluates at a fire)
template <typename T>
__global__ void umpc_taskref_kernel(DevParams<T> prm, int B_, T t, const T *ref, T *out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B_) return;
  const size_t B = (size_t)B_;
  T r[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) r[i] = ref[(size_t)i * B + b];
  umpc::task_reference(prm.task, prm.task_p, t, r);
#pragma unroll
  for (int i = 0; i < 9; ++i) out[(size_t)i * B + b] = r[i];
}

// Task table of the all-assembly step kernel: the time-dependent part of (pdes, dpdes, sdes) is the same for every
// robot (template/flight_tasks.py:6-49 add it to initialPos), so it is evaluated ONCE per closed-loop step of the
// launch -- 8 floats per step: dp[3], dpdes[3], sdes_x, sdes_z (sdes_y is 0 in every task) -- with the same
// task_reference() and the same fire time the C++ kernel uses per robot.
__global__ void umpc_taskf_kernel(DevParams<float> prm, int K, float t0, float *tab) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  float r[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float tnow = t0 + float(k) * (float(prm.nsub) * prm.dtsim);
  umpc::task_reference(prm.task, prm.task_p, tnow, r);
  float *o = tab + 8 * (size_t)k;
  o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3]; o[4] = r[4]; o[5] = r[5]; o[6] = r[6]; o[7] = r[8];
}

template <typename T>
__global__ void umpc_init_ctrl_kernel(int B_, T *ctrl) {
  const size_t B = (size_t)B_;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)UMPC_CTRL_ROWS * B;
       i += (size_t)gridDim.x * blockDim.x)
    ctrl[i] = (i / B) >= (size_t)(NX + 2 * NC + 1) ? T(1) : T(0);
}

template <typename T>
DevParams<T> make_dev(const umpc_batch_params_t &p) {
  DevParams<T> d;
  d.dt = (T)p.dt; d.g = (T)p.g;
  d.Tmax = (T)p.TtoWmax * (T)p.g;  // uprightmpc2.c:25
  d.wpr = (T)p.wpr; d.wpf = (T)p.wpf; d.ws = (T)p.ws; d.wvr = (T)p.wvr; d.wvf = (T)p.wvf;
  d.wds = (T)p.wds; d.wthrust = (T)p.wthrust; d.wmom = (T)p.wmom;
  for (int i = 0; i < 3; ++i) d.Ib[i] = (T)p.Ib[i];
  d.dtsim = (T)p.dtsim; d.taulim = (T)p.taulim;
  d.maxIter = p.maxIter; d.nsub = p.nsub; d.plant_mode = p.plant_mode;
  d.task = 0;
  for (int i = 0; i < 4; ++i) d.task_p[i] = T(0);
  return d;
}

}  // namespace

struct umpc_batch {
  umpc_batch_params_t prm;
  int B, dtype;
  long long global_B = 0;        // size of the whole (sharded) job, umpcBatchSetGlobalBatch; the lane / quad choice is made from it
  int task = 0;
  double task_p[4] = {0, 0, 0, 0};
  double t_ms = 0;               // time of the next MPC step (advanced by every rollout)
  const void *weights = nullptr;  // [8][B] device table or null
  void *ws;  // [WS_ROWS][B] scratch the step parks Ruiz scalings / x_prev / delta_y in
  int step_kernel = 0;            // 0 = automatic, 1 = always the C++ / loop-assembly kernel, 2 / 3 = the all-assembly stream with one lane / one lane quad per robot
  umpc::WLDev *wl = nullptr;      // device copy of the WL parameters (umpcBatchSetWL), null = no coupling
  void *wlu = nullptr, *wlw = nullptr;
  float wl_md2 = 0.f;             // M0[2,2] of the WL coupling (host copy, for the kernel's mb g)
  const char *last_kernel = "";   // the kernel the last umpcBatchRollout / umpcBatchUpdate dispatched (umpcBatchKernelName)
  float *taskf = nullptr;         // task table of the all-assembly kernel: 8 floats per step of a launch
  int taskf_cap = 0;              // ... steps it holds
};

// Parameter block of the all-assembly step kernel (umpcasm::StepParams, read by the stream with scalar loads)
static umpcasm::StepParams make_step_params(umpc_batch_t *h, int K, int nsub, void *state, void *ctrl, const void *ref,
                                            const void *actualT0, const void *Ib, const void *gain, void *out, void *stats,
                                            int32_t *status, void *info) {
  umpcasm::StepParams p;
  p.state = state; p.ctrl = ctrl; p.ref = ref; p.out = out; p.stats = stats; p.status = status;
  // (the kernel's workspace pointer is the row it parks D, E, c in: 32-bit lane offsets stay inside one array)
  p.ws = (char *)h->ws + (size_t)umpcasm::WS_DS * (size_t)h->B * 4;
  p.info = info; p.Ib = Ib; p.gain = gain; p.aT0 = actualT0;
  p.taskf = nullptr; p.weights = h->weights; p.wl = h->wl; p.wlu = h->wlu; p.wlw = h->wlw;
  p.done = nullptr; p.seq = 0;
  p.stride = h->B * 4; p.K = K; p.maxIter = h->prm.maxIter; p.nsub = nsub; p.plant = h->prm.plant_mode;
  const umpc_batch_params_t &q = h->prm;
  const float one = 1.0f;
  p.dt = (float)q.dt; p.dtg = (float)q.dt * (float)q.g; p.Tmax = (float)q.TtoWmax * (float)q.g;
  p.wpr = (float)q.wpr; p.wpf = (float)q.wpf; p.ws_ = (float)q.ws; p.wvr = (float)q.wvr; p.wvf = (float)q.wvf;
  p.wds = (float)q.wds; p.wthrust = (float)q.wthrust; p.wmom = (float)q.wmom;
  p.iwpr = one / p.wpr; p.iwpf = one / p.wpf; p.iws = one / p.ws_; p.iwvr = one / p.wvr; p.iwvf = one / p.wvf;
  p.iwds = one / p.wds; p.iwthrust = one / p.wthrust; p.iwmom = one / p.wmom;
  p.Ib0 = (float)q.Ib[0]; p.Ib1 = (float)q.Ib[1]; p.Ib2 = (float)q.Ib[2];
  p.Ibi0 = one / p.Ib0; p.Ibi1 = one / p.Ib1; p.Ibi2 = one / p.Ib2;
  p.h = (float)q.dtsim; p.hh = 0.5f * p.h; p.h6 = p.h / 6.0f; p.taulim = (float)q.taulim; p.gpl = 9.81e-3f;
  p.idt = one / p.dt;
  // h0 = Rb' (0, 0, mb g) of the WL coupling: M0[2] is float in WLDev (umpcBatchSetWL), the product as the C++ kernel forms it
  p.mbg = h->wl ? h->wl_md2 * (float)q.g : 0.0f;
  return p;
}

template <typename T>
static int launch_rollout(umpc_batch_t *h, int K, int nsub, void *state, void *ctrl, const void *ref,
                          const void *actualT0, const void *Ib, const void *gain, void *out, void *stats,
                          int32_t *status, void *info, void *stream) {
  if (!state || !ctrl || !ref || !out) { g_err = "umpcBatchRollout: null array"; return -1; }
  umpc::StepIO<T> a;
  a.prm = make_dev<T>(h->prm);
  a.prm.nsub = nsub;
  a.B = h->B;
  a.state = (T *)state; a.ctrl = (T *)ctrl; a.ref = (const T *)ref;
  a.prm.task = h->task;
  for (int i = 0; i < 4; ++i) a.prm.task_p[i] = (T)h->task_p[i];
  a.weights = (const T *)h->weights; a.t0 = (T)h->t_ms;
  if (nsub > 0) h->t_ms += (double)K * nsub * h->prm.dtsim;
  a.Ib = (const T *)Ib; a.gain = (const T *)gain; a.ws = (T *)h->ws; a.out = (T *)out; a.stats = (T *)stats;
  a.status = status; a.info = (T *)info;
  a.wl = h->wl; a.wlu = (T *)h->wlu; a.wlw = (T *)h->wlw;
  const int grid = (h->B + kBlock - 1) / kBlock;
  if constexpr (sizeof(T) == 4) {
    // all-assembly fast path: fp32, no task generator, batch-constant weights, no WL coupling, >= 1 iteration, row
    // offsets within 31 bits (either plant); UMPC_NO_ASM_STEP=1 forces the C++ / assembly-loop kernel
    static const bool no_asm = getenv("UMPC_NO_ASM_STEP") != nullptr;
    // (32-bit lane offsets inside one array: the largest is ctrl, 127 rows; the kernel's workspace pointer is the row it
    // parks D, E, c in, so the 559-row workspace does not count)
    const bool fits = (size_t)UMPC_CTRL_ROWS * (size_t)h->B * 4 < ((size_t)1 << 31);
    if (!no_asm && h->step_kernel != 1 && fits && K >= 1 && h->prm.maxIter >= 1) {
      umpcasm::StepParams p = make_step_params(h, K, nsub, state, ctrl, ref, actualT0, Ib, gain, out, stats, status, info);
      // SURVEY 8(f) options of the same stream: task generator (a table of K entries written by a K-thread kernel ahead
      // of the launch, same stream), per-robot weights, the fused WL step
      if (h->task != 0) {
        if (K > h->taskf_cap) {
          if (h->taskf) (void)hipFree(h->taskf);      // (synchronises: an earlier launch may still read the old table)
          h->taskf = nullptr; h->taskf_cap = 0;
          const int cap = K < 1024 ? 1024 : K;
          const hipError_t em = hipMalloc((void **)&h->taskf, (size_t)cap * 8 * sizeof(float));
          if (em != hipSuccess) return fail(em, "umpcBatchRollout: task table");
          h->taskf_cap = cap;
        }
        hipLaunchKernelGGL(umpc_taskf_kernel, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t)stream, a.prm, K, a.t0, h->taskf);
        p.taskf = h->taskf;
      }
      // wave-group start skew (see the kernel): UMPC_ASM_SKEW_US / UMPC_ASM_SKEW_GROUPS override the measured default
      static const int skew_us10 = [] { const char *e_ = getenv("UMPC_ASM_SKEW_US"); return e_ ? (int)(atof(e_) * 10) : 0; }();
      static const int skew_groups = [] { const char *e_ = getenv("UMPC_ASM_SKEW_GROUPS"); return e_ ? atoi(e_) : 4; }();
      const int skew_ticks = (h->B >= 32768 && K >= 2) ? skew_us10 * 10 : 0;
      if (h->step_kernel == 3 || (h->step_kernel == 0 && h->global_B <= quad_max_b())) {
        hipLaunchKernelGGL(umpc_rollout_asm_quad_kernel, dim3((h->B + kBlock / 4 - 1) / (kBlock / 4)), dim3(kBlock), 0,
                           (hipStream_t)stream, p, h->B);
        h->last_kernel = "umpc_rollout_asm_quad_kernel";
        hipError_t eq = hipGetLastError();
        return eq == hipSuccess ? 0 : fail(eq, "umpcBatchRollout");
      }
      hipLaunchKernelGGL(umpc_rollout_asm_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, p, h->B, skew_ticks,
                         skew_groups < 1 ? 1 : skew_groups);
      h->last_kernel = "umpc_rollout_asm_kernel";
      hipError_t e = hipGetLastError();
      return e == hipSuccess ? 0 : fail(e, "umpcBatchRollout");
    }
  }
  // stagger wave groups by ~1/4 step when a launch carries many steps: the last group ends 3 x skew later than the
  // first, so the stagger only pays when that tail is small against the launch (off below 64 steps)
  const char *env = getenv("UMPC_SKEW_US");
  const int skew_us = env ? atoi(env) : (K >= 64 && sizeof(T) == 4 && h->B >= 32768 ? 80 : 0);
  if constexpr (sizeof(T) == 8) {
    // fp64 (BASELINE configs[1]: B = 4096 = one wave per CU at most): L and 1/D in LDS, not in scratch
    static const bool no_ldsf = getenv("UMPC_NO_F64_LDS") != nullptr;
    static const bool no_asm64 = getenv("UMPC_NO_ASM64") != nullptr;
    // Any grid: a workgroup of this kernel owns its CU's whole LDS, so larger batches run 256 workgroups at a time; measured
    // (tools/f64_big.sh) that is still 5-13x faster than four all-C++ waves per CU: 0.42 / 0.82 / 1.61 ms per step at
    // B = 16 384 / 32 768 / 65 536 against 5.3 / 6.2 / 8.6. UMPC_F64_LDS_MAX_GRID caps it (diagnostics).
    static const int ldsf_max_grid = [] { const char *e_ = getenv("UMPC_F64_LDS_MAX_GRID"); return e_ ? atoi(e_) : 0x7fffffff; }();
    if (!no_ldsf && grid <= ldsf_max_grid) {
      // ... and the ADMM phase as generated fp64 assembly (needs >= 1 iteration and 31-bit row offsets)
      const bool fits = (size_t)umpc::WS_ROWS * (size_t)h->B * 8 < ((size_t)1 << 31);
      // ... with one robot per lane quad when that still is one round of workgroups (256 CUs x 16 robots): BASELINE
      // config 2's 4 096 robots are then 256 waves instead of 64, each iteration ~0.55 of the one-lane loop's time
      // (asmquad64.py). UMPC_QUAD64=0 keeps the lane form, UMPC_QUAD64=<n> moves the switch-over batch size.
      static const int quad64_max_b = [] { const char *e_ = getenv("UMPC_QUAD64"); return e_ ? atoi(e_) : 4096; }();
      const bool want_quad = h->step_kernel == 3 || (h->step_kernel == 0 && h->global_B <= quad64_max_b);
      if (!no_asm64 && h->step_kernel != 1 && h->prm.maxIter >= 2 && fits && want_quad) {
        hipLaunchKernelGGL((umpc_rollout_kernel<T, true, true, true>), dim3((h->B + kBlock / 4 - 1) / (kBlock / 4)), dim3(kBlock), 0,
                           (hipStream_t)stream, a, K, (const T *)actualT0, 0);
        h->last_kernel = "umpc_rollout_kernel<double, LDSF, ASM64, QUAD>";
      } else if (!no_asm64 && h->step_kernel != 1 && h->prm.maxIter >= 1 && fits) {
        hipLaunchKernelGGL((umpc_rollout_kernel<T, true, true>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, a, K,
                           (const T *)actualT0, 0);
        h->last_kernel = "umpc_rollout_kernel<double, LDSF, ASM64>";
      } else {
        hipLaunchKernelGGL((umpc_rollout_kernel<T, true, false>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, a, K,
                           (const T *)actualT0, 0);
        h->last_kernel = "umpc_rollout_kernel<double, LDSF>";
      }
      hipError_t e2 = hipGetLastError();
      return e2 == hipSuccess ? 0 : fail(e2, "umpcBatchRollout");
    }
  }
  hipLaunchKernelGGL((umpc_rollout_kernel<T, false, false>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, a, K,
                     (const T *)actualT0, skew_us * 100);
  h->last_kernel = sizeof(T) == 4 ? "umpc_rollout_kernel<float>" : "umpc_rollout_kernel<double>";
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(e, "umpcBatchRollout");
}

void umpc_set_error(const char *msg) { g_err = msg; }

template <typename T>
static int launch_reactive(umpc_batch_t *h, int nsteps, int every, void *state, const void *ref, const void *gains,
                           const void *Ib, const void *gain, void *out, void *stats, void *stream) {
  DevParams<T> prm = make_dev<T>(h->prm);
  prm.task = h->task;
  for (int i = 0; i < 4; ++i) prm.task_p[i] = (T)h->task_p[i];
  const int grid = (h->B + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(umpc_reactive_kernel<T>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, prm, h->B, nsteps, every,
                     (T)h->t_ms, (T *)state, (const T *)ref, (const T *)gains, (const T *)Ib, (const T *)gain, (T *)out,
                     (T *)stats);
  h->t_ms += (double)nsteps * h->prm.dtsim;
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(e, "umpcBatchReactive");
}

extern "C" {

const char *umpcLastError(void) { return g_err.c_str(); }
const int *umpcAxIdx(void) { return umpcgen::kAxIdx; }
const int *umpcKKTPerm(void) { return umpcgen::kPerm; }
int umpcNnzL(void) { return umpcgen::NNZL; }
const char *umpcKernelName(int dtype, int plant_mode) {
  // the kernel a DEFAULT rollout of this dtype dispatches to (launch_rollout); what a particular handle actually
  // dispatched -- options such as maxIter = 0, step_kernel = 1 or the environment overrides change it -- is
  // umpcBatchKernelName(h)
  (void)plant_mode;
  if (dtype == UMPC_F64)
    return getenv("UMPC_NO_F64_LDS") ? "umpc_rollout_kernel<double>"
           : getenv("UMPC_NO_ASM64") ? "umpc_rollout_kernel<double, LDSF>" : "umpc_rollout_kernel<double, LDSF, ASM64>";
  return !getenv("UMPC_NO_ASM_STEP") ? "umpc_rollout_asm_kernel" : "umpc_rollout_kernel<float>";
}
const char *umpcBatchKernelName(const umpc_batch_t *h) { return h ? h->last_kernel : ""; }

void umpcBatchDefaultParams(umpc_batch_params_t *p) {
  // createMPC, template/template_controllers.py:260-263,279; controlTest, template/uprightmpc2.py:87
  p->dt = 5; p->g = 9.81e-3; p->TtoWmax = 2; p->ws = 1e1; p->wds = 1e3; p->wpr = 1; p->wpf = 5;
  p->wvr = 1e3; p->wvf = 2e3; p->wthrust = 1e-1; p->wmom = 1e-2;
  p->Ib[0] = 3333; p->Ib[1] = 3333; p->Ib[2] = 1000;  // template/genqp.py:22
  p->maxIter = 50; p->dtsim = 0.2; p->taulim = 100; p->nsub = 25; p->plant_mode = 0;
}

umpc_batch_t *umpcBatchCreate(const umpc_batch_params_t *prm, int B, int dtype) {
  if (!prm || B <= 0 || (dtype != UMPC_F32 && dtype != UMPC_F64) || prm->maxIter < 0 || prm->nsub < 0) {
    g_err = "umpcBatchCreate: bad argument";
    return nullptr;
  }
  // The step recovers the Ruiz scaling D from the equilibrated diagonal of P (D_j = sqrt(P_jj / (P_raw,jj c)),
  // DESIGN.md 3.4), which needs every objective weight > 0 (the reference's createMPC defaults are; a zero
  // weight would give 0/0 there). Rejected here rather than producing NaN silently.
  const double w8[8] = {prm->ws, prm->wds, prm->wpr, prm->wpf, prm->wvr, prm->wvf, prm->wthrust, prm->wmom};
  for (double w : w8)
    if (!(w > 0)) { g_err = "umpcBatchCreate: objective weights must be > 0"; return nullptr; }
  if (!(prm->Ib[0] > 0 && prm->Ib[1] > 0 && prm->Ib[2] > 0 && prm->dt > 0)) {
    g_err = "umpcBatchCreate: Ib and dt must be > 0";
    return nullptr;
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    g_err = "umpcBatchCreate: no HIP device (this library has no CPU path)";
    return nullptr;
  }
  umpc_batch *h = new umpc_batch;
  h->prm = *prm; h->B = B; h->global_B = B; h->dtype = dtype; h->ws = nullptr;
  e = hipMalloc(&h->ws, (size_t)umpc::WS_ROWS * (size_t)B * (dtype == UMPC_F64 ? 8 : 4));
  if (e != hipSuccess) { fail(e, "umpcBatchCreate: workspace"); delete h; return nullptr; }
  return h;
}
void umpcBatchDestroy(umpc_batch_t *h) {
  if (!h) return;
  if (h->ws) (void)hipFree(h->ws);
  if (h->wl) (void)hipFree(h->wl);
  if (h->taskf) (void)hipFree(h->taskf);
  delete h;
}
int umpcBatchSetTask(umpc_batch_t *h, int task, const double params[4], double t_ms) {
  if (!h || task < 0 || task > 4) { g_err = "umpcBatchSetTask: bad argument"; return -1; }
  h->task = task;
  for (int i = 0; i < 4; ++i) h->task_p[i] = params ? params[i] : 0.0;
  h->t_ms = t_ms;
  return 0;
}
int umpcBatchSetWeights(umpc_batch_t *h, const void *weights) {
  if (!h) return -1;
  if (weights) {
    // one-time validation (synchronous D2H copy of 8 x B scalars): every weight must be > 0, see umpcBatchCreate
    const size_t n = (size_t)8 * h->B, esz = h->dtype == UMPC_F64 ? 8 : 4;
    std::string buf(n * esz, '\0');
    const hipError_t e = hipMemcpy(&buf[0], weights, n * esz, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(e, "umpcBatchSetWeights");
    bool ok = true;
    for (size_t i = 0; i < n && ok; ++i)
      ok = h->dtype == UMPC_F64 ? reinterpret_cast<const double *>(buf.data())[i] > 0
                                : reinterpret_cast<const float *>(buf.data())[i] > 0;
    if (!ok) { g_err = "umpcBatchSetWeights: objective weights must be > 0"; return -1; }
  }
  h->weights = weights;
  return 0;
}
int umpcBatchSetStepKernel(umpc_batch_t *h, int mode) {
  if (!h || mode < 0 || mode > 3) { g_err = "umpcBatchSetStepKernel: bad argument"; return -1; }
  h->step_kernel = mode;
  return 0;
}
int umpcBatchSetGlobalBatch(umpc_batch_t *h, long long global_B) {
  if (!h || global_B < (long long)h->B) { g_err = "umpcBatchSetGlobalBatch: bad argument (the whole job cannot be smaller than its block)"; return -1; }
  h->global_B = global_B;
  return 0;
}
long long umpcBatchGlobalBatch(const umpc_batch_t *h) { return h ? h->global_B : 0; }
double umpcBatchTime(const umpc_batch_t *h) { return h->t_ms; }
int umpcBatchSize(const umpc_batch_t *h) { return h->B; }
int umpcBatchDtype(const umpc_batch_t *h) { return h->dtype; }

int umpcBatchInitCtrl(umpc_batch_t *h, void *ctrl, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (h->dtype == UMPC_F32) hipLaunchKernelGGL(umpc_init_ctrl_kernel<float>, dim3(256), dim3(256), 0, s, h->B, (float *)ctrl);
  else hipLaunchKernelGGL(umpc_init_ctrl_kernel<double>, dim3(256), dim3(256), 0, s, h->B, (double *)ctrl);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(e, "umpcBatchInitCtrl");
}

int umpcBatchRollout(umpc_batch_t *h, int K, void *state, void *ctrl, const void *ref, const void *actualT0,
                     const void *Ib, const void *gain, void *out, void *stats, int32_t *status, void *info,
                     void *stream) {
  if (K < 0) { g_err = "umpcBatchRollout: K < 0"; return -1; }
  return h->dtype == UMPC_F32
             ? launch_rollout<float>(h, K, h->prm.nsub, state, ctrl, ref, actualT0, Ib, gain, out, stats, status, info, stream)
             : launch_rollout<double>(h, K, h->prm.nsub, state, ctrl, ref, actualT0, Ib, gain, out, stats, status, info, stream);
}

int umpcBatchTaskReference(umpc_batch_t *h, double t_ms, const void *ref, void *out, void *stream) {
  if (!h || !ref || !out) { g_err = "umpcBatchTaskReference: bad argument"; return -1; }
  const int grid = (h->B + 255) / 256;
  if (h->dtype == UMPC_F32) {
    DevParams<float> prm = make_dev<float>(h->prm);
    prm.task = h->task;
    for (int i = 0; i < 4; ++i) prm.task_p[i] = (float)h->task_p[i];
    hipLaunchKernelGGL(umpc_taskref_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, prm, h->B, (float)t_ms,
                       (const float *)ref, (float *)out);
  } else {
    DevParams<double> prm = make_dev<double>(h->prm);
    prm.task = h->task;
    for (int i = 0; i < 4; ++i) prm.task_p[i] = h->task_p[i];
    hipLaunchKernelGGL(umpc_taskref_kernel<double>, dim3(grid), dim3(256), 0, (hipStream_t)stream, prm, h->B, t_ms,
                       (const double *)ref, (double *)out);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(e, "umpcBatchTaskReference");
}

int umpcBatchReactive(umpc_batch_t *h, int nsteps, int every, void *state, const void *ref, const void *gains,
                      const void *Ib, const void *gain, void *out, void *stats, void *stream) {
  if (!h || nsteps < 0 || every < 1 || !state || !ref) { g_err = "umpcBatchReactive: bad argument"; return -1; }
  return h->dtype == UMPC_F32 ? launch_reactive<float>(h, nsteps, every, state, ref, gains, Ib, gain, out, stats, stream)
                              : launch_reactive<double>(h, nsteps, every, state, ref, gains, Ib, gain, out, stats, stream);
}

int umpcBatchUpdate(umpc_batch_t *h, const void *state, void *ctrl, const void *ref, const void *actualT0,
                    const void *Ib, void *out, int32_t *status, void *info, void *stream) {
  return h->dtype == UMPC_F32
             ? launch_rollout<float>(h, 1, 0, (void *)state, ctrl, ref, actualT0, Ib, nullptr, out, nullptr, status, info, stream)
             : launch_rollout<double>(h, 1, 0, (void *)state, ctrl, ref, actualT0, Ib, nullptr, out, nullptr, status, info, stream);
}

int umpcBatchPlant(umpc_batch_t *h, int nsub, void *state, const void *u, const void *Ib, const void *gain,
                   void *stream) {
  const int grid = (h->B + kBlock - 1) / kBlock;
  if (h->dtype == UMPC_F32)
    hipLaunchKernelGGL(umpc_plant_kernel<float>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream,
                       make_dev<float>(h->prm), h->B, nsub, (float *)state, (const float *)u, (const float *)Ib,
                       (const float *)gain);
  else
    hipLaunchKernelGGL(umpc_plant_kernel<double>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream,
                       make_dev<double>(h->prm), h->B, nsub, (double *)state, (const double *)u, (const double *)Ib,
                       (const double *)gain);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(e, "umpcBatchPlant");
}

static int assemble_launch(umpc_batch_t *h, const void *state, const void *ctrl, const void *ref, const void *Ib,
                           const void *actualT0, void *l, void *u, void *q, void *Px, void *Ax, void *stream) {
  const int grid = (h->B + kBlock - 1) / kBlock;
  if (h->dtype == UMPC_F32)
    hipLaunchKernelGGL(umpc_assemble_kernel<float>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream,
                       make_dev<float>(h->prm), h->B, (const float *)state, (const float *)ctrl, (const float *)ref,
                       (const float *)Ib, (const float *)actualT0, (float *)l, (float *)u, (float *)q, (float *)Px,
                       (float *)Ax);
  else
    hipLaunchKernelGGL(umpc_assemble_kernel<double>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream,
                       make_dev<double>(h->prm), h->B, (const double *)state, (const double *)ctrl,
                       (const double *)ref, (const double *)Ib, (const double *)actualT0, (double *)l, (double *)u,
                       (double *)q, (double *)Px, (double *)Ax);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(e, "umpcBatchAssemble");
}

int umpcBatchAssemble(umpc_batch_t *h, const void *state, const void *ctrl, const void *ref, const void *Ib,
                      void *l, void *u, void *q, void *Px, void *Ax, void *stream) {
  return assemble_launch(h, state, ctrl, ref, Ib, nullptr, l, u, q, Px, Ax, stream);
}

// ---------------------------------------------------------------------------
// Part 1: the reference's own three symbols, B = 1 on the GPU
// ---------------------------------------------------------------------------
namespace {
// Per-controller state of the drop-in. The caller's POD carries an opaque id in the two words of `smin` the
// reference never touches (uprightmpc2.h:32; no .c file reads smin / smax), so a host that copies or moves the
// POD -- the reference's pybind class holds it by value (py/uprightmpc2py.cpp:32) -- keeps its controller.
// Inputs and outputs of a call travel through ONE pinned, mapped host buffer the kernels read and write
// directly: no staging copies, one stream synchronisation per call.
struct Single {
  umpc_batch_t *h = nullptr;
  float *ctrl = nullptr;   // device: the 127-word controller record
  float *host = nullptr;   // pinned + mapped: inputs | outputs (layout below)
  float *hdev = nullptr;   // device alias of `host`
  hipStream_t stream = nullptr, stream2 = nullptr;      // step kernel | debug-field kernel
  int status = umpc::ST_UNSOLVED;
  unsigned seq = 0;        // call counter: the kernels write it to the completion words of the mapped buffer
  const UprightMPC_t *owner = nullptr;   // the POD umpcInit built this controller for (a by-value COPY of that POD aliases it)
  int compat = 0;          // opt-in reference compatibility flags (umpcSetCompat)
  void *gqp = nullptr;     // general-structure solver handle (created on the first rejected call)
  float *gbuf = nullptr;   // its device arrays (layout: compat_reject_step)
};
std::mutex g_mu;
std::map<uint32_t, Single> g_single;
uint32_t g_next_id = 1;
constexpr uint32_t kMagic = 0x554d5043u;  // "UMPC"
constexpr int O_STATE = 0, O_REF = 18, O_AT0 = O_REF + 9, O_OUT = O_AT0 + 1, O_INFO = O_OUT + 9, O_L = O_INFO + 2,
              O_U = O_L + 39, O_Q = O_U + 39, O_PX = O_Q + 45, O_AX = O_PX + 45, O_STATUS = O_AX + 48,
              O_T0DBG = O_STATUS + 1, O_DONE0 = (O_T0DBG + 1 + 15) / 16 * 16 /* each completion word on a 64-byte line of its own */,
              O_DONE1 = O_DONE0 + 16, O_LU = O_DONE1 + 16 /* compat reject path: the kept bounds l | u */,
              O_TOTAL = O_LU + 2 * 39;
static_assert(O_DONE0 % 16 == 0 && O_DONE0 > O_T0DBG, "completion words: own cache lines behind the I/O words");

uint32_t pod_id(const UprightMPC_t *up) {
  uint32_t w[2];
  memcpy(w, up->smin, sizeof(w));
  return w[0] == kMagic ? w[1] : 0u;
}
void release_locked(uint32_t id) {
  auto it = g_single.find(id);
  if (it == g_single.end()) return;
  Single &s = it->second;
  if (s.stream) (void)hipStreamSynchronize(s.stream);
  if (s.stream2) { (void)hipStreamSynchronize(s.stream2); (void)hipStreamDestroy(s.stream2); }
  if (s.gqp) umpcQPDestroy(s.gqp);
  if (s.gbuf) (void)hipFree(s.gbuf);
  if (s.h) umpcBatchDestroy(s.h);
  if (s.ctrl) (void)hipFree(s.ctrl);
  if (s.host) (void)hipHostFree(s.host);
  if (s.stream) (void)hipStreamDestroy(s.stream);
  g_single.erase(it);
}

// One call of the reference's REJECT path (osqp.c:801-808: any l_new[i] > u_new[i] => osqp_update_bounds returns 1 before
// touching anything; uprightmpc2.c:246 drops the value and goes on): the step solves with the bounds the workspace still
// holds and the NEW q, P, A. A controller's Tmax is fixed by umpcInit here (the kernels take it from the parameter block, not
// from the POD), so crossed bounds mean EVERY call of this controller is rejected and the workspace still holds the generated
// placeholder l = 0, u = 1e30 (workspace.c:476-557) (osqp_update_lin_cost / osqp_update_P_A are not skipped). The kept bounds need not make the dynamics rows
// equalities, so this runs on the general-structure solver (umpc_bqp.hip) with the controller's own warm start x, y, z,
// T0 and thrust-row E. Device arrays in s.gbuf (floats): Pv 45 | q 45 | l 39 | u 39 | par 10 | Av 111 | cst 111 | E 39 |
// sol_x 45 | sol_y 39 | info 6 | src 111 (int32).
namespace reject {
constexpr int O_PV = 0, O_QV = O_PV + NX, O_LV = O_QV + NX, O_UV = O_LV + NC, O_PAR = O_UV + NC, O_AV = O_PAR + 10,
              O_CST = O_AV + NNZA, O_E = O_CST + NNZA, O_SX = O_E + NC, O_SY = O_SX + NX, O_INF = O_SY + NC,
              O_SRC = O_INF + 6, O_END = O_SRC + NNZA;
}
// The general-solver handle and the device arrays of the reject path. Created by umpcSetCompat (a set-up call), or by
// the first rejected umpcUpdate of a controller whose switch came from UMPC_COMPAT in the environment. Nothing is kept
// unless EVERYTHING succeeded (a failed table copy used to leave the handle set and later calls solved with
// uninitialised tables).
int compat_reject_init(Single &s) {
  using namespace reject;
  if (s.gqp && s.gbuf) return 0;
  umpcQPSettings st;
  umpcQPDefaultSettings(&st);
  st.max_iter = s.h->prm.maxIter;
  void *gqp = umpcQPCreate(umpcn3::kBlob, umpcn3::BLOB_WORDS, 1, UMPC_F32, &st);
  if (!gqp) return 1;
  float *gbuf = nullptr;
  hipError_t e = hipMalloc((void **)&gbuf, O_END * sizeof(float));
  if (e == hipSuccess) {
    float tmp[O_END];
    memset(tmp, 0, sizeof(tmp));
    for (int k = 0; k < NNZA; ++k) tmp[O_CST + k] = umpcn3::kIsDt[k] ? (float)s.h->prm.dt : umpcn3::kCst[k];
    for (int k = 0; k < NC; ++k) tmp[O_E + k] = 1.0f;
    memcpy(tmp + O_SRC, umpcn3::kSrc, NNZA * sizeof(int32_t));
    e = hipMemcpy(gbuf, tmp, sizeof(tmp), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    fail(e, "umpcUpdate (compat bounds-reject path: tables)");
    if (gbuf) (void)hipFree(gbuf);
    umpcQPDestroy(gqp);
    return 1;
  }
  s.gqp = gqp; s.gbuf = gbuf;
  return 0;
}
int compat_reject_step(Single &s, const UprightMPC_t *up) {
  static_assert(umpcn3::N == N && umpcn3::NX == NX && umpcn3::NC == NC && umpcn3::NNZA == NNZA, "umpc_n3_general.h is the N = 3 structure");
  using namespace reject;
  if (compat_reject_init(s)) return 1;
  float *g = s.gbuf, *d = s.hdev;
  umpcNParams np;
  const umpc_batch_params_t &bp = s.h->prm;
  np.dt = bp.dt; np.g = bp.g; np.TtoWmax = bp.TtoWmax; np.ws = bp.ws; np.wds = bp.wds; np.wpr = bp.wpr; np.wpf = bp.wpf;
  np.wvr = bp.wvr; np.wvf = bp.wvf; np.wthrust = bp.wthrust; np.wmom = bp.wmom;
  for (int i = 0; i < 3; ++i) np.Ib[i] = bp.Ib[i];
  (void)up;
  // T0 of this call sits in the mapped word O_AT0 (already actualT0-or-accumulator); the kernels update it in place
  int rc = umpcNAssemble(1, UMPC_F32, N, &np, d + O_STATE, d + O_REF, d + O_AT0, nullptr, g + O_PV, g + O_QV, g + O_LV,
                         g + O_UV, g + O_PAR, s.stream);
  if (!rc) rc = umpcQPGather(1, UMPC_F32, NNZA, g + O_CST, (const int32_t *)(g + O_SRC), g + O_PAR, g + O_AV, s.stream);
  if (rc) return 1;
  // the bounds the reference's workspace still holds, handed over in the mapped buffer
  float *lu = s.host + O_LU;
  for (int i = 0; i < NC; ++i) { lu[i] = 0.0f; lu[NC + i] = 1e30f; }
  // thrust-row E of the previous call lives in the controller record (rows 124..126); rows 0..35 keep the general solver's own
  if (hipMemcpyAsync(g + O_E + 36, s.ctrl + 124, 3 * sizeof(float), hipMemcpyDeviceToDevice, s.stream) != hipSuccess) return 1;
  rc = umpcQPSolve(s.gqp, g + O_PV, g + O_AV, g + O_QV, d + O_LU, d + O_LU + NC, s.ctrl, s.ctrl + NX, s.ctrl + NX + NC, g + O_E,
                   g + O_SX, g + O_SY, (int32_t *)(d + O_STATUS), g + O_INF, s.stream);
  if (!rc) rc = umpcNExtract(1, UMPC_F32, N, bp.dt, d + O_STATE, g + O_SX, d + O_AT0, d + O_OUT, s.stream);
  if (rc) return 1;
  if (hipMemcpyAsync(s.ctrl + 124, g + O_E + 36, 3 * sizeof(float), hipMemcpyDeviceToDevice, s.stream) != hipSuccess ||
      hipMemcpyAsync(s.ctrl + 123, d + O_AT0, sizeof(float), hipMemcpyDeviceToDevice, s.stream) != hipSuccess ||
      hipMemcpyAsync(d + O_INFO, g + O_INF, 2 * sizeof(float), hipMemcpyDeviceToDevice, s.stream) != hipSuccess)
    return 1;
  const hipError_t e = hipStreamSynchronize(s.stream);
  if (e != hipSuccess) { fail(e, "umpcUpdate (compat bounds-reject path)"); return 1; }
  s.h->last_kernel = "bqp_solve_kernel";
  return 0;
}
}  // namespace

void umpcInit(UprightMPC_t *up, float dt, float g, float TtoWmax, float ws, float wds, float wpr, float wpf,
              float wvr, float wvf, float wthrust, float wmom, const float Ib[3], int maxIter) {
  std::lock_guard<std::mutex> lk(g_mu);
  // Re-initialising a POD that already carries a live controller (the reference allows it: a gain sweep, a Simulink
  // or MCU start / stop) releases the previous controller first -- its batch handle, ctrl record, pinned buffer and
  // stream -- instead of orphaning them. An uninitialised POD matching both the magic and a live id is not a concern.
  // Only when THIS address is the POD the controller was built for: a by-value copy of a live POD (`b = a;
  // umpcInit(&b, ...)`, which the reference's plain-value struct allows) carries a's id but must not destroy a's
  // controller -- it simply gets its own. (Copies that are not re-initialised alias one controller.)
  {
    auto it = g_single.find(pod_id(up));
    if (it != g_single.end() && it->second.owner == up) release_locked(it->first);
  }
  // host-visible part of uprightmpc2.c:19-118
  memset(up, 0, sizeof(*up));
  up->dt = dt; up->g = g; up->Tmax = TtoWmax * g;
  for (int i = 0; i < 3; ++i) {
    up->Qyr[i] = wpr; up->Qyf[i] = wpf; up->Qyr[3 + i] = up->Qyf[3 + i] = ws;
    up->Qdyr[i] = wvr; up->Qdyf[i] = wvf; up->Qdyr[3 + i] = up->Qdyf[3 + i] = wds;
  }
  up->R[0] = wthrust; up->R[1] = up->R[2] = wmom;
  up->c0[2] = -g;
  up->T0 = 0;
  up->e3h[1] = 1; up->e3h[3] = -1;
  up->e3hIbi[1] = 1.0f / Ib[0]; up->e3hIbi[3] = -(1.0f / Ib[1]);
  for (int i = 0; i < UMPC_nAdata; ++i) up->Ax_idx[i] = umpcgen::kAxIdx[i];
  up->nAxT0dt = 3 * (UMPC_N - 2);
  up->nAxdt = up->nAxT0dt + 6 * UMPC_N;

  Single s;
  umpc_batch_params_t p;
  umpcBatchDefaultParams(&p);
  p.dt = dt; p.g = g; p.TtoWmax = TtoWmax; p.ws = ws; p.wds = wds; p.wpr = wpr; p.wpf = wpf; p.wvr = wvr;
  p.wvf = wvf; p.wthrust = wthrust; p.wmom = wmom; p.maxIter = maxIter; p.nsub = 0;
  for (int i = 0; i < 3; ++i) p.Ib[i] = Ib[i];
  s.h = umpcBatchCreate(&p, 1, UMPC_F32);
  if (!s.h) { fprintf(stderr, "umpcInit: %s\n", g_err.c_str()); return; }
  if (hipMalloc((void **)&s.ctrl, UMPC_CTRL_ROWS * sizeof(float)) != hipSuccess ||
      hipHostMalloc((void **)&s.host, O_TOTAL * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer((void **)&s.hdev, s.host, 0) != hipSuccess ||
      hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&s.stream2, hipStreamNonBlocking) != hipSuccess) {
    fprintf(stderr, "umpcInit: device / pinned allocation failed\n");
    if (s.ctrl) (void)hipFree(s.ctrl);
    if (s.host) (void)hipHostFree(s.host);
    umpcBatchDestroy(s.h);
    return;
  }
  memset(s.host, 0, O_TOTAL * sizeof(float));
  umpcBatchInitCtrl(s.h, s.ctrl, s.stream);
  (void)hipStreamSynchronize(s.stream);
  const uint32_t id = g_next_id++, w[2] = {kMagic, id};
  memcpy(up->smin, w, sizeof(w));
  s.owner = up;
  if (const char *c = getenv("UMPC_COMPAT")) s.compat = atoi(c);
  g_single[id] = s;
}

int umpcUpdate(UprightMPC_t *up, float uquad[3], float accdes[6], const float p0[3], const float R0[9],
               const float dq0[6], const float pdes[3], const float dpdes[3], const float sdes[3],
               float actualT0) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_single.find(pod_id(up));
  if (it == g_single.end() || !it->second.h) {
    fprintf(stderr, "umpcUpdate: controller not initialised or no GPU (no CPU path exists)\n");
    return 1;
  }
  Single &s = it->second;
  float *hb = s.host;
  memcpy(hb + O_STATE, p0, 12); memcpy(hb + O_STATE + 3, R0, 36); memcpy(hb + O_STATE + 12, dq0, 24);
  memcpy(hb + O_REF, pdes, 12); memcpy(hb + O_REF + 3, dpdes, 12); memcpy(hb + O_REF + 6, sdes, 12);
  // The POD's T0 is the accumulator of record (uprightmpc2.c:215-216, 256-257): a host that edits up->T0
  // between calls is honoured, and actualT0 >= 0 overrides it for this call.
  hb[O_AT0] = actualT0 >= 0 ? actualT0 : up->T0;
  hb[O_T0DBG] = hb[O_AT0];
  float *d = s.hdev;
  // UMPC_DROPIN_TWO_LAUNCHES=1: the round-2 form (assembly kernel + step kernel + stream synchronisation), for A/B timing
  static const bool two_launches = getenv("UMPC_DROPIN_TWO_LAUNCHES") != nullptr || getenv("UMPC_NO_ASM_STEP") != nullptr;
  bool rejected = false;
  if (s.compat & UMPC_COMPAT_BOUNDS_REJECT) {
    // osqp.c:801-808 on the bounds as assembled (the debug-field kernel's l, u: what up->l / up->u show in the reference too)
    int rc = assemble_launch(s.h, d + O_STATE, s.ctrl, d + O_REF, nullptr, d + O_AT0, d + O_L, d + O_U, d + O_Q,
                             d + O_PX, d + O_AX, s.stream);
    if (rc) return 1;
    const hipError_t e = hipStreamSynchronize(s.stream);
    if (e != hipSuccess) { fail(e, "umpcUpdate"); return 1; }
    for (int i = 0; i < UMPC_NC; ++i) rejected = rejected || (hb[O_L + i] > hb[O_U + i]);
  }
  if (rejected) {
    if (compat_reject_step(s, up)) return 1;
  } else if (two_launches || s.h->prm.maxIter < 1) {
    int rc = assemble_launch(s.h, d + O_STATE, s.ctrl, d + O_REF, nullptr, d + O_AT0, d + O_L, d + O_U, d + O_Q,
                             d + O_PX, d + O_AX, s.stream);
    if (!rc)
      rc = umpcBatchUpdate(s.h, d + O_STATE, s.ctrl, d + O_REF, d + O_AT0, nullptr, d + O_OUT,
                           (int32_t *)(d + O_STATUS), d + O_INFO, s.stream);
    if (rc) return 1;
    const hipError_t e = hipStreamSynchronize(s.stream);
    if (e != hipSuccess) { fail(e, "umpcUpdate"); return 1; }
  } else {
    // the step kernel (B = 1, all assembly) and the debug-field kernel on two streams, completion by polling the two words
    // they release at system scope: no stream synchronisation on the critical path
    const unsigned seq = ++s.seq ? s.seq : ++s.seq;       // never 0 (the words start at 0)
    umpcasm::StepParams p = make_step_params(s.h, 1, 0, d + O_STATE, s.ctrl, d + O_REF, d + O_AT0, nullptr, nullptr, d + O_OUT,
                                             nullptr, (int32_t *)(d + O_STATUS), d + O_INFO);
    p.done = d + O_DONE0; p.seq = (int)seq;
    // UMPC_DROPIN_TWO_STREAMS=1: round 4's form (step kernel + debug-field kernel on two streams, two completion words), A/B
    static const bool two_streams = getenv("UMPC_DROPIN_TWO_STREAMS") != nullptr;
    const bool one_launch = quad_max_b() >= 1 && !two_streams;
    if (one_launch) {
      hipLaunchKernelGGL(umpc_dropin_quad_kernel, dim3(1), dim3(kBlock), 0, s.stream, p, make_dev<float>(s.h->prm),
                         (const float *)(d + O_STATE), (const float *)(d + O_REF), (const float *)(d + O_T0DBG), d + O_L,
                         d + O_U, d + O_Q, d + O_PX, d + O_AX);
      *(unsigned *)(hb + O_DONE1) = seq;           // (one completion word in this form: the second one is the host's own)
    } else {
      if (quad_max_b() >= 1)
        hipLaunchKernelGGL(umpc_rollout_asm_quad_kernel, dim3(1), dim3(kBlock), 0, s.stream, p, 1);
      else
        hipLaunchKernelGGL(umpc_rollout_asm_kernel, dim3(1), dim3(kBlock), 0, s.stream, p, 1, 0, 1);
      hipLaunchKernelGGL(umpc_dropin_debug_kernel, dim3(1), dim3(64), 0, s.stream2, make_dev<float>(s.h->prm),
                         (const float *)(d + O_STATE), (const float *)(d + O_REF), (const float *)(d + O_T0DBG), d + O_L, d + O_U,
                         d + O_Q, d + O_PX, d + O_AX, (unsigned *)(d + O_DONE1), seq);
    }
    s.h->last_kernel = one_launch ? "umpc_dropin_quad_kernel" : quad_max_b() >= 1 ? "umpc_rollout_asm_quad_kernel" : "umpc_rollout_asm_kernel";
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fail(e, "umpcUpdate"); return 1; }
    volatile unsigned *f0 = (volatile unsigned *)(hb + O_DONE0), *f1 = (volatile unsigned *)(hb + O_DONE1);
    // a healthy call completes in ~0.15 ms: poll for at most 50 ms of wall clock (checked every 256 polls), then fall
    // back to the streams, which also report a fault
    bool done = false;
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(50);
    for (unsigned spin = 1;; ++spin) {
      if (*f0 == seq && *f1 == seq) { done = true; break; }
      if ((spin & 255u) == 0 && std::chrono::steady_clock::now() > t_end) break;
#if defined(__x86_64__) || defined(__i386__)
      __builtin_ia32_pause();
#endif
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (!done) {
      e = hipStreamSynchronize(s.stream);
      const hipError_t e2 = hipStreamSynchronize(s.stream2);
      if (e != hipSuccess || e2 != hipSuccess) {
        fail(e != hipSuccess ? e : e2, "umpcUpdate: kernel did not complete");
        return 1;
      }
      if (*f0 != seq || *f1 != seq) {
        g_err = "umpcUpdate: both streams drained without error but a completion word was not written";
        return 1;
      }
    }
  }
  memcpy(uquad, hb + O_OUT, 12); memcpy(accdes, hb + O_OUT + 3, 24);
  memcpy(up->l, hb + O_L, 39 * 4); memcpy(up->u, hb + O_U, 39 * 4);
  memcpy(up->q, hb + O_Q, 45 * 4); memcpy(up->Px_data, hb + O_PX, 45 * 4);
  memcpy(up->Ax_data, hb + O_AX, 48 * 4);
  memcpy(&s.status, hb + O_STATUS, 4);
  up->T0 = uquad[0];  // uprightmpc2.c:256-257
  return 0;
}

namespace { UprightMPC_t g_up; int g_inited = 0; }
void umpcS(float uquad[3], float accdes[6], const float p0[3], const float R0[9], const float dq0[6],
           const float pdes[3], const float dpdes[3], const float sdes[3], float dt, float g, float TtoWmax,
           float ws, float wds, float wpr, float wpf, float wvr, float wvf, float wthrust, float wmom,
           const float Ib[3], int maxIter, float actualT0) {
  if (!g_inited) {  // uprightmpc2.c:279-283
    umpcInit(&g_up, dt, g, TtoWmax, ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom, Ib, maxIter);
    g_inited = 1;
  }
  umpcUpdate(&g_up, uquad, accdes, p0, R0, dq0, pdes, dpdes, sdes, actualT0);
}

int umpcLastStatus(const UprightMPC_t *up) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_single.find(pod_id(up));
  return it == g_single.end() ? umpc::ST_UNSOLVED : it->second.status;
}
int umpcLiveControllers(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  return (int)g_single.size();
}
int umpcSetCompat(UprightMPC_t *up, int flags) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_single.find(pod_id(up));
  if (it == g_single.end()) return -1;
  const int prev = it->second.compat;
  it->second.compat = flags;
  // the reject path's solver handle and tables are set up HERE, not inside the first rejected umpcUpdate (a failure
  // leaves the switch on and is retried, and reported, by that call)
  if ((flags & UMPC_COMPAT_BOUNDS_REJECT) && it->second.h) (void)compat_reject_init(it->second);
  return prev;
}
void umpcRelease(UprightMPC_t *up) {
  std::lock_guard<std::mutex> lk(g_mu);
  release_locked(pod_id(up));
  memset(up->smin, 0, 2 * sizeof(float));
}

// ---------------------------------------------------------------------------
// Part 3: wrench-linearisation step, funapprox.c
// ---------------------------------------------------------------------------
}  // extern "C"

namespace {
using umpc::WLDev;

// one lane = one robot: w0 = w(u0), A1 = dw/du(u0), one clipped gradient step (funapprox.c:118-165)
template <typename T>
__global__ __launch_bounds__(256) void umpc_wl_kernel(WLDev p, int B_, T *u, const T *h0, const T *pd, T *w0out) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B_) return;
  const size_t B = (size_t)B_;
  T u0[4], h[6], d[6], w[6];
#pragma unroll
  for (int j = 0; j < 4; ++j) u0[j] = u[(size_t)j * B + b];
#pragma unroll
  for (int i = 0; i < 6; ++i) { h[i] = h0[(size_t)i * B + b]; d[i] = pd[(size_t)i * B + b]; }
  umpc::wl_step(p, u0, h, d, w);
#pragma unroll
  for (int i = 0; i < 6; ++i) w0out[(size_t)i * B + b] = w[i];
#pragma unroll
  for (int j = 0; j < 4; ++j) u[(size_t)j * B + b] = u0[j];
}

// a19 / a20 vector fields: nsub == 0 evaluates ydot (ca6 also appends wrench and bias h), nsub > 0
// advances y by nsub RK4 steps of dt with u held
template <typename T, int MODEL>
__global__ __launch_bounds__(256) void umpc_model_kernel(int B_, int nsub, T dt, T *y, const T *u, T *aux) {
  constexpr int NYV = MODEL == 0 ? 18 : 12, NUV = MODEL == 0 ? 6 : 4;
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B_) return;
  const size_t B = (size_t)B_;
  T yv[NYV], uv[NUV];
#pragma unroll
  for (int i = 0; i < NYV; ++i) yv[i] = y[(size_t)i * B + b];
#pragma unroll
  for (int i = 0; i < NUV; ++i) uv[i] = u[(size_t)i * B + b];
  auto vf = [&](const T (&ys)[NYV], T (&k)[NYV]) {
    if constexpr (MODEL == 0) {
      T w[6], h[6];
      umpc::ca6_vf(ys, uv, k, w, h);
    } else {
      umpc::tsd_vf(ys, uv, k);
    }
  };
  if (nsub == 0) {
    T k[NYV];
    vf(yv, k);
#pragma unroll
    for (int i = 0; i < NYV; ++i) aux[(size_t)i * B + b] = k[i];
    if constexpr (MODEL == 0) {
      T w[6], h[6], kk[NYV];
      umpc::ca6_vf(yv, uv, kk, w, h);
#pragma unroll
      for (int i = 0; i < 6; ++i) { aux[(size_t)(18 + i) * B + b] = w[i]; aux[(size_t)(24 + i) * B + b] = h[i]; }
    }
    return;
  }
#pragma nounroll
  for (int s = 0; s < nsub; ++s) umpc::rk4_step<T, NYV>(yv, dt, vf);
#pragma unroll
  for (int i = 0; i < NYV; ++i) y[(size_t)i * B + b] = yv[i];
}

WLDev make_wl(const WLCon_t *wl) {
  WLDev d;
  for (int j = 0; j < 4; ++j) { d.umin[j] = wl->umin[j]; d.umax[j] = wl->umax[j]; d.dumax[j] = wl->dumax[j]; }
  for (int i = 0; i < 6; ++i) {
    d.Qw[i] = wl->Qw[i + 6 * i];
    d.a0[i] = wl->fa[i].a0;
    for (int j = 0; j < 4; ++j) d.a1[i][j] = wl->fa[i].a1[j];
    for (int j = 0; j < 16; ++j) d.A2[i][j] = wl->fa[i].A2[j];
    d.Md[i] = 0.f;
  }
  return d;
}
float *g_wl_dev = nullptr;  // 4 + 6 + 6 + 6 floats for the B = 1 entry point
}  // namespace

extern "C" {

int umpcBatchWLUpdate(const WLCon_t *wl, int B, int dtype, void *u, const void *h0, const void *pdotdes, void *w0,
                      void *stream) {
  if (!wl || B <= 0 || !u || !h0 || !pdotdes || !w0) { g_err = "umpcBatchWLUpdate: bad argument"; return -1; }
  const WLDev d = make_wl(wl);
  const int grid = (B + 255) / 256;
  if (dtype == UMPC_F32)
    hipLaunchKernelGGL(umpc_wl_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, d, B, (float *)u,
                       (const float *)h0, (const float *)pdotdes, (float *)w0);
  else
    hipLaunchKernelGGL(umpc_wl_kernel<double>, dim3(grid), dim3(256), 0, (hipStream_t)stream, d, B, (double *)u,
                       (const double *)h0, (const double *)pdotdes, (double *)w0);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(e, "umpcBatchWLUpdate");
}

int umpcBatchSetWL(umpc_batch_t *h, const WLCon_t *wl, const double Mdiag[6], void *u4, void *w0) {
  if (!h) return -1;
  if (!wl) {
    h->wlu = h->wlw = nullptr;
    if (h->wl) { (void)hipDeviceSynchronize(); (void)hipFree(h->wl); h->wl = nullptr; }
    return 0;
  }
  if (!Mdiag || !u4 || !(Mdiag[2] > 0)) { g_err = "umpcBatchSetWL: bad argument"; return -1; }
  WLDev d = make_wl(wl);
  for (int i = 0; i < 6; ++i) d.Md[i] = (float)Mdiag[i];
  h->wl_md2 = d.Md[2];
  hipError_t e = hipSuccess;
  if (!h->wl) e = hipMalloc((void **)&h->wl, sizeof(WLDev));
  if (e == hipSuccess) e = hipMemcpy(h->wl, &d, sizeof(WLDev), hipMemcpyHostToDevice);
  if (e != hipSuccess) return fail(e, "umpcBatchSetWL");
  h->wlu = u4; h->wlw = w0;
  return 0;
}

int umpcBatchModel(int model, int B, int dtype, int nsub, double dt, void *y, const void *u, void *aux, void *stream) {
  if ((model != UMPC_MODEL_CA6 && model != UMPC_MODEL_TSD) || B <= 0 || nsub < 0 || !y || !u || (nsub == 0 && !aux)) {
    g_err = "umpcBatchModel: bad argument";
    return -1;
  }
  const dim3 grid((B + 255) / 256), blk(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UMPC_F32) {
    if (model == UMPC_MODEL_CA6)
      hipLaunchKernelGGL((umpc_model_kernel<float, 0>), grid, blk, 0, s, B, nsub, (float)dt, (float *)y, (const float *)u, (float *)aux);
    else
      hipLaunchKernelGGL((umpc_model_kernel<float, 1>), grid, blk, 0, s, B, nsub, (float)dt, (float *)y, (const float *)u, (float *)aux);
  } else {
    if (model == UMPC_MODEL_CA6)
      hipLaunchKernelGGL((umpc_model_kernel<double, 0>), grid, blk, 0, s, B, nsub, dt, (double *)y, (const double *)u, (double *)aux);
    else
      hipLaunchKernelGGL((umpc_model_kernel<double, 1>), grid, blk, 0, s, B, nsub, dt, (double *)y, (const double *)u, (double *)aux);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(e, "umpcBatchModel");
}

void wlConInit(WLCon_t *wl, const float u0[4], const float umin[4], const float umax[4], const float dumax[4],
               const float Qw[6], float controlRate, const float popts[90]) {
  // funapprox.c:102-116 + funApproxInit :35-51 (host-side unpacking; the struct carries all state)
  memset(wl, 0, sizeof(*wl));
  for (int i = 0; i < 4; ++i) {
    wl->u0[i] = u0[i]; wl->umin[i] = umin[i]; wl->umax[i] = umax[i];
    wl->dumax[i] = dumax[i] / controlRate;
  }
  for (int i = 0; i < 6; ++i) {
    const float *p = &popts[15 * i];
    wl->fa[i].k = NDELU;
    wl->fa[i].a0 = p[0];
    memcpy(wl->fa[i].a1, &p[1], 4 * sizeof(float));
    int kk = 0;
    for (int r = 0; r < 4; ++r)
      for (int c = r; c < 4; ++c) wl->fa[i].A2[r + 4 * c] = wl->fa[i].A2[c + 4 * r] = p[5 + kk++];
    wl->Qw[i + 6 * i] = Qw[i];
  }
}

void wlConUpdate(WLCon_t *wl, float u1[4], float w0[6], const float h0[6], const float pdotdes[6]) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_wl_dev && hipMalloc((void **)&g_wl_dev, 22 * sizeof(float)) != hipSuccess) {
    fprintf(stderr, "wlConUpdate: no GPU (no CPU path exists)\n");
    return;
  }
  float hin[16];
  memcpy(hin, wl->u0, 16); memcpy(hin + 4, h0, 24); memcpy(hin + 10, pdotdes, 24);
  (void)hipMemcpy(g_wl_dev, hin, sizeof(hin), hipMemcpyHostToDevice);
  if (umpcBatchWLUpdate(wl, 1, UMPC_F32, g_wl_dev, g_wl_dev + 4, g_wl_dev + 10, g_wl_dev + 16, nullptr)) return;
  float hout[22];
  if (hipMemcpy(hout, g_wl_dev, sizeof(hout), hipMemcpyDeviceToHost) != hipSuccess) return;
  memcpy(wl->u0, hout, 16); memcpy(u1, hout, 16); memcpy(w0, hout + 16, 24);
}

namespace { WLCon_t g_wl; int g_wl_inited = 0; }
void wlconS(float u1[4], float w0[6], const float u0init[4], const float umin[4], const float umax[4],
            const float dumax[4], const float Qw[6], float controlRate, const float popts[90], const float h0[6],
            const float pdotdes[6]) {
  if (!g_wl_inited) {  // funapprox.c:172-175
    wlConInit(&g_wl, u0init, umin, umax, dumax, Qw, controlRate, popts);
    g_wl_inited = 1;
  }
  wlConUpdate(&g_wl, u1, w0, h0, pdotdes);
}

}  // extern "C"
