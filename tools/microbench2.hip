// Issue-rate microbenchmarks, round 2: what ONE wave and what 2 / 4 waves per SIMD sustain on gfx950 for the
// instruction kinds the step kernel is made of. Occupancy is forced by dynamic LDS (160 KiB / n per 256-thread
// block => exactly n blocks per CU => n waves per SIMD), the body is 64 instructions, timed with s_memtime.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench2 tools/microbench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, long long *cyc, int iters) {
  extern __shared__ float dyn[];
  const int lane = threadIdx.x;
  float a[16], b[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = out[i * 256 + lane]; b[i] = out[(16 + i) * 256 + lane]; }
  if (iters < 0) dyn[lane] = a[0];
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
      } else if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
      } else if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          f2 x = {a[i], a[i + 1]}, y = {b[i], b[i + 1]};
          asm volatile("v_pk_fma_f32 %0, %1, %1, %0\n v_pk_fma_f32 %0, %1, %1, %0" : "+v"(x) : "v"(y));
          a[i] = x.x; a[i + 1] = x.y;
        }
      } else if (MODE == 3) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_accvgpr_read_b32 %0, a%1" : "=v"(a[i]) : "n"(0) : "a0");
      } else if (MODE == 4) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
      } else if (MODE == 5) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_rsq_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
      } else if (MODE == 6) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
      } else if (MODE == 7) {  // dependent chains: 1 chain
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[0]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
      } else if (MODE == 8) {  // 2 interleaved chains
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i & 1]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
      } else if (MODE == 9) {  // 3 interleaved chains
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i % 3]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
      } else if (MODE == 10) {  // 4 interleaved chains
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i & 3]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
      } else if (MODE == 11) {  // RAW on a multiplicand: chain through src
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[0]) : "v"(b[i]));
      } else if (MODE == 12) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          f2 x = {a[i], a[i + 1]}, y = {b[i], b[i + 1]};
          asm volatile("v_pk_mul_f32 %0, %1, %0\n v_pk_add_f32 %0, %1, %0" : "+v"(x) : "v"(y));
          a[i] = x.x; a[i + 1] = x.y;
        }
      } else if (MODE == 13) {  // compare + select
#pragma unroll
        for (int i = 0; i < 16; i += 2)
          asm volatile("v_cmp_gt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
      } else if (MODE == 14) {  // fmac with an s_nop 0 after each
#pragma unroll
        for (int i = 0; i < 16; i += 2) asm volatile("v_fmac_f32 %0, %1, %2\n s_nop 0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
      } else if (MODE == 15) {  // accvgpr_read feeding a fmac three instructions later (the loop's pattern)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          asm volatile("v_accvgpr_read_b32 %0, a%1" : "=v"(a[8 + ((i / 2 + 3) & 7)]) : "n"(1) : "a1");
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[(i / 2) & 3]) : "v"(a[8 + ((i / 2) & 7)]), "v"(b[i]));
        }
      } else if (MODE == 16) {  // v_med3 clamp
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
      } else if (MODE == 17) {  // v_rcp_f32
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
      } else if (MODE == 18) {  // pk_fma, dependent pairs at distance 2
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          f2 x = {a[(i & 2)], a[(i & 2) + 1]}, y = {b[i], b[i + 1]};
          asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(x) : "v"(y));
          a[(i & 2)] = x.x; a[(i & 2) + 1] = x.y;
        }
      } else if (MODE == 20) {  // accvgpr_read + s_nop 0 pairs: does a scalar-side instruction fit in the read's shadow?
#pragma unroll
        for (int i = 0; i < 16; i += 2) asm volatile("v_accvgpr_read_b32 %0, a0\n s_nop 0" : "=v"(a[i]) : : "a0");
      } else if (MODE == 21) {  // accvgpr_read + s_mov pairs
#pragma unroll
        for (int i = 0; i < 16; i += 2) asm volatile("v_accvgpr_read_b32 %0, a0\n s_mov_b32 s40, 1" : "=v"(a[i]) : : "a0", "s40");
      } else if (MODE == 22) {  // accvgpr_read + ds_read_b128 pairs (LDS instruction in the shadow), one wait per 8 reads
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          float4 q;
          asm volatile("v_accvgpr_read_b32 %0, a0\n ds_read_b128 %1, %2" : "=v"(a[i]), "=v"(q) : "v"(lane * 16) : "a0");
          if (i == 14) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          b[i & 3] += 0.f * q.x;
        }
      } else if (MODE == 23) {  // VOP2 with DPP: partner-lane operand
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
      } else if (MODE == 24) {  // ds_read_b128 stream, one wait per 16
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float4 q;
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(lane * 16), "n"(0));
          if (i == 15) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          b[i & 3] += 0.f * q.x;
        }
      } else if (MODE == 19) {  // v_accvgpr_write
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_accvgpr_write_b32 a%1, %0" : : "v"(a[i]), "n"(2) : "a2");
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  out[(blockIdx.x * 256 + lane) % (32 * 256)] = s;
  if ((lane & 63) == 0) cyc[blockIdx.x * 4 + lane / 64] = t1 - t0;
}

template <int M> void launch(int blocks, int lds, float *d, long long *c, int iters) {
  hipFuncSetAttribute((const void *)k<M>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(256), lds, 0, d, c, iters);
}

int main() {
  float *d; long long *c;
  const int iters = 2000;
  CHECK(hipMalloc(&d, 32 * 256 * sizeof(float)));
  CHECK(hipMemset(d, 0, 32 * 256 * sizeof(float)));
  CHECK(hipMalloc(&c, 256 * 8 * 4 * sizeof(long long)));
  const char *names[] = {"v_fma_f32 (VOP3) indep", "v_fmac_f32 (VOP2) indep", "v_pk_fma_f32 indep", "v_accvgpr_read indep",
                         "v_max3_f32 |a|,|b|", "v_rsq_f32", "v_mul_f32 (VOP2)", "fmac 1 chain", "fmac 2 chains", "fmac 3 chains",
                         "fmac 4 chains", "mul chain via src", "pk_mul + pk_add", "v_cmp + v_cndmask", "fmac + s_nop 0 (32+32)",
                         "accread -> fmac 3 later", "v_med3_f32", "v_rcp_f32", "pk_fma 2 chains", "v_accvgpr_write", "accread + s_nop", "accread + s_mov", "accread + ds_read_b128", "v_fmac_dpp quad_perm", "ds_read_b128 stream"};
  const int per_body[] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 32, 64, 64, 64, 64, 64, 64};
  for (int n : {1, 2}) {
    const int blocks = 256 * n, lds = (160 * 1024) / n - (n > 1 ? 1024 : 0);
    for (int m = 0; m < 25; ++m) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        switch (m) {
#define CASE(M) case M: launch<M>(blocks, lds, d, c, iters); break;
          CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12)
          CASE(13) CASE(14) CASE(15) CASE(16) CASE(17) CASE(18) CASE(19) CASE(20) CASE(21) CASE(22) CASE(23) CASE(24)
        }
        hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> h(blocks * 4); CHECK(hipMemcpy(h.data(), c, blocks * 4 * sizeof(long long), hipMemcpyDeviceToHost));
      double avg = 0; for (auto v : h) avg += v; avg /= h.size();
      printf("waves/SIMD=%d  %-28s : %6.2f ticks/instr/wave  -> %5.2f per SIMD-instr, wall %.3f ms (%.2f ns per SIMD-instr)\n", n,
             names[m], avg / iters / per_body[m], avg / iters / per_body[m] / n, ms, ms * 1e6 / iters / per_body[m] / n);
    }
  }
  return 0;
}
