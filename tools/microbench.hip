// Issue-rate microbenchmarks for the lane-per-robot design (one wave per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, long long *cyc, int iters) {
  __shared__ float4 lds[40 * 64];
  const int lane = threadIdx.x;
  float a[16], b[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = out[i * 64 + lane]; b[i] = out[(16 + i) * 64 + lane]; }
  for (int i = 0; i < 40; ++i) lds[i * 64 + lane] = make_float4(a[i & 15], b[i & 15], a[(i + 1) & 15], b[(i + 3) & 15]);
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // 64 independent v_fma_f32
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
    } else if (MODE == 1) {  // 32 v_pk_fma_f32 (= 64 FMAs)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          typedef float f2 __attribute__((ext_vector_type(2)));
          f2 x = {a[i], a[i + 1]}, y = {b[i], b[i + 1]};
          asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(x) : "v"(y));
          a[i] = x.x; a[i + 1] = x.y;
        }
    } else if (MODE == 2) {  // dependent chain of 64 v_fma_f32
#pragma unroll
      for (int r = 0; r < 64; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b[0]), "v"(b[1]));
    } else if (MODE == 3) {  // 64 fma each fed by an accvgpr read
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float t;
          asm volatile("v_accvgpr_read_b32 %0, a%1" : "=v"(t) : "n"(0));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(t), "v"(b[i]));
        }
    } else if (MODE == 4) {  // 64 fma fed by 16 ds_read_b128
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float4 v = lds[((r + it) % 40) * 64 + lane];
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[(4 * r) & 15]) : "v"(v.x), "v"(b[0]));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[(4 * r + 1) & 15]) : "v"(v.y), "v"(b[1]));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[(4 * r + 2) & 15]) : "v"(v.z), "v"(b[2]));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[(4 * r + 3) & 15]) : "v"(v.w), "v"(b[3]));
      }
    } else if (MODE == 5) {  // 64 v_rsq_f32
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_rsq_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
    } else if (MODE == 7) {  // 64 independent v_fmac_f32: the 4-byte VOP2 encoding of d += a*b
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
    } else if (MODE == 6) {  // 64 IEEE divisions (compiler sequence)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __fdiv_rn(b[i], a[i] + 2.0f);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * 64 + lane] = s;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float *d; long long *c;
  const int blocks_full = 1024, iters = 2000;
  CHECK(hipMalloc(&d, 4096 * 64 * sizeof(float)));
  CHECK(hipMemset(d, 0, 4096 * 64 * sizeof(float)));
  CHECK(hipMalloc(&c, blocks_full * 8 * sizeof(long long)));
  const char *names[] = {"64 indep v_fma_f32", "32 v_pk_fma_f32 (64 FMA)", "64 dependent v_fma_f32", "64 x (accvgpr_read + fma)", "16 ds_read_b128 + 64 fma", "64 v_rsq_f32", "64 IEEE div", "64 indep v_fmac_f32 (VOP2)"};
  for (int wavesPerCU : {1, 4, 8}) {
    int blocks = 256 * wavesPerCU;
    for (int m = 0; m < 8; ++m) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        switch (m) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, d, c, iters); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d, c, iters); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, d, c, iters); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, d, c, iters); break;
          case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, d, c, iters); break;
          case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(64), 0, 0, d, c, iters); break;
          case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(64), 0, 0, d, c, iters); break;
          case 7: hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(64), 0, 0, d, c, iters); break;
        }
        hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> h(blocks); CHECK(hipMemcpy(h.data(), c, blocks * sizeof(long long), hipMemcpyDeviceToHost));
      double avg = 0; for (auto v : h) avg += v; avg /= blocks;
      printf("waves/CU=%d  %-28s : %.1f memtime ticks per iteration of 64 ops, wall %.3f ms\n", wavesPerCU, names[m], avg / iters, ms);
    }
  }
  return 0;
}
