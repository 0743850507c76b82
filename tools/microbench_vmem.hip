// What does it cost ONE wave per SIMD (the step kernel's occupancy) to fetch 16 words per 64 VALU instructions from
// (a) AGPRs, (b) L2 with global_load_dword, (c) L2 with global_load_dwordx4 (+ the two SALU pointer updates), (d) LDS?
// The body is 64 independent-enough v_fmac (4 chains); loads land in a 16-register ring and are waited for one body
// THREE bodies later (s_waitcnt vmcnt(3 bodies' loads); LDS: one body later), so only ISSUE cost and memory-pipe throughput show, as in a software-pipelined loop.
// Each wave reads its own 16 KiB window (1 024 waves: 16 MiB, 2 MiB per XCD: L2 hits after the first pass).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench_vmem tools/microbench_vmem.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const float *src, long long *cyc, int iters) {
  extern __shared__ float dyn[];
  const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float a[4], b[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = out[i * 256 + threadIdx.x];
#pragma unroll
  for (int i = 0; i < 16; ++i) b[i] = out[(4 + i) * 256 + threadIdx.x];
  if (iters < 0) dyn[threadIdx.x] = a[0];
  const float *base = src + (size_t)wave * 4096;          // 16 KiB window per wave
  float4 r0 = {0, 0, 0, 0}, r1 = r0, r2 = r0, r3 = r0;
  float acc = 0.f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const unsigned long long pa = (unsigned long long)(base + ((it & 3) * 1024));
    const float *p = (const float *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(pa >> 32)) << 32) |
                                     (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)pa));   // (unsigned): the builtin returns int
    if (MODE == 1) {        // 16 v_accvgpr_read
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(b[i]) : : "a0");
    } else if (MODE == 2) { // 16 global_load_dword, lane-contiguous rows of 256 B
      float *q = (float *)&r0;
      asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
      acc += r0.x + r1.y + r2.z + r3.w;
      asm volatile("global_load_dword %0, %4, %5\n global_load_dword %1, %4, %5 offset:256\n global_load_dword %2, %4, %5 offset:512\n global_load_dword %3, %4, %5 offset:768"
                   : "=v"(r0.x), "=v"(r0.y), "=v"(r0.z), "=v"(r0.w) : "v"(lane * 4), "s"(p) : "memory");
      asm volatile("global_load_dword %0, %4, %5 offset:1024\n global_load_dword %1, %4, %5 offset:1280\n global_load_dword %2, %4, %5 offset:1536\n global_load_dword %3, %4, %5 offset:1792"
                   : "=v"(r1.x), "=v"(r1.y), "=v"(r1.z), "=v"(r1.w) : "v"(lane * 4), "s"(p) : "memory");
      asm volatile("global_load_dword %0, %4, %5 offset:2048\n global_load_dword %1, %4, %5 offset:2304\n global_load_dword %2, %4, %5 offset:2560\n global_load_dword %3, %4, %5 offset:2816"
                   : "=v"(r2.x), "=v"(r2.y), "=v"(r2.z), "=v"(r2.w) : "v"(lane * 4), "s"(p) : "memory");
      asm volatile("global_load_dword %0, %4, %5 offset:3072\n global_load_dword %1, %4, %5 offset:3328\n global_load_dword %2, %4, %5 offset:3584\n global_load_dword %3, %4, %5 offset:3840"
                   : "=v"(r3.x), "=v"(r3.y), "=v"(r3.z), "=v"(r3.w) : "v"(lane * 4), "s"(p) : "memory");
      (void)q;
    } else if (MODE == 3 || MODE == 4) { // 4 global_load_dwordx4 (16 B per lane, 1 KiB per instruction); MODE 4: + 2 SALU per load
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      acc += r0.x + r1.y + r2.z + r3.w;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r0) : "v"(lane * 16), "s"(p) : "memory");
      if (MODE == 4) asm volatile("s_add_u32 s40, s40, 1024\n s_addc_u32 s41, s41, 0" ::: "s40", "s41", "scc");
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(r1) : "v"(lane * 16), "s"(p) : "memory");
      if (MODE == 4) asm volatile("s_add_u32 s40, s40, 1024\n s_addc_u32 s41, s41, 0" ::: "s40", "s41", "scc");
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(r2) : "v"(lane * 16), "s"(p) : "memory");
      if (MODE == 4) asm volatile("s_add_u32 s40, s40, 1024\n s_addc_u32 s41, s41, 0" ::: "s40", "s41", "scc");
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(r3) : "v"(lane * 16), "s"(p) : "memory");
      if (MODE == 4) asm volatile("s_add_u32 s40, s40, 1024\n s_addc_u32 s41, s41, 0" ::: "s40", "s41", "scc");
    } else if (MODE == 5) { // 4 ds_read_b128
      asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      acc += r0.x + r1.y + r2.z + r3.w;
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072"
                   : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(lane * 16) : "memory");
    } else if (MODE == 7) { // ONE global_load_dwordx4 per body (the step kernel's q / l rate: ~4 words per 70 instructions)
      asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      acc += r0.x;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r0) : "v"(lane * 16), "s"(p) : "memory");
    } else if (MODE == 8) { // 4 v_accvgpr_read per body
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(b[i]) : : "a0");
    } else if (MODE == 9) { // 4 global_load_dword per body
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      acc += r0.x + r0.w;
      asm volatile("global_load_dword %0, %4, %5\n global_load_dword %1, %4, %5 offset:256\n global_load_dword %2, %4, %5 offset:512\n global_load_dword %3, %4, %5 offset:768"
                   : "=v"(r0.x), "=v"(r0.y), "=v"(r0.z), "=v"(r0.w) : "v"(lane * 4), "s"(p) : "memory");
    } else if (MODE == 6) { // 8 global_load_dwordx2
      asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      acc += r0.x + r1.y + r2.z + r3.w;
      asm volatile("global_load_dwordx2 %0, %2, %3\n global_load_dwordx2 %1, %2, %3 offset:512" : "=v"(*(float2 *)&r0.x), "=v"(*(float2 *)&r0.z) : "v"(lane * 8), "s"(p) : "memory");
      asm volatile("global_load_dwordx2 %0, %2, %3 offset:1024\n global_load_dwordx2 %1, %2, %3 offset:1536" : "=v"(*(float2 *)&r1.x), "=v"(*(float2 *)&r1.z) : "v"(lane * 8), "s"(p) : "memory");
      asm volatile("global_load_dwordx2 %0, %2, %3 offset:2048\n global_load_dwordx2 %1, %2, %3 offset:2560" : "=v"(*(float2 *)&r2.x), "=v"(*(float2 *)&r2.z) : "v"(lane * 8), "s"(p) : "memory");
      asm volatile("global_load_dwordx2 %0, %2, %3 offset:3072\n global_load_dwordx2 %1, %2, %3 offset:3584" : "=v"(*(float2 *)&r3.x), "=v"(*(float2 *)&r3.z) : "v"(lane * 8), "s"(p) : "memory");
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i & 3]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = acc + r0.x + r1.x + r2.x + r3.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += a[i];
  out[(blockIdx.x * 256 + threadIdx.x) % (32 * 256)] = s;
  if (lane == 0) cyc[wave] = t1 - t0;
}

template <int M> void launch(int blocks, int lds, float *d, const float *s, long long *c, int iters) {
  hipFuncSetAttribute((const void *)k<M>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(256), lds, 0, d, s, c, iters);
}

int main() {
  float *d, *s; long long *c;
  const int iters = 4000, blocks = 256, lds = 160 * 1024;
  CHECK(hipMalloc(&d, 32 * 256 * sizeof(float)));
  CHECK(hipMemset(d, 0, 32 * 256 * sizeof(float)));
  CHECK(hipMalloc(&s, (size_t)blocks * 4 * 4096 * sizeof(float)));
  CHECK(hipMemset(s, 0, (size_t)blocks * 4 * 4096 * sizeof(float)));
  CHECK(hipMalloc(&c, blocks * 4 * sizeof(long long)));
  const char *names[] = {"64 fmac (4 chains) alone", "+ 16 v_accvgpr_read", "+ 16 global_load_dword", "+ 4 global_load_dwordx4",
                         "+ 4 global_load_dwordx4 + 8 SALU", "+ 4 ds_read_b128", "+ 8 global_load_dwordx2", "+ 1 global_load_dwordx4 (4 words)", "+ 4 v_accvgpr_read (4 words)", "+ 4 global_load_dword (4 words)"};
  const int words[] = {16, 16, 16, 16, 16, 16, 16, 4, 4, 4};
  double base = 0;
  for (int m = 0; m < 10; ++m) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      switch (m) {
#define CASE(M) case M: launch<M>(blocks, lds, d, s, c, iters); break;
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9)
      }
      hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 4); CHECK(hipMemcpy(h.data(), c, blocks * 4 * sizeof(long long), hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    const double per = avg / iters;   // s_memtime ticks (100 MHz) per body
    if (m == 0) base = ms;
    printf("%-36s : wall %.3f ms, %.1f ns per body, +%.1f ns for the %d words (%.2f ns per word)\n", names[m], ms, ms * 1e6 / iters,
           (ms - base) * 1e6 / iters, words[m], (ms - base) * 1e6 / iters / words[m]);
    (void)per;
  }
  return 0;
}
