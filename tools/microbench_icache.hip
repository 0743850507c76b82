// Does a straight-line instruction stream that exceeds the instruction cache slow a lone wave per SIMD down?
// K passes over N independent v_fma_f32 (8 bytes each; 8 chains), 1024 waves of 64 (one per SIMD on 256 CUs), for several N.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench_icache tools/microbench_icache.hip ; run: tools/microbench_icache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int N8>
__global__ void __launch_bounds__(64) body(float *out, int K) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  float c0 = 0.f, c1 = 1.f, c2 = 2.f, c3 = 3.f, c4 = 4.f, c5 = 5.f, c6 = 6.f, c7 = 7.f;
  for (int k = 0; k < K; ++k) {
    asm volatile(
        ".rept %10\n"
        "v_fma_f32 %0, %8, %9, %0\n"
        "v_fma_f32 %1, %8, %9, %1\n"
        "v_fma_f32 %2, %8, %9, %2\n"
        "v_fma_f32 %3, %8, %9, %3\n"
        "v_fma_f32 %4, %8, %9, %4\n"
        "v_fma_f32 %5, %8, %9, %5\n"
        "v_fma_f32 %6, %8, %9, %6\n"
        "v_fma_f32 %7, %8, %9, %7\n"
        ".endr\n"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
        : "v"(a), "v"(b), "n"(N8));
  }
  out[blockIdx.x * 64 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}

template <int N8>
void run(float *out, int waves) {
  const int K = 200;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(body<N8>, dim3(waves), dim3(64), 0, 0, out, K);   // warm-up
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(body<N8>, dim3(waves), dim3(64), 0, 0, out, K);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)N8 * 8 * K;
  printf("waves %5d  body %6d instr = %7.1f KB : %8.3f ms, %6.2f ns per instruction (%.2f cycles at 2.4 GHz)\n", waves, N8 * 8,
         N8 * 64 / 1024.0, ms, ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
}

int main() {
  float *out; hipMalloc(&out, 4096 * 64 * sizeof(float));
  for (int waves : {1024, 256}) {
    run<250>(out, waves); run<500>(out, waves); run<750>(out, waves); run<875>(out, waves); run<1000>(out, waves);
    run<1125>(out, waves); run<1250>(out, waves); run<1500>(out, waves); run<2000>(out, waves);
  }
  return 0;
}
