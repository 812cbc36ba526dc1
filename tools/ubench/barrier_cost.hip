// Cost of a workgroup barrier as a function of the workgroup's size, with and without role asymmetry (loader waves that
// spin through their loop much faster than the compute waves) - what a per-K-tile barrier costs the fused kernels.
//   hipcc --offload-arch=gfx950 -O3 barrier_cost.hip -o bin/barrier_cost && bin/barrier_cost
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NT, int WORK>
__global__ __launch_bounds__(NT) void k_bar(float* out, int iters, int ldr_waves) {
  extern __shared__ unsigned char smem[];
  const int wave = threadIdx.x >> 6;
  float v = (float)threadIdx.x;
  const bool is_cmp = wave < (NT / 64 - ldr_waves);
  for (int i = 0; i < iters; ++i) {
    if (is_cmp) {
#pragma unroll
      for (int j = 0; j < WORK; ++j) v = v * 1.0001f + 0.5f;     // dependent VALU chain: ~4-8 cycles each
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (v == 12345.f) out[threadIdx.x] = v;
}

template <typename K>
float run(K kern, int grid, int nt, int iters, int ldr) {
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
  float* d; (void)hipMalloc(&d, 4096);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), 160 << 10, 0, d, iters, ldr);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), 160 << 10, 0, d, iters, ldr);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipFree(d);
  return 1e6f * ms / iters;    // ns per iteration
}

int main() {
  const int iters = 2000;
  printf("ns per loop iteration (one barrier each), 256 workgroups, one per CU (160 KB LDS)\n");
  printf("threads  work=0      work=16     work=64     | with 4 idle 'loader' waves: work=0  work=16  work=64\n");
#define ROW(NT) printf("%5d    %8.1f    %8.1f    %8.1f    |  %8.1f  %8.1f  %8.1f\n", NT, run(k_bar<NT, 0>, 256, NT, iters, 0), \
  run(k_bar<NT, 16>, 256, NT, iters, 0), run(k_bar<NT, 64>, 256, NT, iters, 0), run(k_bar<NT, 0>, 256, NT, iters, 4), \
  run(k_bar<NT, 16>, 256, NT, iters, 4), run(k_bar<NT, 64>, 256, NT, iters, 4));
  ROW(256) ROW(512) ROW(768) ROW(1024)
  return 0;
}
