// Launch-to-launch time of an (almost) empty kernel as a function of its resources: threads per workgroup, dynamic LDS,
// registers (via launch bounds), grid.  What a persistent igemm launch pays before its first instruction and after its last.
//   hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o bin/launch_floor && bin/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NT, int VG>
__global__ __launch_bounds__(NT) void k_empty(int* out, int n) {
  extern __shared__ unsigned char smem[];
  // keep VG registers alive so that the allocation is real
  float v[VG];
#pragma unroll
  for (int i = 0; i < VG; ++i) v[i] = (float)(threadIdx.x + i);
  if (n == 12345) {   // never true
    float s = 0;
#pragma unroll
    for (int i = 0; i < VG; ++i) s += v[i] * smem[i];
    out[threadIdx.x] = (int)s;
  }
}

template <typename K>
float run(K kern, int grid, int nt, size_t lds, int iters) {
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int* d; (void)hipMalloc(&d, 4096);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), lds, 0, d, 0);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), lds, 0, d, 0);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipFree(d);
  return 1e3f * ms / iters;
}

int main() {
  const int iters = 200;
  printf("empty kernel, microseconds per launch (back to back on one stream)\n");
  for (int grid : {256, 512, 1024, 2048}) {
    for (size_t lds : {(size_t)0, (size_t)(48 << 10), (size_t)(114 << 10), (size_t)(160 << 10)}) {
      printf("grid %4d lds %3zu KB:  256 thr/16 regs %5.2f   512 thr/16 regs %5.2f   768 thr/16 regs %5.2f   512 thr/200 regs %5.2f\n", grid,
             lds >> 10, run(k_empty<256, 16>, grid, 256, lds, iters), run(k_empty<512, 16>, grid, 512, lds, iters),
             run(k_empty<768, 16>, grid, 768, lds, iters), run(k_empty<512, 200>, grid, 512, lds, iters));
    }
  }
  return 0;
}
