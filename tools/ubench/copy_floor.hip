// Floor for the HBM-bound glue kernels (GroupNorm apply / statistics, split-K finish): how long does the plainest possible
// streaming kernel take on tensors of the UNet's sizes (1.3 - 63 MB)?  Read-only (sum), and read + write (copy), 16 B per
// lane, launched back to back (so the source is cache/MALL-warm like an activation the previous kernel just wrote) and
// rotated over 16 buffers (cold).
//   hipcc --offload-arch=gfx950 -O3 -o bin/copy_floor copy_floor.hip && ./bin/copy_floor
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void copy_k(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void sum_k(const uint4* __restrict__ src, unsigned* out, size_t n) {
  unsigned a = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint4 v = src[i]; a += v.x ^ v.y ^ v.z ^ v.w; }
  if (a == 0x12345678u) out[0] = a;
}

int main() {
  const size_t sizes[] = {1310720, 5242880, 10485760, 20971520, 41943040, 62914560};
  const int NB = 16;
  char* buf; unsigned* out;
  (void)hipMalloc(&buf, (size_t)NB * 62914560 * 2); (void)hipMalloc(&out, 64);
  (void)hipMemset(buf, 1, (size_t)NB * 62914560 * 2);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (size_t bytes : sizes) {
    const size_t n = bytes / 16;
    for (int blocks : {512, 1024, 2048, 4096}) {
      float ms[4];
      for (int mode = 0; mode < 4; ++mode) {       // 0 copy warm, 1 copy cold, 2 sum warm, 3 sum cold
        const bool cold = mode & 1, sum = mode >= 2;
        const int iters = 64;
        for (int rep = 0; rep < 2; ++rep) {
          (void)hipEventRecord(e0);
          for (int it = 0; it < iters; ++it) {
            const uint4* s = (const uint4*)(buf + (cold ? (size_t)(it % NB) * 62914560 * 2 : 0));
            uint4* d = (uint4*)(buf + (cold ? (size_t)(it % NB) * 62914560 * 2 : 0) + 62914560);
            if (sum) hipLaunchKernelGGL(sum_k, dim3(blocks), dim3(256), 0, 0, s, out, n);
            else hipLaunchKernelGGL(copy_k, dim3(blocks), dim3(256), 0, 0, s, d, n);
          }
          (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
          (void)hipEventElapsedTime(&ms[mode], e0, e1);
          ms[mode] = ms[mode] * 1e3f / iters;
        }
      }
      printf("%6.1f MB, %4d workgroups: copy %6.1f us warm (%4.2f TB/s r+w) %6.1f us cold | read-only %6.1f us warm (%4.2f TB/s) %6.1f us cold\n",
             bytes / 1048576.0, blocks, ms[0], 2.0 * bytes / ms[0] * 1e-6, ms[1], ms[2], bytes / ms[2] * 1e-6, ms[3]);
    }
  }
  return 0;
}
