#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = in[i];
  __syncthreads();
  const int l = threadIdx.x;
  // lane i of a 16-group points at row (i>>2), cols (i&3)*4 of a row-major [4][16] block; group g -> block g
  const unsigned short* p = lds + (l >> 4) * 64 + ((l & 15) >> 2) * 16 + (l & 3) * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short h[1024], *d, *o, r[256];
  for (int i = 0; i < 1024; ++i) h[i] = i;
  hipMalloc(&d, 2048); hipMalloc(&o, 512);
  hipMemcpy(d, h, 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", r[l * 4 + j]); printf("\n"); }
}
