#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned char* in, unsigned char* out, float* fo) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = in[i];
  __syncthreads();
  const int l = threadIdx.x;
  // guess: a 16-lane group reads an [8 rows][16 cols] byte block; lane i points at row (i>>1), 8 bytes (i&1); group g -> block g
  const unsigned char* p = lds + (l >> 4) * 128 + ((l & 15) >> 1) * 16 + (l & 1) * 8;
  i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)p);
  unsigned long long u = ((unsigned long long)(unsigned)v[1] << 32) | (unsigned)v[0];
  for (int j = 0; j < 8; ++j) out[l * 8 + j] = (unsigned char)(u >> (8 * j));
  // fp8 conversion + mfma check: A = identity-ish
  float a0 = 1.5f, a1 = -3.0f;
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a0, a1, 0, false);      // low half
  w = __builtin_amdgcn_cvt_pk_fp8_f32(448.0f, 1000.0f, w, true);  // high half
  fo[l] = __builtin_bit_cast(float, w);
  long A = 0x3838383838383838L;  // fp8 e4m3 1.0 = 0x38
  long Bv = 0x4040404040404040L; // 2.0
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(A, Bv, c, 0, 0, 0);
  fo[64 + l] = c[0];
}
int main() {
  unsigned char h[2048], *d, *o, r[512]; float* fo; float fr[128];
  for (int i = 0; i < 2048; ++i) h[i] = i & 255;
  hipMalloc(&d, 2048); hipMalloc(&o, 512); hipMalloc(&fo, 512);
  hipMemcpy(d, h, 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, fo);
  hipMemcpy(r, o, 512, hipMemcpyDeviceToHost); hipMemcpy(fr, fo, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 34; ++l) { printf("lane %2d:", l); for (int j = 0; j < 8; ++j) printf(" %4d", r[l * 8 + j]); printf("\n"); }
  unsigned w; memcpy(&w, &fr[0], 4); printf("cvt word %08x  mfma %f (expect 64)\n", w, fr[64]);
}
