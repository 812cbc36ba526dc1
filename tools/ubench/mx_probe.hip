// Operand layout of the block-scaled fp8 MFMAs on gfx950, determined by experiment (the instruction reference is not
// available offline): v_mfma_scale_f32_32x32x64_f8f6f4 and v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands, unit
// E8M0 scales (0x7f).  Hypothesis checked: lane l holds 32 CONSECUTIVE k of one row / column -
//   32x32x64 : row (col) = l & 31, k = 32 * (l >> 5) + byte;     D: col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
//   16x16x128: row (col) = l & 15, k = 32 * (l >> 4) + byte;     D: col = l & 15, row = 4 * (l >> 4) + r
//   hipcc --offload-arch=gfx950 -O3 mx_probe.hip -o bin/mx_probe && bin/mx_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void k32(const unsigned char* A, const unsigned char* B, float* D, int scale) {
  const int l = threadIdx.x;
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = ((const int*)(A + l * 32))[i]; b[i] = ((const int*)(B + l * 32))[i]; }
  v16f c = {};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale, 0, scale);
  for (int r = 0; r < 16; ++r) D[l * 16 + r] = c[r];
}
__global__ void k16(const unsigned char* A, const unsigned char* B, float* D, int scale) {
  const int l = threadIdx.x;
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = ((const int*)(A + l * 32))[i]; b[i] = ((const int*)(B + l * 32))[i]; }
  v4f c = {};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale, 0, scale);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}

static unsigned char e4m3(int v) {          // small non-negative integers 0..8 as OCP e4m3 (bias 7)
  if (v == 0) return 0;
  int e = 0; while ((1 << (e + 1)) <= v) ++e;             // v = 2^e * (1 + m/8)
  const int m = (v * 8 >> e) - 8;
  return (unsigned char)(((e + 7) << 3) | m);
}

int main() {
  srand(1);
  std::vector<unsigned char> A(64 * 32), B(64 * 32);
  std::vector<int> Ai(64 * 32), Bi(64 * 32);
  for (int i = 0; i < 64 * 32; ++i) { Ai[i] = rand() % 8; Bi[i] = rand() % 8; A[i] = e4m3(Ai[i]); B[i] = e4m3(Bi[i]); }
  unsigned char *dA, *dB; float* dD;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 64 * 16 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  std::vector<float> D(64 * 16);
  for (int scale : {(int)0x7f7f7f7f, (int)0x80808080}) {
    hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dD, scale);
    hipMemcpy(D.data(), dD, 64 * 16 * 4, hipMemcpyDeviceToHost);
    double err = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
      const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      double ref = 0;
      for (int k = 0; k < 64; ++k) ref += Ai[(row + 32 * (k >> 5)) * 32 + (k & 31)] * Bi[(col + 32 * (k >> 5)) * 32 + (k & 31)];
      if (scale == (int)0x80808080) ref *= 4.0;      // 2^1 on each side
      err = fmax(err, fabs(ref - D[l * 16 + r]));
    }
    printf("32x32x64  scale %08x: max |D - ref(hypothesis)| = %g  (D[0] = %g)\n", scale, err, D[0]);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dD, scale);
    hipMemcpy(D.data(), dD, 64 * 4 * 4, hipMemcpyDeviceToHost);
    err = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
      const int col = l & 15, row = 4 * (l >> 4) + r;
      double ref = 0;
      for (int k = 0; k < 128; ++k) ref += Ai[(row + 16 * (k >> 5)) * 32 + (k & 31)] * Bi[(col + 16 * (k >> 5)) * 32 + (k & 31)];
      if (scale == (int)0x80808080) ref *= 4.0;
      err = fmax(err, fabs(ref - D[l * 4 + r]));
    }
    printf("16x16x128 scale %08x: max |D - ref(hypothesis)| = %g  (D[0] = %g)\n", scale, err, D[0]);
  }
  // throughput: back-to-back independent MFMAs, one wave per SIMD x 4 accumulators
  return 0;
}
