// Micro-benchmark: issue cost of v_exp_f32 (transcendental) vs plain VALU vs MFMA on gfx950, alone and interleaved.
// Answers: how many cycles does one wave64 v_exp_f32 occupy its SIMD, and does it overlap with FMA / MFMA issue?
//   hipcc --offload-arch=gfx950 -O3 -o valu_trans valu_trans.hip && ./valu_trans
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  float r[8], f[8];
  for (int i = 0; i < 8; ++i) { r[i] = -0.001f * (threadIdx.x + i); f[i] = 1.0f + 0.001f * i; }
  f32x4 acc[4] = {};
  bf16x8 a = {}, b = {};
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * threadIdx.x); b[i] = (__bf16)0.5f; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
#define MFMA(i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
#define CVT(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
#define MAX(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&f[(i & 3) * 2]) : "v"(*(double*)&f[((i + 1) & 3) * 2]));
#define FRACT(i) asm volatile("v_fract_f32 %0, %0" : "+v"(f[i]));
#define LDEXP(i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(f[i]) : "v"(3));
#define CVTI(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(f[i]));
    if constexpr (MODE == 0) { REP8(EXP) }                                   // 8 exp
    if constexpr (MODE == 1) { REP8(FMA) }                                   // 8 fma
    if constexpr (MODE == 2) { EXP(0) FMA(0) EXP(1) FMA(1) EXP(2) FMA(2) EXP(3) FMA(3) EXP(4) FMA(4) EXP(5) FMA(5) EXP(6) FMA(6) EXP(7) FMA(7) }   // 8 exp + 8 fma
    if constexpr (MODE == 3) { EXP(0) FMA(0) FMA(1) FMA(2) FMA(3) EXP(1) FMA(4) FMA(5) FMA(6) FMA(7) }   // 2 exp + 8 fma
    if constexpr (MODE == 4) { MFMA(0) MFMA(1) MFMA(2) MFMA(3) }             // 4 mfma
    if constexpr (MODE == 5) { MFMA(0) EXP(0) EXP(1) MFMA(1) EXP(2) EXP(3) MFMA(2) EXP(4) EXP(5) MFMA(3) EXP(6) EXP(7) }   // 4 mfma + 8 exp
    if constexpr (MODE == 6) { MFMA(0) FMA(0) FMA(1) MFMA(1) FMA(2) FMA(3) MFMA(2) FMA(4) FMA(5) MFMA(3) FMA(6) FMA(7) }   // 4 mfma + 8 fma
    if constexpr (MODE == 7) { REP8(CVT) }
    if constexpr (MODE == 8) { REP8(MAX) }
    if constexpr (MODE == 9) { REP8(PKFMA) }                                 // 8 pk_fma (16 fma)
    if constexpr (MODE == 10) { REP8(FRACT) }
    if constexpr (MODE == 11) { REP8(LDEXP) }
    if constexpr (MODE == 12) { REP8(CVTI) }
    if constexpr (MODE == 13) { MFMA(0) EXP(0) FMA(0) FMA(1) EXP(1) FMA(2) MFMA(1) EXP(2) FMA(3) FMA(4) EXP(3) FMA(5) MFMA(2) EXP(4) FMA(6) FMA(7) EXP(5) FMA(0) MFMA(3) EXP(6) FMA(1) FMA(2) EXP(7) FMA(3) }  // 4 mfma + 8 exp + 12 fma
    if constexpr (MODE == 14) { EXP(0) EXP(1) EXP(2) EXP(3) REP8(FMA) EXP(4) EXP(5) EXP(6) EXP(7) REP8(FMA) }   // 8 exp + 16 fma blocked
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += r[i] + f[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int instr_per_iter, int waves_per_simd) {
  const int iters = 20000;
  const int blocks = 256 * waves_per_simd;   // 256-thread blocks = 1 wave per SIMD each
  float* out; unsigned long long* cyc;
  hipMalloc(&out, blocks * 256 * sizeof(float));
  hipMalloc(&cyc, blocks * sizeof(unsigned long long));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, 100);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += (double)v; avg /= blocks;
  // s_memtime ticks per iteration of one wave; with w waves per SIMD the SIMD executes w*instr in that time
  printf("%-44s waves/SIMD %d  ticks/iter/wave %8.1f  ticks per instr (SIMD-level) %6.2f   wall %.3f ms\n", name, waves_per_simd,
         avg / iters, avg / iters / (instr_per_iter * waves_per_simd), ms);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0>("8 x v_exp_f32", 8, w);
    run<1>("8 x v_fma_f32", 8, w);
    run<2>("8 exp + 8 fma interleaved", 16, w);
    run<3>("2 exp + 8 fma", 10, w);
    run<14>("8 exp + 16 fma (blocks of 4/8)", 24, w);
    run<4>("4 x mfma_16x16x32_bf16", 4, w);
    run<5>("4 mfma + 8 exp", 12, w);
    run<6>("4 mfma + 8 fma", 12, w);
    run<13>("4 mfma + 8 exp + 12 fma", 24, w);
    run<7>("8 x v_cvt_pk_bf16_f32", 8, w);
    run<8>("8 x v_max_f32", 8, w);
    run<9>("8 x v_pk_fma_f32", 8, w);
    run<10>("8 x v_fract_f32", 8, w);
    run<11>("8 x v_ldexp_f32", 8, w);
    run<12>("8 x v_cvt_i32_f32", 8, w);
  }
  return 0;
}
