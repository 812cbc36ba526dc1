// Micro-benchmark: issue cost of v_exp_f32 (transcendental) vs plain VALU vs MFMA on gfx950, alone and interleaved,
// at 1..8 waves per SIMD.  Bodies are 64+ instructions per loop trip (branch overhead < 3 %), sources and destinations
// are distinct registers (no RAW chains) unless the mode says "dep".
//   hipcc --offload-arch=gfx950 -O3 -o bin/valu_trans valu_trans.hip && ./bin/valu_trans
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define R4(X) X(0) X(1) X(2) X(3)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  float s[8], d[8];
  for (int i = 0; i < 8; ++i) { s[i] = -0.001f * (threadIdx.x + i + 1); d[i] = 0.f; }
  f32x4 acc[4] = {};
  bf16x8 a = {}, b = {};
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * threadIdx.x); b[i] = (__bf16)0.5f; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define EXP(i) asm volatile("v_exp_f32 %0, %1" : "=v"(d[i]) : "v"(s[i]));
#define EXPD(i) asm volatile("v_exp_f32 %0, %0" : "+v"(s[i]));          /* dependent on its own previous result */
#define FMA(i) asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(d[i]) : "v"(s[i]));
#define FMAD(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(s[i]));
#define SUB(i) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d[i]) : "v"(s[i]), "v"(s[(i + 1) & 7]));
#define MAX(i) asm volatile("v_max_f32 %0, %1, %2" : "=v"(d[i]) : "v"(s[i]), "v"(s[(i + 1) & 7]));
#define MAX3(i) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d[i]) : "v"(s[i]), "v"(s[(i + 1) & 7]), "v"(s[(i + 2) & 7]));
#define CVT(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d[i]) : "v"(s[i]), "v"(s[(i + 1) & 7]));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %1, %1, %1" : "=v"(*(double*)&d[(i & 3) * 2]) : "v"(*(double*)&s[(i & 3) * 2]));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(*(double*)&d[(i & 3) * 2]) : "v"(*(double*)&s[(i & 3) * 2]));
#define MFMA(i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
#define X8(B) B B B B B B B B
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) { X8(R8(EXP)) }                              // 64 exp
    if constexpr (MODE == 1) { X8(R8(FMA)) }                              // 64 fma
    if constexpr (MODE == 2) { X8(EXP(0) FMA(1) EXP(2) FMA(3) EXP(4) FMA(5) EXP(6) FMA(7)) }   // 32 exp + 32 fma
    if constexpr (MODE == 3) { X8(EXP(0) FMA(1) SUB(2) MAX(3) EXP(4) FMA(5) SUB(6) MAX(7)) }   // 16 exp + 48 plain
    if constexpr (MODE == 4) { X8(R4(MFMA) R4(MFMA)) }                    // 64 mfma
    if constexpr (MODE == 5) { X8(MFMA(0) EXP(0) EXP(1) MFMA(1) EXP(2) EXP(3) MFMA(2) EXP(4) EXP(5) MFMA(3) EXP(6) EXP(7)) }   // 32 mfma + 64 exp
    if constexpr (MODE == 6) { X8(MFMA(0) FMA(0) FMA(1) MFMA(1) FMA(2) FMA(3) MFMA(2) FMA(4) FMA(5) MFMA(3) FMA(6) FMA(7)) }   // 32 mfma + 64 fma
    if constexpr (MODE == 7) { X8(R8(CVT)) }
    if constexpr (MODE == 8) { X8(R8(MAX)) }
    if constexpr (MODE == 9) { X8(R8(PKFMA)) }                            // 64 pk_fma = 128 fma
    if constexpr (MODE == 10) { X8(R8(MAX3)) }
    if constexpr (MODE == 11) { X8(R8(EXPD)) }                            // 8 dependent chains of exp
    if constexpr (MODE == 12) { X8(R8(FMAD)) }                            // 8 dependent chains of fma
    if constexpr (MODE == 13) { X8(MFMA(0) SUB(0) EXP(1) MAX(2) SUB(3) EXP(4) MFMA(1) CVT(5) SUB(6) EXP(7) MAX(0) SUB(1)
                                   MFMA(2) EXP(2) CVT(3) SUB(4) EXP(5) MAX(6) MFMA(3) SUB(7) EXP(0) CVT(1) MAX(2) SUB(3)) }  // softmax-like: 32 mfma, 48 exp, 112 plain
    if constexpr (MODE == 14) { X8(R8(PKMUL)) }
    if constexpr (MODE == 15) { X8(EXPD(0) EXPD(0) EXPD(0) EXPD(0) EXPD(0) EXPD(0) EXPD(0) EXPD(0)) }   // one dependent chain
    if constexpr (MODE == 16) { X8(FMAD(0) FMAD(0) FMAD(0) FMAD(0) FMAD(0) FMAD(0) FMAD(0) FMAD(0)) }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
  for (int i = 0; i < 8; ++i) r += s[i] + d[i];
  for (int i = 0; i < 4; ++i) r += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int instr_per_iter, int waves_per_simd) {
  const int iters = 4000;
  const int blocks = 256 * waves_per_simd;   // 256-thread blocks = 1 wave per SIMD each
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, blocks * 256 * sizeof(float));
  (void)hipMalloc(&cyc, blocks * sizeof(unsigned long long));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, 50);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  (void)hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += (double)v; avg /= blocks;
  // SIMD-level cycles per instruction: the SIMD retires waves_per_simd * instr_per_iter instructions per (avg / iters) cycles
  printf("%-40s w/SIMD %d  cyc/instr/SIMD %6.2f  (wave-cycles/iter %8.1f, clock %.2f GHz)\n", name, waves_per_simd,
         avg / iters / (instr_per_iter * waves_per_simd), avg / iters, avg / (ms * 1e6));
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  const int ws[] = {1, 2, 3, 4, 8};
  for (int w : ws) {
    run<0>("64 exp (indep)", 64, w);
    run<1>("64 fma (indep)", 64, w);
    run<2>("32 exp + 32 fma", 64, w);
    run<3>("16 exp + 48 plain", 64, w);
    run<4>("64 mfma 16x16x32 bf16", 64, w);
    run<5>("32 mfma + 64 exp", 96, w);
    run<6>("32 mfma + 64 fma", 96, w);
    run<13>("softmax mix 32 mfma+48 exp+112 plain", 192, w);
    run<7>("64 cvt_pk_bf16", 64, w);
    run<8>("64 max", 64, w);
    run<10>("64 max3", 64, w);
    run<9>("64 pk_fma", 64, w);
    run<14>("64 pk_mul", 64, w);
    run<11>("64 exp, 8 dep chains", 64, w);
    run<12>("64 fma, 8 dep chains", 64, w);
    run<15>("64 exp, 1 dep chain", 64, w);
    run<16>("64 fma, 1 dep chain", 64, w);
  }
  return 0;
}
