// Model of a 256x160 igemm K tile on FOUR compute waves (2 x 2, wave tile 128 x 80: 5 W x 8 X fragments, 160 accumulator
// registers) + 4 LDS-DMA loader waves, against the shipped 8-compute-wave form (64 x 80 wave tiles) modelled in
// mfma_lds.hip: the LDS fragment reads per tile drop from 147 KB to 106 KB.  The X fragments are streamed through a
// 3-deep register ring (one ds_read_b128 in front of each group of 5 MFMAs, two fragments ahead), the next half tile's
// W fragments are fetched meanwhile; one s_barrier per tile, placed mid-tile so that the reads that run ahead into the
// next tile are covered.
//   hipcc --offload-arch=gfx950 -O3 -o bin/mfma_lds2 mfma_lds2.hip && ./bin/mfma_lds2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union Frag { uint4 u; bf16x8 h; };

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory");
}

constexpr int NF = 5, MF = 8;
// MODE 0: compute waves only (no DMA)   1: + loader waves   2: loader waves only
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, const unsigned char* __restrict__ gsrc, size_t foot) {
  extern __shared__ unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 39 * 1024; i += 512) ((unsigned*)smem)[i] = 0x3c003c00u;
  __syncthreads();
  if (wave >= 4) {
    if constexpr (MODE >= 1) {
      const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(smem) + (wave - 4) * 13 * 1024);
      const unsigned char* g = gsrc + (wave - 4) * 13 * 1024 + lane * 16;
      const size_t boff = (size_t)blockIdx.x * 45 * 52 * 1024;
      auto fetch = [&](int t, int st) {
        const size_t o = (boff + (size_t)t * 52 * 1024) % foot;
#pragma unroll
        for (int j = 0; j < 13; ++j) glds16(g + o + j * 1024, __builtin_amdgcn_readfirstlane(lds0 + st * 52 * 1024 + j * 1024));
      };
      fetch(0, 0); fetch(1, 1);
      asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
      int st = 2;
      for (int it = 0; it < iters; ++it) {
        fetch(it + 2, st);
        st = st == 2 ? 0 : st + 1;
        asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_barrier();
    }
    return;
  }
  if constexpr (MODE == 2) {
    for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_barrier();
    return;
  }
  const unsigned char* base = smem + wave * 16 * 1024 + lane * 16;   // conflict-free 1 KB reads
  Frag wc[NF], wn[NF], xr[3];
  f32x4 acc[NF][MF] = {};
#define RD(F, OFF) F.u = *(const uint4*)(base + (OFF));
#pragma unroll
  for (int a = 0; a < NF; ++a) RD(wc[a], a * 1024)
  RD(xr[0], 5 * 1024) RD(xr[1], 6 * 1024)
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (half == 1) __builtin_amdgcn_s_barrier();       // next tile landed / previous stage released
#define GROUP(b, NRD)                                                                               \
  {                                                                                                    \
    RD(xr[((b) + 2) % 3], ((5 + (b) + 2) % 13) * 1024 + half * 13 * 1024 % 3072)                       \
    if ((b) < NF) RD(wn[(b) < NF ? (b) : 0], (b) * 1024 + (1 - half) * 7 * 1024 % 2048)                  \
    _Pragma("unroll") for (int a = 0; a < NF; ++a)                                                      \
      acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[a].h, xr[(b) % 3].h, acc[a][b], 0, 0, 0);  \
    __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);                                               \
    __builtin_amdgcn_sched_group_barrier(0x008, NF, 0);                                                \
  }
      GROUP(0, 2) GROUP(1, 2) GROUP(2, 2) GROUP(3, 2) GROUP(4, 2) GROUP(5, 1) GROUP(6, 1) GROUP(7, 1)
#pragma unroll
      for (int a = 0; a < NF; ++a) wc[a] = wn[a];
    }
  }
  float r = 0;
  for (int a = 0; a < NF; ++a) for (int b = 0; b < MF; ++b) r += acc[a][b][0] + acc[a][b][3];
  out[blockIdx.x * 256 + (threadIdx.x & 255)] = r;
}

template <int MODE>
void run(const char* name, size_t foot) {
  const int iters = 2000, blocks = 256;
  float* out; (void)hipMalloc(&out, blocks * 256 * sizeof(float));
  static unsigned char* g = nullptr;
  if (!g) { (void)hipMalloc(&g, (200u << 20)); (void)hipMemset(g, 0x3c, 200u << 20); }
  const int lds = 156 * 1024;
  (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), lds, 0, out, 20, g, foot);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), lds, 0, out, iters, g, foot);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-70s footprint %4zu MB: %7.1f ns per K tile (MFMA alone 610; shipped form, mfma_lds modes 9/10: 908/817 L2, 1306/1158 MALL)\n", name, foot >> 20, ms * 1e6 / iters);
  (void)hipFree(out);
}

int main() {
  hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void*)k<1>);
  printf("k<1>: %d VGPRs (+AGPRs), %zu B scratch\n", fa.numRegs, (size_t)fa.localSizeBytes);
  run<0>("4 compute waves x (128x80), streamed X ring, no DMA", (size_t)1 << 20);
  for (size_t f : {(size_t)1 << 20, (size_t)8 << 20, (size_t)96 << 20}) {
    run<1>("4 compute waves x (128x80) + 4 loader waves", f);
    run<2>("loader waves only", f);
  }
  return 0;
}
