// Micro-benchmark: how v_mfma_f32_16x16x32_bf16 and ds_read_b128 overlap on one CU of gfx950 when they are issued the way
// igemm's pipelined K loop issues them (8 compute waves = 2 per SIMD, per K tile and wave: 2 x [9 ds_read_b128, 20 MFMA],
// one s_barrier), against other orders of the same work.  One 512- or 768-thread workgroup per CU, 256 workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o bin/mfma_lds mfma_lds.hip && ./bin/mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union Frag { uint4 u; bf16x8 h; };

constexpr int NF = 5, MF = 4;
// MODE 0: MFMA only          1: LDS reads only       2: igemm order (burst of 9 reads, 20 MFMA), no barrier
//      3: igemm order + s_barrier per half tile      4: interleaved: one read every 2 MFMA, no barrier
//      5: interleaved + barrier per half             6: as 3 plus 4 extra waves writing 13 KB of LDS each per tile (the DMA's LDS writes)
//      7: as 5 plus the writer waves                 8: as 2, but only ONE compute wave per SIMD (4 waves, twice the trips)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory");
}
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void blds16(unsigned voff, i32x4 srd, unsigned soff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 3\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(srd), "s"(soff), "s"(lds_dst) : "memory");
}
__device__ int g_use_buffer;
//      9: as 3, the writer waves replaced by 4 loader waves streaming 52 KB per tile from global memory with LDS-DMA into a
//         3-stage ring (counted vmcnt, the real kernel's loader loop)      10: as 5 with those loader waves
//      11: loader waves only (no reads, no MFMA; compute waves only take the barriers)
template <int MODE>
__global__ __launch_bounds__(768) void k(float* out, int iters, const unsigned char* __restrict__ gsrc = nullptr, size_t foot = 24u << 20) {
  extern __shared__ unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool writer = wave >= 8;
  Frag wA[NF], xA[MF], wB[NF], xB[MF];
  f32x4 acc[NF][MF] = {};
  for (int i = 0; i < 1024; ++i) ((unsigned*)smem)[threadIdx.x + i * 768 < 36 * 1024 ? threadIdx.x + i * 768 : 0] = 0x3c003c00u;
  __syncthreads();
  // conflict-free ds_read_b128: 64 lanes x 16 B contiguous (what the XOR-swizzled tile gives the real kernel)
  const unsigned char* base = smem + (wave & 7) * 16 * 1024 + lane * 16;
  constexpr bool kDma = MODE >= 9 && MODE <= 11;
  constexpr bool kMfma = MODE != 1 && MODE != 11, kLds = MODE != 0 && MODE != 11,
                 kBar = MODE == 3 || MODE == 5 || MODE == 6 || MODE == 7 || kDma,
                 kInter = MODE == 4 || MODE == 5 || MODE == 7 || MODE == 10, kWr = MODE == 6 || MODE == 7;
#define READ(W, X, OFF)                                                                  \
  if constexpr (kLds) {                                                                  \
    _Pragma("unroll") for (int a = 0; a < NF; ++a) W[a].u = *(const uint4*)(base + (OFF) + a * 1024);     \
    _Pragma("unroll") for (int b = 0; b < MF; ++b) X[b].u = *(const uint4*)(base + (OFF) + (NF + b) * 1024); \
  }
#define MMA(W, X)                                                                        \
  if constexpr (kMfma) {                                                                 \
    _Pragma("unroll") for (int a = 0; a < NF; ++a)                                        \
      _Pragma("unroll") for (int b = 0; b < MF; ++b)                                      \
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[a].h, X[b].h, acc[a][b], 0, 0, 0); \
  } else {                                                                               \
    _Pragma("unroll") for (int a = 0; a < NF; ++a) asm volatile("" ::"v"(W[a].u.x), "v"(W[a].u.w)); \
    _Pragma("unroll") for (int b = 0; b < MF; ++b) asm volatile("" ::"v"(X[b].u.x), "v"(X[b].u.w)); \
  }
  if (writer) {
    if constexpr (kDma) {
      // stage s of the ring = smem + s * 52 KB; this wave fills 13 KB of it per tile
      const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(smem) + (wave - 8) * 13 * 1024);
      const unsigned char* g = gsrc + (wave - 8) * 13 * 1024 + lane * 16;
      const size_t boff = (size_t)blockIdx.x * 45 * 52 * 1024;
      const bool use_buf = g_use_buffer != 0;
      i32x4 srd;
      {
        const unsigned long long b = (unsigned long long)(uintptr_t)gsrc;
        srd[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b); srd[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
        srd[2] = 0x7fffffff; srd[3] = 0x00020000;
      }
      const unsigned vo = (wave - 8) * 13 * 1024 + lane * 16;
      auto fetch = [&](int t, int st) {
        const size_t o = (boff + (size_t)t * 52 * 1024) % foot;
        if (use_buf) {
          const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)o);
#pragma unroll
          for (int j = 0; j < 13; ++j) blds16(vo + j * 1024, srd, so, __builtin_amdgcn_readfirstlane(lds0 + st * 52 * 1024 + j * 1024));
        } else {
#pragma unroll
          for (int j = 0; j < 13; ++j) glds16(g + o + j * 1024, __builtin_amdgcn_readfirstlane(lds0 + st * 52 * 1024 + j * 1024));
        }
      };
      fetch(0, 0); fetch(1, 1);
      asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
      int st = 2;
      for (int it = 0; it < iters; ++it) {
        fetch(it + 2, st);
        st = st == 2 ? 0 : st + 1;
        asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();     // (the compute waves take two barriers per tile in this model: one per half)
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if constexpr (kWr) {
      const uint4 v = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
      unsigned char* dst = smem + 128 * 1024 + (wave - 8) * 6 * 1024 + lane * 16;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int j = 0; j < 6; ++j) *(uint4*)(dst + (j % 6) * 1024) = v;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
      }
    }
    return;
  }
  if (MODE == 8 && wave >= 4) return;
  for (int a = 0; a < NF; ++a) { wA[a].u = make_uint4(0x3c003c00u, 0, 0, 0); wB[a].u = wA[a].u; }
  for (int b = 0; b < MF; ++b) { xA[b].u = make_uint4(0x3c003c00u, 0, 0, 0); xB[b].u = xA[b].u; }
  const int trips = MODE == 8 ? 2 * iters : iters;
  for (int it = 0; it < trips; ++it) {
    if constexpr (!kInter) {
      READ(wB, xB, 0)
      __builtin_amdgcn_sched_barrier(0);
      MMA(wA, xA)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kLds) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (kBar) __builtin_amdgcn_s_barrier();
      READ(wA, xA, 9 * 1024 % 7168)
      __builtin_amdgcn_sched_barrier(0);
      MMA(wB, xB)
      __builtin_amdgcn_sched_barrier(0);
    } else {
      // the same 18 reads and 40 MFMA, one read in front of every second MFMA (sched_group_barrier pins the order)
      READ(wB, xB, 0)
      MMA(wA, xA)
#pragma unroll
      for (int g = 0; g < 9; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMA
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kBar) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
      READ(wA, xA, 9 * 1024 % 7168)
      MMA(wB, xB)
#pragma unroll
      for (int g = 0; g < 9; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = 0;
  for (int a = 0; a < NF; ++a) for (int b = 0; b < MF; ++b) r += acc[a][b][0] + acc[a][b][3];
  r += (float)(wA[0].u.x + xA[0].u.y + wB[1].u.z + xB[2].u.w);
  out[blockIdx.x * 512 + (threadIdx.x & 511)] = r;
}

size_t g_foot = 24u << 20;
template <int MODE>
void run(const char* name) {
  const int iters = 2000, blocks = 256;
  const int threads = (MODE == 6 || MODE == 7 || (MODE >= 9 && MODE <= 11)) ? 768 : 512;
  static unsigned char* g = nullptr;
  if (!g) { (void)hipMalloc(&g, (2048u << 20) + (1u << 20)); (void)hipMemset(g, 0x3c, (2048u << 20) + (1u << 20)); }
  float* out; (void)hipMalloc(&out, blocks * 512 * sizeof(float));
  const int lds = 156 * 1024;
  (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), lds, 0, out, 20, g, g_foot);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), lds, 0, out, iters, g, g_foot);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms * 1e6 / iters;
  printf("%-78s %7.1f ns per K tile = %6.0f cycles at 2.4 GHz  (MFMA alone needs 1280, 147 KB of ds_read at 128 B/clk 1152)\n", name, ns, ns * 2.4);
  (void)hipFree(out);
}

int main() {
  run<0>("0 MFMA only (8 waves x 40 MFMA per tile)");
  run<1>("1 ds_read_b128 only (8 waves x 18 KB per tile)");
  run<2>("2 igemm order: [9 reads | 20 MFMA] x2, no barrier");
  run<3>("3 igemm order + s_barrier per half tile");
  run<4>("4 one read per 2 MFMA, no barrier");
  run<5>("5 one read per 2 MFMA + s_barrier per half tile");
  run<6>("6 = 3 + 4 writer waves (48 KB of ds_write_b128 per tile)");
  run<7>("7 = 5 + 4 writer waves");
  run<9>("9 = 3 with 4 LDS-DMA loader waves (52 KB per tile from global, 3-stage ring)");
  run<10>("10 = 5 with the LDS-DMA loader waves");
  run<11>("11 LDS-DMA loader waves only (compute waves idle at the barriers)");
  for (int ub = 0; ub < 2; ++ub)
  for (size_t f : {(size_t)1 << 20, (size_t)8 << 20, (size_t)96 << 20}) {
    g_foot = f;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_use_buffer), &ub, sizeof(int));
    printf("-- DMA source footprint %zu MB, %s\n", f >> 20, ub ? "buffer_load ... lds" : "global_load_lds");
    run<11>("11 loader waves only");
    run<9>("9 igemm order + loaders");
    run<10>("10 interleaved + loaders");
  }
  run<8>("8 igemm order, ONE compute wave per SIMD (per-wave rate, 2 tiles' worth of trips)");
  return 0;
}
