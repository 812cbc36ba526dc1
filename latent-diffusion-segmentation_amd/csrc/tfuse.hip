// Row-local fusion of the transformer feed-forward at the 320-channel level (64x64 maps at 512x512, 128x128 at 1024x1024):
//
//     h   <- h + ff.net.2( GEGLU( ff.net.0.proj( LayerNorm_3(h) ) ) )            (diffusers BasicTransformerBlock,
//     out <- proj_out(h) + x                                                      /root/reference/ldmseg/models/unet.py:401-425)
//
// in ONE launch per transformer instead of five (rowstats, GEGLU GEMM, ff.net.2 GEMM, proj_out GEMM + their HBM round
// trips: the [M, 1280] hidden tensor - 84 MB at B = 8 - was written and read back, h was read three times).
//
// A workgroup owns 128 whole rows (tokens) of h.  The tile (128 x 320 bf16 = 80 KB) is brought into the LDS once, its rows
// are LayerNorm-ed in place (statistics + normalisation by the compute waves, two-pass like launch_rowstats; gamma / beta
// are folded into the GEGLU weights and bias at create time), and it then serves as the X operand of EVERY hidden chunk:
//
//     for each chunk of 64 hidden columns (20 of them):
//        acc1[128 x 128]  = X_ln[128 x 320] . W1_chunk^T      (value | gate columns interleaved in 16-column pairs)
//        H[128 x 64]      = (value + b) * gelu_erf(gate + b)   -> bf16 into the LDS (the next product's X operand)
//        acc2[128 x 320] += H . W2_chunk^T                      (accumulators live in registers for the whole kernel)
//     h_new = acc2 + b2 + h;   (proj_out:) the bf16 tile goes back into the LDS, acc3 = h_new . Wp^T + bp + x
//
// Only weights stream.  They are pre-packed at create time into ONE linear byte stream in exactly the order they are
// consumed (per chunk: five [128 rows][128 B] K tiles of W1, then [320 rows][128 B] of W2; then five [320][128 B] K tiles of
// proj_out), with the LDS XOR swizzle already applied, so the four loader waves are a memcpy: piece g (1 KiB, one
// LDS-DMA wave instruction) goes from stream + g KiB to ring + (g KiB mod 64 KiB).  Every workgroup reads the same 2.6 MB,
// which therefore lives in the L2 of its XCD.  8 compute waves (2 x 4: 64-row x 32 / 80-column wave tiles) + 4 loader
// waves, <= 168 registers; LDS = 80 KB tile + 16 KB hidden chunk + 64 KB ring = all 160 KB of the CU.
#include <hip/hip_runtime.h>

#include "common.h"
#include "kernels.h"

namespace ldmseg {
namespace {

constexpr int kBM = 128;                 // rows per workgroup
constexpr int kC = 320;                  // channels
constexpr int kKT = kC / 64;             // K tiles of the tile
constexpr int kHC = 64;                  // hidden columns per chunk
constexpr int kXT = kKT * kBM * 128;     // 81920: the row tile, [K tile][row][128 B], 16-B chunk index XOR (row & 7)
constexpr int kHT = kBM * 128;           // 16384: one hidden chunk, same layout
constexpr int kRing = 65536;
constexpr int kLds = kXT + kHT + kRing;  // 163840 = the CU's whole LDS
constexpr int kW1Tile = 128 * 128;       // bytes of one W1 K tile (128 packed GEGLU rows)
constexpr int kW2Unit = kC * 128;        // bytes of a [320 rows][128 B] unit (W2 chunk, proj_out K tile)
constexpr int kChunkBytes = kKT * kW1Tile + kW2Unit;   // 122880
constexpr int kW2A = 192 * 128, kW2B = 128 * 128;     // the W2 chunk as two units: output fragments 0..2 / 3..4 of every wave
constexpr int kPUnit = 192 * 128;        // 24576: proj_out unit = [160 rows of one N half + 32 rows of padding][128 B] (two consecutive
                                         // 40 KB units would not fit the ring; 24 KB keeps every unit a multiple of the loaders' 8 KB round)

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off"
      :
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
__device__ __forceinline__ void glds16_sbase(unsigned voff, const void* sbase, unsigned lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 3\n\t"   // (M0 write -> LDS-DMA: 1 wait state; SGPR base written by SALU -> VMEM: 5; see igemm.hip)
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}
// two consecutive pieces (2 KiB) from one base: the instruction's immediate offset moves the LDS address along with the global one
__device__ __forceinline__ void glds16x2_sbase(unsigned voff, const void* sbase, unsigned lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 3\n\t"
      "global_load_lds_dwordx4 %0, %1\n\t"
      "global_load_lds_dwordx4 %0, %1 offset:1024"
      :
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}
// at most n of this wave's LDS-DMA pieces still in flight (n <= 16: the wave's share of the ring)
__device__ __forceinline__ void wait_pieces(int n) {
  switch (n) {
#define TF_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    TF_W(0) TF_W(1) TF_W(2) TF_W(3) TF_W(4) TF_W(5) TF_W(6) TF_W(7) TF_W(8) TF_W(9) TF_W(10) TF_W(11) TF_W(12) TF_W(13)
    TF_W(14) TF_W(15)
#undef TF_W
    default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
  }
}

struct MlpFusedParams {
  const bf16_t* x;         // [M][320] h (un-normalised): LayerNorm input and the residual of ff.net.2
  bf16_t* out;             // [M][320]: h_new (PROJ = false; may alias x) or proj_out(h_new) + x2 (PROJ = true)
  const bf16_t* x2;        // PROJ: residual of proj_out (the transformer's input), [M][320]
  const unsigned char* stream;   // packed weights (tf_pack_stream_kernel)
  const float* bias1;      // [2560] GEGLU bias in packed row order (W beta + b)
  const float* bias2;      // [320]
  const float* bias3;      // [320] proj_out bias (PROJ)
  const void* zeros;       // >= 16 B of zeros
  int M, nchunks;
  float eps;
  int rot_tiles;           // > 0: tiles per image; the workgroup starts its hidden-chunk sweep at a chunk derived from its tile index inside the image
  int dbg;                 // -DLDMSEG_TFUSE_ABLATE builds: phase-ablation flags (results are wrong when != 0)
  int ldr_delay;           // tuning: loader waves sleep 64 * {0, 2, 4, 8}[ldr_delay] cycles after a barrier before they issue
};
#ifdef LDMSEG_TFUSE_ABLATE
#define TFDBG(p, bit) ((LDMSEG_TFUSE_ABLATE) & (bit))     // compile-time mask: run-time flags cost registers (the 168-budget kernel spilled)
__device__ unsigned long long g_tf_ts[64];
#define TFSTAMP(slot) if (blockIdx.x == 0 && threadIdx.x == 0) g_tf_ts[slot] = wall_clock64();
#define TFSTAMP_L(slot, cond) if ((cond) && blockIdx.x == 0 && threadIdx.x == 512) g_tf_ts[slot] = wall_clock64();
#else
#define TFDBG(p, bit) 0
#define TFSTAMP(slot)
#define TFSTAMP_L(slot, cond)
#endif

template <bool PROJ>
__global__ __launch_bounds__(768, 3) void mlp_fused_kernel(const MlpFusedParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * kBM;
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
  // units of the weight stream, in consumption order: per chunk W1 K tiles {0,1} (8 pieces per loader wave), {2,3} (8), {4} (4),
  // W2 part A (6), W2 part B (4); PROJ: 10 x 6
  const int nunits = p.nchunks * 5 + (PROJ ? 2 * kKT : 0);
  // Every workgroup consumes the same stream; started at the same chunk they would all ask the L2 for the same lines at the
  // same moment.  The sweep over the hidden chunks is a sum, so each tile starts it elsewhere (a function of the tile's index
  // INSIDE its image only: an image's result does not depend on its position in the batch).
  const int start = p.rot_tiles > 0 ? (int)(((unsigned)blockIdx.x % (unsigned)p.rot_tiles) * 7u % (unsigned)p.nchunks) : 0;

  if (wave_id >= 8) {
    // ================= loader waves: the row tile, then the weight stream through the ring =================
    const int lw = wave_id - 8;
    {
      const int ld_r = lane >> 3, ld_j = (lane & 7) ^ ld_r;
#pragma unroll 4
      for (int i = 0; i < (kKT * kBM / 8) / 4; ++i) {                 // 80 pieces of [8 rows][128 B], 20 per wave
        const int q = lw + 4 * i;
        const int kt = q >> 4, rg = q & 15;
        const int m = m0 + rg * 8 + ld_r;
        const void* src = m < p.M ? (const void*)((const unsigned char*)p.x + (size_t)m * (kC * 2) + kt * 128 + ld_j * 16) : p.zeros;
        glds16(src, __builtin_amdgcn_readfirstlane(lds0 + kt * (kBM * 128) + rg * 1024));
      }
    }
    // This wave's share of a unit's pieces: 4 (W1 tile), 10 (W2 chunk), 5 (proj_out unit); its share of the ring: 16.
    // What the first version got wrong (s_memrealtime stamps, tools/ff_ablate.py): the compute waves sat at every barrier
    // waiting for THIS loop - bookkeeping with runtime divisions, the issue of the next pieces in front of the barrier, and a
    // loader wave gets an issue slot only every ~12 cycles next to two busy compute waves of its SIMD.  Now: raised priority,
    // pointer increments instead of index arithmetic, and per step  wait(unit t+1 landed) -> barrier -> issue into the space
    // unit t just freed  (the issue runs in the shadow of the compute waves' next step).
    __builtin_amdgcn_s_setprio(3);
    const int nchunk_units = p.nchunks * 5;
    const int chunk_pieces = p.nchunks * 30;                          // this wave's pieces of the chunk region (circular: rotation)
    int left = chunk_pieces + (PROJ ? 2 * kKT * 6 : 0);               // pieces this wave still has to issue
    int iss = 0;                                                      // issued so far
    const unsigned voff = (unsigned)lane * 16u;
    // Every unit is a multiple of 8 KiB, so the stream is a sequence of 8 KiB rounds in which wave lw owns the 2 KiB block
    // (two pieces) at offset 2 KiB * lw: per block one 64-bit add, one ring-offset add + and, one M0 write and two DMA
    // instructions (the first version spent ~22 scalar instructions and three taken branches per piece).
    const unsigned char* src = p.stream + (size_t)start * kChunkBytes + (size_t)lw * 2048;
    int to_wrap = (p.nchunks - start) * 15;                           // blocks until the source wraps to chunk 0 / moves on to proj_out
    unsigned dst = (unsigned)lw * 2048u;                              // ring offset of the next block
    auto issue_n = [&](int n) __attribute__((always_inline)) {
      for (int i = 0; i < n && left > 0; i += 2) {
        if (!TFDBG(p, 1)) {
          const unsigned long long su = (unsigned long long)(uintptr_t)src;
          const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)su), hi = __builtin_amdgcn_readfirstlane((unsigned)(su >> 32));
          glds16x2_sbase(voff, (const void*)(uintptr_t)(((unsigned long long)hi << 32) | lo),
                         __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(kXT + kHT) + dst));
        }
        src += 8192;
        dst = (dst + 8192u) & (unsigned)(kRing - 1);
        iss += 2;
        left -= 2;
        if (--to_wrap == 0) {
          if (iss < chunk_pieces) { src = p.stream + (size_t)lw * 2048; to_wrap = start * 15; }                        // chunk 0 follows the last chunk
          else { src = p.stream + (size_t)p.nchunks * kChunkBytes + (size_t)lw * 2048; to_wrap = 0x7fffffff; }         // proj_out region
        }
      }
    };
    issue_n(16);
    wait_pieces(iss);                       // the row tile has landed (its pieces were issued first)
    __syncthreads();                        // A: tile ready for the LayerNorm pass
    int land = 8;                           // this wave's pieces through the unit that must have landed next (unit 0: two W1 tiles)
    wait_pieces(iss - land);
    __syncthreads();                        // B: tile normalised, unit 0 landed
    int pos = 0;                            // position of unit t inside its chunk
    for (int t = 0; t < nunits; ++t) {
      if (PROJ && t == nchunk_units) asm volatile("s_barrier" ::: "memory");   // pairs with the barrier between the ff epilogue and proj_out's K loop
      const int sh_t = (t < nchunk_units) ? ((0x46488 >> (4 * pos)) & 15) : 6;          // shares by position: 8, 8, 4, 6, 4
      if (t + 1 < nunits) {
        const int pn = pos == 4 ? 0 : pos + 1;
        land += (t + 1 < nchunk_units) ? ((0x46488 >> (4 * pn)) & 15) : 6;              // share of unit t+1
        wait_pieces(iss - land);            // (steady state of the chunk loop: 0, 4, 6, 6, 4 pieces may stay in flight)
      }
      asm volatile("s_barrier" ::: "memory");           // end of step t: unit t+1 has landed, unit t is free
      if (p.ldr_delay == 1) __builtin_amdgcn_s_sleep(2);
      issue_n(sh_t);
      pos = (pos == 4) ? 0 : pos + 1;
    }
    return;
  }

  // ================= compute waves =================
  const int wm = wave_id >> 2, wn = wave_id & 3;
  const int lg = lane >> 4, lq = lane & 15;
  unsigned char* XT = smem;
  unsigned char* HT = smem + kXT;
  unsigned char* RING = smem + kXT + kHT;
  const int fr_row = lq * 128;
  const int fr_c0 = ((lg) ^ (lane & 7)) * 16;
  const int fr_c1 = ((lg + 4) ^ (lane & 7)) * 16;

  TFSTAMP(0)
  __syncthreads();                          // A: the row tile has landed
  TFSTAMP(1)
  {
    // ---- LayerNorm of the tile's rows in place (no affine: gamma / beta live in the GEGLU weights and bias): 4 lanes per
    // row, 10 x 16 B each; two-pass mean / centred variance like launch_rowstats
    const int r = wave_id * 16 + (lane >> 2), q = lane & 3;
    uint4 raw[10];
    unsigned char* rowp = XT + r * 128;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int j = q + 4 * i;
      raw[i] = *(const uint4*)(rowp + (j >> 3) * (kBM * 128) + (((j & 7) ^ (r & 7)) << 4));
    }
    float f[10][8];
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      Chunk<bf16_t>::unpack(raw[i], f[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) sm += f[i][e];
    }
    sm += dpp_f<0xB1>(sm);
    sm += dpp_f<0x4E>(sm);
    const float mean = sm * (1.0f / kC);
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = f[i][e] - mean; qq += d * d; }
    qq += dpp_f<0xB1>(qq);
    qq += dpp_f<0x4E>(qq);
    const float rstd = 1.0f / sqrtf(qq * (1.0f / kC) + p.eps);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int j = q + 4 * i;
      float y[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = (f[i][e] - mean) * rstd;
      *(uint4*)(rowp + (j >> 3) * (kBM * 128) + (((j & 7) ^ (r & 7)) << 4)) = Chunk<bf16_t>::pack(y);
    }
  }
  TFSTAMP(2)
  __syncthreads();                          // B: normalised tile visible, unit 0 landed
  TFSTAMP(3)

  f32x4 acc2[5][4];
#pragma unroll
  for (int a = 0; a < 5; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc2[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  unsigned uoff = 0;                        // ring offset of the current unit
  const unsigned char* xrow = XT + (wm * 64) * 128 + fr_row;
  const unsigned char* hrow = HT + (wm * 64) * 128 + fr_row;
  constexpr unsigned RM = (unsigned)(kRing - 1);
  int c = start;
  for (int ci = 0; ci < p.nchunks; ++ci, c = (c + 1 == p.nchunks) ? 0 : c + 1) {
    f32x4 acc1[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc1[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ci == 5) { TFSTAMP(11) }
    const float* bp = p.bias1 + c * 128 + wn * 32 + lg * 4;
    const f32x4 bv = *(const f32x4*)bp, bg = *(const f32x4*)(bp + 16);
    // one K tile of GEMM1; W1 tile rows of this wave: [wn*32, wn*32 + 32) = one (value | gate) pair of 16-row blocks (a 2 KB
    // row block never wraps around the ring: units start on 8 KB boundaries)
    auto gemm1_tile = [&](int kt, unsigned tile_off) __attribute__((always_inline)) {
      const unsigned char* w0 = RING + ((uoff + tile_off + (unsigned)(wn * 32) * 128u) & RM) + fr_row;
      const unsigned char* w1 = RING + ((uoff + tile_off + (unsigned)(wn * 32 + 16) * 128u) & RM) + fr_row;
      const unsigned char* xs = xrow + kt * (kBM * 128);
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        const int co = kg ? fr_c1 : fr_c0;
        uint4 xf[4], wf[2];
        wf[0] = *(const uint4*)(w0 + co);
#pragma unroll
        for (int b = 0; b < 4; ++b) xf[b] = *(const uint4*)(xs + b * 2048 + co);
        wf[1] = *(const uint4*)(w1 + co);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) mma_kgroup<bf16_t>(wf[a], xf[b], acc1[a][b]);
      }
    };
    // Steps are TWO K tiles wide where the ring allows it: a barrier step costs ~0.4 us of loop / issue / barrier overhead
    // whatever is in it (stamps: an empty step took as long), and one K tile is only 16 MFMAs per wave (0.12 us)
    gemm1_tile(0, 0u);
    gemm1_tile(1, (unsigned)kW1Tile);
    __syncthreads();
    uoff = (uoff + 2u * kW1Tile) & RM;
    gemm1_tile(2, 0u);
    gemm1_tile(3, (unsigned)kW1Tile);
    __syncthreads();
    uoff = (uoff + 2u * kW1Tile) & RM;
    gemm1_tile(4, 0u);
    // GEGLU on the finished chunk: value and gate of 16 hidden columns sit in the same lane.  The bf16 result is this
    // chunk's X operand for ff.net.2: row m, hidden k = wn*16 + lg*4 + {0..3} -> 8 bytes of HT[m][k]
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const f32x4 av = acc1[0][b] + bv, gv = acc1[1][b] + bg;
      const f32x4 o = TFDBG(p, 2) ? av * gv : av * gelu_erf_bf16_f4(gv);
      const int row = wm * 64 + b * 16 + lq;
      const int ch = wn * 2 + (lg >> 1);
      *(uint2*)(HT + row * 128 + ((ch ^ (row & 7)) << 4) + (lg & 1) * 8) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
    }
    __syncthreads();
    uoff = (uoff + (unsigned)kW1Tile) & RM;
    if (ci == 5) { TFSTAMP(12) }
    // ---- acc2 += H . W2_chunk^T in two units: A = this wave's output fragments 0..2 (rows wn*48 + a*16 of a 192-row unit),
    //      B = fragments 3, 4 (rows wn*32 + (a-3)*16 of a 128-row unit): every wave works in both steps
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      const int co = kg ? fr_c1 : fr_c0;
      uint4 hf[4], wf[3];
#pragma unroll
      for (int b = 0; b < 4; ++b) hf[b] = *(const uint4*)(hrow + b * 2048 + co);
#pragma unroll
      for (int a = 0; a < 3; ++a) wf[a] = *(const uint4*)(RING + ((uoff + (unsigned)(wn * 48 + a * 16) * 128u) & RM) + fr_row + co);
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) mma_kgroup<bf16_t>(wf[a], hf[b], acc2[a][b]);
    }
    __syncthreads();
    uoff = (uoff + (unsigned)kW2A) & RM;
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      const int co = kg ? fr_c1 : fr_c0;
      uint4 hf[4], wf[2];
#pragma unroll
      for (int b = 0; b < 4; ++b) hf[b] = *(const uint4*)(hrow + b * 2048 + co);
#pragma unroll
      for (int a = 0; a < 2; ++a) wf[a] = *(const uint4*)(RING + ((uoff + (unsigned)(wn * 32 + a * 16) * 128u) & RM) + fr_row + co);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) mma_kgroup<bf16_t>(wf[a], hf[b], acc2[3 + a][b]);
    }
    __syncthreads();
    uoff = (uoff + (unsigned)kW2B) & RM;
    if (ci == 0) { TFSTAMP(4) }
    if (ci == 5) { TFSTAMP(13) }
    if (ci == 9) { TFSTAMP(5) }
  }
  TFSTAMP(6)

  // ================= ff.net.2 epilogue: + bias + h =================
  const int nl = wn * 80 + lg * 4;          // this lane's first column of fragment 0 (+ a*16)
  if constexpr (!PROJ) {
    // row-major through the LDS (the tile area is free: the last chunk's barriers are behind us): each wave parks a
    // 16-row x 80-column fp32 block, then every lane moves 16 output bytes of one row (see igemm.hip epilogue_rows)
    constexpr int SROW = 80 * 4 + 16, CPR = 10, NCH = 160, NI = 3;
    unsigned char* stg = XT + wave_id * (16 * SROW);
    f32x4 b2[5];
#pragma unroll
    for (int a = 0; a < 5; ++a) b2[a] = *(const f32x4*)(p.bias2 + nl + a * 16);
    const int m_wave = m0 + wm * 64;
    uint4 rcur[NI], rnext[NI];
    auto load_res = [&](int b, uint4 (&r)[NI]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int cidx = lane + i * 64;
        const int row = cidx / CPR, cc = cidx - row * CPR;
        const int m = m_wave + b * 16 + row;
        if (cidx < NCH && m < p.M) r[i] = *(const uint4*)(p.x + (size_t)m * kC + wn * 80 + cc * 8);
      }
    };
    load_res(0, rcur);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
      for (int a = 0; a < 5; ++a) *(f32x4*)(stg + lq * SROW + (a * 16 + lg * 4) * 4) = acc2[a][b] + b2[a];
      if (b + 1 < 4) load_res(b + 1, rnext);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int cidx = lane + i * 64;
        if (cidx < NCH) {
          const int row = cidx / CPR, cc = cidx - row * CPR;
          const int m = m_wave + b * 16 + row;
          if (m < p.M) {
            float v[8], r[8];
            const unsigned char* sp = stg + row * SROW + cc * 32;
            const f32x4 t0 = *(const f32x4*)sp, t1 = *(const f32x4*)(sp + 16);
            Chunk<bf16_t>::unpack(rcur[i], r);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = t0[e] + r[e]; v[4 + e] = t1[e] + r[4 + e]; }
            *(uint4*)(p.out + (size_t)m * kC + wn * 80 + cc * 8) = Chunk<bf16_t>::pack(v);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < NI; ++i) rcur[i] = rnext[i];
    }
  } else {
    // (`zero` is opaque to the optimiser: every lane-dependent address of this branch is derived from it, so none of them is computed
    // in front of the chunk loop and kept alive through it - that cost the 4 VGPRs by which this variant exceeded its 168-register
    // budget, i.e. an accumulator fragment parked in scratch and reloaded, with a vmcnt(0) wait, in every chunk)
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    const int elane = lane + zero;
    const int elg = elane >> 4, elq = elane & 15;
    const int efr_row = elq * 128;
    const int efr_c0 = ((elg) ^ (elane & 7)) * 16, efr_c1 = ((elg + 4) ^ (elane & 7)) * 16;
    const int enl = wn * 80 + elg * 4;
    // ---- h_new = acc2 + b2 + h as bf16 back into the tile area (it is proj_out's X operand), then
    //      out = h_new . Wp^T + bp + x2 on the same wave tiles (acc2's registers are reused)
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      const int n = enl + a * 16;
      const f32x4 b2 = *(const f32x4*)(p.bias2 + n);
      const int kt = n >> 6, ch = (n & 63) >> 3, half = (n >> 2) & 1;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int row = wm * 64 + b * 16 + elq;
        const int m = m0 + row;
        uint2 rr = make_uint2(0u, 0u);
        if (m < p.M) rr = *(const uint2*)(p.x + (size_t)m * kC + n);
        f32x4 v = acc2[a][b] + b2;
        v[0] += bits_f32(rr.x << 16); v[1] += bits_f32(rr.x & 0xffff0000u);
        v[2] += bits_f32(rr.y << 16); v[3] += bits_f32(rr.y & 0xffff0000u);
        *(uint2*)(XT + kt * (kBM * 128) + row * 128 + ((ch ^ (row & 7)) << 4) + half * 8) =
            make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      }
    }
    TFSTAMP(7)
    __syncthreads();                        // the new tile is visible to every wave
    TFSTAMP(8)
    // proj_out on 4 x 2 waves per N half (units are [160 rows of half h][128 B]): wave tile 32 rows x 80 columns per half
    const int pm = wave_id >> 1, pn = wave_id & 1;
    f32x4 acc3[2][5][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc3[h][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* prow = XT + (pm * 32) * 128 + efr_row;
#pragma unroll 1
    for (int kt = 0; kt < kKT; ++kt) {
      const unsigned char* xs = prow + kt * (kBM * 128);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
          const int co = kg ? efr_c1 : efr_c0;
          uint4 xf[2], wf[5];
#pragma unroll
          for (int b = 0; b < 2; ++b) xf[b] = *(const uint4*)(xs + b * 2048 + co);
#pragma unroll
          for (int a = 0; a < 5; ++a)
            wf[a] = *(const uint4*)(RING + ((uoff + (unsigned)(pn * 80 + a * 16) * 128u) & (unsigned)(kRing - 1)) + efr_row + co);
#pragma unroll
          for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) mma_kgroup<bf16_t>(wf[a], xf[b], acc3[h][a][b]);
        }
        __syncthreads();
        uoff = (uoff + kPUnit) & (unsigned)(kRing - 1);
      }
    }
    TFSTAMP(9)
    // epilogue of proj_out: + bp + x2, row-major through the hidden-chunk / ring area (every unit has been consumed and
    // every wave is past its last fragment read: the barrier above)
    constexpr int SROW = 80 * 4 + 16, CPR = 10, NCH = 160, NI = 3;
    unsigned char* stg = HT + wave_id * (16 * SROW);
    uint4 rcur[NI], rnext[NI];
    auto load_res = [&](int blk, uint4 (&r)[NI]) __attribute__((always_inline)) {   // blk = h * 2 + b
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int cidx = elane + i * 64;
        const int row = cidx / CPR, cc = cidx - row * CPR;
        const int m = m0 + pm * 32 + (blk & 1) * 16 + row;
        if (cidx < NCH && m < p.M) r[i] = *(const uint4*)(p.x2 + (size_t)m * kC + (blk >> 1) * 160 + pn * 80 + cc * 8);
      }
    };
    load_res(0, rcur);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 b3[5];
#pragma unroll
      for (int a = 0; a < 5; ++a) b3[a] = *(const f32x4*)(p.bias3 + h * 160 + pn * 80 + a * 16 + elg * 4);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int blk = h * 2 + b;
#pragma unroll
        for (int a = 0; a < 5; ++a) *(f32x4*)(stg + elq * SROW + (a * 16 + elg * 4) * 4) = acc3[h][a][b] + b3[a];
        if (blk + 1 < 4) load_res(blk + 1, rnext);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int cidx = elane + i * 64;
          if (cidx < NCH) {
            const int row = cidx / CPR, cc = cidx - row * CPR;
            const int m = m0 + pm * 32 + b * 16 + row;
            if (m < p.M) {
              float v[8], r[8];
              const unsigned char* sp = stg + row * SROW + cc * 32;
              const f32x4 t0 = *(const f32x4*)sp, t1 = *(const f32x4*)(sp + 16);
              Chunk<bf16_t>::unpack(rcur[i], r);
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[e] = t0[e] + r[e]; v[4 + e] = t1[e] + r[4 + e]; }
              *(uint4*)(p.out + (size_t)m * kC + h * 160 + pn * 80 + cc * 8) = Chunk<bf16_t>::pack(v);
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NI; ++i) rcur[i] = rnext[i];
      }
    }
  }
  TFSTAMP(10)
}

// One 16-byte chunk of the stream per thread.  w1: GEGLU weights [8C][C] in the engine's packed row order (gamma folded in),
// w2: ff.net.2 [C][4C], wp: proj_out [C][C] (may be null: the stream then ends after the chunks).
__global__ __launch_bounds__(256) void tf_pack_stream_kernel(const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2,
                                                             const bf16_t* __restrict__ wp, uint4* __restrict__ out,
                                                             int nchunks, long long nvec) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= nvec) return;
  const long long byte = gid * 16;
  const long long chunk_end = (long long)nchunks * kChunkBytes;
  const bf16_t* src;
  if (byte < chunk_end) {
    const int c = (int)(byte / kChunkBytes);
    const int o = (int)(byte - (long long)c * kChunkBytes);
    if (o < kKT * kW1Tile) {
      const int kt = o / kW1Tile, rem = o - kt * kW1Tile;
      const int r = rem >> 7, pc = (rem & 127) >> 4, lc = pc ^ (r & 7);
      src = w1 + (size_t)(c * 128 + r) * kC + kt * 64 + lc * 8;
    } else {
      const int rem = o - kKT * kW1Tile;
      const int r = rem >> 7, pc = (rem & 127) >> 4, lc = pc ^ (r & 7);
      // stream row r -> output column n: unit A holds every wave's fragments 0..2 (48 rows per wave), unit B fragments 3, 4
      const int n = r < 192 ? (r / 48) * 80 + (r % 48) : ((r - 192) / 32) * 80 + 48 + ((r - 192) % 32);
      src = w2 + (size_t)n * (4 * kC) + c * kHC + lc * 8;
    }
  } else {
    const int o = (int)(byte - chunk_end);
    const int kt = o / (2 * kPUnit), rem = o - kt * (2 * kPUnit);
    const int h = rem / kPUnit, rem2 = rem - h * kPUnit;
    const int r = rem2 >> 7, pc = (rem2 & 127) >> 4, lc = pc ^ (r & 7);
    if (r >= kC / 2) { out[gid] = make_uint4(0u, 0u, 0u, 0u); return; }     // padding rows of the 24 KB unit
    src = wp + (size_t)(h * 160 + r) * kC + kt * 64 + lc * 8;
  }
  out[gid] = *(const uint4*)src;
}

int g_tfuse_dbg = 0;      // phase-ablation flags (LDMSEG_TFUSE_ABLATE builds) | bit 8: no start-chunk rotation
int g_tfuse_mode = 3;     // bit 0: fuse LayerNorm_3 -> GEGLU -> ff.net.2 (+h); bit 1: also proj_out (+x)

}  // namespace

bool mlp_fused_ok(int C, int dtype) { return (g_tfuse_mode & 1) && C == kC && dtype == DT_BF16; }
bool mlp_fused_proj() { return (g_tfuse_mode & 2) != 0; }
void mlp_fused_set_mode(int m) { g_tfuse_mode = m & 3; }
void mlp_fused_set_dbg(int f) { g_tfuse_dbg = f; }
int mlp_fused_get_mode() { return g_tfuse_mode; }
size_t mlp_fused_stream_bytes(int C) { return C == kC ? (size_t)(4 * kC / kHC) * kChunkBytes + (size_t)2 * kKT * kPUnit : 0; }

int launch_pack_mlp_stream(const void* w1, const void* w2, const void* wp, void* out, int C, hipStream_t s) {
  if (C != kC || !w1 || !w2 || !wp || !out) return -2;
  const long long nvec = (long long)mlp_fused_stream_bytes(C) / 16;
  hipLaunchKernelGGL(tf_pack_stream_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, s, (const bf16_t*)w1,
                     (const bf16_t*)w2, (const bf16_t*)wp, (uint4*)out, 4 * kC / kHC, nvec);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// h [M][320] bf16 in place (proj = 0: out = h + ff(LN(h)))  or  out = proj_out(h + ff(LN(h))) + x2 (proj = 1)
int launch_mlp_fused(const void* h, void* out, const void* x2, const void* stream, const float* bias1, const float* bias2,
                     const float* bias3, const void* zeros, int M, int C, float eps, int proj, int rows_per_image, hipStream_t s) {
  if (C != kC || M < 1 || !h || !out || !stream || !bias1 || !bias2 || !zeros || (proj && (!x2 || !bias3))) return -2;
  MlpFusedParams p;
  p.x = (const bf16_t*)h; p.out = (bf16_t*)out; p.x2 = (const bf16_t*)x2;
  p.stream = (const unsigned char*)stream; p.bias1 = bias1; p.bias2 = bias2; p.bias3 = bias3; p.zeros = zeros;
  p.M = M; p.nchunks = 4 * kC / kHC; p.eps = eps;
  p.dbg = g_tfuse_dbg & 0xff;
  p.ldr_delay = (g_tfuse_dbg >> 16) & 3;
  // start-chunk rotation by the tile's index inside its image (whole tiles per image only; otherwise by nothing)
  p.rot_tiles = (!(g_tfuse_dbg & 256) && rows_per_image > 0 && rows_per_image % kBM == 0) ? rows_per_image / kBM : 0;
  const dim3 grid((M + kBM - 1) / kBM), block(768);
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute((const void*)mlp_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    (void)hipFuncSetAttribute((const void*)mlp_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    attr_set[dev] = true;
  }
  igemm_log_note(proj ? "mlp_fused<bf16,proj=1>" : "mlp_fused<bf16,proj=0>");
  if (proj) hipLaunchKernelGGL(mlp_fused_kernel<true>, grid, block, kLds, s, p);
  else hipLaunchKernelGGL(mlp_fused_kernel<false>, grid, block, kLds, s, p);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace ldmseg

#ifdef LDMSEG_TFUSE_ABLATE
extern "C" int ldmseg_debug_tf_stamps(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(ldmseg::g_tf_ts), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
