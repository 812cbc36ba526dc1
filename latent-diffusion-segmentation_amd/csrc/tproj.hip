// Row-local fusion of a transformer's entry at the 320-channel level (64x64 maps at 512x512, 128x128 at 1024x1024):
//
//     h       <- proj_in(x_gn)                          1x1 conv, + bias        (diffusers Transformer2DModel,
//     q|k|v   <- [to_q | to_k | to_v]( LayerNorm_1(h) )  no bias                  /root/reference/ldmseg/models/unet.py:361-373, 401-425)
//
// in ONE launch instead of three (proj_in GEMM, row statistics, folded-LayerNorm q|k|v GEMM): h was written, then read twice.
// All of these launches are memory-side bound (42 MB / 105 MB of HBM traffic behind 6.7 / 20 GFLOP): the fused kernel reads
// the GroupNorm-ed tile once and writes h and q|k|v once.
//
// Same machinery as tfuse.hip: a workgroup owns 128 whole rows; the tile (128 x 320 bf16 = 80 KB) is brought into the LDS once
// and is the X operand of FOUR [128 x 320] x [320 x 320]^T passes - proj_in, then (after the pass-0 result has replaced the
// tile and has been LayerNorm-ed in place; gamma / beta live in the q|k|v weights and bias) to_q, to_k, to_v.  Only weights
// stream: the four matrices are packed at create time into one byte stream of 40 KB units ([320 rows][128 B of one K tile], LDS
// swizzle applied) that four loader waves copy through a two-unit ring, 8 compute waves (2 x 4: 64-row x 80-column wave tiles)
// consume one unit per barrier step.  (A first version with 24 KB units of one N half - 32 x 80 wave tiles, 40 steps - ran at
// 50 us for M = 32768: LDS fragment reads per MFMA and the ~0.4 us a barrier step costs whatever is in it.)
// Stores go straight from the accumulator layout: two v_permlane16_swap per pair of 16-column fragments give every lane 8
// consecutive bf16 of one row (16-byte stores, 64-byte runs) - no LDS staging, the ring keeps streaming through the epilogues.
#include <hip/hip_runtime.h>

#include "common.h"
#include "kernels.h"

namespace ldmseg {
namespace {

constexpr int kBM = 128, kC = 320, kKT = kC / 64;
constexpr int kXT = kKT * kBM * 128;      // 81920: the row tile, [K tile][row][128 B], 16-B chunk index XOR (row & 7)
constexpr int kRing = 2 * 40960;          // two units: everything the CU has left beside the tile
constexpr int kLds = kXT + kRing;         // 163840
__device__ __forceinline__ unsigned ring_wrap(unsigned u) { return u >= (unsigned)kRing ? u - (unsigned)kRing : u; }
constexpr int kUnit = kC * 128;            // 40960: one K tile of a matrix, [320 rows][128 B]
constexpr int kNPass = 4;                 // proj_in, to_q, to_k, to_v
constexpr int kUnitsPerPass = kKT;        // 5
constexpr int kStreamBytes = kNPass * kUnitsPerPass * kUnit;   // 819200

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off"
      :
      : "v"(gsrc), "s"(lds_dst)
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
__device__ __forceinline__ void wait_pieces(int n) {
  switch (n) {
#define TP_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    TP_W(0) TP_W(1) TP_W(2) TP_W(3) TP_W(4) TP_W(5) TP_W(6) TP_W(7) TP_W(8) TP_W(9) TP_W(10) TP_W(11) TP_W(12) TP_W(13)
    TP_W(14) TP_W(15)
#undef TP_W
    default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
  }
}

#ifdef LDMSEG_TPROJ_ABLATE
#define TPDBG(bit) ((LDMSEG_TPROJ_ABLATE) & (bit))     // compile-time phase ablation (results wrong): 1 no weight DMA, 2 no global stores, 4 no MFMA, 8 no fragment reads
#else
#define TPDBG(bit) 0
#endif

// workgroup barrier that waits for this wave's LDS traffic only (what __syncthreads() compiles to on gfx950 outside tgsplit mode;
// spelled out because the compute waves must never wait for vmcnt here: their epilogue stores are still in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct ProjQkvParams {
  const bf16_t* x;         // [M][320]: GroupNorm(transformer input)
  bf16_t* h;               // [M][320]: proj_in(x) + bias (the residual stream of the block)
  bf16_t* qkv;             // [M][960]
  const unsigned char* stream;
  const float* bias;       // [4][320]: proj_in bias | W_q beta | W_k beta | W_v beta
  const void* zeros;
  int M;
  float eps;
  int rot, stagger;
  // GroupNorm fold (round 5): x is the RAW transformer input; gn_partial = launch_groupnorm_stats' {mean, M2} per (image, pixel chunk,
  // group), combined per wave and applied to the tile in the LDS before pass 0.  Null: x is already normalised.
  const float* gn_partial;
  const float* gn_gamma;
  const float* gn_beta;
  int gn_nchunk, gn_per, gn_hw;
  float gn_eps;
};
constexpr int kCPG = kC / 32;             // channels per GroupNorm group
constexpr int kGnMaxChunks = 64;          // pixel chunks per image the in-kernel combination takes (32 records per lane, held in registers)

__global__ __launch_bounds__(768, 3) void proj_ln_qkv_kernel(const ProjQkvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * kBM;
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
  constexpr int nunits = kNPass * kUnitsPerPass;
  // All workgroups of the single wave run in lockstep: every epilogue is a chip-wide store burst that drains at HBM write speed with
  // the matrix pipes idle, then the K loops run with the memory system idle (ablation: the stores cost 21 of 45 us).  Every second
  // workgroup of an XCD therefore starts half a pass late, so that one half's stores meet the other half's K loop.
  if ((p.rot & 2) && ((blockIdx.x >> 3) & 1)) {
    for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }

  if (wave_id >= 8) {
    // ================= loader waves: the row tile, then the weight stream through the ring (tfuse.hip's protocol) =================
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
    __builtin_amdgcn_s_setprio(3);
    int left = nunits * 10;                                           // pieces this wave still has to issue (10 per unit)
    int iss = 0;
    const unsigned voff = (unsigned)lane * 16u;
    // wave lw owns one 2 KiB block of every 8 KiB round; which one rotates with the workgroup's index inside its XCD, so that the 32
    // workgroups that stream the same bytes through one L2 do not all ask for the same lines at the same moment
    const int blk = (lw + ((p.rot & 1) ? (int)(blockIdx.x >> 3) : 0)) & 3;
    const unsigned char* src = p.stream + (size_t)blk * 2048;
    unsigned dst = (unsigned)blk * 2048u;
    auto issue_n = [&](int n) __attribute__((always_inline)) {
      for (int i = 0; i < n && left > 0; i += 2) {
        const unsigned long long su = (unsigned long long)(uintptr_t)src;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)su), hi = __builtin_amdgcn_readfirstlane((unsigned)(su >> 32));
        if (!TPDBG(1)) glds16x2_sbase(voff, (const void*)(uintptr_t)(((unsigned long long)hi << 32) | lo), __builtin_amdgcn_readfirstlane(lds0 + (unsigned)kXT + dst));
        src += 8192;
        dst = ring_wrap(dst + 8192u);
        iss += 2;
        left -= 2;
      }
    };
    issue_n(20);
    int land = 10;                          // this wave's ring pieces through the unit that must have landed next
    wait_pieces(iss - land);                // the row tile (issued first) and unit 0 have landed
    __syncthreads();                        // A
    if (p.gn_partial) asm volatile("s_barrier" ::: "memory");        // A2: pairs with the barrier behind the GroupNorm sweep
    for (int t = 0; t < nunits; ++t) {
      if (t == kUnitsPerPass) {             // pairs with the two barriers around the in-place LayerNorm after pass 0
        asm volatile("s_barrier" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
      }
      if (t + 1 < nunits) {
        land += 10;
        wait_pieces(iss - land);
      }
      asm volatile("s_barrier" ::: "memory");           // end of step t: unit t+1 has landed, unit t is free
      issue_n(10);
    }
    return;
  }

  // ================= compute waves =================
  const int lg = lane >> 4, lq = lane & 15;
  unsigned char* XT = smem;
  unsigned char* RING = smem + kXT;
  const int fr_row = lq * 128;
  const int fr_c0 = ((lg) ^ (lane & 7)) * 16;
  const int fr_c1 = ((lg + 4) ^ (lane & 7)) * 16;
  const int wm = wave_id >> 2, wn = wave_id & 3;
  const unsigned char* prow = XT + (wm * 64) * 128 + fr_row;
  unsigned uoff = 0;

  // the bias of a pass is what its accumulators start from; it is requested one epilogue ahead (pass 0: behind the tile's DMA)
  f32x4 bnext[5];
#pragma unroll
  for (int a = 0; a < 5; ++a) bnext[a] = *(const f32x4*)(p.bias + wn * 80 + a * 16 + lg * 4);
  // ---- GroupNorm fold, part 1 (in the shadow of the tile's DMA): every wave combines the image's per-chunk {mean, M2} itself - lane l
  // takes every second chunk of group l >> 1, the pair is summed in a fixed order (closed form of Chan's update against the first
  // chunk's mean, as norm.hip's gn_reduce_stats) - and every thread turns the statistics of ITS 8 channels (one 16-byte chunk of
  // the tile, at most two groups) into y = x * ga + gb.  There is no LDS left for a table: the exchange is ds_bpermute.
  float ga[8], gb[8];
  const int gn_j = tid % 40, gn_rg = tid / 40;          // chunk of 8 channels, row group (12 of them; threads 480.. idle)
  if (p.gn_partial) {
    const int img = m0 / p.gn_hw;                       // (a tile never straddles images: HW % 128 == 0, launch_proj_qkv_fused checks)
    const int g = lane >> 1, half = lane & 1;
    const float2* src = (const float2*)p.gn_partial + (size_t)img * p.gn_nchunk * 32 + g;
    // all of the lane's chunk records are requested at once (one round trip to the L2; a rolled loop of dependent trips cost more than
    // the GroupNorm launch this replaces)
    float2 e[kGnMaxChunks / 2];
    float cnt[kGnMaxChunks / 2];
#pragma unroll
    for (int i = 0; i < kGnMaxChunks / 2; ++i) {
      const int ch = half + 2 * i;
      const int p0 = ch * p.gn_per, np = min(p.gn_hw, p0 + p.gn_per) - p0;
      const bool ok = ch < p.gn_nchunk && np > 0;
      e[i] = src[(size_t)(ok ? ch : 0) * 32];
      cnt[i] = ok ? (float)(np * kCPG) : 0.f;
    }
    const float K = __shfl(e[0].x, lane & ~1);          // chunk 0 of the group (always populated)
    float n = 0.f, sd = 0.f;
#pragma unroll
    for (int i = 0; i < kGnMaxChunks / 2; ++i) { n += cnt[i]; sd += cnt[i] * (e[i].x - K); }
    auto pair_sum = [&](float v) __attribute__((always_inline)) {
      const float w = __shfl_xor(v, 1);
      return half ? w + v : v + w;                      // same operand order on both partners
    };
    n = pair_sum(n);
    sd = pair_sum(sd);
    const float dbar = n > 0.f ? sd / n : 0.f;
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < kGnMaxChunks / 2; ++i) {
      const float d = (e[i].x - K) - dbar;
      m2 += cnt[i] > 0.f ? e[i].y + cnt[i] * d * d : 0.f;
    }
    m2 = pair_sum(m2);
    const float g_mean = K + dbar;
    const float g_rstd = 1.0f / sqrtf((n > 0.f ? m2 / n : 0.f) + p.gn_eps);
    const int c0 = gn_j * 8;
    const int g0 = c0 / kCPG, g1 = min(g0 + 1, 31);
    const int split = min(8, (g0 + 1) * kCPG - c0);     // channels [0, split) of the chunk belong to g0, the rest to g0 + 1
    const float m_lo = __shfl(g_mean, 2 * g0), r_lo = __shfl(g_rstd, 2 * g0);
    const float m_hi = __shfl(g_mean, 2 * g1), r_hi = __shfl(g_rstd, 2 * g1);
    const f32x4 gm0 = *(const f32x4*)(p.gn_gamma + c0), gm1 = *(const f32x4*)(p.gn_gamma + c0 + 4);
    const f32x4 bt0 = *(const f32x4*)(p.gn_beta + c0), bt1 = *(const f32x4*)(p.gn_beta + c0 + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float gam = e < 4 ? gm0[e & 3] : gm1[e & 3], bet = e < 4 ? bt0[e & 3] : bt1[e & 3];
      ga[e] = (e < split ? r_lo : r_hi) * gam;
      gb[e] = bet - (e < split ? m_lo : m_hi) * ga[e];
    }
  }
  lds_barrier();                          // A: the row tile and unit 0 have landed
  if (p.gn_partial) {
    // ---- part 2: the affine on the tile in place (the arithmetic of gn_apply_kernel: one fused multiply-add per element, bf16 result)
    if (gn_rg < 12) {
      unsigned char* colp = XT + (gn_j >> 3) * (kBM * 128);
#pragma unroll 2
      for (int row = gn_rg; row < kBM; row += 12) {
        unsigned char* q = colp + row * 128 + (((gn_j & 7) ^ (row & 7)) << 4);
        float f[8];
        Chunk<bf16_t>::unpack(*(const uint4*)q, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = f[e] * ga[e] + gb[e];
        *(uint4*)q = Chunk<bf16_t>::pack(f);
      }
    }
    lds_barrier();                        // A2: the normalised tile is visible
  }

#pragma unroll 1
  for (int pass = 0; pass < kNPass; ++pass) {
    f32x4 acc[5][4];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = bnext[a];
#pragma unroll 1
    for (int kt = 0; kt < kKT; ++kt) {
      const unsigned char* xs = prow + kt * (kBM * 128);
      const unsigned char* ws = RING + uoff + (wn * 80) * 128 + fr_row;
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        const int co = kg ? fr_c1 : fr_c0;
        uint4 xf[4], wf[5];
        if (TPDBG(8)) {
#pragma unroll
          for (int b = 0; b < 4; ++b) xf[b] = make_uint4(lane, kt, kg, b);
#pragma unroll
          for (int a = 0; a < 5; ++a) wf[a] = make_uint4(lane, kt, kg, a);
        } else {
#pragma unroll
          for (int b = 0; b < 4; ++b) xf[b] = *(const uint4*)(xs + b * 2048 + co);
#pragma unroll
          for (int a = 0; a < 5; ++a) wf[a] = *(const uint4*)(ws + a * 2048 + co);
        }
        if (TPDBG(4)) {
#pragma unroll
          for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b][0] += bits_f32(wf[a].x ^ xf[b].y);
        } else {
#pragma unroll
          for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) mma_kgroup<bf16_t>(wf[a], xf[b], acc[a][b]);
        }
      }
      lds_barrier();                      // end of the unit's step (the loader refills its slot)
      uoff = uoff ? 0u : (unsigned)kUnit;
    }
    // ---- epilogue of the pass: + bias, bf16, 16-byte stores from the accumulator layout; pass 0 also replaces the tile.
    // Lane (lq, lg) holds columns f*16 + lg*4 .. +3 of row lq of fragment f.  Swapping (frag a, frag a+1) dwords between the
    // lane rows gives row-0 / row-2 lanes columns [0..7] / [8..15] of fragment a and row-1 / row-3 lanes the same of fragment a+1.
    bf16_t* obase = pass == 0 ? p.h : p.qkv + (pass - 1) * kC;
    const int ldo = pass == 0 ? kC : 3 * kC;
    // (`zero` is opaque to the optimiser and defined here: every address below depends on it, so none of them is computed outside
    // the pass loop, kept alive across the K loop and spilled - a scratch reload in this block waits for vmcnt(0), i.e. for every
    // store issued before it: the first version paid ~5 us per pass for that)
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    const int elq = lq + zero, elg = lg + zero;
    const int n0 = wn * 80;
    if (pass + 1 < kNPass) {
#pragma unroll
      for (int a = 0; a < 5; ++a) bnext[a] = *(const f32x4*)(p.bias + (pass + 1) * kC + n0 + a * 16 + elg * 4);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int row = wm * 64 + b * 16 + elq;
      uint2 pk[5];
#pragma unroll
      for (int a = 0; a < 5; ++a) {
        const f32x4 v = acc[a][b];
        pk[a] = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      }
      if (pass == 0) {
        // (every wave is past its last fragment read of the old tile: the barrier that ended the last step)
#pragma unroll
        for (int a = 0; a < 5; ++a) {
          const int n = n0 + a * 16 + elg * 4;
          const int kt = n >> 6, ch = (n & 63) >> 3, half = (n >> 2) & 1;
          *(uint2*)(XT + kt * (kBM * 128) + row * 128 + ((ch ^ (row & 7)) << 4) + half * 8) = pk[a];
        }
      }
      bf16_t* orow = obase + (size_t)(m0 + row) * ldo + n0;
#pragma unroll
      for (int a = 0; a < 4; a += 2) {
        const auto sx = __builtin_amdgcn_permlane16_swap(pk[a].x, pk[a + 1].x, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(pk[a].y, pk[a + 1].y, false, false);
        if (!TPDBG(2)) *(uint4*)(orow + (a + (elg & 1)) * 16 + (elg >> 1) * 8) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
      }
      if (!TPDBG(2)) *(uint2*)(orow + 4 * 16 + elg * 4) = pk[4];
    }
    if (pass == 0) {
      lds_barrier();                      // X1: the new tile (h) is complete
      // ---- LayerNorm_1 of the tile's rows in place (no affine: gamma / beta live in the q|k|v weights and bias): 4 lanes per
      // row, 10 x 16 B each; two-pass mean / centred variance like launch_rowstats (tfuse.hip)
      const int r = wave_id * 16 + (lane >> 2), q = lane & 3;
      uint4 raw[10];
      unsigned char* rowp = XT + r * 128;
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const int j = q + 4 * i;
        raw[i] = *(const uint4*)(rowp + (j >> 3) * (kBM * 128) + (((j & 7) ^ (r & 7)) << 4));
      }
      // (unpacked on the fly in each of the three sweeps: holding the 80 floats next to the 40 packed registers spilled, and a
      // scratch reload here waits - vmcnt is in order - for every store of pass 0 issued before it)
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        float f[8];
        Chunk<bf16_t>::unpack(raw[i], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) sm += f[e];
      }
      sm += dpp_f<0xB1>(sm);
      sm += dpp_f<0x4E>(sm);
      const float mean = sm * (1.0f / kC);
      float qq = 0.f;
#pragma unroll
      for (int i = 0; i < 10; ++i) asm volatile("" : "+v"(raw[i].x), "+v"(raw[i].y), "+v"(raw[i].z), "+v"(raw[i].w));   // (no CSE of the unpacked values across sweeps)
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        float f[8];
        Chunk<bf16_t>::unpack(raw[i], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; qq += d * d; }
      }
      qq += dpp_f<0xB1>(qq);
      qq += dpp_f<0x4E>(qq);
      const float rstd = 1.0f / sqrtf(qq * (1.0f / kC) + p.eps);
#pragma unroll
      for (int i = 0; i < 10; ++i) asm volatile("" : "+v"(raw[i].x), "+v"(raw[i].y), "+v"(raw[i].z), "+v"(raw[i].w));
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const int j = q + 4 * i;
        float f[8], y[8];
        Chunk<bf16_t>::unpack(raw[i], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = (f[e] - mean) * rstd;
        *(uint4*)(rowp + (j >> 3) * (kBM * 128) + (((j & 7) ^ (r & 7)) << 4)) = Chunk<bf16_t>::pack(y);
      }
      lds_barrier();                      // X2: normalised tile visible
    }
  }
}

// One 16-byte chunk of the stream per thread.  wp: proj_in [C][C]; wqkv: [3C][C] (gamma folded in), both [n][k] bf16.
__global__ __launch_bounds__(256) void tp_pack_stream_kernel(const bf16_t* __restrict__ wp, const bf16_t* __restrict__ wqkv, uint4* __restrict__ out,
                                                             long long nvec) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= nvec) return;
  const int byte = (int)(gid * 16);
  const int pass = byte / (kUnitsPerPass * kUnit), o = byte - pass * (kUnitsPerPass * kUnit);
  const int kt = o / kUnit, rem = o - kt * kUnit;
  const int r = rem >> 7, pc = (rem & 127) >> 4, lc = pc ^ (r & 7);
  const bf16_t* w = pass == 0 ? wp : wqkv + (size_t)(pass - 1) * kC * kC;
  out[gid] = *(const uint4*)(w + (size_t)r * kC + kt * 64 + lc * 8);
}

int g_tproj_mode = 3;     // bit 0: on; bit 1: loader block rotation

}  // namespace

bool proj_qkv_fused_ok(int C, int M, int dtype) { return (g_tproj_mode & 1) && C == kC && dtype == DT_BF16 && M > 0 && M % kBM == 0; }
// the GroupNorm in front of the transformer as a statistics pass + a sweep inside the kernel: whole tiles of one image (32 groups of 10)
bool proj_qkv_gn_fold_ok(int HW) { return HW > 0 && HW % kBM == 0; }
int proj_qkv_gn_chunks(int B, int HW) {    // pixel chunks per image for launch_groupnorm_stats: <= 64, about two workgroups per CU in total
  int n = 512 / (B > 0 ? B : 1);
  if (n > kGnMaxChunks) n = kGnMaxChunks;
  if (n < 4) n = 4;
  while (n > 1 && HW / n < 32) n >>= 1;
  return n;
}
void proj_qkv_set_mode(int m) { g_tproj_mode = m & 0xff; }
int proj_qkv_get_mode() { return g_tproj_mode; }
size_t proj_qkv_stream_bytes(int C) { return C == kC ? (size_t)kStreamBytes : 0; }

int launch_pack_proj_qkv_stream(const void* wp, const void* wqkv, void* out, int C, hipStream_t s) {
  if (C != kC || !wp || !wqkv || !out) return -2;
  const long long nvec = kStreamBytes / 16;
  hipLaunchKernelGGL(tp_pack_stream_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, s, (const bf16_t*)wp, (const bf16_t*)wqkv, (uint4*)out, nvec);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// x [M][320] bf16 -> h [M][320], qkv [M][960].  bias4: [4][320] fp32 (proj_in bias | W_q beta | W_k beta | W_v beta)
int launch_proj_qkv_fused(const void* x, void* h, void* qkv, const void* stream, const float* bias4, const void* zeros, int M, int C, float eps,
                          const GnFold* gn, hipStream_t s) {
  if (C != kC || M < 1 || M % kBM || !x || !h || !qkv || !stream || !bias4 || !zeros) return -2;
  ProjQkvParams p;
  p.x = (const bf16_t*)x; p.h = (bf16_t*)h; p.qkv = (bf16_t*)qkv; p.stream = (const unsigned char*)stream; p.bias = bias4; p.zeros = zeros;
  p.M = M; p.eps = eps; p.rot = (g_tproj_mode >> 1) & 3; p.stagger = (g_tproj_mode >> 4) & 15;
  p.gn_partial = nullptr; p.gn_gamma = p.gn_beta = nullptr; p.gn_nchunk = p.gn_per = p.gn_hw = 0; p.gn_eps = 0.f;
  if (gn) {
    if (!gn->partial || !gn->gamma || !gn->beta || !proj_qkv_gn_fold_ok(gn->HW) || M % gn->HW || gn->nchunk < 1 || gn->nchunk > kGnMaxChunks) return -2;
    p.gn_partial = gn->partial; p.gn_gamma = gn->gamma; p.gn_beta = gn->beta;
    p.gn_nchunk = gn->nchunk; p.gn_per = (gn->HW + gn->nchunk - 1) / gn->nchunk; p.gn_hw = gn->HW; p.gn_eps = gn->eps;
  }
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute((const void*)proj_ln_qkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    attr_set[dev] = true;
  }
  igemm_log_note(gn ? "proj_ln_qkv<bf16,gn>" : "proj_ln_qkv<bf16>");
  hipLaunchKernelGGL(proj_ln_qkv_kernel, dim3(M / kBM), dim3(768), kLds, s, p);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace ldmseg
