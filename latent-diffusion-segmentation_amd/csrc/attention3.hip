// Self-attention for the bf16 perf mode, head dims 40 and 80 (the 64x64 / 32x32 levels of the UNet: N = 4096 / 1024
// tokens at 512x512, 16384 / 4096 at 1024x1024) - the third generation of the kernel in attention.hip, restructured
// after measuring what actually bounds it on gfx950 (tools/ubench/valu_trans.hip): not the matrix core and not
// v_exp_f32 as such (a transcendental costs ~1.7 plain VALU slots), but VALU *issue* - one wave issues a VALU
// instruction every ~5 cycles, so at 2 waves per SIMD a softmax of ~4.2 VALU slots per score runs at half the MFMA
// rate.  Hence:
//   * K and V tiles arrive by LDS-DMA (global_load_lds_dwordx4) as "chunk planes": plane c holds the 16-byte chunk c
//     (8 head-dim columns) of all 64 keys of the tile, 1 KiB = one wave instruction, lane = key.  No staging
//     registers, no ds_write pass, no transpose: V^T fragments are read with ds_read_b64_tr_b16 (hardware transpose),
//     K fragments with ds_read_b128; both are bank-conflict free by construction (K planes 1024 B apart, V planes
//     1152 B apart).  The kernel fits 128 VGPRs: 4 waves per SIMD instead of 2.
//   * The running row maximum is folded into the matrix product: head dim D < 32*KG leaves spare contraction columns,
//     column D of K is a constant plane of ones and column D of Q holds -m (bf16-exact by construction), so the scores
//     leave the MFMA already relative to m and the per-score subtract disappears.  m is only moved when a row's tile
//     maximum exceeds it by more than THR (= 2^6 headroom for p, harmless with fp32 accumulation); everything that
//     was accumulated against the old m is rescaled exactly once at that point.
//   * Row maxima with v_max3_f32 (two scores per instruction); row sums come out of the PV product through a ones
//     plane of V (column D of O^T), as in attention.hip.
// Per score that leaves: 1/2 max3 + 1 exp + 1/2 cvt_pk.  Scores are computed transposed (A = K, B = Q) so a lane owns
// one query row; P never leaves registers.  Same arithmetic contract as attention.hip (bf16 operands, fp32
// accumulation, softmax(q k^T d^-1/2) v per head); tests/test_ops_gpu.py::test_attention* cover both.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace ldmseg {
namespace {

constexpr int BKV3 = 64;
typedef short s16x4 __attribute__((ext_vector_type(4)));

// max of three, written for the compiler (it forms v_max3_f32).  NOT inline asm: the operands are MFMA results, an MFMA
// write needs wait states before a VALU reads it, and hipcc's hazard recognizer pads those only for instructions it can
// see - an asm v_max3_f32 here read half-written accumulators now and then (wrong maxima -> overflow -> NaN, run-to-run
// different).  This file is built with -fno-honor-nans so no canonicalising v_max(x, x) is emitted in front.
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
// smallest bf16-representable value >= x (finite x), returned as f32
__device__ __forceinline__ float bf16_ceil(float x) {
  const uint32_t u = f32_bits(x);
  return bits_f32(((u & 0x80000000u) ? u : u + 0xffffu) & 0xffff0000u);
}
// LDS-DMA of 64 x 16 B: lane l fetches gsrc(l) + OFF into LDS lds_dst + 16 l.  The instruction's immediate offset is
// applied to the global AND the LDS address, so M0 carries lds_dst - OFF (callers keep lds_dst >= OFF).
template <int OFF>
__device__ __forceinline__ void glds16_off(const void* gsrc, unsigned lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off offset:%2"
      :
      : "v"(gsrc), "s"(lds_dst - OFF), "i"(OFF)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

template <int D, int NWV = 4> struct A3Cfg {
  static constexpr int DCH = D / 8;                  // data planes per tile and operand (16-byte chunks per row)
  static constexpr int KG = (D + 1 + 31) / 32;       // 32-wide k-groups of the QK^T contraction (incl. the -m column)
  static constexpr int DF = (D + 1 + 15) / 16;       // 16-row fragments of O^T (incl. the row-sum row D)
  static constexpr int NPL = DCH + 1;                // planes incl. the constant ones plane
  static constexpr int KPS = 1024, VPS = 1152;       // plane strides (bytes)
  static constexpr int KBYTES = NPL * KPS, VBYTES = NPL * VPS;
  static constexpr int STAGE = KBYTES + VBYTES;
  static constexpr int VSH = (NWV - DCH % NWV) % NWV; // V planes start at wave VSH's successor... balances the DMA instructions over the waves
  static constexpr int JMAX = (DCH + NWV - 1) / NWV;
  static_assert(D % 8 == 0 && D % 32 != 0, "needs a spare contraction column");
};

// NST LDS stages; the DMA stream runs NST-1 key tiles ahead of the compute stream
template <int D, int QF, int WPS, int NST, bool PIPE, int LAZY, int NWV>
__global__ __launch_bounds__(64 * NWV, WPS) void attn3_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int N, int C,
                                                         int heads, float scale_log2e) {
  using Cfg = A3Cfg<D, NWV>;
  static_assert(!PIPE, "the software-pipelined loop (QK^T of tile t+1 issued before the softmax of tile t; 248 VGPRs, measured 11 % slower) was removed in round 3");
  constexpr float THR = 6.0f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 15, lg = lane >> 4;

  // XCD-aware: all query blocks of one (image, head) share an XCD's L2
  const int nqb = (N + 16 * NWV * QF - 1) / (16 * NWV * QF);
  int wg;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int qb = wg % nqb;
  const int bh = wg / nqb;
  const int h = bh % heads, b = bh / heads;
  const size_t ld = (size_t)3 * C;
  const bf16_t* qbase = qkv + (size_t)b * N * ld + (size_t)h * D;
  const unsigned char* kbase = (const unsigned char*)(qbase + C);
  const size_t ldb = ld * sizeof(bf16_t);
  const int voff_bytes = C * (int)sizeof(bf16_t);          // V sits C elements after K in a token's row

  // ---- constant planes: chunk DCH of every key is [1, 0, 0, 0, 0, 0, 0, 0] in all stages, for K (the -m column) and V
  // (the row-sum row); the DMA never touches them
  for (int i = tid; i < NST * 2 * BKV3; i += 64 * NWV) {
    const int st = i / (2 * BKV3), rem = i - st * (2 * BKV3), which = rem / BKV3, key = rem - which * BKV3;
    unsigned char* p = smem + st * Cfg::STAGE + (which ? Cfg::KBYTES + Cfg::DCH * Cfg::VPS : Cfg::DCH * Cfg::KPS) + key * 16;
    *(uint4*)p = make_uint4(0x3f80u, 0u, 0u, 0u);
  }

  // ---- DMA stream: K plane c = wave + NWV j, V plane c = ((wave + VSH) % NWV) + NWV j; lane = key of the tile
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
  const int wv = (wave + Cfg::VSH) % NWV;
  int my_cnt = 0;                                           // DMA instructions this wave issues per tile
#pragma unroll
  for (int j = 0; j < Cfg::JMAX; ++j) my_cnt += (wave + NWV * j < Cfg::DCH) + (wv + NWV * j < Cfg::DCH);
  my_cnt = __builtin_amdgcn_readfirstlane(my_cnt);
  auto issue_tile = [&](int t, int stage) __attribute__((always_inline)) {
    const int row = min(t * BKV3 + lane, N - 1);           // keys past N are clamped (finite data) and masked below
    const unsigned char* rp = kbase + (size_t)row * ldb;
    const unsigned char* kp = rp + wave * 16;
    const unsigned char* vp = rp + voff_bytes + wv * 16;
    const unsigned kdst = __builtin_amdgcn_readfirstlane(lds0 + stage * Cfg::STAGE + wave * Cfg::KPS);
    const unsigned vdst = __builtin_amdgcn_readfirstlane(lds0 + stage * Cfg::STAGE + Cfg::KBYTES + wv * Cfg::VPS);
    static_for_n<Cfg::JMAX>([&](auto jc) __attribute__((always_inline)) {
      constexpr int J = decltype(jc)::value;
      if (wave + NWV * J < Cfg::DCH) glds16_off<J * NWV * 16>(kp, kdst + J * NWV * Cfg::KPS);
      if (wv + NWV * J < Cfg::DCH) glds16_off<J * NWV * 16>(vp, vdst + J * NWV * Cfg::VPS);
    });
  };
  // wait until at most `ahead` tiles of this wave's DMA are still in flight (my_cnt in {2,3,5}: wave-uniform)
  auto wait_tiles_ahead = [&](int ahead) __attribute__((always_inline)) {
    if (ahead == 0) { wait_vm<0>(); return; }
    if (my_cnt == 0) return;
    if (my_cnt == 1) wait_vm<1>();
    else if (my_cnt == 2) wait_vm<2>();
    else if (my_cnt == 3) wait_vm<3>();
    else if (my_cnt == 4) wait_vm<4>();
    else if (my_cnt == 5) wait_vm<5>();
    else wait_vm<6>();
  };

  const int ntiles = (N + BKV3 - 1) / BKV3;
  const int nfull = N / BKV3;                // tiles without masked keys
  static_for_n<NST - 1>([&](auto jc) __attribute__((always_inline)) {
    constexpr int J = decltype(jc)::value;
    if (J < ntiles) issue_tile(J, J);
  });

  // ---- Q fragments (MFMA B operand): lane (q = lq, g = lg) holds chunk kg*4+g of its row, pre-scaled by d^-1/2 log2 e
  const int q0 = qb * 16 * NWV * QF + wave * 16 * QF;
  uint4 qf[QF][Cfg::KG];
#pragma unroll
  for (int a = 0; a < QF; ++a) {
    const int q = q0 + a * 16 + lq;
#pragma unroll
    for (int kg = 0; kg < Cfg::KG; ++kg) {
      const int ch = kg * 4 + lg;
      uint4 raw = (q < N && ch < Cfg::DCH) ? *(const uint4*)((const unsigned char*)(qbase + (size_t)q * ld) + ch * 16)
                                           : make_uint4(0, 0, 0, 0);
      float qv[8];
      Chunk<bf16_t>::unpack(raw, qv);
#pragma unroll
      for (int e = 0; e < 8; ++e) qv[e] *= scale_log2e;
      qf[a][kg] = Chunk<bf16_t>::pack(qv);
    }
  }
  constexpr int MKG = Cfg::DCH / 4, MLG = Cfg::DCH % 4;      // the (k-group, lane group) that holds column D of Q
  const bool holds_m = (lg == MLG);

  f32x4 o[QF][Cfg::DF];
  float mrow[QF];
#pragma unroll
  for (int a = 0; a < QF; ++a) {
    mrow[a] = 0.f;
#pragma unroll
    for (int d = 0; d < Cfg::DF; ++d) o[a][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // per-lane LDS offsets.  K fragment f, k-group kg: plane min(kg*4+lg, DCH), row 16f + lq  (planes past DCH would be
  // zero columns: the ones plane is read instead and meets zero columns of Q)
  int koff[Cfg::KG];
#pragma unroll
  for (int kg = 0; kg < Cfg::KG; ++kg) koff[kg] = min(kg * 4 + lg, Cfg::DCH) * Cfg::KPS + lq * 16;
  // V^T fragment df via ds_read_b64_tr_b16: lane i of a 16-lane group points at key row (i >> 2), 8 bytes (i & 1) of
  // chunk plane 2df + ((i & 3) >> 1); group g covers keys 4g.. of each 16-key block
  int voff[Cfg::DF];
#pragma unroll
  for (int d = 0; d < Cfg::DF; ++d)
    voff[d] = Cfg::KBYTES + min(2 * d + ((lq & 3) >> 1), Cfg::DCH) * Cfg::VPS + (4 * lg + (lq >> 2)) * 16 + (lq & 1) * 8;

  // ---- the three pieces of a key tile, as macros over a named score array (lambdas taking the arrays by reference made
  // hipcc index them dynamically -> scratch) ----
  // S^T = K Q^T for the K planes of LDS stage ST (already relative to the folded row maxima)
#define A3_QK(SARR, ST)                                                                                         \
  {                                                                                                             \
    const unsigned char* kb_ = smem + (ST) * Cfg::STAGE;                                                        \
    _Pragma("unroll") for (int a = 0; a < QF; ++a)                                                              \
      _Pragma("unroll") for (int f = 0; f < 4; ++f) SARR[a][f] = f32x4{0.f, 0.f, 0.f, 0.f};                      \
    _Pragma("unroll") for (int kg = 0; kg < Cfg::KG; ++kg) {                                                    \
      _Pragma("unroll") for (int f = 0; f < 4; ++f) {                                                           \
        const uint4 kf = *(const uint4*)(kb_ + koff[kg] + f * 256);                                             \
        _Pragma("unroll") for (int a = 0; a < QF; ++a) mma_kgroup<bf16_t>(kf, qf[a][kg], SARR[a][f]);           \
      }                                                                                                         \
    }                                                                                                           \
  }
  // softmax numerators of tile T_ in place (SARR <- p), moving the folded maxima when a row outgrew them.  SNXT: scores of
  // the NEXT tile that were already computed against the old maxima (pipelined loop) and must move with them, or SARR.
#define A3_SOFTMAX(SARR, SNXT, HAS_NXT, T_, RAGGED_, CHECK_)                                                    \
  {                                                                                                             \
    if constexpr (RAGGED_) {                                                                                    \
      _Pragma("unroll") for (int a = 0; a < QF; ++a)                                                            \
        _Pragma("unroll") for (int f = 0; f < 4; ++f)                                                           \
          _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                         \
            if ((T_) * BKV3 + 16 * f + 4 * lg + r >= N) SARR[a][f][r] = -INFINITY;                              \
    }                                                                                                           \
    if constexpr (CHECK_) {                                                                                     \
    float tmax[QF];                                                                                             \
    bool need_any = ((T_) == 0);                                                                                \
    _Pragma("unroll") for (int a = 0; a < QF; ++a) {                                                            \
      float m0 = max3f(SARR[a][0][0], SARR[a][0][1], SARR[a][0][2]);                                            \
      float m1 = max3f(SARR[a][0][3], SARR[a][1][0], SARR[a][1][1]);                                            \
      float m2 = max3f(SARR[a][1][2], SARR[a][1][3], SARR[a][2][0]);                                            \
      float m3 = max3f(SARR[a][2][1], SARR[a][2][2], SARR[a][2][3]);                                            \
      m0 = max3f(m0, SARR[a][3][0], SARR[a][3][1]);                                                             \
      m1 = max3f(m1, SARR[a][3][2], SARR[a][3][3]);                                                             \
      m0 = max3f(m0, m1, m2);                                                                                   \
      m0 = fmaxf(m0, m3);                                                                                       \
      m0 = xor32_max(xor16_max(m0)); /* the 4 lanes (l, l^16, l^32, l^48) share the query row */                \
      tmax[a] = m0;                                                                                             \
      need_any |= tmax[a] > THR;                                                                                \
    }                                                                                                           \
    if (__any(need_any)) {                                                                                      \
      /* Move the reference maximum of the rows that outgrew it (first tile: of every row): everything */       \
      /* accumulated against the old one is rescaled exactly once, the scores in flight are shifted and */      \
      /* column D of Q is rewritten.  `one` is opaque to the optimiser and defined inside this block: every */  \
      /* value below depends on it, so none of this arithmetic can be speculated into the straight-line */      \
      /* path (hipcc otherwise turns the block into ~50 unconditional VALU instructions per tile). */           \
      float one;                                                                                                \
      asm volatile("v_mov_b32 %0, 1.0" : "=v"(one));                                                            \
      _Pragma("unroll") for (int a = 0; a < QF; ++a) {                                                          \
        const bool need = ((T_) == 0) | (tmax[a] > THR);                                                        \
        const float mnew = need ? bf16_ceil(mrow[a] + tmax[a]) : mrow[a];                                       \
        const float delta = (mnew - mrow[a]) * one; /* exact: both are bf16 values */                           \
        mrow[a] += delta;                                                                                       \
        _Pragma("unroll") for (int f = 0; f < 4; ++f)                                                           \
          _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                       \
            SARR[a][f][r] -= delta;                                                                             \
            if (HAS_NXT) SNXT[a][f][r] -= delta;                                                                \
          }                                                                                                     \
        if ((T_) > 0) { /* (nothing accumulated yet on the first tile) */                                       \
          const float alpha = __builtin_amdgcn_exp2f(-delta);                                                   \
          _Pragma("unroll") for (int d = 0; d < Cfg::DF; ++d) o[a][d] *= alpha;                                 \
        }                                                                                                       \
        if (holds_m) qf[a][MKG].x = f32_bits(-mrow[a]) >> 16; /* column D of Q: bf16(-m); D+1.. stay zero */    \
      }                                                                                                         \
    }                                                                                                           \
    }                                                                                                           \
    _Pragma("unroll") for (int a = 0; a < QF; ++a)                                                              \
      _Pragma("unroll") for (int f = 0; f < 4; ++f)                                                             \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) SARR[a][f][r] = __builtin_amdgcn_exp2f(SARR[a][f][r]);    \
  }
  // O^T += V^T P^T with the V planes of stage ST; 32-key group hh = score fragments 2hh, 2hh+1
#define A3_PV(SARR, ST)                                                                                         \
  {                                                                                                             \
    const unsigned char* vb_ = smem + (ST) * Cfg::STAGE;                                                        \
    _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                                          \
      uint4 pb[QF];                                                                                             \
      _Pragma("unroll") for (int a = 0; a < QF; ++a) {                                                          \
        const float p0 = SARR[a][2 * hh][0], p1 = SARR[a][2 * hh][1], p2 = SARR[a][2 * hh][2], p3 = SARR[a][2 * hh][3]; \
        const float p4 = SARR[a][2 * hh + 1][0], p5 = SARR[a][2 * hh + 1][1], p6 = SARR[a][2 * hh + 1][2],     \
                    p7 = SARR[a][2 * hh + 1][3];                                                                \
        pb[a] = make_uint4(pack_bf16x2(p0, p1), pack_bf16x2(p2, p3), pack_bf16x2(p4, p5), pack_bf16x2(p6, p7)); \
      }                                                                                                         \
      _Pragma("unroll") for (int d = 0; d < Cfg::DF; ++d) {                                                     \
        typedef __attribute__((address_space(3))) s16x4* lds_v4;                                                \
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(vb_ + voff[d] + (32 * hh) * 16));      \
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(vb_ + voff[d] + (32 * hh + 16) * 16)); \
        const uint2 a0 = __builtin_bit_cast(uint2, v0), a1 = __builtin_bit_cast(uint2, v1);                     \
        const uint4 vf = make_uint4(a0.x, a0.y, a1.x, a1.y);                                                    \
        _Pragma("unroll") for (int a = 0; a < QF; ++a) mma_kgroup<bf16_t>(vf, pb[a], o[a][d]);                  \
      }                                                                                                         \
    }                                                                                                           \
  }

  f32x4 sA[QF][4];
  // One key tile.  STAGE, RAGGED and CHECK are compile-time: the LDS offsets of every fragment read fold into immediates,
  // the -inf masking of keys past N exists only in the instantiation the last, partial tile runs (as a run-time branch
  // the compiler if-converts it into ~45 VALU instructions on every tile), and so does the row-maximum code.
  auto tile = [&](int t, auto stage_c, auto ragged_c, auto check_c) __attribute__((always_inline)) {
    constexpr int ST = decltype(stage_c)::value;
    constexpr bool RAGGED = decltype(ragged_c)::value;
    constexpr bool CHECK = decltype(check_c)::value;
    if (t + NST - 1 < ntiles) issue_tile(t + NST - 1, (ST + NST - 1) % NST);
    A3_QK(sA, ST)
    A3_SOFTMAX(sA, sA, false, t, RAGGED, CHECK)
    A3_PV(sA, ST)
    // the next tile must have landed before anyone reads it; the one after may stay in flight
    if (t + 1 < ntiles) wait_tiles_ahead((NST > 2 && t + 2 < ntiles) ? 1 : 0);
    __syncthreads();
  };
  // All key tiles (the DMA prologue - tiles 0 .. NST-2 - is in flight).  PERIOD: the row maxima are looked at on tile 0, on
  // every PERIOD-th tile and on the ragged tail only.  The folded maximum m only has to keep exp2(s - m) inside fp32 / bf16
  // range - a score may exceed it by up to ~2^7 - and the relative accuracy of p does not depend on m, so between looks the
  // scores are exponentiated against the m of the last look (max3 + the cross-lane reductions were 45 % of the VALU
  // instructions of a tile: d = 40, N = 4096 at B = 8: 239 -> 220 us).  A row whose scores outgrew m by more than that between
  // two looks shows up as a row sum >= 2^100 (inf included) and sends the whole workgroup through the exact pass below.
  auto all_tiles = [&](auto period_c) __attribute__((always_inline)) {
    constexpr int PERIOD = decltype(period_c)::value;
    static_assert(PERIOD >= 1 && (PERIOD & (PERIOD - 1)) == 0, "power of two");
    // tiles in flight beyond tile 0 after the prologue: min(NST - 1, ntiles) - 1
    wait_tiles_ahead((NST > 2 && ntiles > 1) ? 1 : 0);
    __syncthreads();
    for (int t0 = 0; t0 < ntiles; t0 += NST) {
      static_for_n<NST>([&](auto sc) __attribute__((always_inline)) {
        const int t = t0 + decltype(sc)::value;
        if (t < nfull) {
          if (PERIOD == 1 || (t & (PERIOD - 1)) == 0) tile(t, sc, std::false_type{}, std::true_type{});
          else tile(t, sc, std::false_type{}, std::false_type{});
        } else if (t < ntiles) {
          tile(t, sc, std::true_type{}, std::true_type{});
        }
      });
    }
  };
  all_tiles(std::integral_constant<int, LAZY>{});
  if constexpr (LAZY > 1) {
    bool bad = false;
#pragma unroll
    for (int a = 0; a < QF; ++a) {
      // (an exponent test on the bits: this file is built with -fno-honor-nans, a float compare may be folded for NaN)
      const unsigned lb = f32_bits(o[a][D / 16][(D % 16) % 4]);
      bad |= (lg == (D % 16) / 4) && ((lb & 0x7f800000u) >= ((127u + 100u) << 23));
    }
    if (__syncthreads_or(bad ? 1 : 0)) {            // rare: start over with the maxima tracked on every tile
#pragma unroll
      for (int a = 0; a < QF; ++a) {
        mrow[a] = 0.f;
        if (holds_m) qf[a][MKG].x = 0u;
#pragma unroll
        for (int d = 0; d < Cfg::DF; ++d) o[a][d] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      static_for_n<NST - 1>([&](auto jc) __attribute__((always_inline)) {
        constexpr int J = decltype(jc)::value;
        if (J < ntiles) issue_tile(J, J);
      });
      all_tiles(std::integral_constant<int, 1>{});
    }
  }
#undef A3_QK
#undef A3_SOFTMAX
#undef A3_PV

  // ---- normalise and store: lane (q, g) holds d = df*16 + 4g + r; the row sum sits in row D of O^T ----
#pragma unroll
  for (int a = 0; a < QF; ++a) {
    const float l = __shfl(o[a][D / 16][(D % 16) % 4], lq + 16 * ((D % 16) / 4), 64);
    const float inv = 1.0f / l;
    const int q = q0 + a * 16 + lq;
    if (q >= N) continue;
    bf16_t* op = out + ((size_t)b * N + q) * C + (size_t)h * D;
#pragma unroll
    for (int d = 0; d < Cfg::DF; ++d) {
      const int dd = d * 16 + 4 * lg;
      if (dd >= D) continue;
      const f32x4 v = o[a][d] * inv;
      const float v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
      *(uint2*)(op + dd) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
    }
  }
}

template <int D, int QF, int WPS, int NST, bool PIPE = false, int LAZY = 1, int NWV = 4>
int run3(const void* qkv, void* out, int B, int N, int C, int heads, hipStream_t s) {
  using Cfg = A3Cfg<D, NWV>;
  const size_t lds = (size_t)NST * Cfg::STAGE;
  auto kern = attn3_kernel<D, QF, WPS, NST, PIPE, LAZY, NWV>;
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set[dev] = true;
  }
  const int nqb = (N + 16 * NWV * QF - 1) / (16 * NWV * QF);
  const float scale_log2e = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  hipLaunchKernelGGL(kern, dim3(nqb * heads * B), dim3(64 * NWV), lds, s, (const bf16_t*)qkv, (bf16_t*)out, N, C, heads, scale_log2e);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

// bf16, head dim 40 or 80; returns -100 when the shape is not handled here (caller falls through to attention.hip)
int launch_attention4(const void* qkv, void* out, int B, int N, int C, int heads, int variant, hipStream_t s);   // attention4.hip

int launch_attention3(const void* qkv, void* out, int B, int N, int C, int heads, int variant, hipStream_t s) {
  const int d = C / heads;
  // variant: 0 = shipped choice; the others are alternatives kept for A/B measurements and the parity tests
  // (template arguments: head dim, query fragments per wave, waves per SIMD, ring stages, -, maxima look period, waves per workgroup)
  if (d == 40) {
    if (variant == 1) return run3<40, 1, 4, 3>(qkv, out, B, N, C, heads, s);
    if (variant == 4) return run3<40, 2, 3, 2>(qkv, out, B, N, C, heads, s);
    if (variant == 5) return run3<40, 2, 4, 3>(qkv, out, B, N, C, heads, s);
    if (variant == 6) return run3<40, 2, 3, 3>(qkv, out, B, N, C, heads, s);                  // the round-2 kernel: 4 waves, maxima on every tile
    if (variant == 8) return run3<40, 2, 4, 3, false, 1, 8>(qkv, out, B, N, C, heads, s);     // 8 waves, maxima on every tile
    if (variant == 9) return run3<40, 2, 4, 3, false, 4, 8>(qkv, out, B, N, C, heads, s);
    if (variant == 10) return run3<40, 2, 3, 3, false, 16, 4>(qkv, out, B, N, C, heads, s);
    if (variant >= 11 && variant <= 14) return launch_attention4(qkv, out, B, N, C, heads, variant - 11, s);   // attention4.hip, forced form
    // 8-wave workgroups: 256 query rows share every K / V tile (half the LDS-DMA instructions per score: issuing one parks
    // the wave for 60-185 cycles); short sequences keep the 4-wave form (too few workgroups otherwise)
    const bool big = (long)B * heads * ((N + 255) / 256) >= 256;
    // shipped (0): the 32x32x16 score-block kernel of attention4.hip (round 4: N = 4096 at B = 8 224 -> 196 us, N = 16384 at B = 4
    // 1.67 -> 1.49 ms, N = 1000 15.8 -> 14.1 us); 7: this file's kernel under the same rule (the round-3 choice), for A/B
    if (variant == 0) return launch_attention4(qkv, out, B, N, C, heads, big ? 0 : 2, s);
    if (big) return run3<40, 2, 4, 3, false, 16, 8>(qkv, out, B, N, C, heads, s);
    return run3<40, 2, 3, 3, false, 16, 4>(qkv, out, B, N, C, heads, s);
  }
  if (d == 80) {
    if (variant == 1) return run3<80, 1, 3, 2>(qkv, out, B, N, C, heads, s);
    if (variant == 4) return run3<80, 2, 3, 2>(qkv, out, B, N, C, heads, s);
    if (variant == 5) return run3<80, 1, 2, 3>(qkv, out, B, N, C, heads, s);
    if (variant == 6) return run3<80, 2, 2, 3>(qkv, out, B, N, C, heads, s);                  // the round-2 kernel
    if (variant == 8) return run3<80, 1, 4, 3, false, 16, 8>(qkv, out, B, N, C, heads, s);
    if (variant == 9) return run3<80, 2, 2, 3, false, 16, 8>(qkv, out, B, N, C, heads, s);
    if (variant == 10) return run3<80, 1, 3, 3, false, 16, 8>(qkv, out, B, N, C, heads, s);
    if (variant == 0 && (long)B * heads * ((N + 255) / 256) >= 256) return run3<80, 2, 2, 3, false, 16, 8>(qkv, out, B, N, C, heads, s);
    return run3<80, 2, 2, 3, false, 16, 4>(qkv, out, B, N, C, heads, s);
  }
  return -100;
}

}  // namespace ldmseg
