// Tiled online-softmax ("flash") self-attention for the UNet transformer blocks
// (8 heads, head dims 40 / 80 / 160, N = H*W tokens) on gfx950 MFMA.
//
// Input is the fused projection qkv [B, N, 3C] (q | k | v, channel = head*d + i,
// SURVEY App. A); output [B, N, C].  Per workgroup: 4 waves x QF x 16 query rows;
// K/V tiles of 64 keys staged in LDS (V transposed so that both MFMA operands are
// contraction-contiguous).  Scores are computed TRANSPOSED (MFMA A = K, B = Q):
// a lane then owns one query row's scores, so the softmax row statistics are
// in-lane reductions plus two wave shuffles (xor 16, 32), the probabilities stay
// in registers as the next MFMA's B operand (no LDS round trip for P), and the
// output accumulator O^T is rescaled by a per-lane scalar.  The K-tile rows are
// read through a fixed permutation so that the 8 bf16 probabilities a lane packs
// are 8 consecutive keys of the V^T row it multiplies.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace ldmseg {
namespace {

constexpr int BKV = 64;

template <typename T, int D> struct AttnCfg {
  static constexpr int S = (int)sizeof(T);
  static constexpr int KGE = 64 / S;                        // elements per 64-B k-group
  static constexpr int KG = (D + KGE - 1) / KGE;            // k-groups covering D (QK^T)
  static constexpr int DPAD = KG * KGE;
  static constexpr int DCH = D * S / 16;                    // real 16-B chunks per row
  static constexpr int KCH = DPAD * S / 16;                 // chunks per padded row
  static constexpr int KROW = DPAD * S + 16;                // Ks row stride, bytes
  static constexpr int DF = (D + 15) / 16;                  // output d fragments
  static constexpr int VROW = BKV * S + 16;                 // Vt row stride, bytes
  static constexpr int PG = BKV / KGE;                      // P k-groups per tile (PV)
  static constexpr int KS_BYTES = BKV * KROW;
  static constexpr int VT_BYTES = DF * 16 * VROW;
};

// row statistics reduce over the 4 lanes (l, l^16, l^32, l^48) that share a query row with
// v_permlane16_swap / v_permlane32_swap (common.h) instead of LDS round trips (ds_bpermute)
// X3 (compute_dtype "bf16x3", round 5): T = bf16 inside, fp32 in HBM.  q, k, v are split into bf16 hi + lo (lo = bf16(x - hi)) - K and V
// once per tile when they are committed to the LDS, as two planes of the bf16 layout; Q once per workgroup; the probabilities in
// registers - and every product block is three MFMAs, Xl.Yh + Xh.Yl + Xh.Yh, with fp32 accumulation: the arithmetic of igemm.hip's
// x3 path on the bf16 kernel's structure (8 keys per 16-byte chunk, ones row of V^T for the row sums).
template <typename T, int D, int QF, bool X3 = false>
__global__ __launch_bounds__(256) void attention_kernel(const std::conditional_t<X3, float, T>* __restrict__ qkv,
                                                        std::conditional_t<X3, float, T>* __restrict__ out,
                                                        int N, int C, int heads, float scale_log2e) {
  using IO = std::conditional_t<X3, float, T>;
  static_assert(!X3 || sizeof(T) == 2, "the split-bf16 form runs on the bf16 layout");
  using Cfg = AttnCfg<T, D>;
  constexpr int PC = Chunk<T>::N;
  constexpr int NCH = BKV * Cfg::DCH;          // real 16-B chunks of one K (or V) tile
  constexpr int KIT = (NCH + 255) / 256;       // chunks per thread per tile
  constexpr int KR_N = X3 ? 2 * KIT : KIT;     // X3: a chunk of 8 elements is two 16-byte fp32 loads
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int PLANE = Cfg::KS_BYTES + Cfg::VT_BYTES;
  constexpr int STAGE = X3 ? 2 * PLANE : PLANE;   // X3: the lo plane sits PLANE bytes behind the hi plane
  constexpr int NST = (2 * STAGE <= 144 * 1024) ? 2 : 1;   // two stages (one barrier per key tile) when they fit

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 15, lg = lane >> 4;

  // XCD-aware: all query blocks of one (image, head) share an XCD's L2
  const int nqb = (N + 64 * QF - 1) / (64 * QF);
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
  const IO* qbase = qkv + (size_t)b * N * ld + (size_t)h * D;
  const unsigned char* kbase = (const unsigned char*)(qbase + C);
  const unsigned char* vbase = (const unsigned char*)(qbase + 2 * C);
  const size_t ldb = ld * sizeof(IO);
  // hi / lo split of eight floats (two 16-byte chunks) into two bf16 chunks
  auto split8 = [&](const u32x4& c0, const u32x4& c1, u32x4& hi, u32x4& lo) __attribute__((always_inline)) {
    const unsigned w0[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float f0 = bits_f32(w0[2 * i]), f1 = bits_f32(w0[2 * i + 1]);
      const unsigned hh = pack_bf16x2(f0, f1);
      hi[i] = hh;
      lo[i] = pack_bf16x2(f0 - bits_f32(hh << 16), f1 - bits_f32(hh & 0xffff0000u));
    }
  };

  // K/V tile staging: global -> registers (issued before the MFMA block of the previous tile, so
  // the L2/HBM latency hides under it) -> LDS after that tile's last LDS read.
  // K chunk i -> (row i / DCH, chunk i % DCH): coalesced 16-B loads, ds_write_b128 rows.
  // V chunk i -> (row i % 64, chunk i / 64): a wave writes 64 consecutive keys of one d row of
  // V^T, i.e. consecutive 2-/4-byte LDS addresses - no bank conflicts in the transpose.
  // Two register sets (A, B) so that two K/V tiles can be in flight.  The bodies are macros over
  // the set name: passing the arrays by reference into lambdas made hipcc index them dynamically
  // (-> scratch memory, 6x slower).
  u32x4 kregA[KR_N], vregA[KR_N], kregB[KR_N], vregB[KR_N];
#define ATTN_PREFETCH(T_, KR, VR)                                                                          \
  {                                                                                                        \
    const int kv0_ = (T_) * BKV;                                                                           \
    _Pragma("unroll") for (int it = 0; it < KIT; ++it) {                                                   \
      const int i = tid + it * 256;                                                                        \
      if (it + 1 < KIT || i < NCH) {                                                                       \
        /* rows past N are clamped to the last key: their scores are masked to -inf below, so the */      \
        /* (finite) duplicate K/V data never contributes */                                                \
        const int krow = i / Cfg::DCH, kch = i - krow * Cfg::DCH;                                          \
        const int vrow = i & (BKV - 1), vch = i / BKV;                                                     \
        if constexpr (X3) {                                                                                \
          const unsigned char* kp_ = kbase + (size_t)min(kv0_ + krow, N - 1) * ldb + kch * 32;             \
          const unsigned char* vp_ = vbase + (size_t)min(kv0_ + vrow, N - 1) * ldb + vch * 32;             \
          KR[2 * it] = *(const u32x4*)kp_; KR[2 * it + 1] = *(const u32x4*)(kp_ + 16);                     \
          VR[2 * it] = *(const u32x4*)vp_; VR[2 * it + 1] = *(const u32x4*)(vp_ + 16);                     \
        } else {                                                                                           \
          KR[it] = *(const u32x4*)(kbase + (size_t)min(kv0_ + krow, N - 1) * ldb + kch * 16);              \
          VR[it] = *(const u32x4*)(vbase + (size_t)min(kv0_ + vrow, N - 1) * ldb + vch * 16);              \
        }                                                                                                  \
      }                                                                                                    \
    }                                                                                                      \
  }
#define ATTN_COMMIT(STAGE_, KR, VR)                                                                        \
  {                                                                                                        \
    unsigned char* Ks_ = smem + (STAGE_) * STAGE;                                                          \
    unsigned char* Vt_ = Ks_ + Cfg::KS_BYTES;                                                              \
    _Pragma("unroll") for (int it = 0; it < KIT; ++it) {                                                   \
      const int i = tid + it * 256;                                                                        \
      if (it + 1 < KIT || i < NCH) {                                                                       \
        const int krow = i / Cfg::DCH, kch = i - krow * Cfg::DCH;                                          \
        const int vrow = i & (BKV - 1), vch = i / BKV;                                                     \
        if constexpr (X3) {                                                                                \
          u32x4 kh_, kl_, vh_, vl_;                                                                        \
          split8(KR[2 * it], KR[2 * it + 1], kh_, kl_);                                                    \
          split8(VR[2 * it], VR[2 * it + 1], vh_, vl_);                                                    \
          *(u32x4*)(Ks_ + krow * Cfg::KROW + kch * 16) = kh_;                                              \
          *(u32x4*)(Ks_ + PLANE + krow * Cfg::KROW + kch * 16) = kl_;                                      \
          _Pragma("unroll") for (int k = 0; k < PC; ++k) {                                                 \
            *(T*)(Vt_ + (vch * PC + k) * Cfg::VROW + vrow * sizeof(T)) = chunk_elem<T>(vh_, k);            \
            *(T*)(Vt_ + PLANE + (vch * PC + k) * Cfg::VROW + vrow * sizeof(T)) = chunk_elem<T>(vl_, k);    \
          }                                                                                                \
        } else {                                                                                           \
          *(u32x4*)(Ks_ + krow * Cfg::KROW + kch * 16) = KR[it];                                           \
          const u32x4 vv_ = VR[it];                                                                        \
          _Pragma("unroll") for (int k = 0; k < PC; ++k)                                                   \
              *(T*)(Vt_ + (vch * PC + k) * Cfg::VROW + vrow * sizeof(T)) = chunk_elem<T>(vv_, k);          \
        }                                                                                                  \
      }                                                                                                    \
    }                                                                                                      \
  }

  ATTN_PREFETCH(0, kregA, vregA)
  // zero the whole LDS once: K pad chunks (d >= D), V^T rows d >= D and row pads stay zero
  for (int i = tid * 16; i < NST * STAGE; i += 256 * 16) *(uint4*)(smem + i) = make_uint4(0, 0, 0, 0);
  // ONES: the head dim leaves spare rows in the last 16-row fragment of V^T (d = 40 -> rows 40..47).  Row D holds
  // ones, so the PV matrix product accumulates the softmax denominator sum_k p[q][k] into O^T[D][q] for free -
  // rescaled together with O - and the per-score row-sum adds disappear from the VALU-bound softmax.
  // (bf16 mode only: the fp32 parity mode keeps the exact VALU row sums)
  constexpr bool ONES = (D % 16) != 0 && sizeof(T) == 2;
  if constexpr (ONES) {
    __syncthreads();
    if (tid < NST * BKV) {
      const int st = tid / BKV, key = tid % BKV;
      *(T*)(smem + st * STAGE + Cfg::KS_BYTES + D * Cfg::VROW + key * sizeof(T)) = from_f32<T>(1.0f);
    }
  }

  // Q fragments (MFMA B operand): lane (q = lq, g = lg) holds chunk kg*4+g of its row
  const int q0 = qb * 64 * QF + wave * 16 * QF;
  uint4 qf[QF][Cfg::KG];
  uint4 ql[X3 ? QF : 1][X3 ? Cfg::KG : 1];       // X3: lo halves of Q
#pragma unroll
  for (int a = 0; a < QF; ++a) {
    const int q = q0 + a * 16 + lq;
#pragma unroll
    for (int kg = 0; kg < Cfg::KG; ++kg) {
      const int ch = kg * 4 + lg;
      if constexpr (X3) {
        u32x4 c0 = {0u, 0u, 0u, 0u}, c1 = c0;
        if (q < N && ch < Cfg::DCH) {
          const unsigned char* qp = (const unsigned char*)(qbase + (size_t)q * ld) + ch * 32;
          c0 = *(const u32x4*)qp;
          c1 = *(const u32x4*)(qp + 16);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {                     // d^-1/2 * log2(e) folded into Q before the split
          c0[e] = f32_bits(bits_f32(c0[e]) * scale_log2e);
          c1[e] = f32_bits(bits_f32(c1[e]) * scale_log2e);
        }
        u32x4 hh, ll;
        split8(c0, c1, hh, ll);
        qf[a][kg] = make_uint4(hh[0], hh[1], hh[2], hh[3]);
        ql[a][kg] = make_uint4(ll[0], ll[1], ll[2], ll[3]);
        continue;
      }
      qf[a][kg] = (q < N && ch < Cfg::DCH) ? *(const uint4*)((const unsigned char*)(qbase + (size_t)q * ld) + ch * 16)
                                            : make_uint4(0, 0, 0, 0);
      // fold d^-1/2 * log2(e) into Q once, so that the scores come out of the MFMA in exp2 units
      float qv[PC];
      Chunk<T>::unpack(qf[a][kg], qv);
#pragma unroll
      for (int e = 0; e < PC; ++e) qv[e] *= scale_log2e;
      qf[a][kg] = Chunk<T>::pack(qv);
    }
  }

  f32x4 o[QF][Cfg::DF];
  float mrow[QF], lrow[QF];
#pragma unroll
  for (int a = 0; a < QF; ++a) {
    mrow[a] = -INFINITY;
    lrow[a] = 0.f;
#pragma unroll
    for (int d = 0; d < Cfg::DF; ++d) o[a][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // K-tile row read by MFMA row i of score fragment f (see header comment)
  int krow_f[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    if constexpr (sizeof(T) == 2) krow_f[f] = 32 * (f >> 1) + 8 * (lq >> 2) + (lq & 3) + 4 * (f & 1);
    else krow_f[f] = 16 * f + lq;
  }

  const int ntiles = (N + BKV - 1) / BKV;
  const int nfull = N / BKV;                 // tiles without masked keys
  __syncthreads();
  ATTN_COMMIT(0, kregA, vregA)
  if (NST == 2 && ntiles > 1) ATTN_PREFETCH(1, kregA, vregA)
  __syncthreads();

  auto tile = [&](int kv0, int stage, auto ragged_tag) __attribute__((always_inline)) {
    constexpr bool RAGGED = decltype(ragged_tag)::value;
    const unsigned char* Ks = smem + stage * STAGE;
    const unsigned char* Vt = Ks + Cfg::KS_BYTES;
    // ---- S^T = K Q^T ----
    f32x4 s[QF][4];
#pragma unroll
    for (int a = 0; a < QF; ++a)
#pragma unroll
      for (int f = 0; f < 4; ++f) s[a][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kg = 0; kg < Cfg::KG; ++kg) {
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const uint4 kf = *(const uint4*)(Ks + krow_f[f] * Cfg::KROW + (kg * 4 + lg) * 16);
        if constexpr (X3) {                              // small terms first
          const uint4 kfl = *(const uint4*)(Ks + PLANE + krow_f[f] * Cfg::KROW + (kg * 4 + lg) * 16);
#pragma unroll
          for (int a = 0; a < QF; ++a) mma_kgroup<T>(kfl, qf[a][kg], s[a][f]);
#pragma unroll
          for (int a = 0; a < QF; ++a) mma_kgroup<T>(kf, ql[a][kg], s[a][f]);
        }
#pragma unroll
        for (int a = 0; a < QF; ++a) mma_kgroup<T>(kf, qf[a][kg], s[a][f]);
      }
    }

    // ---- online softmax (lane = one query row, 16 of its 64 scores); scores are already in
    // exp2 units (scale folded into Q), so each costs one subtract + one raw v_exp_f32 ----
    float alpha[QF];
    bool grew = false;
#pragma unroll
    for (int a = 0; a < QF; ++a) {
      float tmax = -INFINITY;
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float z = s[a][f][r];
          if constexpr (RAGGED) {
            int kv;
            if constexpr (sizeof(T) == 2) kv = kv0 + 32 * (f >> 1) + 8 * lg + r + 4 * (f & 1);
            else kv = kv0 + 16 * f + 4 * lg + r;
            if (kv >= N) z = -INFINITY;
            s[a][f][r] = z;
          }
          tmax = fmaxf(tmax, z);
        }
      tmax = xor32_max(xor16_max(tmax));
      const float mnew = fmaxf(mrow[a], tmax);   // finite: key kv0 is always valid
      grew |= (mnew > mrow[a]);
      alpha[a] = __builtin_amdgcn_exp2f(mrow[a] - mnew);
      mrow[a] = mnew;
      float ps = 0.f;
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float z = s[a][f][r];
          const float pv = __builtin_amdgcn_exp2f(z - mnew);
          s[a][f][r] = pv;
          if constexpr (!ONES) ps += pv;
        }
      if constexpr (!ONES) lrow[a] = lrow[a] * alpha[a] + ps;
    }
    // rescale O only when some row's running max actually grew in this tile (alpha == 1 exactly
    // otherwise): after the first few tiles this skips the accumulator round trip entirely
    if (__any(grew)) {
#pragma unroll
      for (int a = 0; a < QF; ++a)
#pragma unroll
        for (int d = 0; d < Cfg::DF; ++d) o[a][d] *= alpha[a];
    }

    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int g = 0; g < Cfg::PG; ++g) {
      uint4 pb[QF];
      uint4 pbl[X3 ? QF : 1];
#pragma unroll
      for (int a = 0; a < QF; ++a) {
        if constexpr (sizeof(T) == 2) {
          const float p0 = s[a][2 * g][0], p1 = s[a][2 * g][1], p2 = s[a][2 * g][2], p3 = s[a][2 * g][3];
          const float p4 = s[a][2 * g + 1][0], p5 = s[a][2 * g + 1][1], p6 = s[a][2 * g + 1][2], p7 = s[a][2 * g + 1][3];
          pb[a] = make_uint4(pack_bf16x2(p0, p1), pack_bf16x2(p2, p3), pack_bf16x2(p4, p5), pack_bf16x2(p6, p7));
          if constexpr (X3) {
            const unsigned hx = pb[a].x, hy = pb[a].y, hz = pb[a].z, hw = pb[a].w;
            pbl[a] = make_uint4(pack_bf16x2(p0 - bits_f32(hx << 16), p1 - bits_f32(hx & 0xffff0000u)),
                                pack_bf16x2(p2 - bits_f32(hy << 16), p3 - bits_f32(hy & 0xffff0000u)),
                                pack_bf16x2(p4 - bits_f32(hz << 16), p5 - bits_f32(hz & 0xffff0000u)),
                                pack_bf16x2(p6 - bits_f32(hw << 16), p7 - bits_f32(hw & 0xffff0000u)));
          }
        } else {
          const float p0 = s[a][g][0], p1 = s[a][g][1], p2 = s[a][g][2], p3 = s[a][g][3];
          pb[a] = make_uint4(f32_bits(p0), f32_bits(p1), f32_bits(p2), f32_bits(p3));
        }
      }
#pragma unroll
      for (int d = 0; d < Cfg::DF; ++d) {
        const uint4 vf = *(const uint4*)(Vt + (d * 16 + lq) * Cfg::VROW + (g * 4 + lg) * 16);
        if constexpr (X3) {
          const uint4 vfl = *(const uint4*)(Vt + PLANE + (d * 16 + lq) * Cfg::VROW + (g * 4 + lg) * 16);
#pragma unroll
          for (int a = 0; a < QF; ++a) mma_kgroup<T>(vfl, pb[a], o[a][d]);
#pragma unroll
          for (int a = 0; a < QF; ++a) mma_kgroup<T>(vf, pbl[a], o[a][d]);
        }
#pragma unroll
        for (int a = 0; a < QF; ++a) mma_kgroup<T>(vf, pb[a], o[a][d]);
      }
    }
  };

  auto run_tile = [&](int t, int st) __attribute__((always_inline)) {
    if (t < nfull) tile(t * BKV, st, std::false_type{});
    else tile(t * BKV, st, std::true_type{});
  };
  if constexpr (NST == 2) {
    // K/V tiles are fetched TWO tiles ahead into alternating register sets (the L2 round trip is
    // longer than one tile of work at 2-3 waves/SIMD) and committed to the free LDS stage one tile
    // ahead.  Invariant at the top of the even half: stage 0 holds tile t, set A holds tile t+1.
    int t = 0;
    for (; t + 1 < ntiles; t += 2) {
      if (t + 2 < ntiles) ATTN_PREFETCH(t + 2, kregB, vregB)
      run_tile(t, 0);
      ATTN_COMMIT(1, kregA, vregA)
      __syncthreads();
      if (t + 3 < ntiles) ATTN_PREFETCH(t + 3, kregA, vregA)
      run_tile(t + 1, 1);
      if (t + 2 < ntiles) ATTN_COMMIT(0, kregB, vregB)
      __syncthreads();
    }
    if (t < ntiles) run_tile(t, 0);
  } else {
    for (int t = 0; t < ntiles; ++t) {
      const bool more = t + 1 < ntiles;
      if (more) ATTN_PREFETCH(t + 1, kregA, vregA)
      run_tile(t, 0);
      if (more) {
        __syncthreads();
        ATTN_COMMIT(0, kregA, vregA)
        __syncthreads();
      }
    }
  }
#undef ATTN_PREFETCH
#undef ATTN_COMMIT

  // ---- normalise and store: lane (q, g) holds d = df*16 + 4g + r ----
#pragma unroll
  for (int a = 0; a < QF; ++a) {
    float l;
    if constexpr (ONES) {
      // O^T[D][q] sits in lane (q, g = (D % 16) / 4), element (D % 16) % 4 of fragment D / 16
      l = __shfl(o[a][D / 16][(D % 16) % 4], lq + 16 * ((D % 16) / 4), 64);
    } else {
      l = xor32_sum(xor16_sum(lrow[a]));
    }
    const float inv = 1.0f / l;
    const int q = q0 + a * 16 + lq;
    if (q >= N) continue;
    IO* op = out + ((size_t)b * N + q) * C + (size_t)h * D;
#pragma unroll
    for (int d = 0; d < Cfg::DF; ++d) {
      const int dd = d * 16 + 4 * lg;
      if (dd >= D) continue;
      const f32x4 v = o[a][d] * inv;
      const float v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
      if constexpr (sizeof(IO) == 2) {
        *(uint2*)(op + dd) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
      } else {
        *(f32x4*)(op + dd) = v;
      }
    }
  }
}

template <typename T, int D, int QF, bool X3 = false>
int run(const void* qkv, void* out, int B, int N, int C, int heads, hipStream_t s) {
  using Cfg = AttnCfg<T, D>;
  using IO = std::conditional_t<X3, float, T>;
  const size_t stage = (size_t)(Cfg::KS_BYTES + Cfg::VT_BYTES) * (X3 ? 2 : 1);
  const size_t lds = (2 * stage <= 144 * 1024 ? 2 : 1) * stage;
  auto kern = attention_kernel<T, D, QF, X3>;
  static bool attr_set[64] = {};     // per device: a process may hold handles on several GPUs
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set[dev] = true;
  }
  const int nqb = (N + 64 * QF - 1) / (64 * QF);
  const float scale_log2e = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  hipLaunchKernelGGL(kern, dim3(nqb * heads * B), dim3(256), lds, s, (const IO*)qkv, (IO*)out, N, C, heads, scale_log2e);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

int g_attn_qf1 = 0;

template <typename T>
int dispatch(const void* qkv, void* out, int B, int N, int C, int heads, hipStream_t s) {
  const int d = C / heads;
  const bool big = N >= 256 && g_attn_qf1 != 1 && g_attn_qf1 != 3;
  switch (d) {
    case 40: return big ? run<T, 40, 2>(qkv, out, B, N, C, heads, s) : run<T, 40, 1>(qkv, out, B, N, C, heads, s);
    case 80: return big ? run<T, 80, 2>(qkv, out, B, N, C, heads, s) : run<T, 80, 1>(qkv, out, B, N, C, heads, s);
    case 160: return run<T, 160, 1>(qkv, out, B, N, C, heads, s);
    default: return -2;
  }
}

// fp32 tensors, split-bf16 products (compute_dtype "bf16x3")
int dispatch_x3(const void* qkv, void* out, int B, int N, int C, int heads, hipStream_t s) {
  const int d = C / heads;
  const bool big = N >= 256;
  switch (d) {
    case 40: return big ? run<bf16_t, 40, 2, true>(qkv, out, B, N, C, heads, s) : run<bf16_t, 40, 1, true>(qkv, out, B, N, C, heads, s);
    case 80: return big ? run<bf16_t, 80, 2, true>(qkv, out, B, N, C, heads, s) : run<bf16_t, 80, 1, true>(qkv, out, B, N, C, heads, s);
    case 160: return run<bf16_t, 160, 1, true>(qkv, out, B, N, C, heads, s);
    default: return -2;
  }
}

}  // namespace

void attention_set_qf1(int v) { g_attn_qf1 = v; }

int launch_attention3(const void* qkv, void* out, int B, int N, int C, int heads, int variant, hipStream_t s);   // attention3.hip

int launch_attention(const void* qkv, void* out, int B, int N, int C, int heads, int dtype, hipStream_t s) {
  if (C % heads != 0 || N <= 0) return -2;
  if (dtype == 2) return dispatch_x3(qkv, out, B, N, C, heads, s);      // LDMSEG_BF16X3: fp32 in HBM, three bf16 MFMAs per product block
  // bf16 perf mode, head dims 40 / 80: the LDS-DMA + folded-max kernel of attention3.hip (knob value 2 forces this file's
  // kernel for A/B measurements; 1 = 16 query rows per wave)
  if (dtype == DT_BF16 && g_attn_qf1 != 2 && g_attn_qf1 != 3) {      // 0, 1, 4..7: variants of attention3.hip
    const int r = launch_attention3(qkv, out, B, N, C, heads, g_attn_qf1, s);
    if (r != -100) return r;
  }
  return dtype == DT_BF16 ? dispatch<bf16_t>(qkv, out, B, N, C, heads, s) : dispatch<float>(qkv, out, B, N, C, heads, s);
}

}  // namespace ldmseg
