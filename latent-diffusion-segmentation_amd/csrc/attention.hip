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

template <typename T, int D, int QF>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                        int N, int C, int heads, float scale_log2e) {
  using Cfg = AttnCfg<T, D>;
  constexpr int PC = Chunk<T>::N;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;
  unsigned char* Vt = smem + Cfg::KS_BYTES;

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
  const T* qbase = qkv + (size_t)b * N * ld + (size_t)h * D;
  const T* kbase = qbase + C;
  const T* vbase = qbase + 2 * C;

  // zero V^T once (rows d >= D and the pad stay zero for the whole kernel)
  for (int i = tid * 16; i < Cfg::VT_BYTES; i += 256 * 16) *(uint4*)(Vt + i) = make_uint4(0, 0, 0, 0);

  // Q fragments (MFMA B operand): lane (q = lq, g = lg) holds chunk kg*4+g of its row
  const int q0 = qb * 64 * QF + wave * 16 * QF;
  uint4 qf[QF][Cfg::KG];
#pragma unroll
  for (int a = 0; a < QF; ++a) {
    const int q = q0 + a * 16 + lq;
#pragma unroll
    for (int kg = 0; kg < Cfg::KG; ++kg) {
      const int ch = kg * 4 + lg;
      qf[a][kg] = (q < N && ch < Cfg::DCH) ? *(const uint4*)((const unsigned char*)(qbase + (size_t)q * ld) + ch * 16)
                                            : make_uint4(0, 0, 0, 0);
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
  int krow[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    if constexpr (sizeof(T) == 2) krow[f] = 32 * (f >> 1) + 8 * (lq >> 2) + (lq & 3) + 4 * (f & 1);
    else krow[f] = 16 * f + lq;
  }

  const int ntiles = (N + BKV - 1) / BKV;
  for (int t = 0; t < ntiles; ++t) {
    const int kv0 = t * BKV;
    __syncthreads();
    // ---- stage K (row-major, zero padded) and V (transposed) ----
    for (int i = tid; i < BKV * Cfg::KCH; i += 256) {
      const int row = i / Cfg::KCH, ch = i - row * Cfg::KCH;
      const int kv = kv0 + row;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (kv < N && ch < Cfg::DCH) v = *(const uint4*)((const unsigned char*)(kbase + (size_t)kv * ld) + ch * 16);
      *(uint4*)(Ks + row * Cfg::KROW + ch * 16) = v;
    }
    for (int i = tid; i < BKV * Cfg::DCH; i += 256) {
      const int row = i / Cfg::DCH, ch = i - row * Cfg::DCH;
      const int kv = kv0 + row;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (kv < N) v = *(const uint4*)((const unsigned char*)(vbase + (size_t)kv * ld) + ch * 16);
      union { uint4 u; T e[PC]; } cv;
      cv.u = v;
#pragma unroll
      for (int k = 0; k < PC; ++k) *(T*)(Vt + (ch * PC + k) * Cfg::VROW + row * sizeof(T)) = cv.e[k];
    }
    __syncthreads();

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
        const uint4 kf = *(const uint4*)(Ks + krow[f] * Cfg::KROW + (kg * 4 + lg) * 16);
#pragma unroll
        for (int a = 0; a < QF; ++a) mma_kgroup<T>(kf, qf[a][kg], s[a][f]);
      }
    }

    // ---- online softmax (lane = one query row, 16 of its 64 scores) ----
    const bool ragged = (kv0 + BKV > N);
#pragma unroll
    for (int a = 0; a < QF; ++a) {
      float tmax = -INFINITY;
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float z = s[a][f][r] * scale_log2e;
          if (ragged) {
            int kv;
            if constexpr (sizeof(T) == 2) kv = kv0 + 32 * (f >> 1) + 8 * lg + r + 4 * (f & 1);
            else kv = kv0 + 16 * f + 4 * lg + r;
            if (kv >= N) z = -INFINITY;
          }
          s[a][f][r] = z;
          tmax = fmaxf(tmax, z);
        }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float mnew = fmaxf(mrow[a], tmax);   // finite: key kv0 is always valid
      const float alpha = exp2f(mrow[a] - mnew);
      mrow[a] = mnew;
      float ps = 0.f;
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = exp2f(s[a][f][r] - mnew);
          s[a][f][r] = pv;
          ps += pv;
        }
      lrow[a] = lrow[a] * alpha + ps;
#pragma unroll
      for (int d = 0; d < Cfg::DF; ++d) o[a][d] *= alpha;
    }

    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int g = 0; g < Cfg::PG; ++g) {
      uint4 pb[QF];
#pragma unroll
      for (int a = 0; a < QF; ++a) {
        if constexpr (sizeof(T) == 2) {
          pb[a] = make_uint4(pack_bf16x2(s[a][2 * g][0], s[a][2 * g][1]), pack_bf16x2(s[a][2 * g][2], s[a][2 * g][3]),
                             pack_bf16x2(s[a][2 * g + 1][0], s[a][2 * g + 1][1]),
                             pack_bf16x2(s[a][2 * g + 1][2], s[a][2 * g + 1][3]));
        } else {
          const float p0 = s[a][g][0], p1 = s[a][g][1], p2 = s[a][g][2], p3 = s[a][g][3];
          pb[a] = make_uint4(f32_bits(p0), f32_bits(p1), f32_bits(p2), f32_bits(p3));
        }
      }
#pragma unroll
      for (int d = 0; d < Cfg::DF; ++d) {
        const uint4 vf = *(const uint4*)(Vt + (d * 16 + lq) * Cfg::VROW + (g * 4 + lg) * 16);
#pragma unroll
        for (int a = 0; a < QF; ++a) mma_kgroup<T>(vf, pb[a], o[a][d]);
      }
    }
  }

  // ---- normalise and store: lane (q, g) holds d = df*16 + 4g + r ----
#pragma unroll
  for (int a = 0; a < QF; ++a) {
    float l = lrow[a];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    const int q = q0 + a * 16 + lq;
    if (q >= N) continue;
    T* op = out + ((size_t)b * N + q) * C + (size_t)h * D;
#pragma unroll
    for (int d = 0; d < Cfg::DF; ++d) {
      const int dd = d * 16 + 4 * lg;
      if (dd >= D) continue;
      const f32x4 v = o[a][d] * inv;
      if constexpr (sizeof(T) == 2) {
        *(uint2*)(op + dd) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      } else {
        *(f32x4*)(op + dd) = v;
      }
    }
  }
}

template <typename T, int D, int QF>
int run(const void* qkv, void* out, int B, int N, int C, int heads, hipStream_t s) {
  using Cfg = AttnCfg<T, D>;
  const size_t lds = Cfg::KS_BYTES + Cfg::VT_BYTES;
  auto kern = attention_kernel<T, D, QF>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const int nqb = (N + 64 * QF - 1) / (64 * QF);
  const float scale_log2e = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  hipLaunchKernelGGL(kern, dim3(nqb * heads * B), dim3(256), lds, s, (const T*)qkv, (T*)out, N, C, heads, scale_log2e);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <typename T>
int dispatch(const void* qkv, void* out, int B, int N, int C, int heads, hipStream_t s) {
  const int d = C / heads;
  const bool big = N >= 256;
  switch (d) {
    case 40: return big ? run<T, 40, 2>(qkv, out, B, N, C, heads, s) : run<T, 40, 1>(qkv, out, B, N, C, heads, s);
    case 80: return big ? run<T, 80, 2>(qkv, out, B, N, C, heads, s) : run<T, 80, 1>(qkv, out, B, N, C, heads, s);
    case 160: return run<T, 160, 1>(qkv, out, B, N, C, heads, s);
    default: return -2;
  }
}

}  // namespace

int launch_attention(const void* qkv, void* out, int B, int N, int C, int heads, int dtype, hipStream_t s) {
  if (C % heads != 0 || N <= 0) return -2;
  return dtype == DT_BF16 ? dispatch<bf16_t>(qkv, out, B, N, C, heads, s) : dispatch<float>(qkv, out, B, N, C, heads, s);
}

}  // namespace ldmseg
