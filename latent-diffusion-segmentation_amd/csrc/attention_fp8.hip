// fp8 (OCP e4m3) self-attention for the long-context levels of the bf16 perf mode - BASELINE configs[4]: 1024x1024
// images, 128x128x4 latents, N = 16384 tokens at head dim 40 and N = 4096 at head dim 80.  Same structure as
// attention3.hip (LDS-DMA chunk planes, hardware-transposed V^T reads, row maximum folded into the QK^T product, row sum
// out of the PV product, v_max3 maxima) with Q, K, V and P in fp8 on v_mfma_f32_16x16x32_fp8_fp8 (fp32 accumulation):
//   * a pre-pass (kv_to_fp8_kernel) rewrites K and V of the fused bf16 qkv tensor as head-major fp8 rows padded to a
//     multiple of 16 bytes, [B, H, N, DP] with DP = 48 (d = 40) / 96 (d = 80).  The padding carries the constants the
//     folding tricks need as DATA: K[..., d] = K[..., d+1] = 1 (two columns, because the folded maximum must be exact
//     in the operand format and one e4m3 value is too coarse: -m = -(m_hi + m_lo), each part e4m3-exact), V[..., d] = 1
//     (row sum).  Values are clamped to +-448 first (the hardware conversion turns overflow into NaN, it does not
//     saturate); activations behind a LayerNorm'ed projection are O(1-10), so no per-tensor scale is needed - e4m3 is a
//     floating format, a scale would only move the range.
//   * per 64-key tile and operand 3 (6) planes of 1 KiB instead of 5 (10) in bf16: half the LDS-DMA instructions and
//     LDS bytes; K fragments are 8-byte reads, V^T fragments ONE ds_read_b64_tr_b8 per MFMA (two tr_b16 in bf16).
//     The LDS slot of K row kappa is kappa with bits 2 and 4 swapped (the DMA lane -> key map is free), which makes the
//     K fragment reads conflict-free while every lane ends up with 8 CONSECUTIVE keys - the contraction order the V^T
//     transpose-read delivers.
// Non-scaled fp8 MFMA runs at the bf16 rate, so this path does not raise the matrix throughput; what it buys is the
// memory side.  Accuracy: P and V carry 3 mantissa bits - see tests/test_ops_gpu.py::test_attention_fp8* for the bound.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace ldmseg {
namespace {

constexpr int BKV8 = 64;
typedef int i32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float clamp448(float x) { return fminf(fmaxf(x, -448.f), 448.f); }
// four floats -> one dword of e4m3 (RNE; inputs must be within +-448)
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (uint32_t)w;
}
__device__ __forceinline__ float fp8_to_f32(uint32_t byte) { return __builtin_amdgcn_cvt_f32_fp8((int)byte, 0); }

template <int OFF>
__device__ __forceinline__ void glds16_off8(const void* gsrc, unsigned lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off offset:%2"
      :
      : "v"(gsrc), "s"(lds_dst - OFF), "i"(OFF)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm8() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// ---- pre-pass: K and V of qkv [B, N, 3C] bf16 -> k8 / v8 [B, H, N, DP] e4m3 with the constant columns ----
template <int D, int DP>
__global__ __launch_bounds__(256) void kv_to_fp8_kernel(const bf16_t* __restrict__ qkv, unsigned char* __restrict__ k8,
                                                        unsigned char* __restrict__ v8, int B, int N, int C, int heads) {
  constexpr int CH = DP / 16;                              // 16-byte output chunks per row
  // grid = (token blocks, heads, B): thread -> (token, which, chunk) with compile-time divisors only (the flat-index form
  // of this loop spent four 64-bit runtime divisions per 16-byte chunk - more instructions than the conversion itself)
  const int h = blockIdx.y, b = blockIdx.z;
  constexpr int PER = 2 * CH;                              // work items per token
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N * PER; i += gridDim.x * blockDim.x) {
    const int tok = i / PER, rem = i - tok * PER;
    const int which = rem / CH, c = rem - which * CH;      // 0 = K, 1 = V
    const bf16_t* src = qkv + ((size_t)b * N + tok) * 3 * C + (size_t)(1 + which) * C + (size_t)h * D + c * 16;
    float f[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int d = c * 16 + e;
      float v = 0.f;
      if (d < D) v = clamp448(bf16_to_f32(src[e]));
      else if (d == D || (d == D + 1 && which == 0)) v = 1.0f;      // K: two ones columns (-m_hi, -m_lo); V: one (row sum)
      f[e] = v;
    }
    uint4 o = make_uint4(pack_fp8x4(f[0], f[1], f[2], f[3]), pack_fp8x4(f[4], f[5], f[6], f[7]),
                         pack_fp8x4(f[8], f[9], f[10], f[11]), pack_fp8x4(f[12], f[13], f[14], f[15]));
    unsigned char* dst = (which ? v8 : k8) + (((size_t)b * heads + h) * N + tok) * DP + c * 16;
    *(uint4*)dst = o;
  }
}

template <int D> struct A8Cfg {
  static constexpr int DP = (D + 2 + 15) / 16 * 16;  // padded row bytes: 48 (d = 40), 96 (d = 80)
  static constexpr int NPL = DP / 16;                // 16-byte chunk planes per operand
  static constexpr int KK = (D + 2 + 31) / 32;       // fp8 MFMAs (K = 32) covering the QK^T contraction
  static constexpr int DF = NPL;                     // 16-row fragments of O^T (V^T rows = one plane each)
  static constexpr int PS = 1024;                    // plane stride
  static constexpr int STAGE = 2 * NPL * PS;
  static constexpr int JMAX = (NPL + 3) / 4;
  static constexpr int MCOL = D;                     // columns D, D+1 of Q hold -m_hi, -m_lo
  static_assert(D % 8 == 0 && (D % 32) + 2 <= 32 && D % 32 != 0, "needs two spare contraction columns");
};

template <int D, int QF, int WPS, int NST>
__global__ __launch_bounds__(256, WPS) void attn_fp8_kernel(const bf16_t* __restrict__ qkv, const unsigned char* __restrict__ k8,
                                                            const unsigned char* __restrict__ v8, bf16_t* __restrict__ out,
                                                            int N, int C, int heads, float scale_log2e) {
  using Cfg = A8Cfg<D>;
  // e4m3 has 4 exponent bits: probabilities below 2^-9 vanish.  A softmax row over thousands of keys keeps a large part
  // of its mass in terms 2^-10 .. 2^-16 below its maximum (measured: 10 % output error at N = 4096 when they are lost), so
  // the folded reference is the row maximum MINUS PSHIFT: p = 2^(z - m + PSHIFT) uses the top of the e4m3 range (<= 448)
  // and reaches 2^-16 below the maximum; the row sum carries the same factor and the normalisation cancels it.  The
  // reference moves when a row's tile maximum exceeds it by more than PSHIFT + THR (p <= 2^8.25 = 304 plus the rounding
  // slack of the two-part folded value stays under 448; the hardware conversion does not saturate).
  constexpr float PSHIFT = 7.0f, THR = PSHIFT + 1.25f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 15, lg = lane >> 4;
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
  const bf16_t* qbase = qkv + (size_t)b * N * ld + (size_t)h * D;
  const unsigned char* kbase = k8 + (size_t)bh * N * Cfg::DP;
  const unsigned char* vbase = v8 + (size_t)bh * N * Cfg::DP;

  // ---- DMA stream: K plane c = wave + 4j, V plane c = ((wave + 2) & 3) + 4j.  LDS slot = lane; the K lane loads key
  // swap(lane bits 2, 4), the V lane key = lane.
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
  const int wv = (wave + 2) & 3;
  const int kkey = (lane & ~0x14) | ((lane & 4) << 2) | ((lane & 16) >> 2);
  int my_cnt = 0;
#pragma unroll
  for (int j = 0; j < Cfg::JMAX; ++j) my_cnt += (wave + 4 * j < Cfg::NPL) + (wv + 4 * j < Cfg::NPL);
  my_cnt = __builtin_amdgcn_readfirstlane(my_cnt);
  auto issue_tile = [&](int t, int stage) __attribute__((always_inline)) {
    const unsigned char* kp = kbase + (size_t)min(t * BKV8 + kkey, N - 1) * Cfg::DP + wave * 16;   // keys past N: clamped, masked below
    const unsigned char* vp = vbase + (size_t)min(t * BKV8 + lane, N - 1) * Cfg::DP + wv * 16;
    const unsigned kdst = __builtin_amdgcn_readfirstlane(lds0 + stage * Cfg::STAGE + wave * Cfg::PS);
    const unsigned vdst = __builtin_amdgcn_readfirstlane(lds0 + stage * Cfg::STAGE + Cfg::NPL * Cfg::PS + wv * Cfg::PS);
    static_for_n<Cfg::JMAX>([&](auto jc) __attribute__((always_inline)) {
      constexpr int J = decltype(jc)::value;
      if (wave + 4 * J < Cfg::NPL) glds16_off8<J * 64>(kp, kdst + J * 4 * Cfg::PS);
      if (wv + 4 * J < Cfg::NPL) glds16_off8<J * 64>(vp, vdst + J * 4 * Cfg::PS);
    });
  };
  auto wait_tiles_ahead = [&](int ahead) __attribute__((always_inline)) {
    if (ahead == 0) { wait_vm8<0>(); return; }
    if (my_cnt == 1) wait_vm8<1>();
    else if (my_cnt == 2) wait_vm8<2>();
    else if (my_cnt == 3) wait_vm8<3>();
    else wait_vm8<4>();
  };
  const int ntiles = (N + BKV8 - 1) / BKV8;
  const int nfull = N / BKV8;
  static_for_n<NST - 1>([&](auto jc) __attribute__((always_inline)) {
    constexpr int J = decltype(jc)::value;
    if (J < ntiles) issue_tile(J, J);
  });

  // ---- Q fragments (MFMA B operand): lane (q = lq, g = lg) holds columns kk*32 + 8g .. +8 of its row as e4m3,
  // pre-scaled by d^-1/2 log2 e; columns D, D+1 (the folded maximum) start at zero
  const int q0 = qb * 64 * QF + wave * 16 * QF;
  long qf[QF][Cfg::KK];
#pragma unroll
  for (int a = 0; a < QF; ++a) {
    const int q = q0 + a * 16 + lq;
#pragma unroll
    for (int kk = 0; kk < Cfg::KK; ++kk) {
      const int c0 = kk * 32 + lg * 8;
      float qv[8];
      if (q < N && c0 < D) {
        Chunk<bf16_t>::unpack(*(const uint4*)((const unsigned char*)(qbase + (size_t)q * ld) + c0 * 2), qv);
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] = (c0 + e < D) ? clamp448(qv[e] * scale_log2e) : 0.f;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] = 0.f;
      }
      const uint32_t lo = pack_fp8x4(qv[0], qv[1], qv[2], qv[3]), hi = pack_fp8x4(qv[4], qv[5], qv[6], qv[7]);
      qf[a][kk] = (long)(((unsigned long long)hi << 32) | lo);
    }
  }
  constexpr int MKK = Cfg::MCOL / 32, MLG = (Cfg::MCOL % 32) / 8, MB = Cfg::MCOL % 8;   // where columns D, D+1 live
  static_assert(MB + 1 < 8, "both maximum columns must sit in one 8-byte operand half");
  const bool holds_m = (lg == MLG);

  f32x4 o[QF][Cfg::DF];
  float mrow[QF];
#pragma unroll
  for (int a = 0; a < QF; ++a) {
    mrow[a] = 0.f;
#pragma unroll
    for (int d = 0; d < Cfg::DF; ++d) o[a][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // K fragment f, MFMA kk: slot 16f + lq of plane min(2kk + (lg >> 1), NPL-1), byte half lg & 1  (planes past the padded
  // row would be zero columns: a real plane is read instead and meets zero columns of Q; its bytes are finite e4m3)
  int koff[Cfg::KK];
  const int lqs = (lq & 3) | ((lq & 4) << 1) | ((lq & 8) >> 1);     // fragment row lq sits in slot 16f + (lq with bits 2, 3 swapped)
#pragma unroll
  for (int kk = 0; kk < Cfg::KK; ++kk) koff[kk] = min(2 * kk + (lg >> 1), Cfg::NPL - 1) * Cfg::PS + lqs * 16 + (lg & 1) * 8;
  // V^T fragment df (= plane df) through ds_read_b64_tr_b8: lane i of a 16-lane group points at key row (i >> 1), byte half
  // (i & 1) of an [8 keys][16 columns] block; group g takes keys 8g .. 8g+7 of each 32-key block
  const int voff = Cfg::NPL * Cfg::PS + (8 * lg + (lq >> 1)) * 16 + (lq & 1) * 8;

  wait_tiles_ahead((NST > 2 && ntiles > 1) ? 1 : 0);
  __syncthreads();

  f32x4 s[QF][4];
  auto tile = [&](int t, auto stage_c, auto ragged_c) __attribute__((always_inline)) {
    constexpr int ST = decltype(stage_c)::value;
    constexpr bool RAGGED = decltype(ragged_c)::value;
    if (t + NST - 1 < ntiles) issue_tile(t + NST - 1, (ST + NST - 1) % NST);
    const unsigned char* sb = smem + ST * Cfg::STAGE;
    // ---- S^T = K Q^T (relative to the folded maxima); fragment f row i is key 32(f>>1) + 8(i>>2) + 4(f&1) + (i&3) - so that
    // lane group g ends up with the 8 consecutive keys 32hh + 8g .. of every 32-key block - and that key sits in slot
    // (key with bits 2 and 4 swapped) = 16f + (i with bits 2 and 3 swapped): fragment f = slots 16f .. 16f+15, conflict-free
#pragma unroll
    for (int a = 0; a < QF; ++a)
#pragma unroll
      for (int f = 0; f < 4; ++f) s[a][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < Cfg::KK; ++kk) {
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const long kf = *(const long*)(sb + koff[kk] + f * 256);
#pragma unroll
        for (int a = 0; a < QF; ++a) s[a][f] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(kf, qf[a][kk], s[a][f], 0, 0, 0);
      }
    }
    if constexpr (RAGGED) {
#pragma unroll
      for (int a = 0; a < QF; ++a)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (t * BKV8 + 32 * (f >> 1) + 8 * lg + 4 * (f & 1) + r >= N) s[a][f][r] = -INFINITY;
    }
    float tmax[QF];
    bool need_any = (t == 0);
#pragma unroll
    for (int a = 0; a < QF; ++a) {
      float m0 = __builtin_fmaxf(__builtin_fmaxf(s[a][0][0], s[a][0][1]), s[a][0][2]);
      float m1 = __builtin_fmaxf(__builtin_fmaxf(s[a][0][3], s[a][1][0]), s[a][1][1]);
      float m2 = __builtin_fmaxf(__builtin_fmaxf(s[a][1][2], s[a][1][3]), s[a][2][0]);
      float m3 = __builtin_fmaxf(__builtin_fmaxf(s[a][2][1], s[a][2][2]), s[a][2][3]);
      m0 = __builtin_fmaxf(__builtin_fmaxf(m0, s[a][3][0]), s[a][3][1]);
      m1 = __builtin_fmaxf(__builtin_fmaxf(m1, s[a][3][2]), s[a][3][3]);
      m0 = __builtin_fmaxf(__builtin_fmaxf(m0, m1), m2);
      m0 = fmaxf(m0, m3);
      m0 = xor32_max(xor16_max(m0));
      tmax[a] = m0;
      need_any |= tmax[a] > THR;
    }
    if (__any(need_any)) {
      // move the folded maximum of the rows that outgrew it (see attention3.hip); here the new value is the sum of two
      // e4m3 numbers (RNE of the target, RNE of the remainder), which is what columns D, D+1 of Q can hold exactly
      float one;
      asm volatile("v_mov_b32 %0, 1.0" : "=v"(one));
#pragma unroll
      for (int a = 0; a < QF; ++a) {
        const bool need = (t == 0) | (tmax[a] > THR);
        const float target = clamp448(mrow[a] + tmax[a] - PSHIFT);
        const uint32_t b_hi = pack_fp8x4(target, 0.f, 0.f, 0.f) & 0xffu;
        const float m_hi = fp8_to_f32(b_hi);
        const uint32_t b_lo = pack_fp8x4(target - m_hi, 0.f, 0.f, 0.f) & 0xffu;
        const float mnew = need ? (m_hi + fp8_to_f32(b_lo)) : mrow[a];
        const float delta = (mnew - mrow[a]) * one;
        mrow[a] += delta;
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[a][f][r] -= delta;
        if (t > 0) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
          for (int d = 0; d < Cfg::DF; ++d) o[a][d] *= alpha;
        }
        if (holds_m && need) {      // bytes MB, MB+1 of this lane's operand half <- e4m3(-m_hi), e4m3(-m_lo) (sign bit flipped)
          unsigned long long w = (unsigned long long)qf[a][MKK];
          const unsigned long long neg = ((unsigned long long)((b_hi ^ 0x80u) | ((b_lo ^ 0x80u) << 8))) << (8 * MB);
          w = (w & ~(0xffffull << (8 * MB))) | neg;
          qf[a][MKK] = (long)w;
        }
      }
    }
#pragma unroll
    for (int a = 0; a < QF; ++a)
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[a][f][r] = __builtin_amdgcn_exp2f(s[a][f][r]);
    // ---- O^T += V^T P^T: 32-key group hh = score fragments 2hh, 2hh+1; the lane's 8 probabilities are 8 consecutive keys ----
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      long pb[QF];
#pragma unroll
      for (int a = 0; a < QF; ++a) {
        const uint32_t lo = pack_fp8x4(s[a][2 * hh][0], s[a][2 * hh][1], s[a][2 * hh][2], s[a][2 * hh][3]);
        const uint32_t hi = pack_fp8x4(s[a][2 * hh + 1][0], s[a][2 * hh + 1][1], s[a][2 * hh + 1][2], s[a][2 * hh + 1][3]);
        pb[a] = (long)(((unsigned long long)hi << 32) | lo);
      }
#pragma unroll
      for (int d = 0; d < Cfg::DF; ++d) {
        typedef __attribute__((address_space(3))) i32x2* lds_v2;
        const i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2)(sb + voff + d * Cfg::PS + (32 * hh) * 16));
        const long vf = (long)(((unsigned long long)(unsigned)v[1] << 32) | (unsigned)v[0]);
#pragma unroll
        for (int a = 0; a < QF; ++a) o[a][d] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(vf, pb[a], o[a][d], 0, 0, 0);
      }
    }
    if (t + 1 < ntiles) wait_tiles_ahead((NST > 2 && t + 2 < ntiles) ? 1 : 0);
    __syncthreads();
  };
  for (int t0 = 0; t0 < ntiles; t0 += NST) {
    static_for_n<NST>([&](auto sc) __attribute__((always_inline)) {
      const int t = t0 + decltype(sc)::value;
      if (t < nfull) tile(t, sc, std::false_type{});
      else if (t < ntiles) tile(t, sc, std::true_type{});
    });
  }

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

template <int D, int QF, int WPS, int NST>
int run8(const void* qkv, void* kv8, void* out, int B, int N, int C, int heads, hipStream_t s) {
  using Cfg = A8Cfg<D>;
  unsigned char* k8 = (unsigned char*)kv8;
  unsigned char* v8 = k8 + (size_t)B * heads * N * Cfg::DP;
  {
    const long per_bh = (long)N * (Cfg::DP / 16) * 2;
    int bx = (int)((per_bh + 255) / 256);
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL((kv_to_fp8_kernel<D, Cfg::DP>), dim3(bx, heads, B), dim3(256), 0, s, (const bf16_t*)qkv, k8, v8, B, N, C, heads);
  }
  const size_t lds = (size_t)NST * Cfg::STAGE;
  auto kern = attn_fp8_kernel<D, QF, WPS, NST>;
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set[dev] = true;
  }
  const int nqb = (N + 64 * QF - 1) / (64 * QF);
  const float scale_log2e = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  hipLaunchKernelGGL(kern, dim3(nqb * heads * B), dim3(256), lds, s, (const bf16_t*)qkv, k8, v8, (bf16_t*)out, N, C, heads,
                     scale_log2e);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

// bytes of the fp8 K/V scratch launch_attention_fp8 needs (0 = head dim not supported by the fp8 path)
size_t attention_fp8_scratch_bytes(int B, int N, int C, int heads) {
  const int d = C / heads;
  if (const size_t mx = attention_mx_scratch_bytes(B, N, C, heads)) return mx;       // head dim 40, whole 128-key tiles: the scaled-MFMA path
  if (d == 40) return (size_t)2 * B * heads * N * A8Cfg<40>::DP;
  if (d == 80) return (size_t)2 * B * heads * N * A8Cfg<80>::DP;
  return 0;
}

// qkv bf16 [B, N, 3C] -> out bf16 [B, N, C] with fp8 (e4m3) Q/K/V/P operands; kv8 = scratch of attention_fp8_scratch_bytes
int launch_attention_fp8(const void* qkv, void* kv8, void* out, int B, int N, int C, int heads, hipStream_t s) {
  const int d = C / heads;
  if (attention_mx_ok(N, C, heads)) return launch_attention_mx(qkv, kv8, out, B, N, C, heads, s);   // 2x-rate block-scaled MFMAs (attention_mx.hip)
  if (d == 40) return run8<40, 2, 4, 3>(qkv, kv8, out, B, N, C, heads, s);
  if (d == 80) return run8<80, 2, 2, 3>(qkv, kv8, out, B, N, C, heads, s);
  return -2;
}

}  // namespace ldmseg
