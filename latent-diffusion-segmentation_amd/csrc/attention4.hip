// bf16 self-attention, head dim 40, fourth generation: the structure of attention_mx.hip (32x32 score blocks in which a lane
// owns ONE query, probabilities moved into the PV operand layout by v_permlane16_swap) on the bf16 MFMAs.
//   * S^T = K Q^T on v_mfma_f32_32x32x16_bf16: the contraction of d = 40 (+1 column for the folded maximum) takes THREE
//     K = 16 steps (48 columns) instead of the K = 64 of two 16x16x32 k-groups: 192 instead of 256 matrix cycles per 64 keys x 32
//     queries (remove padding, not instructions).  PV stays on v_mfma_f32_16x16x32_bf16 (192 cycles).
//   * The 32x32 result gives lane (q = lane & 31, h = lane >> 5) sixteen keys of one query per block; as bf16 they are 2 x 4
//     dwords (U, W), and four v_permlane16_swap per 32-key block turn them into the P^T operands of the two 16-query blocks
//     (k-group g = 16-lane row).  The contraction slot -> key permutation (slot (g, j) = key 16 (g & 1) + 4 (g >> 1) + (j & 3) +
//     8 (j >> 2)) is absorbed by the addresses of the hardware-transposed V^T reads (ds_read_b64_tr_b16 takes per-lane rows).
// Everything else is attention3.hip's: chunk planes by LDS-DMA, the row maximum folded into the product through a ones plane
// of K, lazily tracked maxima with the exact redo pass, row sums through a ones plane of V.  Same arithmetic contract.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace ldmseg {
namespace {

constexpr int BKV4 = 64;
constexpr int D4 = 40, DCH4 = 5, NPL4 = 6, DF4 = 3;
constexpr int KPS4 = 1024, VPS4 = 1152;
constexpr int KBYTES4 = NPL4 * KPS4, VBYTES4 = NPL4 * VPS4, STAGE4 = KBYTES4 + VBYTES4;
typedef short s16x4b __attribute__((ext_vector_type(4)));
typedef float v16f4 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8b __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf16_ceil4(float x) {
  const uint32_t u = f32_bits(x);
  return bits_f32(((u & 0x80000000u) ? u : u + 0xffffu) & 0xffff0000u);
}
template <int OFF>
__device__ __forceinline__ void glds16_off4(const void* gsrc, unsigned lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off offset:%2"
      :
      : "v"(gsrc), "s"(lds_dst - OFF), "i"(OFF)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm4() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

template <int NST, int LAZY, int NWV>
__global__ __launch_bounds__(64 * NWV, NWV == 8 ? 4 : 3) void attn4_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int N, int C, int heads,
                                                            float scale_log2e) {
  constexpr float THR = 6.0f;
  constexpr int VSH = (NWV - DCH4 % NWV) % NWV, JMAX = (DCH4 + NWV - 1) / NWV;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, hh = lane >> 5;                 // score block: query column, key half
  const int lq = lane & 15, lg = lane >> 4;                 // PV operands / O^T: query within a 16-block, k-group

  const int nqb = (N + 32 * NWV - 1) / (32 * NWV);          // 32 queries per wave
  int wg;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int qb = wg % nqb, bh = wg / nqb;
  const int h = bh % heads, b = bh / heads;
  const size_t ld = (size_t)3 * C;
  const bf16_t* qbase = qkv + (size_t)b * N * ld + (size_t)h * D4;
  const unsigned char* kbase = (const unsigned char*)(qbase + C);
  const size_t ldb = ld * sizeof(bf16_t);
  const int voff_bytes = C * (int)sizeof(bf16_t);

  // ---- constant planes: chunk 5 of every key is [1, 0, 0, 0, 0, 0, 0, 0] in all stages, for K (the -m column) and V (row sum)
  for (int i = tid; i < NST * 2 * BKV4; i += 64 * NWV) {
    const int st = i / (2 * BKV4), rem = i - st * (2 * BKV4), which = rem / BKV4, key = rem - which * BKV4;
    unsigned char* p = smem + st * STAGE4 + (which ? KBYTES4 + DCH4 * VPS4 : DCH4 * KPS4) + key * 16;
    *(uint4*)p = make_uint4(0x3f80u, 0u, 0u, 0u);
  }

  // ---- DMA stream (attention3.hip): K plane c = wave + NWV j, V plane c = ((wave + VSH) % NWV) + NWV j; lane = key of the tile
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
  const int wv = (wave + VSH) % NWV;
  int my_cnt = 0;
#pragma unroll
  for (int j = 0; j < JMAX; ++j) my_cnt += (wave + NWV * j < DCH4) + (wv + NWV * j < DCH4);
  my_cnt = __builtin_amdgcn_readfirstlane(my_cnt);
  auto issue_tile = [&](int t, int stage) __attribute__((always_inline)) {
    const int row = min(t * BKV4 + lane, N - 1);
    const unsigned char* rp = kbase + (size_t)row * ldb;
    const unsigned char* kp = rp + wave * 16;
    const unsigned char* vp = rp + voff_bytes + wv * 16;
    const unsigned kdst = __builtin_amdgcn_readfirstlane(lds0 + stage * STAGE4 + wave * KPS4);
    const unsigned vdst = __builtin_amdgcn_readfirstlane(lds0 + stage * STAGE4 + KBYTES4 + wv * VPS4);
    static_for_n<JMAX>([&](auto jc) __attribute__((always_inline)) {
      constexpr int J = decltype(jc)::value;
      if (wave + NWV * J < DCH4) glds16_off4<J * NWV * 16>(kp, kdst + J * NWV * KPS4);
      if (wv + NWV * J < DCH4) glds16_off4<J * NWV * 16>(vp, vdst + J * NWV * VPS4);
    });
  };
  auto wait_tiles_ahead = [&](int ahead) __attribute__((always_inline)) {
    if (ahead == 0) { wait_vm4<0>(); return; }
    if (my_cnt == 0) return;
    if (my_cnt == 1) wait_vm4<1>();
    else if (my_cnt == 2) wait_vm4<2>();
    else if (my_cnt == 3) wait_vm4<3>();
    else wait_vm4<4>();
  };
  const int ntiles = (N + BKV4 - 1) / BKV4;
  const int nfull = N / BKV4;
  static_for_n<NST - 1>([&](auto jc) __attribute__((always_inline)) {
    constexpr int J = decltype(jc)::value;
    if (J < ntiles) issue_tile(J, J);
  });

  // ---- Q operand (B of the 32x32x16 product): lane (q = ql, kg = hh) holds chunk 2 ks + kg of its row for ks = 0..2, scaled by
  // d^-1/2 log2 e; chunk 5 = [-m, 0, ...] lives in the kg = 1 lanes' third fragment
  const int q0 = qb * 32 * NWV + wave * 32;
  uint4 qf[3];
  {
    const int q = q0 + ql;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const int ch = 2 * ks + hh;
      const uint4 raw = (q < N && ch < DCH4) ? *(const uint4*)((const unsigned char*)(qbase + (size_t)q * ld) + ch * 16) : make_uint4(0, 0, 0, 0);
      float qv[8];
      Chunk<bf16_t>::unpack(raw, qv);
#pragma unroll
      for (int e = 0; e < 8; ++e) qv[e] *= scale_log2e;
      qf[ks] = Chunk<bf16_t>::pack(qv);
    }
  }
  f32x4 o[2][DF4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int d = 0; d < DF4; ++d) o[a][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float mrow = 0.f;

  // K fragment (A operand): key = 32 kb + ql, plane 2 ks + hh
  int koff[3];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) koff[ks] = (2 * ks + hh) * KPS4 + ql * 16;
  // V^T fragment of d-block d through ds_read_b64_tr_b16: lane i of 16-lane group g points at key row base(g) + (i >> 2) of the
  // 32-key block (and 8 keys further for the second read), 8 bytes (i & 1) of chunk plane 2 d + ((i & 3) >> 1)
  int voff[DF4];
  const int kbase_g = 16 * (lg & 1) + 4 * (lg >> 1);
#pragma unroll
  for (int d = 0; d < DF4; ++d)
    voff[d] = KBYTES4 + min(2 * d + ((lq & 3) >> 1), DCH4) * VPS4 + (kbase_g + (lq >> 2)) * 16 + (lq & 1) * 8;

  auto tile = [&](int t, auto stage_c, auto ragged_c, auto check_c) __attribute__((always_inline)) {
    constexpr int ST = decltype(stage_c)::value;
    constexpr bool RAGGED = decltype(ragged_c)::value;
    constexpr bool CHECK = decltype(check_c)::value;
    if (t + NST - 1 < ntiles) issue_tile(t + NST - 1, (ST + NST - 1) % NST);
    const unsigned char* sb = smem + ST * STAGE4;
    // S^T = K Q^T of one 32-key block: three K = 16 steps
    auto qk = [&](int kb) __attribute__((always_inline)) {
      v16f4 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const uint4 kf = *(const uint4*)(sb + koff[ks] + kb * 512);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8b, kf), __builtin_bit_cast(bf16x8b, qf[ks]), acc, 0, 0, 0);
      }
      return acc;
    };
    // p = 2^s as bf16: U = values 0..7, W = values 8..15 of the lane; after the exchange U / W are the P^T operands of query
    // blocks 0 / 1 (k-group g = 16-lane row: [U, W of lane q'; U, W of lane q' + 32]); then O^T += V^T P^T for the block
    auto softmax_pv = [&](const v16f4& sc, int kb) __attribute__((always_inline)) {
      float pv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) pv[r] = __builtin_amdgcn_exp2f(sc[r]);
      unsigned U[4], W[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        U[i] = pack_bf16x2(pv[2 * i], pv[2 * i + 1]);
        W[i] = pack_bf16x2(pv[8 + 2 * i], pv[8 + 2 * i + 1]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const auto r = __builtin_amdgcn_permlane16_swap(U[i], W[i], false, false);
        U[i] = r[0];
        W[i] = r[1];
      }
      const uint4 pu = make_uint4(U[0], U[1], U[2], U[3]), pw = make_uint4(W[0], W[1], W[2], W[3]);
#pragma unroll
      for (int d = 0; d < DF4; ++d) {
        typedef __attribute__((address_space(3))) s16x4b* lds_v4;
        const s16x4b v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sb + voff[d] + (32 * kb) * 16));
        const s16x4b v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sb + voff[d] + (32 * kb + 8) * 16));
        const uint2 a0 = __builtin_bit_cast(uint2, v0), a1 = __builtin_bit_cast(uint2, v1);
        const uint4 vf = make_uint4(a0.x, a0.y, a1.x, a1.y);
        mma_kgroup<bf16_t>(vf, pu, o[0][d]);
        mma_kgroup<bf16_t>(vf, pw, o[1][d]);
      }
    };
    if constexpr (!CHECK) {
      // 15 tiles of 16: nothing couples the two 32-key blocks, one score block is live at a time
      const v16f4 s0 = qk(0);
      const v16f4 s1 = qk(1);
      softmax_pv(s0, 0);
      softmax_pv(s1, 1);
    } else {
      v16f4 s[2];
      s[0] = qk(0);
      s[1] = qk(1);
      if constexpr (RAGGED) {
        const int nrem = N - t * BKV4 - 4 * hh;                    // keys of this tile the lane's first slot still has
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (32 * kb + (r & 3) + 8 * (r >> 2) >= nrem) s[kb][r] = -INFINITY;
      }
      float m0 = s[0][0], m1 = s[1][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) {
        m0 = __builtin_fmaxf(m0, s[0][r]);
        m1 = __builtin_fmaxf(m1, s[1][r]);
      }
      const float tm = xor32_max(__builtin_fmaxf(m0, m1));        // the query's other 32 keys sit in lane ^ 32
      const bool need = (t == 0) | (tm > THR);
      if (__any(need)) {
        float one;
        asm volatile("v_mov_b32 %0, 1.0" : "=v"(one));
        const float mnew = need ? bf16_ceil4(mrow + tm) : mrow;
        const float delta = (mnew - mrow) * one;                   // exact: both are bf16 values
        mrow += delta;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kb][r] -= delta;
        if (t > 0) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const float al = __shfl(alpha, lq + 16 * a, 64);       // O^T of query block a sits in lanes (q' = lq, any g)
#pragma unroll
            for (int d = 0; d < DF4; ++d) o[a][d] *= al;
          }
        }
        if (hh == 1) qf[2].x = f32_bits(-mrow) >> 16;             // column 40 of Q: bf16(-m); 41.. stay zero
      }
      softmax_pv(s[0], 0);
      softmax_pv(s[1], 1);
    }
    if (t + 1 < ntiles) wait_tiles_ahead((NST > 2 && t + 2 < ntiles) ? 1 : 0);
    __syncthreads();
  };
  auto all_tiles = [&](auto period_c) __attribute__((always_inline)) {
    constexpr int PERIOD = decltype(period_c)::value;
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
    // a row whose scores outgrew the folded maximum between two looks shows up as a row sum >= 2^100 (attention3.hip): redo exactly
    bool bad = false;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const unsigned lb = f32_bits(o[a][2][0]);                   // row 40 = d-block 2, g = 2, r = 0
      bad |= (lg == 2) && ((lb & 0x7f800000u) >= ((127u + 100u) << 23));
    }
    if (__syncthreads_or(bad ? 1 : 0)) {
      mrow = 0.f;
      if (hh == 1) qf[2].x &= 0xffff0000u;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int d = 0; d < DF4; ++d) o[a][d] = f32x4{0.f, 0.f, 0.f, 0.f};
      static_for_n<NST - 1>([&](auto jc) __attribute__((always_inline)) {
        constexpr int J = decltype(jc)::value;
        if (J < ntiles) issue_tile(J, J);
      });
      all_tiles(std::integral_constant<int, 1>{});
    }
  }

  // ---- normalise and store: lane (q' = lq, g) holds rows d = 16 db + 4 g + r of query 16 a + q'; the row sum is row 40
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const float l = __shfl(o[a][2][0], lq + 32, 64);
    const float inv = 1.0f / l;
    const int q = q0 + 16 * a + lq;
    if (q >= N) continue;
    bf16_t* op = out + ((size_t)b * N + q) * C + (size_t)h * D4;
#pragma unroll
    for (int d = 0; d < DF4; ++d) {
      const int dd = d * 16 + 4 * lg;
      if (dd >= D4) continue;
      const f32x4 v = o[a][d] * inv;
      *(uint2*)(op + dd) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    }
  }
}

template <int NST, int LAZY, int NWV>
int run4(const void* qkv, void* out, int B, int N, int C, int heads, hipStream_t s) {
  const size_t lds = (size_t)NST * STAGE4;
  auto kern = attn4_kernel<NST, LAZY, NWV>;
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set[dev] = true;
  }
  const int nqb = (N + 32 * NWV - 1) / (32 * NWV);
  const float scale_log2e = (1.0f / sqrtf((float)D4)) * 1.4426950408889634f;
  hipLaunchKernelGGL(kern, dim3(nqb * heads * B), dim3(64 * NWV), lds, s, (const bf16_t*)qkv, (bf16_t*)out, N, C, heads, scale_log2e);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

// bf16, head dim 40 only.  variant 0: 8-wave workgroups (256 queries share every K / V tile), maxima looked at on tile 0 and every 16th;
// 1: 8 waves, maxima on every tile; 2 / 3: the same two on 4-wave workgroups (short sequences).  -100: shape not handled here.
// Two ring stages: with three the 8-wave form needs more than the 128 registers four waves per SIMD leave (17-37 spilled, 1.6x
// slower); 2 stages measured as fast as 3 / 4 on the 4-wave form.
int launch_attention4(const void* qkv, void* out, int B, int N, int C, int heads, int variant, hipStream_t s) {
  if (heads <= 0 || C % heads || C / heads != D4) return -100;
  if (variant == 1) return run4<2, 1, 8>(qkv, out, B, N, C, heads, s);
  if (variant == 2) return run4<2, 16, 4>(qkv, out, B, N, C, heads, s);
  if (variant == 3) return run4<2, 1, 4>(qkv, out, B, N, C, heads, s);
  return run4<2, 16, 8>(qkv, out, B, N, C, heads, s);
}

}  // namespace ldmseg
