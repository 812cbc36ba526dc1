// Shared device/host helpers for the LDMSeg gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "kernels.h"

namespace ldmseg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // native 16-B register chunk (HIP's uint4 struct can end up in scratch)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short bf16_t;  // raw bf16 bits

// One LDS/global "chunk" is 16 bytes: 8 bf16 or 4 f32.
template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kPerChunk = 4;
  static constexpr int kDType = DT_F32;
};
template <> struct Elem<bf16_t> {
  static constexpr int kPerChunk = 8;
  static constexpr int kDType = DT_BF16;
};

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __builtin_bit_cast(float, (uint32_t)v << 16);
}
// fp32 -> bf16 through the gfx950 hardware convert (v_cvt_pk_bf16_f32, round-to-nearest-even)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// NOTE: never apply __builtin_bit_cast directly to an ext_vector element (v[i]): hipcc 7.2 then
// reads element 0.  Go through these by-value helpers instead.
// n / d for 0 <= n < 2^31 with a divisor prepared by fastdiv_make() (kernels.h)
template <typename FD>
__device__ __forceinline__ int fd_div(int n, const FD& f) {
  return f.mul ? (int)(__umulhi((unsigned)n, f.mul) >> f.shift) : n;
}
__device__ __forceinline__ uint32_t f32_bits(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float bits_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf16_to_f32(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f32_to_bf16(v); }

// Unpack a 16-byte chunk into Elem<T>::kPerChunk floats and back.
template <typename T> struct Chunk;
template <> struct Chunk<float> {
  static constexpr int N = 4;
  __device__ __forceinline__ static void unpack(const uint4& c, float* f) {
    f[0] = __builtin_bit_cast(float, c.x); f[1] = __builtin_bit_cast(float, c.y);
    f[2] = __builtin_bit_cast(float, c.z); f[3] = __builtin_bit_cast(float, c.w);
  }
  __device__ __forceinline__ static uint4 pack(const float* f) {
    return make_uint4(__builtin_bit_cast(uint32_t, f[0]), __builtin_bit_cast(uint32_t, f[1]),
                      __builtin_bit_cast(uint32_t, f[2]), __builtin_bit_cast(uint32_t, f[3]));
  }
};
template <> struct Chunk<bf16_t> {
  static constexpr int N = 8;
  __device__ __forceinline__ static void unpack(const uint4& c, float* f) {
    f[0] = __builtin_bit_cast(float, c.x << 16); f[1] = __builtin_bit_cast(float, c.x & 0xffff0000u);
    f[2] = __builtin_bit_cast(float, c.y << 16); f[3] = __builtin_bit_cast(float, c.y & 0xffff0000u);
    f[4] = __builtin_bit_cast(float, c.z << 16); f[5] = __builtin_bit_cast(float, c.z & 0xffff0000u);
    f[6] = __builtin_bit_cast(float, c.w << 16); f[7] = __builtin_bit_cast(float, c.w & 0xffff0000u);
  }
  __device__ __forceinline__ static uint4 pack(const float* f) {
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]),
                      pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
  }
};

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{})
template <int N, typename F>
__device__ __forceinline__ void static_for_n(F&& f) {
  if constexpr (N > 0) {
    static_for_n<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// ---- cross-lane reductions on the VALU (DPP + v_permlane{16,32}_swap), no LDS round trips ----
__device__ __forceinline__ void swap16(float x, float& a, float& b) {
  const unsigned u = f32_bits(x);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  a = bits_f32(r0);
  b = bits_f32(r1);
}
__device__ __forceinline__ void swap32(float x, float& a, float& b) {
  const unsigned u = f32_bits(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  a = bits_f32(r0);
  b = bits_f32(r1);
}
__device__ __forceinline__ float xor16_max(float x) { float a, b; swap16(x, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float xor32_max(float x) { float a, b; swap32(x, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float xor16_sum(float x) { float a, b; swap16(x, a, b); return a + b; }
__device__ __forceinline__ float xor32_sum(float x) { float a, b; swap32(x, a, b); return a + b; }
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
  return bits_f32((unsigned)__builtin_amdgcn_update_dpp(0, (int)f32_bits(x), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);
  v += dpp_f<0x140>(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  return v;
}
__device__ __forceinline__ float wave64_max(float v) { return xor32_max(xor16_max(row16_max(v))); }
// all-lanes sum of a wave64, result in every lane
__device__ __forceinline__ float wave64_sum(float v) { return xor32_sum(xor16_sum(row16_sum(v))); }

// k-th storage element of a 16-byte chunk (k must fold to a constant after unrolling); avoids
// union/array punning, which hipcc leaves in scratch memory
template <typename T> __device__ __forceinline__ T chunk_elem(const u32x4 c, int k);
template <> __device__ __forceinline__ float chunk_elem<float>(const u32x4 c, int k) {
  const uint32_t w = c[k];
  return bits_f32(w);
}
template <> __device__ __forceinline__ bf16_t chunk_elem<bf16_t>(const u32x4 c, int k) {
  const uint32_t w = c[k >> 1];
  return (bf16_t)((k & 1) ? (w >> 16) : (w & 0xffffu));
}

// x * sigmoid(x) with v_exp_f32 + v_rcp_f32 (1 ulp each, far inside the 1e-3 parity budget): the IEEE
// division expands to ~10 instructions per element in every epilogue that carries an optional SiLU.
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// erf-GELU (torch.nn.functional.gelu default).  erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7,
// far inside the 1e-3 parity budget): one v_rcp + one v_exp + a 5-term Horner instead of libm's erff.
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  float poly = 1.061405429f;
  poly = poly * t - 1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t - 0.284496736f;
  poly = poly * t + 0.254829592f;
  const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
  const float erf_abs = 1.0f - poly * t * e;
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// erf-GELU for results that are rounded to bf16 anyway: erf(x / sqrt 2) ~ x * R(x^2) on |x| <= 3 sqrt 2 (degree-8 weighted
// least-squares fit in x^2, saturated beyond), |gelu error| <= 5e-5 absolute, < 6e-6 relative for large |x| - two orders
// below a bf16 ulp of the output.  No v_rcp / v_exp (quarter rate) and the Horner chain runs as v_pk_fma_f32 on pairs: about
// 8 VALU slots per element against ~23 for gelu_erf_f.  The GEGLU epilogue of the 64x64 maps is VALU-bound on exactly this
// (32 activations per lane per item against five K tiles of MFMA).
__device__ __forceinline__ f32x4 gelu_erf_bf16_f4(f32x4 x) {
  const float XC = 4.2426405f;
  f32x4 xc;
#pragma unroll
  for (int r = 0; r < 4; ++r) xc[r] = __builtin_amdgcn_fmed3f(x[r], -XC, XC);
  const f32x4 u = xc * xc;
  f32x4 q = u * 8.812844898e-11f + -8.762744308e-09f;
  q = q * u + 3.867438352e-07f;
  q = q * u + -1.007556784e-05f;
  q = q * u + 1.743170724e-04f;
  q = q * u + -2.139258897e-03f;
  q = q * u + 1.936233975e-02f;
  q = q * u + -1.322318017e-01f;
  q = q * u + 7.975320816e-01f;
  f32x4 e = xc * q;
#pragma unroll
  for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_fmed3f(e[r], -1.0f, 1.0f);
  const f32x4 h = x * 0.5f;
  return h * e + h;
}

// One MFMA "k-group" = 64 bytes of K per operand row (4 lane-groups x 16 B):
// bf16: one v_mfma_f32_16x16x32_bf16; f32: four v_mfma_f32_16x16x4_f32, lane
// group g supplying k = 4*g + t in the t-th instruction (any K permutation is
// legal as long as both operands use the same one).
template <typename T>
__device__ __forceinline__ void mma_kgroup(const uint4& a, const uint4& b, f32x4& acc);
template <>
__device__ __forceinline__ void mma_kgroup<bf16_t>(const uint4& a, const uint4& b, f32x4& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_kgroup<float>(const uint4& a, const uint4& b, f32x4& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), acc, 0, 0, 0);
}

}  // namespace ldmseg
