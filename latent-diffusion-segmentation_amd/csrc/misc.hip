// Small / HBM-bound kernels around the GEMM path: layout packing at the fp32
// NCHW boundary, the timestep-embedding MLP, bilinear x2 logits upsample, the
// DDIM scheduler step, and one-time weight repacking.
#include "common.h"
#include "kernels.h"

namespace ldmseg {
namespace {

// ---------------------------------------------------------------- packing
template <typename T>
__global__ void pack_concat3_kernel(const float* a, int Ca, const float* b, int Cb, const float* c, int Cc,
                                    T* y, int HW, int Cpad, float mul, float add) {
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  const int img = blockIdx.y;
  if (pix >= HW) return;
  T* o = y + ((size_t)img * HW + pix) * Cpad;
  int k = 0;
  for (int ch = 0; ch < Ca; ++ch) o[k++] = from_f32<T>(a[((size_t)img * Ca + ch) * HW + pix] * mul + add);
  for (int ch = 0; ch < Cb; ++ch) o[k++] = from_f32<T>(b[((size_t)img * Cb + ch) * HW + pix] * mul + add);
  for (int ch = 0; ch < Cc; ++ch) o[k++] = from_f32<T>(c[((size_t)img * Cc + ch) * HW + pix] * mul + add);
  for (; k < Cpad; ++k) o[k] = from_f32<T>(0.f);
}

// ---------------------------------------------------------------- time embedding
// diffusers Timesteps(320, flip_sin_to_cos=True, freq_shift=0): [cos(t f_i) | sin(t f_i)], f_i = exp(-ln(1e4) i/160)
__global__ void time_sinus_kernel(const int64_t* t_dev, int t_count, int64_t t_host, int B, float* out) {
  const int i = threadIdx.x;  // 0..159
  const int b = blockIdx.x;
  if (b >= B || i >= 160) return;
  const int64_t t = t_dev ? t_dev[t_count > 1 ? b : 0] : t_host;
  const float expo = (-9.210340371976184f * (float)i) / 160.0f;
  const float arg = (float)t * expf(expo);
  out[b * 320 + i] = cosf(arg);
  out[b * 320 + 160 + i] = sinf(arg);
}

// y[b][n] = act_out( sum_k act_in(x[b][k]) W[n][k] + bias[n] ), f32, B <= 64; one wave per n
template <int BC>
__global__ __launch_bounds__(256) void small_linear_kernel(const float* x, const float* W, const float* bias, float* y,
                                                           int B, int K, int N, int silu_in, int silu_out) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float* w = W + (size_t)n * K;
  for (int b0 = 0; b0 < B; b0 += BC) {
    float acc[BC];
#pragma unroll
    for (int j = 0; j < BC; ++j) acc[j] = 0.f;
    for (int k = lane; k < K; k += 64) {
      const float wv = w[k];
#pragma unroll
      for (int j = 0; j < BC; ++j) {
        if (b0 + j < B) {
          float xv = x[(size_t)(b0 + j) * K + k];
          if (silu_in) xv = silu_f(xv);
          acc[j] += xv * wv;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < BC; ++j) {
      float v = acc[j];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0 && b0 + j < B) {
        v += bias ? bias[n] : 0.f;
        y[(size_t)(b0 + j) * N + n] = silu_out ? silu_f(v) : v;
      }
    }
  }
}

// ---------------------------------------------------------------- bilinear x2, NHWC -> NCHW f32
template <typename T>
__global__ __launch_bounds__(256) void bilinear2x_kernel(const T* x, float* y, int H, int W, int C) {
  constexpr int PC = Chunk<T>::N;
  const int OW = 2 * W, OH = 2 * H;
  const int ox = blockIdx.x * blockDim.x + threadIdx.x;
  const int oy = blockIdx.y;
  const int b = blockIdx.z;
  if (ox >= OW) return;
  // torch upsample_bilinear2d, align_corners=False: src = max(0.5*(dst+0.5)-0.5, 0)
  const float sy = fmaxf(0.5f * (oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * (ox + 0.5f) - 0.5f, 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
  const float ly1 = sy - y0, ly0 = 1.f - ly1, lx1 = sx - x0, lx0 = 1.f - lx1;
  const T* r00 = x + (((size_t)b * H + y0) * W + x0) * C;
  const T* r01 = x + (((size_t)b * H + y0) * W + x1) * C;
  const T* r10 = x + (((size_t)b * H + y1) * W + x0) * C;
  const T* r11 = x + (((size_t)b * H + y1) * W + x1) * C;
  float* o = y + ((size_t)b * C * OH + oy) * OW + ox;
  const size_t plane = (size_t)OH * OW;
  for (int c = 0; c < C; c += PC) {
    float a[PC], bb[PC], cc[PC], d[PC];
    Chunk<T>::unpack(*(const uint4*)(r00 + c), a);
    Chunk<T>::unpack(*(const uint4*)(r01 + c), bb);
    Chunk<T>::unpack(*(const uint4*)(r10 + c), cc);
    Chunk<T>::unpack(*(const uint4*)(r11 + c), d);
#pragma unroll
    for (int e = 0; e < PC; ++e)
      o[(size_t)(c + e) * plane] = ly0 * (lx0 * a[e] + lx1 * bb[e]) + ly1 * (lx0 * cc[e] + lx1 * d[e]);
  }
}

// ---------------------------------------------------------------- fused decode tail
// bilinear x2 (align_corners=False) of NHWC logits [B,H,W,C] fused with argmax over C and the
// max-softmax probability (trainers_ldm_cond.py:427-433): ids [B,2H,2W] int64 (ignore_label where
// prob < mask_th, if mask_th >= 0), prob [B,2H,2W] f32 (optional).  Online softmax: one pass over C,
// the 134 MB/image fp32 logits tensor is never written.
template <typename T>
__global__ __launch_bounds__(256) void bilinear_argmax_kernel(const T* x, int64_t* ids, float* prob, int H, int W, int C,
                                                              float mask_th, int64_t ignore_label) {
  constexpr int PC = Chunk<T>::N;
  const int OW = 2 * W, OH = 2 * H;
  const int ox = blockIdx.x * blockDim.x + threadIdx.x;
  const int oy = blockIdx.y;
  const int b = blockIdx.z;
  if (ox >= OW) return;
  const float sy = fmaxf(0.5f * (oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * (ox + 0.5f) - 0.5f, 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
  const float ly1 = sy - y0, ly0 = 1.f - ly1, lx1 = sx - x0, lx0 = 1.f - lx1;
  const T* r00 = x + (((size_t)b * H + y0) * W + x0) * C;
  const T* r01 = x + (((size_t)b * H + y0) * W + x1) * C;
  const T* r10 = x + (((size_t)b * H + y1) * W + x0) * C;
  const T* r11 = x + (((size_t)b * H + y1) * W + x1) * C;
  float m = -INFINITY, ssum = 0.f;
  int arg = 0;
  for (int c = 0; c < C; c += PC) {
    float a[PC], bb[PC], cc[PC], d[PC];
    Chunk<T>::unpack(*(const uint4*)(r00 + c), a);
    Chunk<T>::unpack(*(const uint4*)(r01 + c), bb);
    Chunk<T>::unpack(*(const uint4*)(r10 + c), cc);
    Chunk<T>::unpack(*(const uint4*)(r11 + c), d);
#pragma unroll
    for (int e = 0; e < PC; ++e) {
      const float v = ly0 * (lx0 * a[e] + lx1 * bb[e]) + ly1 * (lx0 * cc[e] + lx1 * d[e]);
      if (v > m) {            // strict: the first maximum wins, like torch.argmax
        ssum = ssum * __expf(m - v) + 1.f;
        m = v;
        arg = c + e;
      } else {
        ssum += __expf(v - m);
      }
    }
  }
  const float pmax = 1.f / ssum;
  const size_t o = ((size_t)b * OH + oy) * OW + ox;
  ids[o] = (mask_th >= 0.f && pmax < mask_th) ? ignore_label : (int64_t)arg;
  if (prob) prob[o] = pmax;
}

// ---------------------------------------------------------------- row softmax (single-head attention of the image VAE)
// one wave per row; three passes over a row that stays in L2 (16-64 KB): max, sum of exp, normalise
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* s, T* p, int rows, int n, float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* r = s + (size_t)row * n;
  float m = -INFINITY;
  for (int i = lane * 4; i < n; i += 256) {
    const f32x4 v = *(const f32x4*)(r + i);
    m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
  }
  m = wave64_max(m) * scale;     // scale > 0
  float sum = 0.f;
  for (int i = lane * 4; i < n; i += 256) {
    const f32x4 v = *(const f32x4*)(r + i);
    sum += (__expf(v[0] * scale - m) + __expf(v[1] * scale - m)) + (__expf(v[2] * scale - m) + __expf(v[3] * scale - m));
  }
  sum = wave64_sum(sum);
  const float inv = 1.0f / sum;
  T* o = p + (size_t)row * n;
  for (int i = lane * 4; i < n; i += 256) {
    const f32x4 v = *(const f32x4*)(r + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[i + e] = from_f32<T>(__expf(v[e] * scale - m) * inv);
  }
}

// ---------------------------------------------------------------- weight repack
template <typename T>
__global__ void repack_conv_kernel(const float* w, T* out, int Co, int Ci, int KH, int KW, int Npad, int Cipad, int cm_tile) {
  const size_t total = (size_t)Npad * KH * KW * Cipad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c, tap, n;
    if (cm_tile > 0) {                    // [n][channel tile][tap][channel in tile]
      const int ci = (int)(i % cm_tile);
      size_t r = i / cm_tile;
      tap = (int)(r % (KH * KW));
      r /= (KH * KW);
      const int tiles = Cipad / cm_tile;
      c = (int)(r % tiles) * cm_tile + ci;
      n = (int)(r / tiles);
    } else {                              // [n][tap][channel]
      c = (int)(i % Cipad);
      const size_t r = i / Cipad;
      tap = (int)(r % (KH * KW));
      n = (int)(r / (KH * KW));
    }
    float v = 0.f;
    if (n < Co && c < Ci) v = w[(((size_t)n * Ci + c) * KH * KW) + tap];
    out[i] = from_f32<T>(v);
  }
}
template <typename T>
__global__ void repack_rows_kernel(const float* w, T* out, const int* src_row, int Npad, int K) {
  const size_t total = (size_t)Npad * K;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % K);
    const int r = (int)(i / K);
    const int sr = src_row[r];
    out[i] = from_f32<T>(sr >= 0 ? w[(size_t)sr * K + k] : 0.f);
  }
}
template <typename T>
__global__ void repack_rows_scaled_kernel(const float* w, T* out, const int* src_row, int Npad, int K, const float* colscale) {
  const size_t total = (size_t)Npad * K;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % K);
    const int r = (int)(i / K);
    const int sr = src_row[r];
    const float v = sr >= 0 ? w[(size_t)sr * K + k] : 0.f;
    out[i] = from_f32<T>(colscale ? v * colscale[k] : v);
  }
}
// one wave per row: out[n] = sum_k W[n][k] (+ add[n])
template <typename T>
__global__ __launch_bounds__(256) void rowsum_kernel(const T* W, const float* add, float* out, int N, int K) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += to_f32<T>(W[(size_t)n * K + k]);
  s = wave64_sum(s);
  if (lane == 0) out[n] = s + (add ? add[n] : 0.f);
}
template <typename T>
__global__ void repack_convt2_kernel(const float* w, T* out, int Ci, int Co) {
  const size_t total = (size_t)4 * Co * Ci;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Ci);
    const int n = (int)(i / Ci);
    const int tap = n / Co, co = n - tap * Co;
    out[i] = from_f32<T>(w[((size_t)ci * Co + co) * 4 + tap]);
  }
}

// COCO.encode_bitmap (ldmseg/data/coco.py:377-382): ids [B,HW] int64 -> bits [B,n,HW] f32, LSB first,
// ignore_label pixels -> fill_value; optional affine (2x-1 of encode_inputs) and ignore mask.
__global__ void bit_encode_kernel(const int64_t* ids, float* out, uint8_t* ignore, int n, int HW, size_t total,
                                  int64_t ignore_label, float fill, float mul, float add) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / HW, pix = i - b * HW;
    const int64_t v = ids[i];
    const bool ign = (v == ignore_label);
    if (ignore) ignore[i] = ign ? 1 : 0;
    for (int k = 0; k < n; ++k) {
      // torch.remainder(x >> k, 2): python-style modulo, also for negative ids
      const int64_t sh = v >> k;
      const float bit = (float)(sh & 1);
      out[(b * n + k) * HW + pix] = (ign ? fill : bit) * mul + add;
    }
  }
}
// COCO.decode_bitmap (coco.py:384-390): x [B,n,HW] f32 -> ids [B,HW] int64, bit k set iff x[k] > 0
__global__ void bit_decode_kernel(const float* x, int64_t* out, int n, int HW, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / HW, pix = i - b * HW;
    float acc = 0.f;   // the reference sums float(bit) * 2**k in fp32
    for (int k = 0; k < n; ++k) acc += (x[(b * n + k) * HW + pix] > 0.f ? 1.f : 0.f) * (float)(1ll << k);
    out[i] = (int64_t)acc;
  }
}

__global__ void posterior_sample_kernel(const float* mom, const float* noise, float* out, int HW, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i % HW, r = i / HW;
    const size_t c = r % 4, b = r / 4;
    const float mean = mom[(b * 8 + c) * HW + pix];
    const float lv = fminf(fmaxf(mom[(b * 8 + 4 + c) * HW + pix], -30.f), 20.f);
    out[i] = mean + expf(0.5f * lv) * (noise ? noise[i] : 0.f);
  }
}

inline int grid_for(size_t n, int block = 256, int cap = 4096) {
  size_t g = (n + block - 1) / block;
  if (g > (size_t)cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}
inline int ok() { return hipGetLastError() == hipSuccess ? 0 : -3; }

}  // namespace

int launch_pack_concat3(const float* a, int Ca, const float* b, int Cb, const float* c, int Cc, void* y, int B, int HW,
                        int Cpad, int dtype, hipStream_t s) {
  if (Ca + Cb + Cc > Cpad) return -2;
  dim3 grid((HW + 255) / 256, B), block(256);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(pack_concat3_kernel<bf16_t>, grid, block, 0, s, a, Ca, b, Cb, c, Cc, (bf16_t*)y, HW, Cpad, 1.f, 0.f);
  else
    hipLaunchKernelGGL(pack_concat3_kernel<float>, grid, block, 0, s, a, Ca, b, Cb, c, Cc, (float*)y, HW, Cpad, 1.f, 0.f);
  return ok();
}

int launch_pack_nchw(const float* x, void* y, int B, int C, int HW, int Cpad, float mul, float add, int dtype,
                     hipStream_t s) {
  if (C > Cpad) return -2;
  dim3 grid((HW + 255) / 256, B), block(256);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(pack_concat3_kernel<bf16_t>, grid, block, 0, s, x, C, nullptr, 0, nullptr, 0, (bf16_t*)y, HW, Cpad, mul, add);
  else
    hipLaunchKernelGGL(pack_concat3_kernel<float>, grid, block, 0, s, x, C, nullptr, 0, nullptr, 0, (float*)y, HW, Cpad, mul, add);
  return ok();
}

int launch_time_embed(const int64_t* t_dev, int t_count, int64_t t_host, int B, float* sinus, hipStream_t s) {
  hipLaunchKernelGGL(time_sinus_kernel, dim3(B), dim3(192), 0, s, t_dev, t_count, t_host, B, sinus);
  return ok();
}

int launch_small_linear(const float* x, const float* W, const float* bias, float* y, int B, int K, int N, int silu_in,
                        int silu_out, hipStream_t s) {
  hipLaunchKernelGGL(small_linear_kernel<8>, dim3((N + 3) / 4), dim3(256), 0, s, x, W, bias, y, B, K, N, silu_in, silu_out);
  return ok();
}

int launch_bilinear2x_nchw(const void* x, float* y, int B, int H, int W, int C, int dtype, hipStream_t s) {
  dim3 grid((2 * W + 255) / 256, 2 * H, B), block(256);
  if (dtype == DT_BF16) {
    if (C % 8) return -2;
    hipLaunchKernelGGL(bilinear2x_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, y, H, W, C);
  } else {
    if (C % 4) return -2;
    hipLaunchKernelGGL(bilinear2x_kernel<float>, grid, block, 0, s, (const float*)x, y, H, W, C);
  }
  return ok();
}

int launch_bilinear2x_argmax(const void* x, int64_t* ids, float* prob, int B, int H, int W, int C, float mask_th,
                             int64_t ignore_label, int dtype, hipStream_t s) {
  dim3 grid((2 * W + 255) / 256, 2 * H, B), block(256);
  if (dtype == DT_BF16) {
    if (C % 8) return -2;
    hipLaunchKernelGGL(bilinear_argmax_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, ids, prob, H, W, C, mask_th, ignore_label);
  } else {
    if (C % 4) return -2;
    hipLaunchKernelGGL(bilinear_argmax_kernel<float>, grid, block, 0, s, (const float*)x, ids, prob, H, W, C, mask_th, ignore_label);
  }
  return ok();
}

int launch_softmax_rows(const float* sc, void* p, int rows, int n, float scale, int dtype, hipStream_t st) {
  if (n % 4) return -2;
  if (dtype == DT_BF16) hipLaunchKernelGGL(softmax_rows_kernel<bf16_t>, dim3((rows + 3) / 4), dim3(256), 0, st, sc, (bf16_t*)p, rows, n, scale);
  else hipLaunchKernelGGL(softmax_rows_kernel<float>, dim3((rows + 3) / 4), dim3(256), 0, st, sc, (float*)p, rows, n, scale);
  return ok();
}

int launch_repack_conv(const float* w, void* out, int Co, int Ci, int KH, int KW, int Npad, int Cipad, int dtype,
                       hipStream_t s, int cm_tile) {
  const size_t total = (size_t)Npad * KH * KW * Cipad;
  if (cm_tile > 0 && Cipad % cm_tile != 0) return -2;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(repack_conv_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, w, (bf16_t*)out, Co, Ci, KH, KW, Npad, Cipad, cm_tile);
  else
    hipLaunchKernelGGL(repack_conv_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, w, (float*)out, Co, Ci, KH, KW, Npad, Cipad, cm_tile);
  return ok();
}

int launch_repack_rows(const float* w, void* out, const int* src_row_dev, int Npad, int K, int dtype, hipStream_t s) {
  const size_t total = (size_t)Npad * K;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(repack_rows_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, w, (bf16_t*)out, src_row_dev, Npad, K);
  else
    hipLaunchKernelGGL(repack_rows_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, w, (float*)out, src_row_dev, Npad, K);
  return ok();
}

int launch_repack_rows_scaled(const float* w, void* out, const int* src_row_dev, int Npad, int K, const float* colscale,
                              int dtype, hipStream_t s) {
  const size_t total = (size_t)Npad * K;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(repack_rows_scaled_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, w, (bf16_t*)out, src_row_dev, Npad, K, colscale);
  else
    hipLaunchKernelGGL(repack_rows_scaled_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, w, (float*)out, src_row_dev, Npad, K, colscale);
  return ok();
}

int launch_rowsum(const void* W, const float* add, float* out, int N, int K, int dtype, hipStream_t s) {
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(rowsum_kernel<bf16_t>, dim3((N + 3) / 4), dim3(256), 0, s, (const bf16_t*)W, add, out, N, K);
  else
    hipLaunchKernelGGL(rowsum_kernel<float>, dim3((N + 3) / 4), dim3(256), 0, s, (const float*)W, add, out, N, K);
  return ok();
}

int launch_repack_convt2(const float* w, void* out, int Ci, int Co, int dtype, hipStream_t s) {
  const size_t total = (size_t)4 * Co * Ci;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(repack_convt2_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, w, (bf16_t*)out, Ci, Co);
  else
    hipLaunchKernelGGL(repack_convt2_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, w, (float*)out, Ci, Co);
  return ok();
}





int launch_bit_encode(const int64_t* ids, float* out, uint8_t* ignore, int B, int n, int HW, int64_t ignore_label,
                      float fill, float mul, float add, hipStream_t s) {
  const size_t total = (size_t)B * HW;
  hipLaunchKernelGGL(bit_encode_kernel, dim3(grid_for(total)), dim3(256), 0, s, ids, out, ignore, n, HW, total, ignore_label, fill, mul, add);
  return ok();
}
int launch_bit_decode(const float* x, int64_t* out, int B, int n, int HW, hipStream_t s) {
  const size_t total = (size_t)B * HW;
  hipLaunchKernelGGL(bit_decode_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, out, n, HW, total);
  return ok();
}

int launch_posterior_sample(const float* moments, const float* noise, float* out, int B, int HW, hipStream_t s) {
  const size_t total = (size_t)B * 4 * HW;
  hipLaunchKernelGGL(posterior_sample_kernel, dim3(grid_for(total)), dim3(256), 0, s, moments, noise, out, HW, total);
  return ok();
}

}  // namespace ldmseg
