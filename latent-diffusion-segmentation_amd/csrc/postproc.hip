// Panoptic post-processing of the decoded logits on the GPU (reference: the per-image loop in
// ldmseg/trainers/trainers_ldm_cond.py:1277-1313, which runs argmax/softmax on the device, copies the
// [C,H,W] sigmoid volume to the host and filters segments with numpy).
//
//   scan   : one pass over the NCHW fp32 logits: per pixel argmax, max / top-2 softmax probability
//            (threshold -> label -1), and per image the pixel count of every label plus the number of
//            pixels whose sigmoid(logit_c) >= mask_th for every class c.  HBM-bound: the logits are read
//            exactly once (C*H*W*4 bytes per image), everything else is O(H*W) or O(C).
//   filter : per (image, class): void if label == ignore_label, count < count_th or
//            count / mask_count < overlap_th   (double division, like numpy's int/int)
//   remap  : panoptic = kept ? label + 1 : 0
//
// Counters are integers accumulated with integer atomics: the result does not depend on the order.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.h"

namespace ldmseg {
namespace {

constexpr int kMaxClasses = 256;

// grid (ceil(HW / 256), B); thread = pixel; loop over classes (coalesced: consecutive threads read
// consecutive pixels of one class plane)
__global__ __launch_bounds__(256) void panoptic_scan_kernel(const float* logits, int C, int HW, int threshold_output,
                                                            int threshold_mode, float mask_th, int32_t* labels,
                                                            int32_t* counts, int32_t* mask_counts) {
  __shared__ int s_cnt[kMaxClasses];
  __shared__ int s_msk[kMaxClasses];
  const int b = blockIdx.y;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = pix < HW;
  for (int c = threadIdx.x; c < C; c += blockDim.x) { s_cnt[c] = 0; s_msk[c] = 0; }
  __syncthreads();
  const float* base = logits + (size_t)b * C * HW + (live ? pix : 0);
  float m = -INFINITY, m2 = -INFINITY, ssum = 0.f;
  int arg = 0;
  const int lane = threadIdx.x & 63;
  for (int c = 0; c < C; ++c) {
    const float v = live ? base[(size_t)c * HW] : -INFINITY;
    if (v > m) {                 // strict: the first maximum wins, like torch.argmax
      ssum = ssum * __expf(m - v) + 1.f;
      m2 = m;
      m = v;
      arg = c;
    } else {
      ssum += __expf(v - m);
      m2 = fmaxf(m2, v);
    }
    // sigmoid(v) >= mask_th, in fp32 like F.sigmoid on the fp32 logits
    const bool in_mask = live && (1.0f / (1.0f + expf(-v)) >= mask_th);
    const unsigned long long bal = __ballot(in_mask);
    if (lane == 0 && bal) atomicAdd(&s_msk[c], __popcll(bal));
  }
  int label = arg;
  if (threshold_output) {
    const float pmax = 1.f / ssum;
    const float p = threshold_mode == 1 ? pmax - __expf(m2 - m) / ssum : pmax;   // 'topk_diff' : max prob
    if (p < mask_th) label = -1;
  }
  if (live) {
    labels[(size_t)b * HW + pix] = label;
    if (label >= 0) atomicAdd(&s_cnt[label], 1);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    if (s_cnt[c]) atomicAdd(&counts[b * C + c], s_cnt[c]);
    if (s_msk[c]) atomicAdd(&mask_counts[b * C + c], s_msk[c]);
  }
}

__global__ void panoptic_filter_kernel(const int32_t* counts, const int32_t* mask_counts, int total, int C, int count_th,
                                       double overlap_th, int64_t ignore_label, uint8_t* keep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = i % C;
  const int n = counts[i];
  bool k = n > 0 && n >= count_th && (int64_t)c != ignore_label;
  if (k) {
    const double ratio = (double)n / (double)mask_counts[i];   // x/0 = inf, kept (numpy does the same)
    if (ratio < overlap_th) k = false;
  }
  keep[i] = k ? 1 : 0;
}

__global__ void panoptic_remap_kernel(const int32_t* labels, const uint8_t* keep, int C, int HW, size_t total,
                                      int32_t* panoptic) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW);
    const int l = labels[i];
    panoptic[i] = (l >= 0 && keep[b * C + l]) ? l + 1 : 0;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Fused evaluation tail (trainers_ldm_cond.py:1252-1313 without the [B,128,H,W] fp32 volume):
//   decoder output at 4L (NHWC, compute dtype)  -- bilinear x2 (vae.py:270) -->  8L
//   -- bilinear to the network input size (:1252-1257; the identity when it is 8L) -->  (S_h, S_w)
//   -- crop to the non-padded box (:1172-1178, :1263) -- bilinear to the original size (:1266-1271) -->  (h, w)
//   -- argmax / max or top-2 softmax / sigmoid >= mask_th counts (:1277-1306).
// Bilinear interpolation is separable and linear, so the chain of (up to) three resamplings is ONE weighted sum over
// at most 8 x 8 source pixels of the 4L map with per-axis weights that are products of the stage weights (every
// stage with torch's align_corners=False arithmetic: src = max(scale * (dst + 0.5) - 0.5, 0), i1 = min(i0 + 1, n - 1)).
// Taps with the same source index are merged: x2 followed by a near-identity resize leaves 2-3 distinct rows / columns.
// One thread = one output pixel; the 128 classes are walked in 16-byte chunks (4 x 4 .. 3 x 3 loads each, neighbouring
// pixels share them through L1/L2).  Reads B * (4L)^2 * C * 2 B, writes one int32 per output pixel.
struct AxisTaps {
  int idx[8];
  float w[8];
  int n;
};
__device__ __forceinline__ void taps_add(AxisTaps& t, int i, float w) {
  for (int k = 0; k < t.n; ++k)
    if (t.idx[k] == i) { t.w[k] += w; return; }
  t.idx[t.n] = i;
  t.w[t.n] = w;
  ++t.n;
}
// one bilinear stage (align_corners=False): output index o of a length-n_out axis over a length-n_in axis
__device__ __forceinline__ void lin_stage(int o, int n_in, int n_out, int& i0, int& i1, float& w0, float& w1) {
  const float scale = (float)n_in / (float)n_out;
  float src = scale * ((float)o + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  w1 = src - (float)i0;
  w0 = 1.f - w1;
}
// composite taps of output index o:  (n_out over the crop [c0, c0 + cn) of the n_mid grid)  o  (n_mid over 2 * n4)  o  (x2 over n4)
__device__ __forceinline__ void axis_taps(int o, int n_out, int c0, int cn, int n_mid, int n4, AxisTaps& t) {
  t.n = 0;
  int a[2]; float wa[2];
  lin_stage(o, cn, n_out, a[0], a[1], wa[0], wa[1]);
  for (int i = 0; i < 2; ++i) {
    if (wa[i] == 0.f) continue;
    const int m = c0 + a[i];                      // index in the n_mid grid
    int b8[2]; float wb[2];
    if (n_mid == 2 * n4) { b8[0] = m; b8[1] = m; wb[0] = 1.f; wb[1] = 0.f; }     // same size: F.interpolate is the identity
    else lin_stage(m, 2 * n4, n_mid, b8[0], b8[1], wb[0], wb[1]);
    for (int j = 0; j < 2; ++j) {
      if (wb[j] == 0.f) continue;
      int c4[2]; float wc[2];
      lin_stage(b8[j], n4, 2 * n4, c4[0], c4[1], wc[0], wc[1]);
      for (int k = 0; k < 2; ++k)
        if (wc[k] != 0.f) taps_add(t, c4[k], wa[i] * wb[j] * wc[k]);
    }
  }
}

struct ResampleScanParams {
  const void* x4;          // [H4][W4][C] of ONE image, compute dtype
  int H4, W4, C;
  int in_h, in_w;          // the network input size the logits are first resized to (:1252)
  int y0, x0, ch, cw;      // crop box in that grid
  int oh, ow;              // output (original image) size
  int threshold_output, threshold_mode;
  float mask_th;
  int32_t* labels;         // [oh * ow]
  int32_t* counts;         // [C] of this image
  int32_t* mask_counts;    // [C]
  float* volume;           // test hook: when non-null the resampled logits [C][oh * ow] are also written (fp32)
};

template <typename T>
__global__ __launch_bounds__(256) void resample_scan_kernel(const ResampleScanParams p) {
  constexpr int PC = Chunk<T>::N;
  __shared__ int s_cnt[kMaxClasses];
  __shared__ int s_msk[kMaxClasses];
  for (int c = threadIdx.x; c < p.C; c += blockDim.x) { s_cnt[c] = 0; s_msk[c] = 0; }
  __syncthreads();
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  const int HW = p.oh * p.ow;
  const bool live = pix < HW;
  const int oy = live ? pix / p.ow : 0, ox = live ? pix - oy * p.ow : 0;
  AxisTaps ty, tx;
  axis_taps(oy, p.oh, p.y0, p.ch, p.in_h, p.H4, ty);
  axis_taps(ox, p.ow, p.x0, p.cw, p.in_w, p.W4, tx);
  const T* x = (const T*)p.x4;
  const int lane = threadIdx.x & 63;
  float m = -INFINITY, m2 = -INFINITY, ssum = 0.f;
  int arg = 0;
  for (int c = 0; c < p.C; c += PC) {
    float v[PC];
#pragma unroll
    for (int e = 0; e < PC; ++e) v[e] = 0.f;
    for (int iy = 0; iy < ty.n; ++iy) {
      float r[PC];
#pragma unroll
      for (int e = 0; e < PC; ++e) r[e] = 0.f;
      const T* row = x + (size_t)ty.idx[iy] * p.W4 * p.C + c;
      for (int ix = 0; ix < tx.n; ++ix) {
        float f[PC];
        Chunk<T>::unpack(*(const uint4*)(row + (size_t)tx.idx[ix] * p.C), f);
#pragma unroll
        for (int e = 0; e < PC; ++e) r[e] += tx.w[ix] * f[e];
      }
#pragma unroll
      for (int e = 0; e < PC; ++e) v[e] += ty.w[iy] * r[e];
    }
    if (p.volume && live) {
#pragma unroll
      for (int e = 0; e < PC; ++e) p.volume[(size_t)(c + e) * HW + pix] = v[e];
    }
#pragma unroll
    for (int e = 0; e < PC; ++e) {
      const float val = live ? v[e] : -INFINITY;
      if (val > m) {               // strict: the first maximum wins, like torch.argmax
        ssum = ssum * __expf(m - val) + 1.f;
        m2 = m;
        m = val;
        arg = c + e;
      } else {
        ssum += __expf(val - m);
        m2 = fmaxf(m2, val);
      }
      const bool in_mask = live && (1.0f / (1.0f + expf(-val)) >= p.mask_th);
      const unsigned long long bal = __ballot(in_mask);
      if (lane == 0 && bal) atomicAdd(&s_msk[c + e], __popcll(bal));
    }
  }
  int label = arg;
  if (p.threshold_output) {
    const float pmax = 1.f / ssum;
    const float pr = p.threshold_mode == 1 ? pmax - __expf(m2 - m) / ssum : pmax;
    if (pr < p.mask_th) label = -1;
  }
  if (live) {
    p.labels[pix] = label;
    if (label >= 0) atomicAdd(&s_cnt[label], 1);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
    if (s_cnt[c]) atomicAdd(&p.counts[c], s_cnt[c]);
    if (s_msk[c]) atomicAdd(&p.mask_counts[c], s_msk[c]);
  }
}

}  // namespace

// Fused evaluation tail on the decoder's 4L output (see resample_scan_kernel).  boxes [B][4] = (y0, x0, height, width) of the
// non-padded region in the (in_h, in_w) grid, sizes [B][2] = (h, w), offsets [B] = first element of image b in the flat
// labels / panoptic buffers: HOST arrays.
int launch_panoptic_from_decoder(const void* x4, int B, int H4, int W4, int C, int dtype, int in_h, int in_w,
                                 const int32_t* boxes, const int32_t* sizes, const int64_t* offsets, int threshold_output,
                                 int threshold_mode, float mask_th, int count_th, double overlap_th, int64_t ignore_label,
                                 int32_t* labels, int32_t* panoptic, uint8_t* keep, int32_t* counts, int32_t* mask_counts,
                                 hipStream_t s, float* volume) {
  if (C < 1 || C > kMaxClasses || B < 1 || H4 < 1 || W4 < 1 || in_h < 1 || in_w < 1) return -2;
  if (C % (dtype == DT_BF16 ? 8 : 4) != 0) return -2;
  if (hipMemsetAsync(counts, 0, (size_t)B * C * sizeof(int32_t), s) != hipSuccess) return -3;
  if (hipMemsetAsync(mask_counts, 0, (size_t)B * C * sizeof(int32_t), s) != hipSuccess) return -3;
  const size_t img = (size_t)H4 * W4 * C * (dtype == DT_BF16 ? 2 : 4);
  for (int b = 0; b < B; ++b) {
    ResampleScanParams p;
    p.x4 = (const char*)x4 + (size_t)b * img;
    p.H4 = H4; p.W4 = W4; p.C = C; p.in_h = in_h; p.in_w = in_w;
    p.y0 = boxes ? boxes[4 * b] : 0; p.x0 = boxes ? boxes[4 * b + 1] : 0;
    p.ch = boxes ? boxes[4 * b + 2] : in_h; p.cw = boxes ? boxes[4 * b + 3] : in_w;
    p.oh = sizes[2 * b]; p.ow = sizes[2 * b + 1];
    if (p.y0 < 0 || p.x0 < 0 || p.ch < 1 || p.cw < 1 || p.y0 + p.ch > in_h || p.x0 + p.cw > in_w || p.oh < 1 || p.ow < 1) return -2;
    p.threshold_output = threshold_output; p.threshold_mode = threshold_mode; p.mask_th = mask_th;
    p.labels = labels + offsets[b];
    p.counts = counts + (size_t)b * C;
    p.mask_counts = mask_counts + (size_t)b * C;
    p.volume = volume ? volume + (size_t)offsets[b] * C : nullptr;
    const int hw = p.oh * p.ow;
    if (dtype == DT_BF16) hipLaunchKernelGGL(resample_scan_kernel<bf16_t>, dim3((hw + 255) / 256), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(resample_scan_kernel<float>, dim3((hw + 255) / 256), dim3(256), 0, s, p);
  }
  hipLaunchKernelGGL(panoptic_filter_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, counts, mask_counts, B * C, C,
                     count_th, overlap_th, ignore_label, keep);
  for (int b = 0; b < B; ++b) {
    const size_t total = (size_t)sizes[2 * b] * sizes[2 * b + 1];
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(panoptic_remap_kernel, dim3((unsigned)blocks), dim3(256), 0, s, labels + offsets[b], keep + (size_t)b * C, C,
                       (int)total, total, panoptic + offsets[b]);
  }
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_panoptic_postprocess(const float* logits, int B, int C, int HW, int threshold_output, int threshold_mode,
                                float mask_th, int count_th, double overlap_th, int64_t ignore_label, int32_t* labels,
                                int32_t* panoptic, uint8_t* keep, int32_t* counts, int32_t* mask_counts, hipStream_t s) {
  if (C < 1 || C > kMaxClasses || B < 1 || HW < 1) return -2;
  if (hipMemsetAsync(counts, 0, (size_t)B * C * sizeof(int32_t), s) != hipSuccess) return -3;
  if (hipMemsetAsync(mask_counts, 0, (size_t)B * C * sizeof(int32_t), s) != hipSuccess) return -3;
  hipLaunchKernelGGL(panoptic_scan_kernel, dim3((HW + 255) / 256, B), dim3(256), 0, s, logits, C, HW, threshold_output,
                     threshold_mode, mask_th, labels, counts, mask_counts);
  hipLaunchKernelGGL(panoptic_filter_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, counts, mask_counts, B * C, C,
                     count_th, overlap_th, ignore_label, keep);
  const size_t total = (size_t)B * HW;
  size_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(panoptic_remap_kernel, dim3((unsigned)blocks), dim3(256), 0, s, labels, keep, C, HW, total, panoptic);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace ldmseg
