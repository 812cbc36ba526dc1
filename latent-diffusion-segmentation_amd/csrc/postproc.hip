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

}  // namespace

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
