// Single-operator entry points with an fp32 boundary (ldmseg_op_*): the parity tests
// drive each gfx950 kernel in isolation through these, with torch ops as the reference.
// Not on the hot path: every call allocates and frees its own staging buffers.
#include <hip/hip_runtime.h>

#include <string>
#include <cstdlib>
#include <vector>

#include "../../include/ldmseg_hip.h"
#include "common.h"
#include "kernels.h"

using namespace ldmseg;

// Counter region of the in-launch split-K finish for these handle-less launches: one per device, zeroed once.  The operator
// entry points are test support and run one launch at a time per device (a handle owns its own region).
static unsigned long long* op_cf_region() {
  static unsigned long long* r[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!r[dev]) {
    void* q = nullptr;
    if (hipMalloc(&q, igemm_cf_bytes()) != hipSuccess) return nullptr;
    (void)hipMemset(q, 0, igemm_cf_bytes());
    r[dev] = (unsigned long long*)q;
  }
  return r[dev];
}

namespace {

template <typename T>
__global__ void unpack_nhwc_kernel(const T* x, float* y, int C, int HW, int ldx) {
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  const int img = blockIdx.y;
  if (pix >= HW) return;
  const T* r = x + ((size_t)img * HW + pix) * ldx;
  for (int c = 0; c < C; ++c) y[((size_t)img * C + c) * HW + pix] = to_f32<T>(r[c]);
}
template <typename T>
__global__ void convert_rows_kernel(const float* x, T* y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = from_f32<T>(x[i]);
}
template <typename T>
__global__ void convert_back_kernel(const T* x, float* y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = to_f32<T>(x[i]);
}

struct Temp {
  std::vector<void*> v;
  void* get(size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
    v.push_back(p);
    return p;
  }
  ~Temp() {
    (void)hipDeviceSynchronize();
    for (void* p : v) (void)hipFree(p);
  }
};
__global__ void fastdiv_probe_kernel(const int* n, int count, FastDiv f, int* q) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) q[i] = fd_div(n[i], f);
}
inline size_t es(int dt) { return dt == DT_BF16 ? 2 : 4; }
inline int bke(int dt) { return dt == DT_BF16 ? 64 : 32; }
inline int rupi(int v, int a) { return (v + a - 1) / a * a; }

int to_dev_dtype(const float* x, void* y, size_t n, int dt, hipStream_t s) {
  int g = (int)((n + 255) / 256); if (g > 4096) g = 4096; if (g < 1) g = 1;
  if (dt == DT_BF16) hipLaunchKernelGGL(convert_rows_kernel<bf16_t>, dim3(g), dim3(256), 0, s, x, (bf16_t*)y, n);
  else hipLaunchKernelGGL(convert_rows_kernel<float>, dim3(g), dim3(256), 0, s, x, (float*)y, n);
  return 0;
}
int from_dev_dtype(const void* x, float* y, size_t n, int dt, hipStream_t s) {
  int g = (int)((n + 255) / 256); if (g > 4096) g = 4096; if (g < 1) g = 1;
  if (dt == DT_BF16) hipLaunchKernelGGL(convert_back_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)x, y, n);
  else hipLaunchKernelGGL(convert_back_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)x, y, n);
  return 0;
}
int unpack_nhwc(const void* x, float* y, int B, int C, int HW, int ldx, int dt, hipStream_t s) {
  dim3 grid((HW + 255) / 256, B), block(256);
  if (dt == DT_BF16) hipLaunchKernelGGL(unpack_nhwc_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, y, C, HW, ldx);
  else hipLaunchKernelGGL(unpack_nhwc_kernel<float>, grid, block, 0, s, (const float*)x, y, C, HW, ldx);
  return 0;
}

}  // namespace

extern "C" {

// F.conv2d(cat([x, x2], 1) [nearest-x2 upsampled if up], w, bias, stride, padding=k//2) ; NCHW f32 in/out
int ldmseg_op_conv2d(const float* x, const float* x2, const float* w, const float* bias, int B, int Ci, int Ci2, int H,
                     int W, int Co, int k, int stride, int up, int dtype, float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Temp t;
  const int a = bke(dtype);
  const int c0 = rupi(Ci, a), c1 = Ci2 ? rupi(Ci2, a) : 0;
  if (Ci2 && (Ci % a)) return -2;  // a concat boundary must be K-tile aligned
  void* xp = t.get((size_t)B * H * W * c0 * es(dtype));
  void* x2p = Ci2 ? t.get((size_t)B * H * W * c1 * es(dtype)) : nullptr;
  if (launch_pack_nchw(x, xp, B, Ci, H * W, c0, 1.f, 0.f, dtype, s)) return -3;
  if (Ci2 && launch_pack_nchw(x2, x2p, B, Ci2, H * W, c1, 1.f, 0.f, dtype, s)) return -3;
  // weights: build an OIHW tensor with padded input channels by repacking the two halves separately
  const int bn = igemm_pick_bn(Co, EPI_STORE);
  const int Np = rupi(Co, bn);
  const int ct = c0 + c1;
  void* wp = t.get((size_t)Np * k * k * ct * es(dtype));
  if (!Ci2) {
    if (launch_repack_conv(w, wp, Co, Ci, k, k, Np, c0, dtype, s)) return -3;
  } else {
    // channel-concat weights: [Co][Ci+Ci2][k][k] -> pad each part; do it on the host-free way: two strided repacks
    // (Ci % a == 0 so c0 == Ci; only the second part may be padded)
    std::vector<int> dummy;
    void* w_cat = t.get((size_t)Co * (c0 + c1) * k * k * sizeof(float));
    (void)hipMemsetAsync(w_cat, 0, (size_t)Co * (c0 + c1) * k * k * sizeof(float), s);
    (void)hipMemcpy2DAsync(w_cat, (size_t)(c0 + c1) * k * k * sizeof(float), w, (size_t)(Ci + Ci2) * k * k * sizeof(float),
                           (size_t)(Ci + Ci2) * k * k * sizeof(float), Co, hipMemcpyDeviceToDevice, s);
    if (launch_repack_conv((const float*)w_cat, wp, Co, c0 + c1, k, k, Np, ct, dtype, s)) return -3;
  }
  float* bp = (float*)t.get(Np * sizeof(float));
  (void)hipMemsetAsync(bp, 0, Np * sizeof(float), s);
  if (bias) (void)hipMemcpyAsync(bp, bias, Co * sizeof(float), hipMemcpyDeviceToDevice, s);
  const int Hl = up ? 2 * H : H, Wl = up ? 2 * W : W;
  const int Ho = (k == 3 && stride == 2) ? (Hl - 1) / 2 + 1 : Hl, Wo = (k == 3 && stride == 2) ? (Wl - 1) / 2 + 1 : Wl;
  IgemmParams p;
  p.src0 = xp; p.C0 = c0; p.src1 = x2p; p.C1 = c1;
  p.B = B; p.Hi = H; p.Wi = W; p.Ho = Ho; p.Wo = Wo; p.taps = k * k; p.stride = stride; p.up = up;
  p.M = B * Ho * Wo; p.N = Np; p.n_valid = Co; p.W = wp; p.bias = bp; p.out = out; p.epi = EPI_NCHW_F32;
  if (std::getenv("LDMSEG_OP_TIMING_NHWC")) {   // kernel-timing experiments: the engine's NHWC store epilogue (output discarded)
    p.out = t.get((size_t)p.M * Co * es(dtype)); p.ldo = Co; p.epi = EPI_STORE;
  }
  return launch_igemm(p, dtype, s);
}

// out = proj_out(h + ff.net.2(g)) + x  (diffusers BasicTransformerBlock.ff.net[2] + residual, Transformer2DModel.proj_out + residual;
// /root/reference/ldmseg/models/unet.py:401-425) exactly as the bf16 / fp32 engines launch it at the 640- / 1280-channel levels:
// chained matrix [Wp W2 | Wp] and bias bp + Wp b2 formed in fp32 by launch_chain_weights (the create-time kernel), packed to the
// compute dtype like TransformerW::ffp, then ONE two-source 1x1 igemm over [g | h] with x as the residual (the launch plan, K slices
// and instantiation are the engine's).  g [M][4C], h / x / out [M][C], w2 [C][4C], wp [C][C]; b2 / bp may be NULL.
int ldmseg_op_chained_ff_out(const float* g, const float* h, const float* x, const float* w2, const float* b2, const float* wp,
                             const float* bp, int M, int C, int rows_per_image, int dtype, float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Temp t;
  if (C % bke(dtype) || M < 1 || rows_per_image < 1 || M % rows_per_image) return -2;
  float* wcat = (float*)t.get((size_t)C * 5 * C * sizeof(float));
  float* bcat = (float*)t.get((size_t)C * sizeof(float));
  if (launch_chain_weights(wp, w2, b2, bp, wcat, bcat, C, s)) return -3;
  const int bn = igemm_pick_bn(C, EPI_STORE);
  const int Np = rupi(C, bn);
  if (Np != C) return -2;                                   // (640 and 1280 are multiples of the 160-column tile)
  void* wpk = t.get((size_t)C * 5 * C * es(dtype));
  if (launch_repack_conv(wcat, wpk, C, 5 * C, 1, 1, Np, 5 * C, dtype, s)) return -3;
  void* gp = t.get((size_t)M * 4 * C * es(dtype));
  void* hp = t.get((size_t)M * C * es(dtype));
  void* xp = t.get((size_t)M * C * es(dtype));
  void* op = t.get((size_t)M * C * es(dtype));
  to_dev_dtype(g, gp, (size_t)M * 4 * C, dtype, s);
  to_dev_dtype(h, hp, (size_t)M * C, dtype, s);
  to_dev_dtype(x, xp, (size_t)M * C, dtype, s);
  IgemmParams p;
  p.src0 = gp; p.C0 = 4 * C; p.src1 = hp; p.C1 = C;
  p.B = M / rows_per_image; p.Hi = p.Ho = rows_per_image; p.Wi = p.Wo = 1; p.taps = 1;
  p.M = M; p.N = Np; p.n_valid = C; p.W = wpk; p.bias = bcat;
  p.resid = xp; p.ldr = C; p.out = op; p.ldo = C; p.epi = EPI_STORE;
  p.cf_ctr = op_cf_region();
  const int sp = igemm_plan_splits(p, dtype);
  if (sp > 1) { p.splits = sp; p.partial = (float*)t.get((size_t)sp * M * Np * sizeof(float)); }
  const int r = launch_igemm(p, dtype, s);
  if (r) return r;
  from_dev_dtype(op, out, (size_t)M * C, dtype, s);
  return 0;
}

// y = [silu]( x @ w^T + bias [+ rowbias[row / rows_per_image]] [+ resid] )   or GEGLU when geglu=1 (w is [2*Nout, K])
int ldmseg_op_linear(const float* x, const float* w, const float* bias, const float* resid, const float* rowbias,
                     int rows_per_image, int M, int K, int N, int geglu, int silu, int splits, int dtype, float* out,
                     void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Temp t;
  if (K % bke(dtype)) return -2;
  void* xp = t.get((size_t)M * K * es(dtype));
  to_dev_dtype(x, xp, (size_t)M * K, dtype, s);
  const int epi = geglu ? EPI_GEGLU : EPI_STORE;
  const int bn = igemm_pick_bn(N, epi);
  const int Np = rupi(N, bn);
  void* wp = t.get((size_t)Np * K * es(dtype));
  float* bp = (float*)t.get(Np * sizeof(float));
  (void)hipMemsetAsync(bp, 0, Np * sizeof(float), s);
  const int nout = geglu ? N / 2 : N;
  if (geglu) {
    std::vector<int> map(Np);
    for (int r = 0; r < Np; ++r) {
      const int blk = r / 32, q = r % 32;
      map[r] = (q < 16) ? blk * 16 + q : nout + blk * 16 + (q - 16);
    }
    int* dmap = (int*)t.get(Np * sizeof(int));
    (void)hipMemcpy(dmap, map.data(), Np * sizeof(int), hipMemcpyHostToDevice);
    if (launch_repack_rows(w, wp, dmap, Np, K, dtype, s)) return -3;
    if (bias && launch_repack_rows(bias, bp, dmap, Np, 1, DT_F32, s)) return -3;
  } else {
    if (launch_repack_conv(w, wp, N, K, 1, 1, Np, K, dtype, s)) return -3;
    if (bias) (void)hipMemcpyAsync(bp, bias, N * sizeof(float), hipMemcpyDeviceToDevice, s);
  }
  void* rp = nullptr;
  if (resid) { rp = t.get((size_t)M * N * es(dtype)); to_dev_dtype(resid, rp, (size_t)M * N, dtype, s); }
  void* op = t.get((size_t)M * nout * es(dtype));
  IgemmParams p;
  p.src0 = xp; p.C0 = K; p.B = M / rows_per_image; p.Hi = p.Ho = rows_per_image; p.Wi = p.Wo = 1;
  p.M = M; p.N = Np; p.n_valid = nout; p.W = wp; p.bias = bp;
  p.rowbias = rowbias; p.rb_stride = N;
  p.resid = rp; p.ldr = N; p.out = op; p.ldo = nout; p.epi = epi; p.silu = silu;
  p.cf_ctr = op_cf_region();
  if (splits > 1) { p.splits = splits; p.partial = (float*)t.get((size_t)splits * M * Np * sizeof(float)); }
  const int r = launch_igemm(p, dtype, s);
  if (r) return r;
  from_dev_dtype(op, out, (size_t)M * nout, dtype, s);
  return 0;
}

// F.group_norm(cat([x,x2],1), 32, gamma, beta, eps) [+SiLU]; NCHW f32 in/out
int ldmseg_op_groupnorm(const float* x, const float* x2, const float* gamma, const float* beta, int B, int C, int C2, int HW,
                        float eps, int silu, int dtype, float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Temp t;
  void* xp = t.get((size_t)B * HW * C * es(dtype));
  void* x2p = C2 ? t.get((size_t)B * HW * C2 * es(dtype)) : nullptr;
  if (launch_pack_nchw(x, xp, B, C, HW, C, 1.f, 0.f, dtype, s)) return -3;
  if (C2 && launch_pack_nchw(x2, x2p, B, C2, HW, C2, 1.f, 0.f, dtype, s)) return -3;
  void* op = t.get((size_t)B * HW * (C + C2) * es(dtype));
  GNParams g;
  g.src0 = xp; g.C0 = C; g.src1 = x2p; g.C1 = C2; g.B = B; g.HW = HW; g.gamma = gamma; g.beta = beta; g.eps = eps;
  g.silu = silu; g.out = op; g.nchunk = gn_nchunk(B, HW);
  g.partial = (float*)t.get((size_t)B * g.nchunk * 64 * sizeof(float));
  const int r = launch_groupnorm(g, dtype, s);
  if (r) return r;
  return unpack_nhwc(op, out, B, C + C2, HW, C + C2, dtype, s);
}

// silu(GroupNorm32(conv3x3(x, w, bias) + rowbias[image])) through the engine's fused path: the conv as `splits` K slices without
// its own finish, then launch_finish_groupnorm.  -4: the shape has no fused instantiation (the engine then runs conv and norm apart).
int ldmseg_op_conv_groupnorm(const float* x, const float* w, const float* bias, const float* rowbias, const float* gamma,
                             const float* beta, int B, int Ci, int H, int W, int Co, float eps, int silu, int splits, int dtype,
                             float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Temp t;
  const int a = bke(dtype);
  if (Ci % a || splits < 2) return -2;
  if (!finish_groupnorm_ok(B, H * W, Co, dtype)) return -4;
  void* xp = t.get((size_t)B * H * W * Ci * es(dtype));
  if (launch_pack_nchw(x, xp, B, Ci, H * W, Ci, 1.f, 0.f, dtype, s)) return -3;
  const int bn = igemm_pick_bn(Co, EPI_STORE);
  const int Np = rupi(Co, bn);
  void* wp = t.get((size_t)Np * 9 * Ci * es(dtype));
  if (launch_repack_conv(w, wp, Co, Ci, 3, 3, Np, Ci, dtype, s)) return -3;
  float* bp = (float*)t.get(Np * sizeof(float));
  (void)hipMemsetAsync(bp, 0, Np * sizeof(float), s);
  if (bias) (void)hipMemcpyAsync(bp, bias, Co * sizeof(float), hipMemcpyDeviceToDevice, s);
  void* op = t.get((size_t)B * H * W * Co * es(dtype));
  IgemmParams p;
  p.src0 = xp; p.C0 = Ci; p.B = B; p.Hi = H; p.Wi = W; p.Ho = H; p.Wo = W; p.taps = 9;
  p.M = B * H * W; p.N = Np; p.n_valid = Co; p.W = wp; p.bias = bp; p.rowbias = rowbias; p.rb_stride = Co;
  p.epi = EPI_STORE; p.splits = splits; p.no_finish = 1;
  p.partial = (float*)t.get((size_t)splits * p.M * Np * sizeof(float));
  int r = launch_igemm(p, dtype, s);
  if (r) return r;
  GNParams g;
  g.C0 = Co; g.B = B; g.HW = H * W; g.gamma = gamma; g.beta = beta; g.eps = eps; g.silu = silu; g.out = op;
  r = launch_finish_groupnorm(p, g, dtype, s);
  if (r) return r;
  return unpack_nhwc(op, out, B, Co, H * W, Co, dtype, s);
}

// timing of one GroupNorm launch shape the way the engine launches it (tools/kbench.py gn): average microseconds over
// `iters` back-to-back launches between two HIP events; the input is whatever the allocation holds (statistics of garbage
// cost the same), gamma / beta are [C + C2] device vectors
int ldmseg_bench_groupnorm(const float* gamma, const float* beta, int B, int C, int C2, int HW, int silu, int dtype, int iters,
                           float* us_per_launch, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Temp t;
  void* xp = t.get((size_t)B * HW * C * es(dtype));
  void* x2p = C2 ? t.get((size_t)B * HW * C2 * es(dtype)) : nullptr;
  void* op = t.get((size_t)B * HW * (C + C2) * es(dtype));
  if (!xp || !op || (C2 && !x2p)) return -3;
  (void)hipMemsetAsync(xp, 0x3c, (size_t)B * HW * C * es(dtype), s);
  if (C2) (void)hipMemsetAsync(x2p, 0x3c, (size_t)B * HW * C2 * es(dtype), s);
  GNParams g;
  g.src0 = xp; g.C0 = C; g.src1 = x2p; g.C1 = C2; g.B = B; g.HW = HW; g.gamma = gamma; g.beta = beta; g.eps = 1e-5f;
  g.silu = silu; g.out = op; g.nchunk = gn_nchunk(B, HW);
  g.partial = (float*)t.get((size_t)B * g.nchunk * 64 * sizeof(float));
  for (int i = 0; i < 3; ++i) {
    const int r = launch_groupnorm(g, dtype, s);
    if (r) return r;
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < iters; ++i) (void)launch_groupnorm(g, dtype, s);
  (void)hipEventRecord(e1, s);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *us_per_launch = 1e3f * ms / iters;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return 0;
}

// The fused evaluation tail (ldmseg_vae_decode_panoptic) on a given 4L decoder output x4 [B,C,H4,W4] f32 NCHW: packs it to
// NHWC `dtype` and runs the resample + scan / filter / remap kernels.  volume (optional) receives the resampled logits
// [C][h_b * w_b] of every image back to back (image b at C * offsets[b]) so that the interpolation chain itself can be
// compared with F.interpolate o crop o F.interpolate.
int ldmseg_op_panoptic_from_decoder(const float* x4, int B, int C, int H4, int W4, int dtype, int in_h, int in_w,
                                    const int32_t* boxes, const int32_t* sizes, const int64_t* offsets, int threshold_output,
                                    int threshold_mode, float mask_th, int count_th, double overlap_th, int64_t ignore_label,
                                    int32_t* labels, int32_t* panoptic, uint8_t* keep, int32_t* counts, int32_t* mask_counts,
                                    float* volume, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Temp t;
  void* xp = t.get((size_t)B * H4 * W4 * C * es(dtype));
  if (!xp) return -3;
  if (launch_pack_nchw(x4, xp, B, C, H4 * W4, C, 1.f, 0.f, dtype, s)) return -3;
  return launch_panoptic_from_decoder(xp, B, H4, W4, C, dtype, in_h, in_w, boxes, sizes, offsets, threshold_output,
                                      threshold_mode, mask_th, count_th, overlap_th, ignore_label, labels, panoptic, keep, counts,
                                      mask_counts, s, volume);
}

int ldmseg_op_layernorm(const float* x, const float* gamma, const float* beta, int M, int C, float eps, int silu, int dtype,
                        float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Temp t;
  void* xp = t.get((size_t)M * C * es(dtype));
  to_dev_dtype(x, xp, (size_t)M * C, dtype, s);
  const int r = launch_layernorm(xp, xp, gamma, beta, M, C, eps, silu, dtype, s);
  if (r) return r;
  from_dev_dtype(xp, out, (size_t)M * C, dtype, s);
  return 0;
}

// softmax(q k^T d^-0.5) v per head on fused qkv [B,N,3C] f32 -> [B,N,C] f32
int ldmseg_op_attention(const float* qkv, int B, int N, int C, int heads, int dtype, float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Temp t;
  void* qp = t.get((size_t)B * N * 3 * C * es(dtype));
  to_dev_dtype(qkv, qp, (size_t)B * N * 3 * C, dtype, s);
  void* op = t.get((size_t)B * N * C * es(dtype));
  const int r = launch_attention(qp, op, B, N, C, heads, dtype, s);
  if (r) return r;
  from_dev_dtype(op, out, (size_t)B * N * C, dtype, s);
  return 0;
}

// ConvTranspose2d(k=2,s=2): x NCHW f32 [B,Ci,H,W], w [Ci,Co,2,2] -> NCHW f32 [B,Co,2H,2W]
int ldmseg_op_convt2(const float* x, const float* w, const float* bias, int B, int Ci, int H, int W, int Co, int dtype,
                     float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Temp t;
  if (Ci % bke(dtype) || Co % 4) return -2;
  void* xp = t.get((size_t)B * H * W * Ci * es(dtype));
  if (launch_pack_nchw(x, xp, B, Ci, H * W, Ci, 1.f, 0.f, dtype, s)) return -3;
  const int N = 4 * Co;
  const int bn = igemm_pick_bn(N, EPI_STORE);
  const int Np = rupi(N, bn);
  void* wp = t.get((size_t)Np * Ci * es(dtype));
  (void)hipMemsetAsync(wp, 0, (size_t)Np * Ci * es(dtype), s);
  if (launch_repack_convt2(w, wp, Ci, Co, dtype, s)) return -3;
  float* bp = (float*)t.get(Np * sizeof(float));
  (void)hipMemsetAsync(bp, 0, Np * sizeof(float), s);
  if (bias) for (int q = 0; q < 4; ++q) (void)hipMemcpyAsync(bp + q * Co, bias, Co * sizeof(float), hipMemcpyDeviceToDevice, s);
  void* op = t.get((size_t)B * 4 * H * W * Co * es(dtype));
  IgemmParams p;
  p.src0 = xp; p.C0 = Ci; p.B = B; p.Hi = p.Ho = H; p.Wi = p.Wo = W; p.M = B * H * W; p.N = Np; p.n_valid = N;
  p.W = wp; p.bias = bp; p.out = op; p.ldo = Co; p.epi = EPI_CONVT2; p.cout = Co;
  const int r = launch_igemm(p, dtype, s);
  if (r) return r;
  return unpack_nhwc(op, out, B, Co, 4 * H * W, Co, dtype, s);
}

// F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False): NCHW f32 in/out
int ldmseg_op_bilinear2x(const float* x, int B, int C, int H, int W, int dtype, float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Temp t;
  void* xp = t.get((size_t)B * H * W * C * es(dtype));
  if (launch_pack_nchw(x, xp, B, C, H * W, C, 1.f, 0.f, dtype, s)) return -3;
  return launch_bilinear2x_nchw(xp, out, B, H, W, C, dtype, s);
}

// The engine's launch of one conv / GEMM layer, exactly as engine.hip issues it in a forward: NHWC operands in the compute
// dtype, the split-K plan of igemm_plan_splits (splits = 0) or a forced one, the row-major store epilogue with optional
// bias, per-image bias row (time embedding), residual and SiLU, or the GEGLU epilogue.  Boundary tensors are NCHW f32:
// x [B,Ci,H,W], x2 [B,Ci2,H,W] or NULL, w [Co,Ci+Ci2,k,k], resid [B,Cout,Ho,Wo] or NULL, rowbias [B,Co] or NULL,
// out [B,Cout,Ho,Wo] with Cout = Co (Co/2 for GEGLU, whose w rows are [value | gate] like ff.net.0.proj).
static int g_bench_rot = 1, g_bench_ln = 0;
static inline bool sp_gt1(int sp) { return sp > 1; }
static int op_igemm_impl(const float* x, const float* x2, const float* w, const float* bias, const float* resid,
                         const float* rowbias, int B, int Ci, int Ci2, int H, int W, int Co, int k, int stride, int up, int geglu,
                         int silu, int splits, int dtype, float* out, void* stream, int time_iters, float* us_per_launch) {
  hipStream_t s = (hipStream_t)stream;
  Temp t;
  const int x3 = dtype == 2 ? 1 : (dtype == 3 ? 2 : 0);   // LDMSEG_BF16X3: fp32 storage, split-bf16 products; 3 = the same with the weights
  if (x3) dtype = DT_F32;                                 // split into hi | lo planes up front, as the handles hold them (round 6)
  const int a = bke(dtype);
  const int c0 = rupi(Ci, a), c1 = Ci2 ? rupi(Ci2, a) : 0;
  if (Ci2 && (Ci % a)) return -2;
  if (geglu && (k != 1 || Ci2 || (Co % 32))) return -2;
  void* xp = t.get((size_t)B * H * W * c0 * es(dtype));
  void* x2p = Ci2 ? t.get((size_t)B * H * W * c1 * es(dtype)) : nullptr;
  if (launch_pack_nchw(x, xp, B, Ci, H * W, c0, 1.f, 0.f, dtype, s)) return -3;
  if (Ci2 && launch_pack_nchw(x2, x2p, B, Ci2, H * W, c1, 1.f, 0.f, dtype, s)) return -3;
  const int epi = geglu ? EPI_GEGLU : EPI_STORE;
  const int bn = igemm_pick_bn(Co, epi);
  const int Np = rupi(Co, bn);
  const int ct = c0 + c1;
  const int cout = geglu ? Co / 2 : Co;
  const bool cm = !geglu && igemm_conv_cm(H * W, c0 + c1, Np, k, stride, up, dtype) && (Ci % a) == 0 && (!Ci2 || (Ci2 % a) == 0);   // the engine's rule
  void* wp = t.get((size_t)Np * k * k * ct * es(dtype));
  float* bp = (float*)t.get(Np * sizeof(float));
  (void)hipMemsetAsync(bp, 0, Np * sizeof(float), s);
  if (geglu) {
    std::vector<int> map(Np);
    for (int r = 0; r < Np; ++r) {
      const int blk = r / 32, q = r % 32;
      map[r] = (q < 16) ? blk * 16 + q : cout + blk * 16 + (q - 16);
    }
    int* dmap = (int*)t.get(Np * sizeof(int));
    (void)hipMemcpy(dmap, map.data(), Np * sizeof(int), hipMemcpyHostToDevice);
    if (launch_repack_rows(w, wp, dmap, Np, ct, dtype, s)) return -3;
    if (bias && launch_repack_rows(bias, bp, dmap, Np, 1, DT_F32, s)) return -3;
  } else if (!Ci2) {
    if (launch_repack_conv(w, wp, Co, Ci, k, k, Np, c0, dtype, s, cm ? a : 0)) return -3;
    if (bias) (void)hipMemcpyAsync(bp, bias, Co * sizeof(float), hipMemcpyDeviceToDevice, s);
  } else {
    void* w_cat = t.get((size_t)Co * ct * k * k * sizeof(float));
    (void)hipMemsetAsync(w_cat, 0, (size_t)Co * ct * k * k * sizeof(float), s);
    (void)hipMemcpy2DAsync(w_cat, (size_t)ct * k * k * sizeof(float), w, (size_t)(Ci + Ci2) * k * k * sizeof(float),
                           (size_t)(Ci + Ci2) * k * k * sizeof(float), Co, hipMemcpyDeviceToDevice, s);
    if (launch_repack_conv((const float*)w_cat, wp, Co, ct, k, k, Np, ct, dtype, s, cm ? a : 0)) return -3;
    if (bias) (void)hipMemcpyAsync(bp, bias, Co * sizeof(float), hipMemcpyDeviceToDevice, s);
  }
  const int Hl = up ? 2 * H : H, Wl = up ? 2 * W : W;
  const int Ho = (k == 3 && stride == 2) ? (Hl - 1) / 2 + 1 : Hl, Wo = (k == 3 && stride == 2) ? (Wl - 1) / 2 + 1 : Wl;
  void* rp = nullptr;
  if (resid) {
    rp = t.get((size_t)B * Ho * Wo * cout * es(dtype));
    if (launch_pack_nchw(resid, rp, B, cout, Ho * Wo, cout, 1.f, 0.f, dtype, s)) return -3;
  }
  void* op = t.get((size_t)B * Ho * Wo * cout * es(dtype));
  IgemmParams p;
  p.src0 = xp; p.C0 = c0; p.src1 = x2p; p.C1 = c1;
  p.B = B; p.Hi = H; p.Wi = W; p.Ho = Ho; p.Wo = Wo; p.taps = k * k; p.stride = stride; p.up = up; p.cm = cm ? 1 : 0;
  p.M = B * Ho * Wo; p.N = Np; p.n_valid = cout; p.W = wp; p.bias = bp;
  p.rowbias = rowbias; p.rb_stride = Co;
  p.resid = rp; p.ldr = cout; p.out = op; p.ldo = cout; p.epi = epi; p.silu = silu;
  p.x3 = x3;
  if (x3 == 2 && launch_split_planes(wp, (size_t)Np * k * k * ct, s)) return -3;
  if (up && k == 3 && stride == 1 && !Ci2 && !geglu && !resid && !rowbias && !silu && igemm_up4_ok(B, H, W, c0, Np, dtype) && Ci == c0) {
    // the engine's form of an upsampler conv: four 2x2 phase convs on the low-resolution map (IgemmParams::up4)
    void* w4 = t.get((size_t)16 * Np * c0 * es(dtype));
    if (launch_pack_up4(w, w4, Co, Ci, Np, c0, dtype, s)) return -3;
    p.W = w4; p.taps = 4; p.up = 0; p.up4 = 1; p.Ho = H; p.Wo = W; p.M = 4 * B * H * W; p.cm = 0;
  }
  p.cf_ctr = op_cf_region();
  int sp = splits > 0 ? splits : igemm_plan_splits(p, dtype);
  if (sp > 1) { p.splits = sp; p.partial = (float*)t.get((size_t)sp * p.M * Np * sizeof(float)); }
  if (time_iters > 0 && g_bench_ln && !sp_gt1(sp) && !rowbias) {   // time the folded-LayerNorm instantiation (mean 0, rstd 1, c1 0)
    float* st = (float*)t.get((size_t)p.M * 2 * sizeof(float));
    float* c1z = (float*)t.get(Np * sizeof(float));
    (void)hipMemsetAsync(c1z, 0, Np * sizeof(float), s);
    std::vector<float> h((size_t)p.M * 2);
    for (int m = 0; m < p.M; ++m) { h[2 * m] = 0.f; h[2 * m + 1] = 1.f; }
    (void)hipMemcpy(st, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    p.rowstats = st; p.c1 = c1z;
  }
  const int r = launch_igemm(p, dtype, s);
  if (r) return r;
  if (time_iters > 0 && us_per_launch) {        // kernel timing (tools/): launches back to back on the stream, HIP events around
    // the weights of a layer are cold in the real forward (1.6 GB of them stream through per step): rotate over copies
    const size_t wbytes = p.up4 ? (size_t)16 * Np * c0 * es(dtype) : (size_t)Np * k * k * ct * es(dtype);
    const void* wsrc = p.W;                       // (the up4 launch reads its own packing)
    int rot = g_bench_rot < 1 ? 1 : g_bench_rot;
    std::vector<const void*> wc(1, wsrc);
    for (int i = 1; i < rot; ++i) {
      void* c = t.get(wbytes);
      if (!c) break;
      (void)hipMemcpyAsync(c, wsrc, wbytes, hipMemcpyDeviceToDevice, s);
      wc.push_back(c);
    }
    rot = (int)wc.size();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) { p.W = wc[i % rot]; (void)launch_igemm(p, dtype, s); }
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < time_iters; ++i) { p.W = wc[(i + 3) % rot]; (void)launch_igemm(p, dtype, s); }
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *us_per_launch = 1e3f * ms / time_iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  if (!out) return 0;
  return unpack_nhwc(op, out, B, cout, Ho * Wo, cout, dtype, s);
}

// conv2d(h, w2, b2, padding=1) + conv2d(cat([xs, xs2], 1), ws, bs) - the tail of a diffusers ResnetBlock2D with a conv_shortcut -
// as the engine runs it in bf16: ONE implicit-GEMM launch whose K range continues past the nine taps with the shortcut's input
// channels read at the output pixel (IgemmParams::src2 / src3).  xs2 may be NULL (Cs2 = 0).  iters > 0: timing like
// ldmseg_bench_igemm (average microseconds per launch incl. the split-K finish).  Returns -4 when the shape has no such launch.
int ldmseg_op_conv3x3_plus_1x1(const float* h, const float* w2, const float* b2, const float* xs, const float* xs2, const float* ws,
                               const float* bs, int B, int C, int Cs, int Cs2, int H, int W, int Co, int splits, int dtype, float* out,
                               int iters, float* us_per_launch, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Temp t;
  if (dtype != DT_BF16 || C % 64 || Cs % 64 || Cs2 % 64 || !h || !w2 || !xs || !ws) return -2;
  const int HW = H * W, cs = Cs + Cs2;
  void* hp = t.get((size_t)B * HW * C * 2);
  void* xp = t.get((size_t)B * HW * Cs * 2);
  void* x2p = Cs2 ? t.get((size_t)B * HW * Cs2 * 2) : nullptr;
  if (launch_pack_nchw(h, hp, B, C, HW, C, 1.f, 0.f, dtype, s)) return -3;
  if (launch_pack_nchw(xs, xp, B, Cs, HW, Cs, 1.f, 0.f, dtype, s)) return -3;
  if (Cs2 && launch_pack_nchw(xs2, x2p, B, Cs2, HW, Cs2, 1.f, 0.f, dtype, s)) return -3;
  const int Np = rupi(Co, igemm_pick_bn(Co, EPI_STORE));
  void* w2p = t.get((size_t)Np * 9 * C * 2);
  void* wsp = t.get((size_t)Np * cs * 2);
  void* wx = t.get((size_t)Np * (9 * C + cs) * 2);
  if (launch_repack_conv(w2, w2p, Co, C, 3, 3, Np, C, dtype, s)) return -3;
  if (launch_repack_conv(ws, wsp, Co, cs, 1, 1, Np, cs, dtype, s)) return -3;
  if (launch_concat_rows(w2p, 9 * C, wsp, cs, wx, Np, dtype, s)) return -3;
  float* b2p = (float*)t.get(Np * sizeof(float));
  float* bsp = (float*)t.get(Np * sizeof(float));
  float* bx = (float*)t.get(Np * sizeof(float));
  (void)hipMemsetAsync(b2p, 0, Np * sizeof(float), s);
  (void)hipMemsetAsync(bsp, 0, Np * sizeof(float), s);
  if (b2) (void)hipMemcpyAsync(b2p, b2, Co * sizeof(float), hipMemcpyDeviceToDevice, s);
  if (bs) (void)hipMemcpyAsync(bsp, bs, Co * sizeof(float), hipMemcpyDeviceToDevice, s);
  if (launch_vec_add(b2p, bsp, bx, Np, s)) return -3;
  void* op = t.get((size_t)B * HW * Co * 2);
  IgemmParams p;
  p.src0 = hp; p.C0 = C; p.src2 = xp; p.C2 = Cs; p.src3 = x2p; p.C3 = Cs2;
  p.B = B; p.Hi = p.Ho = H; p.Wi = p.Wo = W; p.taps = 9; p.stride = 1;
  p.M = B * HW; p.N = Np; p.n_valid = Co; p.W = wx; p.bias = bx; p.out = op; p.ldo = Co; p.epi = EPI_STORE;
  if (!igemm_xt_ok(p, dtype)) return -4;
  p.cf_ctr = op_cf_region();
  const int sp = splits > 0 ? splits : igemm_plan_splits(p, dtype);
  if (sp > 1) { p.splits = sp; p.partial = (float*)t.get((size_t)sp * p.M * Np * sizeof(float)); }
  const int r = launch_igemm(p, dtype, s);
  if (r) return r;
  if (iters > 0 && us_per_launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) (void)launch_igemm(p, dtype, s);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *us_per_launch = 1e3f * ms / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  if (!out) return 0;
  return unpack_nhwc(op, out, B, Co, HW, Co, dtype, s);
}

// q[i] = n[i] / d through the multiply-shift divisor the kernels use (FastDiv, kernels.h): exactness test hook
int ldmseg_op_fastdiv(const int* n, int count, int d, int* q, void* stream) {
  if (!n || !q || count < 0 || d < 1) return -2;
  const FastDiv f = fastdiv_make(d);
  hipLaunchKernelGGL(fastdiv_probe_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, count, f, q);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
}  // extern "C"
namespace ldmseg {
void ops_bench_knob(int key, int value) {
  if (key == 6) g_bench_rot = value;
  if (key == 7) g_bench_ln = value;
}
}  // namespace ldmseg
extern "C" {
int ldmseg_op_igemm(const float* x, const float* x2, const float* w, const float* bias, const float* resid,
                    const float* rowbias, int B, int Ci, int Ci2, int H, int W, int Co, int k, int stride, int up, int geglu,
                    int silu, int splits, int dtype, float* out, void* stream) {
  return op_igemm_impl(x, x2, w, bias, resid, rowbias, B, Ci, Ci2, H, W, Co, k, stride, up, geglu, silu, splits, dtype, out,
                       stream, 0, nullptr);
}
// same launch, then `iters` more back to back with HIP events around them: average microseconds per launch (incl. the
// split-K finish kernel when the plan has one).  Tuning tool (tools/kbench.py), not a product call.
int ldmseg_bench_igemm(const float* x, const float* x2, const float* w, const float* bias, const float* resid,
                       const float* rowbias, int B, int Ci, int Ci2, int H, int W, int Co, int k, int stride, int up, int geglu,
                       int silu, int splits, int dtype, int iters, float* us_per_launch, void* stream) {
  return op_igemm_impl(x, x2, w, bias, resid, rowbias, B, Ci, Ci2, H, W, Co, k, stride, up, geglu, silu, splits, dtype, nullptr,
                       stream, iters, us_per_launch);
}
int ldmseg_bench_attention(const float* qkv, int B, int N, int C, int heads, int dtype, int iters, float* us_per_launch,
                           void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Temp t;
  void* qp = t.get((size_t)B * N * 3 * C * es(dtype));
  to_dev_dtype(qkv, qp, (size_t)B * N * 3 * C, dtype, s);
  void* op = t.get((size_t)B * N * C * es(dtype));
  for (int i = 0; i < 3; ++i) {
    const int r = launch_attention(qp, op, B, N, C, heads, dtype, s);
    if (r) return r;
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < iters; ++i) (void)launch_attention(qp, op, B, N, C, heads, dtype, s);
  (void)hipEventRecord(e1, s);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *us_per_launch = 1e3f * ms / iters;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return 0;
}

// F.linear(F.layer_norm(x, (K,), gamma, beta, eps), w, bias) - or the GEGLU feed-forward half on the normalised input when
// geglu = 1 - computed the way the engine computes norm1 -> q|k|v and norm3 -> ff.net.0: statistics pass + GEMM on the raw x
// with the LayerNorm folded into weights and epilogue (IgemmParams::rowstats).  x [M,K], w [N,K], out [M, N or N/2], f32.
int ldmseg_op_ln_linear(const float* x, const float* gamma, const float* beta, const float* w, const float* bias, int M, int K,
                        int N, float eps, int geglu, int dtype, float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Temp t;
  const int x3 = dtype == 2 ? 1 : (dtype == 3 ? 2 : 0);   // LDMSEG_BF16X3 (3: weights pre-split into planes)
  if (x3) dtype = DT_F32;
  if (K % bke(dtype)) return -2;
  void* xp = t.get((size_t)M * K * es(dtype));
  to_dev_dtype(x, xp, (size_t)M * K, dtype, s);
  const int epi = geglu ? EPI_GEGLU : EPI_STORE;
  const int bn = igemm_pick_bn(N, epi);
  const int Np = rupi(N, bn);
  const int nout = geglu ? N / 2 : N;
  std::vector<int> map(Np);
  for (int r = 0; r < Np; ++r) {
    if (geglu) { const int blk = r / 32, q = r % 32; map[r] = (q < 16) ? blk * 16 + q : nout + blk * 16 + (q - 16); }
    else map[r] = r < N ? r : -1;
  }
  int* dmap = (int*)t.get(Np * sizeof(int));
  (void)hipMemcpy(dmap, map.data(), Np * sizeof(int), hipMemcpyHostToDevice);
  void* wp = t.get((size_t)Np * K * es(dtype));
  void* wb = t.get((size_t)Np * K * sizeof(float));
  float* pb = (float*)t.get(Np * sizeof(float));
  float* c1 = (float*)t.get(Np * sizeof(float));
  float* c2 = (float*)t.get(Np * sizeof(float));
  (void)hipMemsetAsync(pb, 0, Np * sizeof(float), s);
  if (launch_repack_rows_scaled(w, wp, dmap, Np, K, gamma, dtype, s)) return -3;
  if (launch_repack_rows_scaled(w, wb, dmap, Np, K, beta, DT_F32, s)) return -3;
  if (bias && launch_repack_rows(bias, pb, dmap, Np, 1, DT_F32, s)) return -3;
  if (launch_rowsum(wp, nullptr, c1, Np, K, dtype, s)) return -3;
  if (launch_rowsum(wb, pb, c2, Np, K, DT_F32, s)) return -3;
  float* stats = (float*)t.get((size_t)M * 2 * sizeof(float));
  if (launch_rowstats(xp, stats, M, K, eps, dtype, s)) return -3;
  void* op = t.get((size_t)M * nout * es(dtype));
  IgemmParams p;
  p.src0 = xp; p.C0 = K; p.B = 1; p.Hi = p.Ho = M; p.Wi = p.Wo = 1;
  p.M = M; p.N = Np; p.n_valid = nout; p.W = wp; p.bias = c2; p.rowstats = stats; p.c1 = c1;
  p.out = op; p.ldo = nout; p.epi = epi;
  p.x3 = x3;
  if (x3 == 2 && launch_split_planes(wp, (size_t)Np * K, s)) return -3;      // (c1 / c2 above were taken from the fp32 matrix)
  const int sp = igemm_plan_splits(p, dtype);
  if (sp > 1) { p.splits = sp; p.partial = (float*)t.get((size_t)sp * M * Np * sizeof(float)); }
  const int r = launch_igemm(p, dtype, s);
  if (r) return r;
  from_dev_dtype(op, out, (size_t)M * nout, dtype, s);
  return 0;
}

// The tail of a transformer block on [M, C] token rows (diffusers BasicTransformerBlock + Transformer2DModel.proj_out,
// /root/reference/ldmseg/models/unet.py:401-425):
//     h2  = h + F.linear(GEGLU(F.linear(F.layer_norm(h, (C,), gamma, beta, eps), w1, b1)), w2, b2)
//     out = F.linear(h2, wp, bp) + x
// mode 0: the unfused launches (rowstats + folded-LayerNorm GEGLU GEMM + ff.net.2 GEMM + proj_out GEMM); 1: the row-local
// fused kernel for the feed-forward (tfuse.hip) + proj_out GEMM; 3: everything in the fused kernel.  bf16, C = 320 for 1 / 3.
// time_iters > 0 also times the chosen path (us per call of the whole tail).
int ldmseg_op_transformer_ff(const float* h, const float* x, const float* gamma, const float* beta, const float* w1, const float* b1,
                             const float* w2, const float* b2, const float* wp, const float* bp, int M, int C, float eps, int dtype,
                             int mode, float* out, int time_iters, float* us_per_call, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Temp t;
  const int rows_per_image = (M % 4096 == 0) ? 4096 : 0;      // (tiles rotate their chunk sweep by the index inside a 64x64 image)
  if (C % bke(dtype) || (mode != 0 && (dtype != DT_BF16 || !mlp_fused_stream_bytes(C)))) return -2;
  const int N1 = 8 * C;
  void* hp = t.get((size_t)M * C * es(dtype));
  void* h0 = t.get((size_t)M * C * es(dtype));
  void* xp = t.get((size_t)M * C * es(dtype));
  void* op = t.get((size_t)M * C * es(dtype));
  to_dev_dtype(h, h0, (size_t)M * C, dtype, s);
  to_dev_dtype(x, xp, (size_t)M * C, dtype, s);
  std::vector<int> map(N1), ident(C);
  for (int r = 0; r < N1; ++r) { const int blk = r / 32, q = r % 32; map[r] = (q < 16) ? blk * 16 + q : 4 * C + blk * 16 + (q - 16); }
  for (int r = 0; r < C; ++r) ident[r] = r;
  int* dmap = (int*)t.get(N1 * sizeof(int));
  int* dident = (int*)t.get(C * sizeof(int));
  (void)hipMemcpy(dmap, map.data(), N1 * sizeof(int), hipMemcpyHostToDevice);
  (void)hipMemcpy(dident, ident.data(), C * sizeof(int), hipMemcpyHostToDevice);
  void* w1p = t.get((size_t)N1 * C * es(dtype));
  void* w1b = t.get((size_t)N1 * C * sizeof(float));
  float* pb = (float*)t.get(N1 * sizeof(float));
  float* c1 = (float*)t.get(N1 * sizeof(float));
  float* c2 = (float*)t.get(N1 * sizeof(float));
  if (launch_repack_rows_scaled(w1, w1p, dmap, N1, C, gamma, dtype, s)) return -3;
  if (launch_repack_rows_scaled(w1, w1b, dmap, N1, C, beta, DT_F32, s)) return -3;
  if (launch_repack_rows(b1, pb, dmap, N1, 1, DT_F32, s)) return -3;
  if (launch_rowsum(w1p, nullptr, c1, N1, C, dtype, s)) return -3;
  if (launch_rowsum(w1b, pb, c2, N1, C, DT_F32, s)) return -3;
  void* w2p = t.get((size_t)C * 4 * C * es(dtype));
  void* wpp = t.get((size_t)C * C * es(dtype));
  to_dev_dtype(w2, w2p, (size_t)C * 4 * C, dtype, s);        // Linear weights are already [N][K]
  to_dev_dtype(wp, wpp, (size_t)C * C, dtype, s);
  float* b2d = (float*)t.get(C * sizeof(float));
  float* bpd = (float*)t.get(C * sizeof(float));
  (void)hipMemcpyAsync(b2d, b2, C * sizeof(float), hipMemcpyDeviceToDevice, s);
  (void)hipMemcpyAsync(bpd, bp, C * sizeof(float), hipMemcpyDeviceToDevice, s);
  void* stream_w = nullptr;
  if (mode != 0) {
    stream_w = t.get(mlp_fused_stream_bytes(C));
    if (launch_pack_mlp_stream(w1p, w2p, wpp, stream_w, C, s)) return -3;
  }
  float* stats = (float*)t.get((size_t)M * 2 * sizeof(float));
  void* ff = t.get((size_t)M * 4 * C * es(dtype));
  if (igemm_warm()) return -3;
  auto gemm = [&](const void* src, int K, const void* W, int N, int nvalid, const float* bias, const void* resid, void* o, int epi,
                  const float* rs, const float* cc1) -> int {
    IgemmParams p;
    p.src0 = src; p.C0 = K; p.B = 1; p.Hi = p.Ho = M; p.Wi = p.Wo = 1;
    p.M = M; p.N = N; p.n_valid = nvalid; p.W = W; p.bias = bias; p.rowstats = rs; p.c1 = cc1;
    p.resid = resid; p.ldr = C; p.out = o; p.ldo = nvalid; p.epi = epi;
    const int sp = igemm_plan_splits(p, dtype);
    if (sp > 1) { p.splits = sp; p.partial = (float*)t.get((size_t)sp * M * N * sizeof(float)); }
    return launch_igemm(p, dtype, s);
  };
  auto run = [&]() -> int {
    (void)hipMemcpyAsync(hp, h0, (size_t)M * C * es(dtype), hipMemcpyDeviceToDevice, s);   // the tail updates h in place
    if (mode == 0) {
      if (int r = launch_rowstats(hp, stats, M, C, eps, dtype, s)) return r;
      if (int r = gemm(hp, C, w1p, N1, 4 * C, c2, nullptr, ff, EPI_GEGLU, stats, c1)) return r;
      if (int r = gemm(ff, 4 * C, w2p, C, C, b2d, hp, hp, EPI_STORE, nullptr, nullptr)) return r;
      return gemm(hp, C, wpp, C, C, bpd, xp, op, EPI_STORE, nullptr, nullptr);
    }
    const int proj = (mode & 2) ? 1 : 0;
    if (int r = launch_mlp_fused(hp, proj ? op : hp, xp, stream_w, c2, b2d, bpd, igemm_zero_page(), M, C, eps, proj, rows_per_image, s)) return r;
    return proj ? 0 : gemm(hp, C, wpp, C, C, bpd, xp, op, EPI_STORE, nullptr, nullptr);
  };
  if (int r = run()) return r;
  if (time_iters > 0 && us_per_call) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) (void)run();
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < time_iters; ++i) (void)run();
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *us_per_call = 1e3f * ms / time_iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  from_dev_dtype(op, out, (size_t)M * C, dtype, s);
  return 0;
}

// The entry of a transformer on caller-supplied weights: h = x Wp^T + bp (proj_in as a Linear on [M][C] rows), q|k|v = [Wq | Wk | Wv]
// LayerNorm(h; gamma, beta).  mode 0: the unfused launches (GEMM, row statistics, folded-LayerNorm GEMM); 1: the row-local fused
// kernel (tproj.hip; bf16, C = 320, M % 128 == 0).  Outputs fp32 h [M][C], qkv [M][3C].  time_iters > 0 also times the path.
static int transformer_in_impl(const float* x, const float* gn_gamma, const float* gn_beta, float gn_eps, int gn_images, int gn_mode,
                               const float* wp, const float* bp, const float* gamma, const float* beta, const float* wq,
                               const float* wk, const float* wv, int M, int C, float eps, int dtype, int mode, float* h_out, float* qkv_out,
                               int time_iters, float* us_per_call, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Temp t;
  if (C % bke(dtype) || (mode != 0 && (dtype != DT_BF16 || !proj_qkv_stream_bytes(C) || M % 128))) return -2;
  // GroupNorm in front (gn_gamma != null; x = [gn_images][HW][C] rows): gn_mode 0 = a launch of its own, 1 = statistics pass + the
  // sweep inside the fused kernel (mode 1 only)
  GNParams gp;
  GnFold gf;
  void* xn = nullptr;
  if (gn_gamma) {
    if (gn_images < 1 || M % gn_images || C % 32 || (gn_mode == 1 && (mode != 1 || !proj_qkv_gn_fold_ok(M / gn_images)))) return -2;
    xn = t.get((size_t)M * C * es(dtype));
    float* gg = (float*)t.get(C * sizeof(float));
    float* gb = (float*)t.get(C * sizeof(float));
    (void)hipMemcpyAsync(gg, gn_gamma, C * sizeof(float), hipMemcpyDeviceToDevice, s);
    (void)hipMemcpyAsync(gb, gn_beta, C * sizeof(float), hipMemcpyDeviceToDevice, s);
    gp.C0 = C; gp.B = gn_images; gp.HW = M / gn_images; gp.groups = 32; gp.gamma = gg; gp.beta = gb; gp.eps = gn_eps; gp.silu = 0;
    gp.out = xn;
    gp.nchunk = gn_mode == 1 ? proj_qkv_gn_chunks(gp.B, gp.HW) : gn_nchunk(gp.B, gp.HW);
    gp.partial = (float*)t.get((size_t)gp.B * gp.nchunk * 32 * 2 * sizeof(float));
    gf.partial = gp.partial; gf.gamma = gg; gf.beta = gb; gf.nchunk = gp.nchunk; gf.HW = gp.HW; gf.eps = gn_eps;
  }
  void* xp = t.get((size_t)M * C * es(dtype));
  void* hp = t.get((size_t)M * C * es(dtype));
  void* qp = t.get((size_t)M * 3 * C * es(dtype));
  to_dev_dtype(x, xp, (size_t)M * C, dtype, s);
  std::vector<int> ident(C);
  for (int r = 0; r < C; ++r) ident[r] = r;
  int* dident = (int*)t.get(C * sizeof(int));
  (void)hipMemcpy(dident, ident.data(), C * sizeof(int), hipMemcpyHostToDevice);
  void* wpp = t.get((size_t)C * C * es(dtype));
  to_dev_dtype(wp, wpp, (size_t)C * C, dtype, s);
  void* wqkv = t.get((size_t)3 * C * C * es(dtype));
  void* wb = t.get((size_t)3 * C * C * sizeof(float));
  const float* ws3[3] = {wq, wk, wv};
  for (int i = 0; i < 3; ++i) {
    if (launch_repack_rows_scaled(ws3[i], (char*)wqkv + (size_t)i * C * C * es(dtype), dident, C, C, gamma, dtype, s)) return -3;
    if (launch_repack_rows_scaled(ws3[i], (char*)wb + (size_t)i * C * C * sizeof(float), dident, C, C, beta, DT_F32, s)) return -3;
  }
  float* bias4 = (float*)t.get((size_t)4 * C * sizeof(float));
  float* c1 = (float*)t.get((size_t)3 * C * sizeof(float));
  (void)hipMemcpyAsync(bias4, bp, C * sizeof(float), hipMemcpyDeviceToDevice, s);
  if (launch_rowsum(wqkv, nullptr, c1, 3 * C, C, dtype, s)) return -3;
  if (launch_rowsum(wb, nullptr, bias4 + C, 3 * C, C, DT_F32, s)) return -3;
  void* stream_w = nullptr;
  if (mode != 0) {
    stream_w = t.get(proj_qkv_stream_bytes(C));
    if (launch_pack_proj_qkv_stream(wpp, wqkv, stream_w, C, s)) return -3;
  }
  float* stats = (float*)t.get((size_t)M * 2 * sizeof(float));
  if (igemm_warm()) return -3;
  auto gemm = [&](const void* src, const void* W, int N, const float* bias, void* o, const float* rs, const float* cc1) -> int {
    IgemmParams p;
    p.src0 = src; p.C0 = C; p.B = 1; p.Hi = p.Ho = M; p.Wi = p.Wo = 1;
    p.M = M; p.N = N; p.n_valid = N; p.W = W; p.bias = bias; p.rowstats = rs; p.c1 = cc1;
    p.out = o; p.ldo = N; p.epi = EPI_STORE;
    const int sp = igemm_plan_splits(p, dtype);
    if (sp > 1) { p.splits = sp; p.partial = (float*)t.get((size_t)sp * M * N * sizeof(float)); }
    return launch_igemm(p, dtype, s);
  };
  auto run = [&]() -> int {
    const void* xin = xp;
    if (gn_gamma) {
      gp.src0 = xp;
      if (gn_mode == 1) {
        if (int r = launch_groupnorm_stats(gp, dtype, s)) return r;
        return launch_proj_qkv_fused(xp, hp, qp, stream_w, bias4, igemm_zero_page(), M, C, eps, &gf, s);
      }
      if (int r = launch_groupnorm(gp, dtype, s)) return r;
      xin = xn;
    }
    if (mode == 0) {
      if (int r = gemm(xin, wpp, C, bias4, hp, nullptr, nullptr)) return r;
      if (int r = launch_rowstats(hp, stats, M, C, eps, dtype, s)) return r;
      return gemm(hp, wqkv, 3 * C, bias4 + C, qp, stats, c1);
    }
    return launch_proj_qkv_fused(xin, hp, qp, stream_w, bias4, igemm_zero_page(), M, C, eps, nullptr, s);
  };
  if (int r = run()) return r;
  if (time_iters > 0 && us_per_call) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) (void)run();
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < time_iters; ++i) (void)run();
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *us_per_call = 1e3f * ms / time_iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  from_dev_dtype(hp, h_out, (size_t)M * C, dtype, s);
  from_dev_dtype(qp, qkv_out, (size_t)M * 3 * C, dtype, s);
  return 0;
}
int ldmseg_op_transformer_in(const float* x, const float* wp, const float* bp, const float* gamma, const float* beta, const float* wq,
                             const float* wk, const float* wv, int M, int C, float eps, int dtype, int mode, float* h_out, float* qkv_out,
                             int time_iters, float* us_per_call, void* stream) {
  return transformer_in_impl(x, nullptr, nullptr, 0.f, 0, 0, wp, bp, gamma, beta, wq, wk, wv, M, C, eps, dtype, mode, h_out, qkv_out,
                             time_iters, us_per_call, stream);
}
// The same behind the transformer's GroupNorm (32 groups, no SiLU) over x = [images][M / images][C] rows: gn_mode 0 = a GroupNorm
// launch, then the entry as above; gn_mode 1 (mode 1 only) = statistics pass + the apply sweep inside the fused kernel (tproj.hip).
int ldmseg_op_gn_transformer_in(const float* x, const float* gn_gamma, const float* gn_beta, float gn_eps, int images, int gn_mode,
                                const float* wp, const float* bp, const float* gamma, const float* beta, const float* wq,
                                const float* wk, const float* wv, int M, int C, float eps, int dtype, int mode, float* h_out,
                                float* qkv_out, int time_iters, float* us_per_call, void* stream) {
  if (!gn_gamma || !gn_beta) return -2;
  return transformer_in_impl(x, gn_gamma, gn_beta, gn_eps, images, gn_mode, wp, bp, gamma, beta, wq, wk, wv, M, C, eps, dtype, mode, h_out,
                             qkv_out, time_iters, us_per_call, stream);
}

// The step tail kernel (tail.hip, bf16): eps = conv2d(x, w, bias, padding=1) with 320 -> 4 channels on [B,320,H,W] and, when
// ddim != 0, the scheduler update of `latents` (in place; last != 0: pred_original_sample), the inpainting paste, the
// self-condition write and the next step's packed input (returned as fp32 [B, H*W, 64]).  eps_out / cond / xin_out / known
// may be null.  coef4 = {sqrt_alpha_t, sqrt_beta_t, sqrt_alpha_prev, sqrt_beta_prev} (host).
int ldmseg_op_conv_out_tail(const float* x, const float* w, const float* bias, int B, int H, int W, float* eps_out, int ddim, int last,
                            const float* coef4, int pred_type, int clip, float clip_range, float* latents, float* cond,
                            const float* rgb, const uint8_t* known, const float* z0, const float* noise, float sa, float sb,
                            float* xin_out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Temp t;
  if (!conv_out_tail_ok(320, H, W, DT_BF16)) return -2;
  void* xp = t.get((size_t)B * H * W * 320 * 2);
  if (launch_pack_nchw(x, xp, B, 320, H * W, 320, 1.f, 0.f, DT_BF16, s)) return -3;
  void* wp = t.get((size_t)32 * 9 * 320 * 2);
  if (launch_repack_conv(w, wp, 4, 320, 3, 3, 32, 320, DT_BF16, s)) return -3;
  float* bp = (float*)t.get(32 * sizeof(float));
  (void)hipMemsetAsync(bp, 0, 32 * sizeof(float), s);
  if (bias) (void)hipMemcpyAsync(bp, bias, 4 * sizeof(float), hipMemcpyDeviceToDevice, s);
  if (igemm_warm()) return -3;
  StepTail st;
  st.x = xp; st.w = wp; st.bias = bp; st.zeros = igemm_zero_page(); st.B = B; st.H = H; st.W = W; st.eps_out = eps_out;
  void* xin = nullptr;
  if (ddim) {
    if (!coef4 || !latents) return -2;
    st.ddim = 1; st.last = last;
    st.c = DdimCoef{coef4[0], coef4[1], coef4[2], coef4[3], pred_type, clip, clip_range, 0};
    st.latents = latents; st.cond = cond; st.rgb = rgb; st.known = known; st.z0 = z0; st.noise = noise; st.sa = sa; st.sb = sb;
    if (xin_out) { xin = t.get((size_t)B * H * W * 64 * 2); (void)hipMemsetAsync(xin, 0xff, (size_t)B * H * W * 64 * 2, s); st.xin_next = xin; }
  }
  const int r = launch_conv_out_tail(st, s);
  if (r) return r;
  if (xin && !last) from_dev_dtype(xin, xin_out, (size_t)B * H * W * 64, DT_BF16, s);
  return 0;
}

// the same attention on the fp8 (e4m3) operand path of the bf16 mode (attention_fp8.hip): head dim 40 or 80
int ldmseg_op_attention_fp8(const float* qkv, int B, int N, int C, int heads, float* out, int time_iters, float* us_per_launch,
                            void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Temp t;
  const size_t kv = attention_fp8_scratch_bytes(B, N, C, heads);
  if (!kv) return -2;
  void* qp = t.get((size_t)B * N * 3 * C * 2);
  to_dev_dtype(qkv, qp, (size_t)B * N * 3 * C, DT_BF16, s);
  void* op = t.get((size_t)B * N * C * 2);
  void* kp = t.get(kv);
  int r = launch_attention_fp8(qp, kp, op, B, N, C, heads, s);
  if (r) return r;
  if (time_iters > 0 && us_per_launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) (void)launch_attention_fp8(qp, kp, op, B, N, C, heads, s);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < time_iters; ++i) (void)launch_attention_fp8(qp, kp, op, B, N, C, heads, s);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *us_per_launch = 1e3f * ms / time_iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  if (out) from_dev_dtype(op, out, (size_t)B * N * C, DT_BF16, s);
  return 0;
}

}  // extern "C"
