// The tail of a denoising step in one launch (bf16 mode):
//
//     eps   = conv_out(SiLU(conv_norm_out(h)))            3x3, 320 -> 4       /root/reference/ldmseg/models/unet.py:433-436
//     x0, prev = DDIM(eps, latents, t)                                        /root/reference/ldmseg/schedulers/ddim_scheduler.py:231-267
//     latents <- prev (last step: x0) [inpainting: known pixels <- sa*z0 + sb*noise]
//     condition <- x0                                      self-conditioning  /root/reference/ldmseg/trainers/trainers_ldm_cond.py:1151-1159
//     next step's UNet input  [latents | rgb_latents | condition]  packed     trainers_ldm_cond.py:1128-1138
//
// Round 3 ran this as four launches: GroupNorm, conv_out through the general implicit GEMM (47.8 us: M = 32768, N = 4 - one
// output tile per workgroup walking 45 K tiles alone), ddim_step (4.8 us), pack_concat3 (9.2 us).  conv_out is a 21 MB read with
// 0.75 GFLOP behind it: an HBM-bound stencil, not a GEMM.  Here a workgroup owns an 8 x 16 pixel tile of one image, brings
// the tile plus its one-pixel halo (180 pixels x 320 channels = 113 KB) into the LDS with ONE burst of LDS-DMA pieces (the
// 3x3 neighbourhood is then served on chip instead of nine passes over the L2), runs the 9 x 5 x 2 MFMAs per 16-pixel row
// with the four real output channels in rows 0..3 of the A operand, and finishes the step in the epilogue: the lane that
// holds a pixel's four eps values also does that pixel's DDIM update, paste, self-condition write and next-input pack.
// Built with -ffp-contract=off: the DDIM arithmetic is the same rounded-separately chain as sched.hip's (sched_math.h),
// so the native loop stays bit-identical to scheduler.step() on the same eps.
#include <hip/hip_runtime.h>

#include "common.h"
#include "kernels.h"
#include "sched_math.h"

namespace ldmseg {
namespace {

constexpr int kTH = 8, kTW = 16;                       // pixel tile of a workgroup (one 16-pixel row per wave)
constexpr int kHaloW = kTW + 2, kHaloPix = (kTH + 2) * kHaloW;   // 18, 180
constexpr int kHaloGroups = (kHaloPix + 7) / 8;        // 23 DMA groups of 8 pixels (184 rows, the last 4 unused)
constexpr int kC = 320, kKT = kC / 64;
constexpr int kSlice = kHaloGroups * 8 * 128;          // 23552 B: one 64-channel slice of the halo tile, [pixel][128 B]
constexpr int kXBytes = kKT * kSlice;                  // 117760
constexpr int kWRow = 9 * kC * 2;                      // 5760 B: one output channel's weights, K = (tap, channel)
constexpr int kWBytes = 4 * kWRow;                     // 23040
constexpr int kLds = kXBytes + kWBytes + 16;           // + 16 zero bytes (A-operand rows 4..15)

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off"
      :
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

struct TailParams {
  const bf16_t* x;        // [B, H*W, 320] SiLU(GroupNorm(h))
  const bf16_t* w;        // conv_out weights, packed [>=4 rows][9 * 320], K = (tap, channel); rows 0..3 are the output channels
  const float* bias;      // [>=4]
  const void* zeros;
  int B, H, W;
  float* eps_out;         // [B,4,H,W] fp32 (plain forward) or null
  // fused scheduler step (ddim != 0)
  int ddim, last;
  DdimCoef c;
  float* latents;         // [B,4,H,W] in/out
  float* cond;            // [B,4,H,W] out (pred_original_sample) or null
  const float* rgb;       // [B,4,H,W] (only for the pack)
  bf16_t* xin_next;       // [B, H*W, 64] next step's packed input or null
  const uint8_t* known;   // inpainting: [B,1,H,W] or null
  const float* z0;
  const float* noise;
  float sa, sb;
};

__global__ __launch_bounds__(512) void conv_out_tail_kernel(const TailParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* XT = smem;
  unsigned char* WL = smem + kXBytes;
  unsigned char* ZL = smem + kXBytes + kWBytes;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img = blockIdx.z, ty = blockIdx.y, tx = blockIdx.x;
  const int HW = p.H * p.W;
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);

  // ---- halo tile: 5 slices x 23 groups of 8 pixels, one LDS-DMA piece each; 16-B chunk index XOR (halo pixel & 7)
  {
    const int pr = lane >> 3;
    for (int q = wave; q < kKT * kHaloGroups; q += 8) {
      const int kt = q / kHaloGroups, pg = q - kt * kHaloGroups;
      const int hp = pg * 8 + pr;                                   // halo pixel of this lane
      const int py = hp / kHaloW, px = hp - py * kHaloW;
      const int y = ty * kTH - 1 + py, x = tx * kTW - 1 + px;
      const bool ok = hp < kHaloPix && y >= 0 && y < p.H && x >= 0 && x < p.W;
      const void* src = ok ? (const void*)((const unsigned char*)p.x + ((size_t)img * HW + (size_t)y * p.W + x) * (kC * 2) + kt * 128 +
                                           (((lane & 7) ^ (hp & 7)) << 4))
                           : p.zeros;
      glds16(src, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(kt * kSlice + pg * 1024)));
    }
  }
  // ---- the four weight rows (contiguous in the packed matrix) + the zero chunk
  for (int ch = tid; ch < kWBytes / 16; ch += 512) *(uint4*)(WL + ch * 16) = *(const uint4*)((const unsigned char*)p.w + ch * 16);
  if (tid == 0) *(uint4*)ZL = make_uint4(0u, 0u, 0u, 0u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- 16 pixels of tile row `wave`: D[n][m] += W[n][k] X[m][k] over 9 taps x 320 channels
  const int cpix = lane & 15, q4 = lane >> 4;
  const int nrow = lane & 15;                                        // A-operand row = output channel (0..3 real)
  const unsigned char* wbase = nrow < 4 ? WL + nrow * kWRow + q4 * 16 : ZL;
  const int wmul = nrow < 4 ? 1 : 0;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const int hp0 = wave * kHaloW + cpix;                              // halo pixel of tap (0, 0)
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int hp = hp0 + (tap / 3) * kHaloW + (tap % 3);
    const unsigned char* xrow = XT + hp * 128;
    const int sw = hp & 7;
#pragma unroll
    for (int kt = 0; kt < kKT; ++kt) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const uint4 xf = *(const uint4*)(xrow + kt * kSlice + (((g * 4 + q4) ^ sw) << 4));
        const uint4 wf = *(const uint4*)(wbase + wmul * ((tap * kC + kt * 64 + g * 32) * 2));
        mma_kgroup<bf16_t>(wf, xf, acc);
      }
    }
  }
  if (q4 != 0) return;                                               // lanes 0..15 hold channels 0..3 of pixel `lane`

  const int y = ty * kTH + wave, x = tx * kTW + cpix;
  const int pix = y * p.W + x;
  float e[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) e[r] = acc[r] + p.bias[r];
  if (p.eps_out) {
#pragma unroll
    for (int r = 0; r < 4; ++r) p.eps_out[((size_t)img * 4 + r) * HW + pix] = e[r];
  }
  if (!p.ddim) return;
  float nl[4], x0v[4];
  const bool kn = p.known && p.known[(size_t)img * HW + pix];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const size_t i = ((size_t)img * 4 + r) * HW + pix;
    float pv, x0;
    ddim_update(e[r], p.latents[i], p.c, pv, x0);
    float v = p.last ? x0 : pv;
    if (kn) v = __fadd_rn(__fmul_rn(p.sa, p.z0[i]), __fmul_rn(p.sb, p.noise[i]));
    p.latents[i] = v;
    if (!p.last && p.cond) p.cond[i] = x0;
    nl[r] = v;
    x0v[r] = x0;
  }
  if (p.xin_next && !p.last) {
    float rg[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rg[r] = p.rgb[((size_t)img * 4 + r) * HW + pix];
    uint4* o = (uint4*)(p.xin_next + ((size_t)img * HW + pix) * 64);
    const bool sc = p.cond != nullptr;
    o[0] = make_uint4(pack_bf16x2(nl[0], nl[1]), pack_bf16x2(nl[2], nl[3]), pack_bf16x2(rg[0], rg[1]), pack_bf16x2(rg[2], rg[3]));
    o[1] = sc ? make_uint4(pack_bf16x2(x0v[0], x0v[1]), pack_bf16x2(x0v[2], x0v[3]), 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int j = 2; j < 8; ++j) o[j] = make_uint4(0u, 0u, 0u, 0u);
  }
}

int g_tail_mode = 3;      // bit 0: dedicated conv_out kernel; bit 1: DDIM update / self-condition / next-input pack in its epilogue

}  // namespace

void step_tail_set_mode(int m) { g_tail_mode = m & 3; }
int step_tail_get_mode() { return g_tail_mode; }
bool conv_out_tail_ok(int C, int H, int W, int dtype) {
  return (g_tail_mode & 1) && dtype == DT_BF16 && C == kC && H % kTH == 0 && W % kTW == 0;
}
bool step_tail_fused() { return (g_tail_mode & 3) == 3; }

int launch_conv_out_tail(const StepTail& t, hipStream_t s) {
  if (!t.x || !t.w || !t.bias || !t.zeros || t.B < 1 || t.H % kTH || t.W % kTW) return -2;
  if (!t.ddim && !t.eps_out) return -2;
  if (t.ddim && (!t.latents || (t.xin_next && !t.rgb) || (t.known && (!t.z0 || !t.noise)))) return -2;
  TailParams p;
  p.x = (const bf16_t*)t.x; p.w = (const bf16_t*)t.w; p.bias = t.bias; p.zeros = t.zeros;
  p.B = t.B; p.H = t.H; p.W = t.W; p.eps_out = t.eps_out;
  p.ddim = t.ddim; p.last = t.last; p.c = t.c; p.latents = t.latents; p.cond = t.cond; p.rgb = t.rgb;
  p.xin_next = (bf16_t*)t.xin_next; p.known = t.known; p.z0 = t.z0; p.noise = t.noise; p.sa = t.sa; p.sb = t.sb;
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute((const void*)conv_out_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    attr_set[dev] = true;
  }
  igemm_log_note("conv_out_tail<bf16>");
  hipLaunchKernelGGL(conv_out_tail_kernel, dim3(t.W / kTW, t.H / kTH, t.B), dim3(512), kLds, s, p);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace ldmseg
