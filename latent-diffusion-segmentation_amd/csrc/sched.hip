// DDIM scheduler kernels (ddim_scheduler.py:155-269).  This file is compiled with
// -ffp-contract=off: every product / sum / quotient must be rounded separately, exactly like the
// reference's chain of torch elementwise ops, so that given the same fp32 coefficients the
// outputs are bit-identical to torch (hipcc's default contraction would fuse x - b*eps into an
// FMA, and the in-source pragma does not survive inlining of the __f*_rn helpers).
#include "common.h"
#include "kernels.h"
#include "sched_math.h"

namespace ldmseg {
namespace {

inline int grid_for(size_t n, int block = 256, int cap = 4096) {
  size_t g = (n + block - 1) / block;
  if (g > (size_t)cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}
inline int ok() { return hipGetLastError() == hipSuccess ? 0 : -3; }

// ---------------------------------------------------------------- scheduler
// ddim_scheduler.py:238-267; every product / sum rounded separately like the
// reference's chain of torch ops (no FMA contraction) so results are bit-exact.
__global__ void ddim_step_kernel(const float* eps_in, const float* x_in, float* prev, float* x0_out, size_t n, DdimCoef c) {
#pragma clang fp contract(off)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float pv, x0;
    ddim_update(eps_in[i], x_in[i], c, pv, x0);
    if (prev) prev[i] = pv;
    if (x0_out) x0_out[i] = x0;
  }
}

__global__ void inpaint_paste_kernel(float* cur, const float* z0, const float* noise, const uint8_t* known, float sa,
                                     float sb, int C, int HW, size_t total) {
#pragma clang fp contract(off)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i % HW;
    const size_t b = i / ((size_t)C * HW);
    if (known[b * HW + pix]) cur[i] = __fadd_rn(__fmul_rn(sa, z0[i]), __fmul_rn(sb, noise[i]));
  }
}

// timesteps outside [0, n_train) would be an IndexError in the reference (ddim_scheduler.py:170); the host wrapper
// rejects them when it can see them, the kernel clamps so that a bad device-resident index can never read out of bounds
__global__ void add_noise_kernel(const float* x0, const float* noise, const int64_t* t, const float* ac, int n_train,
                                 float scale, float* out, size_t per, size_t total, int remove) {
#pragma clang fp contract(off)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / per;
    int64_t tb = t[b];
    tb = tb < 0 ? 0 : (tb >= n_train ? n_train - 1 : tb);
    const float a = ac[tb];
    const float sa = __fsqrt_rn(a), sb = __fsqrt_rn(__fsub_rn(1.0f, a));
    if (!remove) out[i] = __fadd_rn(__fmul_rn(__fmul_rn(sa, scale), x0[i]), __fmul_rn(sb, noise[i]));
    else out[i] = __fdiv_rn(__fsub_rn(x0[i], __fmul_rn(sb, noise[i])), __fmul_rn(sa, scale));
  }
}

__global__ void axpby_kernel(const float* x, float a, float b, float* y, size_t n) {
#pragma clang fp contract(off)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = __fadd_rn(__fmul_rn(a, x[i]), b);
}

}  // namespace

int launch_ddim_step(const float* eps, const float* x, float* prev, float* x0, size_t n, DdimCoef c, hipStream_t s) {
  hipLaunchKernelGGL(ddim_step_kernel, dim3(grid_for(n)), dim3(256), 0, s, eps, x, prev, x0, n, c);
  return ok();
}

int launch_inpaint_paste(float* cur, const float* z0, const float* noise, const uint8_t* known, float sa, float sb,
                         int B, int C, int HW, hipStream_t s) {
  const size_t total = (size_t)B * C * HW;
  hipLaunchKernelGGL(inpaint_paste_kernel, dim3(grid_for(total)), dim3(256), 0, s, cur, z0, noise, known, sa, sb, C, HW, total);
  return ok();
}

int launch_add_noise(const float* x0, const float* noise, const int64_t* t_dev, const float* ac_dev, int n_train, float scale,
                     float* out, int B, size_t per, int remove, hipStream_t s) {
  const size_t total = (size_t)B * per;
  hipLaunchKernelGGL(add_noise_kernel, dim3(grid_for(total)), dim3(256), 0, s, x0, noise, t_dev, ac_dev, n_train, scale, out, per, total, remove);
  return ok();
}

int launch_axpby(const float* x, float a, float b, float* y, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, a, b, y, n);
  return ok();
}

}  // namespace ldmseg
