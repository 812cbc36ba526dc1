// The DDIM update of one element (ddim_scheduler.py:238-267), shared by ddim_step_kernel (sched.hip) and the fused step
// tail (tail.hip).  Every product / sum / quotient is rounded separately (__f*_rn; both files are built with
// -ffp-contract=off) so that the result is bit-identical to the reference's chain of torch elementwise ops.
#pragma once
#include "kernels.h"

namespace ldmseg {

__device__ __forceinline__ void ddim_update(float mo, float x, const DdimCoef& c, float& prev, float& x0_out) {
#pragma clang fp contract(off)
  float x0, pe;
  if (c.pred_type == 0) {          // epsilon
    x0 = __fdiv_rn(__fsub_rn(x, __fmul_rn(c.sqrt_b_t, mo)), c.sqrt_a_t);
    pe = mo;
  } else if (c.pred_type == 1) {   // sample
    x0 = mo;
    pe = __fdiv_rn(__fsub_rn(x, __fmul_rn(c.sqrt_a_t, x0)), c.sqrt_b_t);
  } else {                         // v_prediction
    x0 = __fsub_rn(__fmul_rn(c.sqrt_a_t, x), __fmul_rn(c.sqrt_b_t, mo));
    pe = __fadd_rn(__fmul_rn(c.sqrt_a_t, mo), __fmul_rn(c.sqrt_b_t, x));
  }
  if (c.clip) x0 = fminf(fmaxf(x0, -c.clip_range), c.clip_range);
  if (c.use_clipped) pe = __fdiv_rn(__fsub_rn(x, __fmul_rn(c.sqrt_a_t, x0)), c.sqrt_b_t);
  const float dir = __fmul_rn(c.sqrt_b_prev, pe);
  prev = __fadd_rn(__fmul_rn(c.sqrt_a_prev, x0), dir);
  x0_out = x0;
}

}  // namespace ldmseg
