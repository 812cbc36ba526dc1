/* ldmseg_hip_ops.h - single-operator entry points of libldmseg_hip.so (fp32 boundary).
 *
 * Test support, not a product surface: tests/ drives each gfx950 kernel in isolation
 * through these and compares with the torch-CPU op the reference executes at that
 * point.  `dtype` is LDMSEG_F32 | LDMSEG_BF16 (the kernel's
 * compute/storage type); all tensors are fp32 device pointers; every call allocates
 * and frees its own staging buffers and is therefore slow.
 */
#ifndef LDMSEG_HIP_OPS_H_
#define LDMSEG_HIP_OPS_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* F.conv2d(cat([x, x2],1) [after nearest x2 upsample if up], w, bias, stride, padding=k/2); k in {1,3}.
 * ResnetBlock2D.conv1/conv2/conv_shortcut, Downsample2D, Upsample2D, proj_in/out (diffusers 0.16.1). */
int ldmseg_op_conv2d(const float* x, const float* x2, const float* w, const float* bias, int B, int Ci, int Ci2, int H,
                     int W, int Co, int k, int stride, int up, int dtype, float* out, void* stream);
/* F.linear (+row bias per image, +residual, SiLU) or the GEGLU feed-forward half when geglu=1 */
int ldmseg_op_linear(const float* x, const float* w, const float* bias, const float* resid, const float* rowbias,
                     int rows_per_image, int M, int K, int N, int geglu, int silu, int splits, int dtype, float* out,
                     void* stream);
/* F.group_norm(cat([x,x2],1), 32, gamma, beta, eps) [+SiLU] */
int ldmseg_op_groupnorm(const float* x, const float* x2, const float* gamma, const float* beta, int B, int C, int C2,
                        int HW, float eps, int silu, int dtype, float* out, void* stream);
/* F.layer_norm over the last dim [+SiLU] (also vae.py:309-322 LayerNorm2d in NHWC) */
int ldmseg_op_layernorm(const float* x, const float* gamma, const float* beta, int M, int C, float eps, int silu,
                        int dtype, float* out, void* stream);
/* diffusers Attention core on fused qkv [B,N,3C] -> [B,N,C] */
int ldmseg_op_attention(const float* qkv, int B, int N, int C, int heads, int dtype, float* out, void* stream);
/* nn.ConvTranspose2d(Ci, Co, kernel_size=2, stride=2)  (vae.py:154) */
int ldmseg_op_convt2(const float* x, const float* w, const float* bias, int B, int Ci, int H, int W, int Co, int dtype,
                     float* out, void* stream);
/* F.interpolate(scale_factor=2, mode='bilinear', align_corners=False)  (vae.py:270) */
int ldmseg_op_bilinear2x(const float* x, int B, int C, int H, int W, int dtype, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
