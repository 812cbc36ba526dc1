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
/* out = proj_out(h + ff.net.2(g)) + x - the end of a diffusers BasicTransformerBlock inside Transformer2DModel
 * (/root/reference/ldmseg/models/unet.py:401-425) - as the engines launch it at the 640- / 1280-channel levels: the chained matrix
 * [Wp W2 | Wp] and bias bp + Wp b2 formed once in fp32 (launch_chain_weights, the create-time kernel), then ONE Linear over [g | h]
 * with x as the residual.  g [M][4C], h / x / out [M][C], w2 [C][4C], wp [C][C]; b2 / bp may be NULL; C a multiple of 160. */
int ldmseg_op_chained_ff_out(const float* g, const float* h, const float* x, const float* w2, const float* b2, const float* wp,
                             const float* bp, int M, int C, int rows_per_image, int dtype, float* out, void* stream);
/* F.group_norm(cat([x,x2],1), 32, gamma, beta, eps) [+SiLU] */
int ldmseg_op_groupnorm(const float* x, const float* x2, const float* gamma, const float* beta, int B, int C, int C2,
                        int HW, float eps, int silu, int dtype, float* out, void* stream);
/* F.layer_norm over the last dim [+SiLU] (also vae.py:309-322 LayerNorm2d in NHWC) */
int ldmseg_op_layernorm(const float* x, const float* gamma, const float* beta, int M, int C, float eps, int silu,
                        int dtype, float* out, void* stream);
/* diffusers Attention core on fused qkv [B,N,3C] -> [B,N,C] */
int ldmseg_op_attention(const float* qkv, int B, int N, int C, int heads, int dtype, float* out, void* stream);
/* the same on the fp8 (e4m3) operand path of the bf16 mode (K/V pre-quantised, Q and P quantised in the kernel, fp32
 * accumulation; head dim 40 or 80) - BASELINE configs[4].  time_iters > 0 additionally times that many launches (pre-pass
 * + kernel) into *us_per_launch; out may be NULL then. */
int ldmseg_op_attention_fp8(const float* qkv, int B, int N, int C, int heads, float* out, int time_iters, float* us_per_launch,
                            void* stream);
/* nn.ConvTranspose2d(Ci, Co, kernel_size=2, stride=2)  (vae.py:154) */
int ldmseg_op_convt2(const float* x, const float* w, const float* bias, int B, int Ci, int H, int W, int Co, int dtype,
                     float* out, void* stream);
/* F.interpolate(scale_factor=2, mode='bilinear', align_corners=False)  (vae.py:270) */
int ldmseg_op_bilinear2x(const float* x, int B, int C, int H, int W, int dtype, float* out, void* stream);

/* The fused evaluation tail of ldmseg_vae_decode_panoptic on a given decoder output x4 [B,C,H4,W4] (f32 NCHW, packed to
 * NHWC `dtype` first); host geometry arrays as there.  volume (optional, device): the resampled logits [C][h_b*w_b] of
 * image b at C * out_offsets[b] - lets a test compare the interpolation chain with F.interpolate o crop o F.interpolate. */
int ldmseg_op_panoptic_from_decoder(const float* x4, int B, int C, int H4, int W4, int dtype, int in_h, int in_w,
                                    const int32_t* crop_boxes, const int32_t* out_sizes, const int64_t* out_offsets,
                                    int threshold_output, int threshold_mode, float mask_th, int count_th, double overlap_th,
                                    int64_t ignore_label, int32_t* labels, int32_t* panoptic, uint8_t* keep, int32_t* counts,
                                    int32_t* mask_counts, float* volume, void* stream);

/* Entry of a transformer on caller-supplied weights: h = x Wp^T + bp (proj_in, /root/reference/ldmseg/models/unet.py:361-373 via
 * diffusers Transformer2DModel), q|k|v = [Wq | Wk | Wv] LayerNorm(h; gamma, beta).  mode 0: unfused launches; 1: the row-local fused
 * kernel (bf16, C = 320, M % 128 == 0).  h_out [M][C], qkv_out [M][3C] fp32.  time_iters > 0 also times the chosen path. */
int ldmseg_op_transformer_in(const float* x, const float* wp, const float* bp, const float* gamma, const float* beta, const float* wq,
                             const float* wk, const float* wv, int M, int C, float eps, int dtype, int mode, float* h_out, float* qkv_out,
                             int time_iters, float* us_per_call, void* stream);
/* The same behind the transformer's GroupNorm (32 groups, eps gn_eps, no SiLU; diffusers Transformer2DModel.norm) over
 * x = [images][M / images][C] rows.  gn_mode 0: a GroupNorm launch, then the entry as above; gn_mode 1 (mode 1 only, M / images a
 * multiple of 128): the statistics pass alone, the normalisation applied to the row tile inside the fused kernel (tproj.hip). */
int ldmseg_op_gn_transformer_in(const float* x, const float* gn_gamma, const float* gn_beta, float gn_eps, int images, int gn_mode,
                                const float* wp, const float* bp, const float* gamma, const float* beta, const float* wq,
                                const float* wk, const float* wv, int M, int C, float eps, int dtype, int mode, float* h_out,
                                float* qkv_out, int time_iters, float* us_per_call, void* stream);

/* One conv / GEMM layer launched exactly as the engine launches it inside a forward (NHWC operands, the engine's
 * split-K plan when splits == 0, row-major store epilogue with bias / per-image bias row / residual / SiLU, or GEGLU):
 * F.conv2d(cat([x,x2],1) [nearest x2 if up], w, bias, stride, k/2) + rowbias[b,:,None,None] + resid, then SiLU;
 * geglu=1: w [Co,Ci] is ff.net.0.proj, out = a * gelu(gate) with [a | gate] = chunk(2, dim=1), Cout = Co/2.
 * NCHW f32 boundary like the other ops. */
int ldmseg_op_igemm(const float* x, const float* x2, const float* w, const float* bias, const float* resid,
                    const float* rowbias, int B, int Ci, int Ci2, int H, int W, int Co, int k, int stride, int up, int geglu,
                    int silu, int splits, int dtype, float* out, void* stream);
/* F.conv2d(h, w2, b2, padding=1) + F.conv2d(torch.cat([xs, xs2], 1), ws, bs): the tail of diffusers' ResnetBlock2D where
 * cin != cout (conv2 + conv_shortcut, ldmseg/models/unet.py:361-425 run them through diffusers 0.16.1's resnet.py) as the
 * engine's ONE bf16 launch with an extra centre tap.  xs2 NULL / Cs2 = 0 when the shortcut input is not a concat.  iters > 0
 * additionally times `iters` launches (us_per_launch).  -4: the shape has no such launch (the engine then runs two convs). */
int ldmseg_op_conv3x3_plus_1x1(const float* h, const float* w2, const float* b2, const float* xs, const float* xs2, const float* ws,
                               const float* bs, int B, int C, int Cs, int Cs2, int H, int W, int Co, int splits, int dtype, float* out,
                               int iters, float* us_per_launch, void* stream);
/* [silu](F.group_norm(F.conv2d(x, w, bias, padding=1) + rowbias[b,:,None,None], 32, gamma, beta, eps)) the way the engine runs
 * resnet conv1 -> norm2 on small maps: the conv as `splits` (>= 2) K slices, then ONE kernel that sums the slices, adds bias and
 * time-embedding row and normalises (launch_finish_groupnorm).  Returns -4 when the shape has no fused instantiation. */
int ldmseg_op_conv_groupnorm(const float* x, const float* w, const float* bias, const float* rowbias, const float* gamma,
                             const float* beta, int B, int Ci, int H, int W, int Co, float eps, int silu, int splits, int dtype,
                             float* out, void* stream);
/* F.linear(F.layer_norm(x, (K,), gamma, beta, eps), w, bias) - GEGLU on top when geglu=1 - the way the engine runs
 * norm1 -> to_q|k|v and norm3 -> ff.net.0.proj (diffusers BasicTransformerBlock): one statistics pass over x, then the GEMM on
 * the un-normalised x with gamma folded into the weights and rstd*(acc - mean*c1) + c2 in the epilogue. */
int ldmseg_op_ln_linear(const float* x, const float* gamma, const float* beta, const float* w, const float* bias, int M, int K,
                        int N, float eps, int geglu, int dtype, float* out, void* stream);
/* The tail of a transformer block on [M, C] token rows (diffusers BasicTransformerBlock.ff + Transformer2DModel.proj_out,
 * /root/reference/ldmseg/models/unet.py:401-425):  h2 = h + ff.net.2(GEGLU(ff.net.0.proj(LayerNorm(h))));  out = proj_out(h2) + x.
 * mode 0 = the unfused launches, 1 = row-local fused feed-forward (tfuse.hip) + proj_out GEMM, 3 = all in the fused kernel
 * (modes 1 / 3: bf16, C = 320).  time_iters > 0: *us_per_call = average microseconds of the whole tail. */
int ldmseg_op_transformer_ff(const float* h, const float* x, const float* gamma, const float* beta, const float* w1, const float* b1,
                             const float* w2, const float* b2, const float* wp, const float* bp, int M, int C, float eps, int dtype,
                             int mode, float* out, int time_iters, float* us_per_call, void* stream);
/* The step-tail kernel (tail.hip, bf16): eps = conv_out(x) (3x3, 320 -> 4, /root/reference/ldmseg/models/unet.py:433-436) and,
 * with ddim != 0, DDIMNoiseScheduler.step (ddim_scheduler.py:231-267) on `latents` in place (last: pred_original_sample), the
 * inpainting paste, the self-condition write (trainers_ldm_cond.py:1151-1159) and the next step's packed input
 * [latents | rgb | cond | 0] as fp32 [B, H*W, 64].  coef4 is a HOST array; eps_out / cond / known / xin_out may be NULL. */
int ldmseg_op_conv_out_tail(const float* x, const float* w, const float* bias, int B, int H, int W, float* eps_out, int ddim, int last,
                            const float* coef4, int pred_type, int clip, float clip_range, float* latents, float* cond,
                            const float* rgb, const uint8_t* known, const float* z0, const float* noise, float sa, float sb,
                            float* xin_out, void* stream);
/* Kernel timing for tuning (tools/kbench.py): the same launches repeated `iters` times back to back on `stream` between
 * two HIP events; *us_per_launch = average microseconds (igemm: including the split-K finish kernel if the plan has one). */
int ldmseg_bench_igemm(const float* x, const float* x2, const float* w, const float* bias, const float* resid,
                       const float* rowbias, int B, int Ci, int Ci2, int H, int W, int Co, int k, int stride, int up, int geglu,
                       int silu, int splits, int dtype, int iters, float* us_per_launch, void* stream);
int ldmseg_bench_groupnorm(const float* gamma, const float* beta, int B, int C, int C2, int HW, int silu, int dtype, int iters,
                           float* us_per_launch, void* stream);
int ldmseg_bench_attention(const float* qkv, int B, int N, int C, int heads, int dtype, int iters, float* us_per_launch,
                           void* stream);
/* "igemm<dtype,BM,BN,WM,WN,NST,PIPE,LDR> splits=S grid=G": template instantiation and plan of the most recent igemm
 * launch of this process - lets a parity test assert WHICH kernel it just compared with the oracle. */
int ldmseg_igemm_last_kernel(char* buf, int n);
/* enable=1 clears and starts a log of the DISTINCT igemm instantiations launched ("igemm<...>" + "/splitk" for K-sliced
 * launches), enable=0 stops it; _read copies them newline-separated.  The parity suite uses it to prove that every
 * instantiation a full-size forward runs is also compared with the oracle by a per-layer test. */
int ldmseg_igemm_log(int enable);
int ldmseg_igemm_log_read(char* buf, int n);

/* q[i] = n[i] / d (0 <= n[i] < 2^31, d >= 1) computed the way the kernels divide: host-prepared multiply-shift pair */
int ldmseg_op_fastdiv(const int* n, int count, int d, int* q, void* stream);

#ifdef __cplusplus
}
#endif
#endif
