/* ldmseg_hip.h - C ABI of libldmseg_hip.so, the MI355X (gfx950) implementation of
 * the LDMSeg denoising path.
 *
 * The reference (segments-ai/latent-diffusion-segmentation) has no FFI layer: the
 * seam is the Python call surface TrainerDiffusion uses.  Each entry point below
 * names the reference interface it stands behind (paths relative to the
 * reference root); INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - every pointer named *dev* / tensor argument is a DEVICE pointer on the
 *     handle's GPU; boundary tensors are fp32 NCHW contiguous (the reference's
 *     layout); timesteps are int64.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls
 *     only enqueue work.  The one exception is workspace growth: the first call of a
 *     handle at a (B, L) larger than any before (re)allocates its workspace, which
 *     synchronises the device.  ldmseg_unet_reserve() does that up front (workspace,
 *     sampler buffers incl. the time-embedding table of up to 64 steps, the
 *     per-device zero page; the GroupNorm hand-off region belongs to the handle and
 *     is made at *_create), after which forward / sample_loop calls at that size
 *     never allocate or synchronise.  The first launch of each kernel instantiation
 *     in a process still sets a function attribute (host-side, no device work): run
 *     one warm-up forward before capturing a stream.  ldmseg_unet_forward with a
 *     device timestep (t_dev) may be captured into a HIP graph and replayed (the
 *     cooperative GroupNorm draws its hand-off generation on the device, so replays
 *     do not meet each other's records); ldmseg_sample_loop copies cfg->timesteps
 *     from pageable host memory on every call and is not meant to be captured.
 *     *_create calls allocate and synchronise.
 *   - return 0 on success, negative LDMSEG_E_* on failure; ldmseg_last_error()
 *     returns a thread-local message.  No C++ exception crosses the boundary.
 *   - the library owns handles, repacked weights and workspaces; the caller owns
 *     every tensor it passes and keeps it alive until the stream work is done.
 *     Weights are copied/repacked at create time.
 *   - a handle is bound to cfg.device and is not thread-safe (one process per GPU,
 *     like tools/main_ldm.py:69 mp.spawn).  Every call on a handle makes that device
 *     current for its duration and restores the caller's; the handle-free calls
 *     (ddim_step, add_noise, ...) run on the calling thread's current device.
 */
#ifndef LDMSEG_HIP_H_
#define LDMSEG_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDMSEG_OK 0
#define LDMSEG_E_ARG (-1)      /* bad argument / unsupported option */
#define LDMSEG_E_SHAPE (-2)    /* shape not supported by the kernels */
#define LDMSEG_E_HIP (-3)      /* HIP runtime error (message has the hipError string) */
#define LDMSEG_E_WEIGHT (-4)   /* missing / mis-sized state-dict entry */
#define LDMSEG_E_ARCH (-5)     /* device is not gfx950 */
#define LDMSEG_E_OOM (-6)

#define LDMSEG_F32 0           /* fp32 storage, exact-f32 MFMA: the parity mode (1e-3 vs torch-CPU) */
#define LDMSEG_BF16 1          /* bf16 storage + bf16 MFMA, fp32 accumulate/statistics: the perf mode */
#define LDMSEG_BF16X3 2        /* fp32 storage, norms / softmax / scheduler in fp32, GEMM and attention products on the bf16 MFMAs as hi + lo
                                * (three MFMAs per product block, fp32 accumulation; relative error per product < 2^-16):
                                * the parity-grade throughput mode - inside the 1e-3 bound at several times the exact mode's speed */

#define LDMSEG_PRED_EPSILON 0
#define LDMSEG_PRED_SAMPLE 1
#define LDMSEG_PRED_V 2

typedef struct ldmseg_unet ldmseg_unet;  /* opaque */
typedef struct ldmseg_vae ldmseg_vae;    /* opaque */
typedef struct ldmseg_vae_image ldmseg_vae_image;  /* opaque */

/* ---- UNet: ldmseg/models/unet.py::UNet (UNet2DConditionModel, SD-1.x topology) ---------- */
typedef struct {
  int32_t in_channels;      /* 8 | 12 after UNet.modify_encoder (unet.py:178-233); 4 = vanilla */
  int32_t cross_attention;  /* 0: attn2/norm2 removed (unet.py:83-105, base.yaml:71). 1 is rejected (E_ARG) */
  int32_t compute_dtype;    /* LDMSEG_F32 | LDMSEG_BF16 | LDMSEG_BF16X3 */
  int32_t device;           /* HIP device ordinal */
} ldmseg_unet_cfg;

/* Build a UNet from a state dict in the reference's key layout (the `unet` entry of
 * ldmseg.pt, trainers_ldm_cond.py:1791-1814; keys = SURVEY.md App. A).  `names[i]` is
 * the key, `dev_ptrs[i]` an fp32 device tensor in torch layout with `numels[i]`
 * elements.  Unknown keys (e.g. the duplicate `new_conv.*`, unet.py:182,233) are
 * ignored; a missing or mis-sized required key fails with LDMSEG_E_WEIGHT.
 * Replaces: UNet.from_pretrained + remove_cross_attention + modify_encoder +
 * load_state_dict (tools/main_ldm.py:146-168, trainers_ldm_cond.py:1863-1891). */
int ldmseg_unet_create(const ldmseg_unet_cfg* cfg, int n_weights, const char* const* names,
                       const void* const* dev_ptrs, const int64_t* numels, ldmseg_unet** out);
void ldmseg_unet_destroy(ldmseg_unet* h);

/* UNet.forward(sample, timestep, encoder_hidden_states=None).sample  (unet.py:281-436).
 * x: [B, in_channels, L, L] fp32; out: [B, 4, L, L] fp32.  The timestep is read from
 * `t_dev` (int64 on device, `t_count` = 1 -> broadcast like timestep.expand(B), or B)
 * so a GPU-resident scheduler.timesteps[i] needs no D2H sync; if t_dev is NULL,
 * `t_host` is used.  L must be a multiple of 8. */
int ldmseg_unet_forward(ldmseg_unet* h, const float* x, const int64_t* t_dev, int t_count, int64_t t_host,
                        int B, int L, float* out, void* stream);
/* Same forward but the input is given as its channel-concat parts
 * (torch.cat([latents, rgb_latents, condition], dim=1), trainers_ldm_cond.py:1128-1138);
 * `cond` may be NULL when in_channels == 8. */
int ldmseg_unet_forward_parts(ldmseg_unet* h, const float* latents, const float* rgb_latents, const float* cond,
                              const int64_t* t_dev, int t_count, int64_t t_host, int B, int L, float* out,
                              void* stream);
/* bf16 mode only: run the self-attention of every level with at least `min_tokens` tokens (H*W) on the fp8 operand path
 * (OCP e4m3 Q/K/V/P on v_mfma_f32_16x16x32_fp8_fp8, fp32 accumulation and statistics) - the "fp8 MFMA attention path" of
 * the 1024x1024 configuration (128x128 latents: 16384 / 4096 tokens at head dims 40 / 80); 0 switches it off (default).
 * Levels where the fp8 path is not the faster one (head dim 80 / token counts that are not a multiple of 128: only the unscaled
 * fp8 MFMAs serve them, at the bf16 rate) stay on the bf16 kernel whatever `min_tokens` says.
 * The reference has no counterpart (its attention is whatever diffusers' processor does in fp32). */
int ldmseg_unet_set_attention_fp8(ldmseg_unet* h, int min_tokens);
/* Allocate everything forward / sample_loop need at (B, L) now (workspace + the sampler's eps,
 * self-condition and time-embedding buffers); may synchronise the device.  Optional: without it the first call at a new,
 * larger shape does the same lazily. */
int ldmseg_unet_reserve(ldmseg_unet* h, int B, int L);
/* Workgroups of this handle's cooperative GroupNorm launches that did not see their partners within the poll bound and
 * computed the partners' statistics themselves since the handle was created, launches under the back-off's short bound
 * excepted (0 on an undisturbed device: the kernel's workgroups are co-resident).  Synchronises.  ldmseg_sample_loop looks at the same counter (asynchronously) and runs the
 * next calls with a short poll bound while it grows.  No reference counterpart (torch's GroupNorm is one kernel). */
int ldmseg_unet_gn_fallbacks(ldmseg_unet* h, int64_t* count);
/* Workgroups of this handle's K-sliced GEMM launches that gave up waiting for the other slices of their tile (poll bound 200 us) and
 * left their share of the reduction to the tile's last arriver, since the handle was created (0 on an undisturbed device).  Results
 * are the same either way; the counter says whether something else is holding CUs.  Synchronises.  No reference counterpart. */
int ldmseg_unet_cf_fallbacks(ldmseg_unet* h, int64_t* count);
/* ldmseg_sample_loop calls this handle will still run with the short (2 us) partner poll: 8 after a call whose full-bound norms
 * missed their partners at least 256 times, one less after every call; launches under the short bound do not count, so the
 * call after the eighth tries the full bound again.  Does not synchronise (the loop reads the counter one call late). */
int ldmseg_unet_gn_backoff(ldmseg_unet* h, int32_t* calls_left);
/* bytes of device workspace a forward at (B, L) needs (allocated lazily, grown never shrunk) */
size_t ldmseg_unet_workspace_bytes(const ldmseg_unet* h, int B, int L);
/* number of parameters held (815,556,484 for the 12-channel default) */
int64_t ldmseg_unet_num_params(const ldmseg_unet* h);

/* ---- seg-VAE: ldmseg/models/vae.py::GeneralVAESeg (gaussian, num_mid_blocks=0) ------------ */
typedef struct {
  int32_t in_channels;      /* 7  (bit maps, base.yaml:15) */
  int32_t int_channels;     /* 256 */
  int32_t out_channels;     /* 128 */
  int32_t latent_channels;  /* 4 */
  int32_t num_latents;      /* 2 (mean, logvar) */
  int32_t num_upscalers;    /* 2 */
  int32_t upscale_channels; /* 256 */
  int32_t norm_num_groups;  /* 32 */
  int32_t block_out_channels[4]; /* 32,64,128,256 */
  int32_t compute_dtype;
  int32_t device;
} ldmseg_vae_cfg;

/* keys: encoder.{0,2,3,5,6,8,9,11,13,15}.{weight,bias}, decoder.{0,2,3,5,6,8,10}.{weight,bias}
 * (vae.py:123-244; the AE checkpoint's 'module.' prefix is stripped by the caller, vae.py:116-121) */
int ldmseg_vae_create(const ldmseg_vae_cfg* cfg, int n_weights, const char* const* names,
                      const void* const* dev_ptrs, const int64_t* numels, ldmseg_vae** out);
void ldmseg_vae_destroy(ldmseg_vae* h);
/* GeneralVAESeg.decode(z, interpolate) (vae.py:267-271): z [B,4,L,L] -> logits
 * [B,128,4L,4L] (interpolate=0) or [B,128,8L,8L] (bilinear x2, align_corners=False).
 * `z_scale` multiplies z first (decode_latents' 1/scaling_factor, trainers_ldm_cond.py:421). */
int ldmseg_vae_decode(ldmseg_vae* h, const float* z, float z_scale, int B, int L, int interpolate, float* logits,
                      void* stream);
/* Fused tail of TrainerDiffusion.decode_latents(return_logits=False, threshold_output) (trainers_ldm_cond.py:
 * 421-433): decode -> bilinear x2 -> argmax over the 128 classes and max-softmax probability, without writing the
 * logits.  ids [B,8L,8L] int64 (= ignore_label where max prob < mask_th; mask_th < 0 disables the threshold),
 * max_prob [B,8L,8L] fp32 or NULL. */
int ldmseg_vae_decode_argmax(ldmseg_vae* h, const float* z, float z_scale, int B, int L, float mask_th, int64_t ignore_label,
                             int64_t* ids, float* max_prob, void* stream);
/* The evaluation tail of TrainerDiffusion.compute_pq (trainers_ldm_cond.py:1243-1313) in one call, without the
 * [B,128,H,W] fp32 logits: decode (vae.py:267-271 incl. the bilinear x2) -> F.interpolate to the network input size
 * (in_h, in_w) (:1252-1257) -> crop_padding (:1172-1178, :1263) -> F.interpolate to the original size (h_b, w_b)
 * (:1266-1271) -> argmax / thresholds / segment filtering (:1277-1313, see ldmseg_panoptic_postprocess below).  The
 * three bilinear stages (all align_corners=False) are evaluated as one separable weighted sum over the decoder's 4L map.
 *   crop_boxes  [B][4] HOST int32 (y0, x0, height, width) of the non-padded box in the (in_h, in_w) grid; NULL = whole grid
 *   out_sizes   [B][2] HOST int32 (h_b, w_b);  out_offsets [B] HOST int64: first element of image b in labels / panoptic
 *   labels, panoptic: flat int32 device buffers holding the ragged [h_b * w_b] maps; keep / counts / mask_counts [B,C]
 * as in ldmseg_panoptic_postprocess.  Enqueues 2B + 2 small launches behind the decoder. */
int ldmseg_vae_decode_panoptic(ldmseg_vae* h, const float* z, float z_scale, int B, int L, int in_h, int in_w,
                               const int32_t* crop_boxes, const int32_t* out_sizes, const int64_t* out_offsets,
                               int threshold_output, int threshold_mode, float mask_th, int count_th, double overlap_th,
                               int64_t ignore_label, int32_t* labels, int32_t* panoptic, uint8_t* keep, int32_t* counts,
                               int32_t* mask_counts, void* stream);
/* Panoptic post-processing of the evaluation loop (trainers_ldm_cond.py:1277-1313) on logits [B,C,H,W] fp32 that
 * are already at the output size (C <= 256):
 *   labels   [B,H,W] int32  argmax over C; -1 where threshold_output and (threshold_mode 0: max softmax prob,
 *                           1 'topk_diff': top1 - top2 prob) < mask_th                       (:1277-1286)
 *   counts   [B,C]  int32   pixels per label (np.unique counts, :1295)
 *   mask_counts [B,C] int32 pixels with sigmoid(logit_c) >= mask_th (original_mask.sum(), :1303)
 *   keep     [B,C]  uint8   0 where label == ignore_label, counts < count_th or counts / mask_counts < overlap_th
 *                           (:1298-1306); kept labels are the reference's segments_info ids (c + 1)
 *   panoptic [B,H,W] int32  label + 1 where kept, 0 (void) elsewhere                          (:1300,1305,1313)
 * All outputs are device buffers owned by the caller.  One pass over the logits; nothing is copied to the host. */
int ldmseg_panoptic_postprocess(const float* logits, int B, int C, int H, int W, int threshold_output, int threshold_mode,
                                float mask_th, int count_th, double overlap_th, int64_t ignore_label, int32_t* labels,
                                int32_t* panoptic, uint8_t* keep, int32_t* counts, int32_t* mask_counts, void* stream);
/* GeneralVAESeg.encode(x) (vae.py:252-265): x [B,7,H,W] (H=W multiple of 8), moments
 * [B,8,H/8,W/8] = (mean | logvar) before the clamp.  x is used as x*in_mul+in_add
 * (encode_inputs' 2x-1, trainers_ldm_cond.py:369). */
int ldmseg_vae_encode(ldmseg_vae* h, const float* x, float in_mul, float in_add, int B, int H, float* moments,
                      void* stream);
/* DiagonalGaussianDistribution.mode()/.sample() (vae.py:370-413) on `moments`:
 * out[B,4,l,l] = (mean + exp(0.5*clamp(logvar,-30,20))*noise) * out_scale; noise NULL -> mode(). */
int ldmseg_vae_posterior(const float* moments, const float* noise, float out_scale, int B, int l, float* out,
                         void* stream);
int64_t ldmseg_vae_num_params(const ldmseg_vae* h);

/* ---- image VAE encoder: ldmseg/models/vae.py:36-39 GeneralVAEImage(AutoencoderKL), decoder removed
 * (tools/main_ldm.py:137-139); used as encode_func in TrainerDiffusion.encode_inputs (trainers_ldm_cond.py:360-375).
 * SD-1.x VAE encoder: conv_in 3->128, down blocks 128/256/512/512 with two temb-free resnets each (GroupNorm 32,
 * eps 1e-6, SiLU), stride-2 convs with F.pad(0,1,0,1), mid block resnet / single-head attention (C=512) / resnet,
 * conv_norm_out + SiLU + conv_out 512->8, quant_conv 8->8.  The arithmetic is diffusers' (not vendored by the
 * reference): parity is pinned against oracle/vae_image.py only. */
typedef struct {
  int32_t compute_dtype;    /* LDMSEG_F32 | LDMSEG_BF16 | LDMSEG_BF16X3 */
  int32_t device;
} ldmseg_vae_image_cfg;
/* keys: the 'vae_image' entry of ldmseg.pt (trainers_ldm_cond.py:1805) / AutoencoderKL.state_dict():
 * encoder.conv_in, encoder.down_blocks.{0..3}.resnets.{0,1}.{norm1,conv1,norm2,conv2[,conv_shortcut]},
 * encoder.down_blocks.{0..2}.downsamplers.0.conv, encoder.mid_block.resnets.{0,1}.*,
 * encoder.mid_block.attentions.0.{group_norm, query|to_q, key|to_k, value|to_v, proj_attn|to_out.0},
 * encoder.conv_norm_out, encoder.conv_out, quant_conv (decoder.* / post_quant_conv.* are ignored). */
int ldmseg_vae_image_create(const ldmseg_vae_image_cfg* cfg, int n_weights, const char* const* names,
                            const void* const* dev_ptrs, const int64_t* numels, ldmseg_vae_image** out);
void ldmseg_vae_image_destroy(ldmseg_vae_image* h);
int64_t ldmseg_vae_image_num_params(const ldmseg_vae_image* h);
/* AutoencoderKL.encode(x).latent_dist parameters: x [B,3,H,W] fp32 (used as x*in_mul+in_add: encode_inputs' 2x-1,
 * trainers_ldm_cond.py:369), H, W multiples of 8 with (H/8)*(W/8) a multiple of 64 -> moments [B,8,H/8,W/8]
 * (mean | logvar before the clamp); feed ldmseg_vae_posterior for .mode()/.sample() and the scaling factor. */
int ldmseg_vae_image_encode(ldmseg_vae_image* h, const float* x, float in_mul, float in_add, int B, int H, int W,
                            float* moments, void* stream);

/* ---- scheduler: ldmseg/schedulers/ddim_scheduler.py --------------------------------------- */
/* DDIMNoiseScheduler.step (:218-269), elementwise over n floats.  The four coefficients are
 * the 0-d fp32 values the reference computes on the host (alpha_prod_t**0.5, ...); every
 * product/sum is rounded separately, so outputs equal the torch result bit for bit.
 * prev or x0 may be NULL. */
int ldmseg_ddim_step(const float* model_output, const float* sample, float sqrt_alpha_t, float sqrt_beta_t,
                     float sqrt_alpha_prev, float sqrt_beta_prev, int prediction_type, int clip_sample,
                     float clip_sample_range, int use_clipped_model_output, float* prev_sample,
                     float* pred_original_sample, size_t n, void* stream);
/* add_noise (:155-187) / remove_noise (:190-216) with per-sample int64 timesteps [B] and the
 * fp32 alphas_cumprod table [n_train_timesteps] on device.  The reference raises IndexError for a
 * timestep outside [0, n_train_timesteps); device-resident timesteps cannot be inspected without a
 * sync, so the kernel clamps the index into the table (never an out-of-bounds read) and the Python
 * wrapper raises IndexError whenever it is handed host-resident timesteps. */
int ldmseg_add_noise(const float* original, const float* noise, const int64_t* timesteps_dev,
                     const float* alphas_cumprod_dev, int n_train_timesteps, float scale, float* out, int B,
                     size_t per_sample, void* stream);
int ldmseg_remove_noise(const float* noisy, const float* noise, const int64_t* timesteps_dev,
                        const float* alphas_cumprod_dev, int n_train_timesteps, float scale, float* out, int B,
                        size_t per_sample, void* stream);

/* ---- sampler: TrainerDiffusion.sample (trainers_ldm_cond.py:1045-1170) -------------------- */
typedef struct {
  int32_t n_steps;               /* len(scheduler.timesteps) */
  const int64_t* timesteps;      /* host, descending */
  const float* coef;             /* host [n_steps][4]: sqrt_a_t, sqrt_b_t, sqrt_a_prev, sqrt_b_prev */
  int32_t prediction_type, clip_sample, self_condition;
  float clip_sample_range;
  /* build-defined mask inpainting (absent from the reference): all NULL = off */
  const uint8_t* known_dev;      /* [B,1,L,L] 1 = latent given */
  const float* z0_dev;           /* [B,4,L,L] scaled known latents */
  const float* noise_dev;        /* [B,4,L,L] the fixed noise draw */
  const float* paste_coef;       /* host [n_steps][2]: sqrt_a, sqrt_b of the NEXT timestep (last: 1,0) */
} ldmseg_sample_cfg;
/* Runs the whole loop on `stream`: latents [B,4,L,L] holds the initial noise on entry and the
 * final latents on exit (last step = pred_original_sample, :1154-1156).  If all_latents is
 * non-NULL it receives every step's latents [n_steps][B,4,L,L] (return_all_latents). */
int ldmseg_sample_loop(ldmseg_unet* h, const ldmseg_sample_cfg* cfg, float* latents, const float* rgb_latents,
                       int B, int L, float* all_latents, void* stream);

/* ---- bit codec of segment ids: COCO.encode_bitmap / decode_bitmap (ldmseg/data/coco.py:377-390) ---- */
/* ids [B,HW] int64 -> bits [B,n_bits,HW] fp32 (LSB first; ids == ignore_label -> fill_value), then *mul+add
 * (mul=2, add=-1 gives the seg-VAE input of trainers_ldm_cond.py:369); ignore_mask [B,HW] u8 may be NULL. */
int ldmseg_bit_encode(const int64_t* ids, int B, int n_bits, int HW, int64_t ignore_label, float fill_value, float mul,
                      float add, float* bits, uint8_t* ignore_mask, void* stream);
/* x [B,n_bits,HW] fp32 -> ids [B,HW] int64: bit k set iff x[:,k] > 0 */
int ldmseg_bit_decode(const float* x, int B, int n_bits, int HW, int64_t* ids, void* stream);

/* ---- diagnostics --------------------------------------------------------------------------- */
const char* ldmseg_last_error(void);
const char* ldmseg_version(void);
/* Kernel-family timing (HIP events on the launch stream).  enable=1 records an event pair
 * around every launch of the family (0 igemm, 1 attention, 2 groupnorm, 3 layernorm, 4 other);
 * ldmseg_profile_read synchronises the recorded events and returns launches / total ms /
 * algorithmic FLOPs and bytes since the last reset. */
int ldmseg_profile_enable(int enable);
int ldmseg_profile_read(int family, int64_t* launches, double* total_ms, double* flops, double* bytes);
int ldmseg_profile_reset(void);
/* one CSV line per recorded launch (family,label,ms,flops); label carries the launch shape */
int ldmseg_profile_dump(const char* path);
/* measurement knobs (defaults = the shipped configuration): key 1 = igemm tile-policy bits in value[8..12] (and, in
 * -DLDMSEG_IGEMM_ABLATE builds only, phase-ablation flags in value[0..7]); key 2 = attention query-tile choice;
 * keys 3/4 = low/high half of a device buffer for per-workgroup s_memtime stamps (ablate builds); key 5 = run every
 * igemm launch with that entry of the instantiation list (-1 = off; the launch table and the shape rules are bypassed -
 * tools/tune_igemm.py); keys 6/7 = ldmseg_bench_igemm only: number of weight copies the timing loop rotates over, and
 * whether the timed launch carries a folded LayerNorm; key 8 = GroupNorm kernel choice (0 = shipped, bit 0 = two-launch
 * scheme everywhere, bit 1 = cooperative kernel from 16x16 maps up, bit 2 = two-pass instead of one-pass small-map kernel, bit 5 (round 6) = no
 * one-workgroup-per-(image, group) kernel on the 16x16 / 32x32 maps); key 9 = K order of 3x3 conv launches: -1 = shipped
 * rule (channel-major on large maps with many input channels), 0 = (tap, channel) everywhere, 1 = (channel tile, tap,
 * channel) wherever the layer holds that packing; key 10 = cooperative GroupNorm hand-off: 1 = every workgroup computes
 * its partners' statistics itself instead of waiting for them (the path a workgroup takes when its partners are not
 * co-resident; results are bit-identical), 0 = shipped; key 11 = bound of the partner poll in microseconds (default 100);
 * key 12 = row-local fusion of the 320-channel transformer feed-forward (bit 0: LayerNorm_3 -> GEGLU -> ff.net.2, bit 1: +
 * proj_out; default 3, 0 = the unfused launches); key 13 bit 8 = no start-chunk rotation in that kernel; key 14 = step tail
 * (bit 0: dedicated conv_out kernel in bf16, bit 1: the sampling loop's scheduler step / self-condition / next-input pack in
 * its epilogue; default 3); key 15 = fp8 attention on the block-scaled 2x-rate MFMAs where the shape allows (head dim 40,
 * tokens a multiple of 128; default 1, 0 = the unscaled fp8 MFMAs of attention_fp8.hip; 0x111 = scaled MFMAs with exp +
 * convert instead of the direct e4m3 byte - attention_mx.hip); key 16 = row-local fusion of the 320-channel transformer entry
 * (proj_in -> LayerNorm_1 -> q|k|v in one launch, tproj.hip; bit 0 on, bit 1 loader block rotation; default 3, 0 = the unfused
 * launches); key 2 values: 0 = shipped rule, 7 = the round-3 rule (attention3.hip at head dim 40), 11..14 = attention4.hip forced
 * (8 / 4 waves, lazily tracked / every-tile maxima); (key 17 was round 5's weight-streaming kernel - measured level with
 * igemm_kernel, now a record under tools/experiments/);
 * key 19 = resnet conv2 + conv_shortcut as one launch with an extra centre tap (bf16; default 1, 0 = two launches);
 * key 20 = ff.net.2 and proj_out of the 640- / 1280-channel transformers as one chained Linear over [g | h] (bf16; default 1);
 * key 21 = upsampler convs (nearest x2 -> conv3x3) as four 2x2-tap phase convs on the low-resolution map (bf16; default 1);
 * key 22 = the GroupNorm in front of a 320-channel transformer as a statistics pass + a sweep over the fused entry's LDS tile
 * (tproj.hip; bf16, maps of a multiple of 128 pixels; default 1, 0 = a GroupNorm launch of its own, n > 1 = n <= 64 pixel chunks);
 * key 23 (round 6) = K-sliced bf16 igemm launches reduce their slabs inside the launch when every (tile, slice) item has a
 * co-resident workgroup: bit 0 on for the 256-row tile forms (where it measured 2-4 % faster than slabs + a finish launch), bit 3 on
 * for every tile form that has the instantiation (measured 2-20 % slower on the 128-row tiles with 8 slices), bit 1 zero-length
 * partner poll (every workgroup but a tile's last arriver gives up at once and the last arriver reduces their shares - the path that
 * needs no co-residency; results are bit-identical), bit 2 = resnet conv1 -> norm2 on the maps whose conv runs on those tiles as
 * conv (finished in-launch) + GroupNorm instead of slabs + the fused finish-GroupNorm launch (moves where bf16 rounding happens, as on
 * the maps whose convs are not K-sliced), bits 8-23 = poll bound in microseconds (0 = 200); default 5, 0 = slabs + finish launches
 * everywhere.  Bits 0 / 1 / 3 never change a result bit;
 * key 24 = tuning: every plain-store igemm launch runs entry (v & 0xff) of the instantiation list with (v >> 8) K slices as if the
 * launch table held that entry (-1 = off; unlike key 5 the extra-tap / phase-conv routes stay). */
int ldmseg_debug_set(int key, int value);
/* current value of a knob (keys 1, 9, 12, 14, 15, 16, 19, 20, 21, 22, 23); key -1 = the shipped default of key 1.  Tests restore through this, never a literal.
 * key 10 = number of cooperative-GroupNorm workgroups that took the self-computing path in ldmseg_op_* launches so far. */
int ldmseg_debug_get(int key);

#ifdef __cplusplus
}
#endif
#endif /* LDMSEG_HIP_H_ */
