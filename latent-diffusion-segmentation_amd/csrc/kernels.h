// Host-side launch API of the gfx950 kernels (internal; the public boundary is
// include/ldmseg_hip.h).  All tensors are NHWC ("[B, H*W, C]") in the compute
// dtype (bf16 or f32) unless a name says otherwise.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

namespace ldmseg {

enum DType : int { DT_F32 = 0, DT_BF16 = 1 };

enum Epilogue : int {
  EPI_STORE = 0,   // out[m][n] = acc + bias (+rowbias) (+resid), optional SiLU
  EPI_GEGLU = 1,   // packed (a,g) 16-column interleave -> out[m][n/2] = a*gelu(g)
  EPI_NCHW_F32 = 2,  // out is float NCHW [B][n_valid][Ho*Wo]  (conv_out, VAE heads)
  EPI_CONVT2 = 3,  // ConvTranspose2d k2s2: n = tap*Cout + co scattered to (2y+dy,2x+dx)
  EPI_ROWS_F32 = 4,  // out is float row-major [M][ldo] whatever the compute dtype (attention scores of the image VAE)
};

// Exact n / d for 0 <= n < 2^31 by one v_mul_hi_u32 and a shift (Granlund-Montgomery: s = ceil(log2 d),
// M = ceil(2^(31+s) / d) < 2^32, q = (n * M) >> (31 + s)).  A runtime integer division is a ~40-instruction VALU sequence
// on this part; kernels whose first useful instruction sits behind several of them (row -> (image, y, x) of every gathered
// row, item -> tile) pay microseconds per launch for it.  Divisors come from the host, made by fastdiv_make().
struct FastDiv {
  unsigned mul = 0;     // 0: divisor is 1
  int shift = 0;        // applied to the high word
  int d = 1;
};
inline FastDiv fastdiv_make(int d) {
  FastDiv f;
  f.d = d < 1 ? 1 : d;
  if (f.d == 1) return f;
  int s = 0;
  while ((1ll << s) < f.d) ++s;
  const unsigned long long num = 1ull << (31 + s);
  f.mul = (unsigned)((num + (unsigned long long)f.d - 1) / (unsigned long long)f.d);
  f.shift = s - 1;
  return f;
}

struct IgemmParams {
  const void* src0 = nullptr;  // [B, Hi*Wi, C0]
  const void* src1 = nullptr;  // [B, Hi*Wi, C1]  channel-concat partner (torch.cat([h, skip],1))
  int C0 = 0, C1 = 0;
  // Extra centre tap (round 5): K continues past the 3x3 taps with C2 (+ C3) channels of a THIRD (and fourth) tensor read at the
  // output pixel itself - a 1x1 conv over torch.cat([src2, src3], 1) accumulated into the same tile.  The resnet's
  // conv_shortcut(x) + conv2(h) of diffusers' ResnetBlock2D as ONE launch: W rows are [9 * (C0 + C1) | C2 + C3], bias = b2 + bs.
  // 3x3, stride 1, no upsample, tap-major K, bf16 only (launch_igemm checks).
  const void* src2 = nullptr;
  const void* src3 = nullptr;
  int C2 = 0, C3 = 0;
  int B = 0, Hi = 0, Wi = 0, Ho = 0, Wo = 0;
  int taps = 1;     // 1 (1x1 conv / Linear) or 9 (3x3, pad 1)
  int pad = -1;     // top/left zero padding of a 3x3 conv; -1 = 1.  0 with stride 2 = F.pad(x,(0,1,0,1)) + conv(pad 0)
                    // (diffusers Downsample2D(padding=0) of the AutoencoderKL encoder); the bottom/right edge is bounds-checked
  int stride = 1;   // 1 | 2
  int up = 0;       // nearest x2 upsample folded into the gather (Upsample2D)
  // up4 (round 5): conv3x3(nearest_x2(x)) as four 2x2-tap phase convs on the LOW-resolution map.  An output pixel (2i + py, 2j + px)
  // of the upsampled map only ever sees 2 x 2 distinct source pixels - rows {i-1, i} (py = 0) or {i, i+1} (py = 1), columns
  // likewise - so the nine taps collapse into four with pre-summed weights: K = 4 C instead of 9 C, 2.25 x fewer MACs, same
  // result up to the rounding of the summed weights.  The launch walks 4 * B * Hi * Wi virtual rows ordered (phase, image, i, j);
  // taps = 4, Hi = Ho, Wi = Wo = the low-resolution map, W = [4 phases][N][4 taps][C], out = [B, 2Hi * 2Wi, ldo].
  int up4 = 0;
  int cm = 0;       // 3x3 only: W is packed in channel-major K order, k = (channel tile, tap, channel in tile), instead of
                    // (tap, channel); needs stride 1, pad 1, no upsample (launch_igemm checks)
  int M = 0;        // B*Ho*Wo
  int N = 0;        // packed rows of W (multiple of the N tile)
  int n_valid = 0;  // real output channels (store mask)
  const void* W = nullptr;       // [N][taps*(C0+C1)] compute dtype, K = (tap, channel)
  const float* bias = nullptr;   // [N] (packed order) or null
  const float* rowbias = nullptr;  // [B][rb_stride] per-image bias (time embedding) or null
  int rb_stride = 0;
  // LayerNorm folded into the GEMM (transformer norm1 -> q|k|v, norm3 -> GEGLU): X is the un-normalised tensor, W holds
  // gamma_k * W[n][k], c1[n] = sum_k W[n][k] of that packed matrix, bias[n] = sum_k beta_k W0[n][k] + b[n], and
  // out = rstd_m * (acc - mean_m * c1[n]) + bias[n] with (mean_m, rstd_m) = rowstats[m] from launch_rowstats.
  const float* rowstats = nullptr;   // [M][2] or null
  const float* c1 = nullptr;         // [N]
  const void* resid = nullptr;   // [M][ldr] residual, compute dtype, may alias out
  int ldr = 0;
  void* out = nullptr;
  int ldo = 0;
  int epi = EPI_STORE;
  int silu = 0;
  int cout = 0;     // EPI_CONVT2: real Cout per tap
  // split-K: when splits>1 partial sums go to `partial` [splits][M][N] f32 and
  // igemm_splitk_finish applies the epilogue.
  int splits = 1;
  float* partial = nullptr;
  int no_finish = 0;             // splits > 1: leave the slabs as they are, the caller runs its own finish (launch_finish_groupnorm)
  // In-launch cooperative finish (round 6): cf_ctr = the caller's counter region (igemm_cf_bytes() bytes, zeroed once; a handle
  // owns one; launches that share a region must be ordered on one stream) or null = two launches as before.  The launcher decides
  // per launch (every K slice of every tile must have its own co-resident workgroup) and sets cf / cf_bytes / cf_poll_ticks.
  unsigned long long* cf_ctr = nullptr;
  int cf = 0;
  unsigned cf_bytes = 0;         // bytes of the slab set (buffer range of the write-through stores / loads)
  int cf_poll_ticks = 20000;     // bound of the partner poll in 100 MHz ticks
  int cf_diag = 0;               // index of the region's diagnostics word (workgroups that gave up waiting)
  const void* zeros = nullptr;   // >= 16 B of zeros (set by the launcher)
  // set by the launcher: divisions the kernels need (by Ho*Wo, Wo, the number of n tiles / tiles / K slices, K tiles per tap)
  FastDiv fd_hwo, fd_wo, fd_nt, fd_ntiles, fd_nsplit, fd_tpt, fd_mphase;
  int x3 = 0;                    // fp32 launches only: split-bf16 arithmetic (hi + lo, three bf16 MFMAs per product block) instead of
                                 // the exact fp32 MFMA - compute_dtype "bf16x3" of the handles.  2 (round 6): W already holds the
                                 // hi | lo planes (launch_split_planes at create time): no weight split in the K loop
  int w_dynamic = 0;             // host-side: W is an activation tensor of this call (image-VAE attention), never pre-split
  int dbg = 0;                   // ablation flags for profiling experiments (results are wrong when != 0)
  unsigned long long* ts = nullptr;   // LDMSEG_IGEMM_ABLATE builds: per-workgroup s_memtime stamps (wave 0)
};

// returns 0 or a negative error (bad shape)
int launch_igemm(const IgemmParams& p, int dtype, hipStream_t s);
size_t igemm_partial_bytes(const IgemmParams& p);
int igemm_warm();   // per-device lazy state (zero page) created now instead of inside the first launch
int igemm_plan_splits(const IgemmParams& p, int dtype);
// cooperative split-K finish inside the igemm launch (IgemmParams::cf_ctr).  Region size; debug key 23: bit 0 = on for the 256-row
// tile forms (default, with bit 2: engine.hip conv_groupnorm), bit 3 = on for every tile form with the instantiation, bit 1 = zero-length poll (every workgroup that is not
// the last arriver gives up: the last arriver reduces the whole tile), bits 8.. = poll bound in microseconds (0 = default 200);
// fallbacks are counted in the region's last word.
size_t igemm_cf_bytes();
void igemm_set_cf_mode(int mode);
int igemm_get_cf_mode();
// tuning (debug key 24): >= 0 = every plain-store launch runs entry (v & 0xff) of the instantiation list with (v >> 8) K slices, as
// if the launch table held that entry for its shape (unlike key 5 this keeps the extra-tap / launch-table routes); -1 = off
void igemm_set_table_override(int v);
void attention_set_qf1(int v);    // tuning knob: force 16 query rows per wave
void igemm_set_tsbuf(void* dev_buf);   // LDMSEG_IGEMM_ABLATE builds only
void igemm_set_dbg(int flags);
void ops_bench_knob(int key, int value);   // ldmseg_bench_igemm (ops_api.hip): 6 = number of weight copies rotated, 7 = folded-LN launch
// K order a 3x3 conv launch uses: -1 = the shipped rule (channel-major for stride-1 convs without upsample on maps of
// >= 4096 pixels with >= 640 input channels: there one tap sweep over the X operand no longer fits an XCD's L2 and the
// channel-major order measured -6 %; smaller launches measured +2..8 % and keep the tap-major order), 0 = tap-major
// everywhere, 1 = channel-major wherever the kernel supports it.  Layers that can meet the rule hold both packings.
void igemm_set_cm_mode(int mode);
int igemm_get_cm_mode();
bool igemm_conv_cm(int hw, int ctot, int n, int k, int stride, int up, int dtype);
// conv3x3 on a nearest-x2 upsampled map as four 2x2 phase convs (IgemmParams::up4): whether the layer shape / map size has the
// launch and debug key 21 allows it, and the weight packing it reads ([4][Npad][4][Cipad] from OIHW fp32)
bool igemm_up4_ok(int B, int H, int W, int C, int N, int dtype);
void igemm_set_up4_mode(int on);
int igemm_get_up4_mode();
int launch_pack_up4(const float* w, void* out, int Co, int Ci, int Npad, int Cipad, int dtype, hipStream_t s);
bool igemm_xt_ok(const IgemmParams& p, int dtype);   // the launch (with src2 / C2 set) has an extra-tap instantiation and key 19 allows it
void igemm_set_xt_mode(int on);
int igemm_get_xt_mode();
// out[n][0:K1] = a[n][:], out[n][K1:K1+K2] = b[n][:]  (rows of two packed weight matrices side by side), compute dtype
// Weights of two chained Linear layers with nothing in between, y = Wp (W2 g + b2 + h) + bp = [Wp W2 | Wp] [g | h] + (bp + Wp b2):
// wcat [C][5C] fp32 = [Wp W2 | Wp], bcat [C] = bp + Wp b2 (wp [C][C], w2 [C][4C]; fp32 products, create time only)
int launch_chain_weights(const float* wp, const float* w2, const float* b2, const float* bp, float* wcat, float* bcat, int C, hipStream_t s);
int launch_vec_add(const float* a, const float* b, float* out, int n, hipStream_t s);
// packed fp32 weights [rows][K] -> split-bf16 planes in place (IgemmParams::x3 == 2); nfloats a multiple of 32
int launch_split_planes(void* w, size_t nfloats, hipStream_t s);
int launch_concat_rows(const void* a, int K1, const void* b, int K2, void* out, int N, int dtype, hipStream_t s);
void igemm_force_cfg(int cfg);   // tuning tool: >= 0 runs every launch with that entry of the instantiation list, -1 = off
int igemm_get_dbg();       // current (policy << 8) | ablation flags
int igemm_default_dbg();   // the shipped value
// template instantiation + plan of the most recent launch_igemm (test introspection)
struct IgemmDispatch { int dtype, bm, bn, wm, wn, nst, pipe, ldr, splits, grid, lnf, cm, cf, xt, up4; };   // cf: the K slices were finished inside the launch
IgemmDispatch igemm_last_dispatch();
std::string igemm_dispatch_name(const IgemmDispatch& d);
void igemm_log_enable(int on);      // start (and clear) / stop recording the distinct instantiations launched
void igemm_log_note(const char* name);   // recorded while logging is on (the fused feed-forward kernel: "mlp_fused<bf16,proj=P>")
std::string igemm_log_read();       // newline-separated   // ablation flags (profiling only)   // K-loop ring depth (2, 3, 4) - tuning knob
// tile the launcher would pick (for weight padding): N tile size for a given N.
int igemm_pick_bn(int n_real, int epi);

struct GNParams {
  const void* src0 = nullptr; const void* src1 = nullptr;
  int C0 = 0, C1 = 0, B = 0, HW = 0, groups = 32;
  const float* gamma = nullptr; const float* beta = nullptr;
  float eps = 1e-5f; int silu = 0;
  void* out = nullptr;         // [B, HW, C0+C1]
  float* partial = nullptr;    // workspace [B][nchunk][groups][2]
  int nchunk = 0;
  // filled by launch_groupnorm (host): quantities the kernels would otherwise divide for before their first load
  int cpg = 0, vx = 0, ty = 0, per = 0;                 // channels per group, threads per pixel row, pixel rows per trip, pixels per chunk
  FastDiv fd_cpg, fd_vx, fd_aux;                        // aux: units / vectors per pixel of the small / fused kernels
  double inv_n = 0.0;                                   // 1 / (HW * cpg)
  // cooperative one-pass kernel (gn_coop_kernel).  sync_region: the caller's hand-off region (gn_sync_bytes() bytes, zeroed
  // once by gn_sync_init; a handle owns one) or null = a region of the per-device ring.  The launcher derives the rest.
  void* sync_region = nullptr;
  unsigned long long* sync = nullptr;                   // record granules [slab][8 splits][8 entries]
  unsigned long long* sync_ctr = nullptr;               // per-slab ticket counters (the generation tag is drawn on the device)
  unsigned long long* sync_diag = nullptr;              // workgroups that computed a missing partner's record themselves
  int splits = 1;                                       // workgroups that share one (image, group block)
  int coop_mode = 0;                                    // 1: never poll, always take the self-computing path (tests)
  int poll_ticks = 10000;                               // bound of the partner poll in 100 MHz wall-clock ticks
  int poll_us = -1;                                     // >= 0: this launch's poll bound instead of the process-wide one (a handle backing off)
};
size_t gn_sync_bytes();
int gn_sync_init(void* region, hipStream_t s);
int gn_warm();                                          // per-device ring allocated now instead of inside the first launch
// test / diagnostic knobs of the cooperative kernel: mode 1 = always self-compute, poll_us = bound of the poll (default 100)
void groupnorm_set_coop(int mode, int poll_us);
// workgroups that took the self-computing path since the region was initialised (region null: sum over the ring of the current device)
long long gn_coop_fallbacks(const void* region);
const void* gn_sync_diag_ptr(const void* region);      // device address of a region's 8-byte fallback counter
int gn_nchunk(int B, int HW);
int launch_groupnorm(const GNParams& p, int dtype, hipStream_t s);
// statistics only: p.partial[b][chunk][group] = {mean, M2} of p.nchunk pixel chunks of ceil(HW / nchunk) pixels (gn_partial_kernel);
// out / gamma / beta / silu are not used.  Consumer: launch_proj_qkv_fused with a GnFold (tproj.hip)
int launch_groupnorm_stats(const GNParams& p, int dtype, hipStream_t s);
// Split-K finish of a conv fused with the GroupNorm (+SiLU) that consumes it (resnet conv1 -> norm2 on the 8x8 / 16x16
// maps): out = GN(sum_z partial[z] + bias + rowbias[image]) in ONE launch; the conv's own output is never stored.
// ip: the igemm launch (partial, splits, M, N, n_valid, bias, rowbias, rb_stride); g: B, HW, C0 = channels, gamma, beta,
// eps, silu, out.  finish_groupnorm_ok: whether the shape has an instantiation (otherwise: igemm's finish + launch_groupnorm).
bool finish_groupnorm_ok(int B, int HW, int C, int dtype);
int launch_finish_groupnorm(const IgemmParams& ip, const GNParams& g, int dtype, hipStream_t s);
void groupnorm_set_variant(int v);   // tuning knob: bit0 = no cooperative one-pass kernel (64x64 maps on the two-launch path)

// LayerNorm over the last dim of [M][C] (+ optional SiLU) - also LayerNorm2d in NHWC
int launch_layernorm(const void* x, void* y, const float* gamma, const float* beta, int M, int C,
                     float eps, int silu, int dtype, hipStream_t s);

// self-attention on fused qkv [B, N, 3C] (q | k | v, channel = head*d + i) -> out [B, N, C]
// dtype 2 = fp32 tensors with the products as three bf16 MFMAs on hi/lo splits (compute_dtype "bf16x3")
int launch_attention(const void* qkv, void* out, int B, int N, int C, int heads, int dtype, hipStream_t s);

// Row-local fusion of the transformer feed-forward at the 320-channel level (tfuse.hip): LayerNorm_3 -> GEGLU -> ff.net.2 (+h)
// [-> proj_out (+x)] in one launch.  mlp_fused_ok: the shape / dtype has the kernel and the mode (debug key 12) allows it.
bool mlp_fused_ok(int C, int dtype);
bool mlp_fused_proj();                      // mode bit 1: proj_out rides along
void mlp_fused_set_mode(int m);             // bit 0: fuse norm3 -> GEGLU -> ff.net.2; bit 1: + proj_out  (default 3)
int mlp_fused_get_mode();
size_t mlp_fused_stream_bytes(int C);
// w1: packed GEGLU weights [8C][C] (16-row value | gate interleave, gamma folded in), w2: [C][4C], wp: proj_out [C][C]; bf16
int launch_pack_mlp_stream(const void* w1, const void* w2, const void* wp, void* out, int C, hipStream_t s);
// tproj.hip (bf16, C = 320): proj_in -> LayerNorm_1 -> q|k|v of a transformer in one row-local launch (writes h and q|k|v).
// proj_qkv_fused_ok: the shape / dtype has the kernel and the mode (debug key 16) allows it.
bool proj_qkv_fused_ok(int C, int M, int dtype);
void proj_qkv_set_mode(int m);
int proj_qkv_get_mode();
size_t proj_qkv_stream_bytes(int C);
int launch_pack_proj_qkv_stream(const void* wp, const void* wqkv, void* out, int C, hipStream_t s);
// gn != null: x is the transformer's RAW input and its GroupNorm (32 groups, no SiLU) runs inside the kernel from the statistics of
// launch_groupnorm_stats (partial [B][nchunk][32][2]); M % HW == 0, HW % 128 == 0 (proj_qkv_gn_fold_ok)
struct GnFold {
  const float* partial = nullptr;
  const float* gamma = nullptr;
  const float* beta = nullptr;
  int nchunk = 0, HW = 0;
  float eps = 1e-6f;
};
bool proj_qkv_gn_fold_ok(int HW);
int proj_qkv_gn_chunks(int B, int HW);    // nchunk for launch_groupnorm_stats (<= 64: the kernel keeps a lane's records in registers)
int launch_proj_qkv_fused(const void* x, void* h, void* qkv, const void* stream, const float* bias4, const void* zeros, int M, int C, float eps,
                          const GnFold* gn, hipStream_t s);
int launch_mlp_fused(const void* h, void* out, const void* x2, const void* stream, const float* bias1, const float* bias2,
                     const float* bias3, const void* zeros, int M, int C, float eps, int proj, int rows_per_image, hipStream_t s);
void mlp_fused_set_dbg(int flags);          // bit 8: no start-chunk rotation; bits 0-7: phase ablation (LDMSEG_TFUSE_ABLATE builds only)
const void* igemm_zero_page();              // >= 64 KB of zeros on the current device (igemm's padding page)

// per-row (mean, rstd) of [M][C] (LayerNorm statistics, two-pass centred variance) -> stats [M][2] f32
int launch_rowstats(const void* x, float* stats, int M, int C, float eps, int dtype, hipStream_t s);
// out[r][k] = colscale[k] * (src_row[r] >= 0 ? w[src_row[r]][k] : 0)   (colscale null = 1)
int launch_repack_rows_scaled(const float* w, void* out, const int* src_row_dev, int Npad, int K, const float* colscale,
                              int dtype, hipStream_t s);
// out[n] = sum_k W[n][k] (+ add[n]) over a packed [N][K] matrix in `dtype`, fp32 accumulation
int launch_rowsum(const void* W, const float* add, float* out, int N, int K, int dtype, hipStream_t s);

// fp8 (e4m3) operand path of the bf16 attention for the long-context levels (attention_fp8.hip): head dims 40 / 80 only
size_t attention_fp8_scratch_bytes(int B, int N, int C, int heads);
int launch_attention_fp8(const void* qkv, void* kv8_scratch, void* out, int B, int N, int C, int heads, hipStream_t s);
// the same on the 2x-rate block-scaled MFMAs (attention_mx.hip): head dim 40, N a multiple of 128; launch_attention_fp8 routes to it
bool attention_mx_ok(int N, int C, int heads);
size_t attention_mx_scratch_bytes(int B, int N, int C, int heads);
int launch_attention_mx(const void* qkv, void* kv8_scratch, void* out, int B, int N, int C, int heads, hipStream_t s);
void attention_mx_set_mode(int m);          // 0: keep attention_fp8.hip's unscaled MFMAs everywhere (A/B)
int attention_mx_get_mode();

// NCHW f32 [B,C,HW] -> NHWC compute dtype [B,HW,Cpad] (zero padded channels), optional affine a*x+b
int launch_pack_nchw(const float* x, void* y, int B, int C, int HW, int Cpad, float mul, float add,
                     int dtype, hipStream_t s);
// three NCHW f32 sources concatenated on channels (torch.cat([latents, rgb, cond],1)); null = skip
int launch_pack_concat3(const float* a, int Ca, const float* b, int Cb, const float* c, int Cc,
                        void* y, int B, int HW, int Cpad, int dtype, hipStream_t s);

// timestep embedding + time MLP + all time_emb_proj: see misc.hip
int launch_time_embed(const int64_t* t_dev, int t_count, int64_t t_host, int B, float* sinus /*[B,320]*/,
                      hipStream_t s);
// y[B][N] = act_in(x[B][K]) @ W[N][K]^T + bias ; f32 small-M GEMV (M<=64)
int launch_small_linear(const float* x, const float* W, const float* bias, float* y, int B, int K, int N,
                        int silu_in, int silu_out, hipStream_t s);

// bilinear x2 (align_corners=False) NHWC compute dtype [B,H,W,C] -> NCHW f32 [B,C,2H,2W]
int launch_bilinear2x_nchw(const void* x, float* y, int B, int H, int W, int C, int dtype, hipStream_t s);

// fused decode tail: bilinear x2 + argmax over C + max-softmax prob; mask_th < 0 disables the threshold
int launch_bilinear2x_argmax(const void* x, int64_t* ids, float* prob, int B, int H, int W, int C, float mask_th,
                             int64_t ignore_label, int dtype, hipStream_t s);

// weight repack (f32 torch layout -> compute dtype [N][K])
// conv OIHW [Co][Ci][kh][kw] -> [Npad][kh*kw][Cipad] ; rows >= Co and channels >= Ci are zero
// cm_tile > 0: channel-major K order [Npad][Cipad / cm_tile][kh*kw][cm_tile] (IgemmParams::cm; cm_tile = K elements per tile)
int launch_repack_conv(const float* w, void* out, int Co, int Ci, int KH, int KW, int Npad, int Cipad,
                       int dtype, hipStream_t s, int cm_tile = 0);
// generic row gather: out[r][:] = (src_row[r] >= 0) ? w[src_row[r]][0:K] : 0  (Linear, GEGLU interleave, qkv concat)
int launch_repack_rows(const float* w, void* out, const int* src_row_dev, int Npad, int K, int dtype, hipStream_t s);
// ConvTranspose2d [Ci][Co][2][2] -> [4*Co][Ci]  (n = (dy*2+dx)*Co + co)
int launch_repack_convt2(const float* w, void* out, int Ci, int Co, int dtype, hipStream_t s);

// DDIM step (ddim_scheduler.py:218-269), elementwise, bit-exact op order
struct DdimCoef { float sqrt_a_t, sqrt_b_t, sqrt_a_prev, sqrt_b_prev; int pred_type; int clip; float clip_range; int use_clipped; };
int launch_ddim_step(const float* eps, const float* x, float* prev, float* x0, size_t n, DdimCoef c, hipStream_t s);
// The tail of a denoising step in one launch (tail.hip, bf16): conv_out on a halo-resident pixel tile and, when ddim != 0,
// the scheduler update, inpainting paste, self-condition write and the next step's packed UNet input in its epilogue.
struct StepTail {
  const void* x = nullptr;        // [B, H*W, 320] SiLU(conv_norm_out(h)), bf16
  const void* w = nullptr;        // conv_out weights packed [N][9*320] bf16 (rows 0..3 real)
  const float* bias = nullptr;
  const void* zeros = nullptr;
  int B = 0, H = 0, W = 0;
  float* eps_out = nullptr;       // [B,4,H,W] fp32: the model output (plain forward); may be null when ddim != 0
  int ddim = 0, last = 0;
  DdimCoef c{};
  float* latents = nullptr;       // [B,4,H,W] fp32, updated in place (last step: pred_original_sample)
  float* cond = nullptr;          // self-condition <- pred_original_sample, or null
  const float* rgb = nullptr;
  void* xin_next = nullptr;       // [B, H*W, 64] bf16: next step's packed input [latents | rgb | cond | 0], or null
  const uint8_t* known = nullptr; const float* z0 = nullptr; const float* noise = nullptr;   // inpainting paste
  float sa = 0.f, sb = 0.f;
};
bool conv_out_tail_ok(int C, int H, int W, int dtype);     // the shape has the kernel and debug key 14 bit 0 allows it
bool step_tail_fused();                                      // key 14 bits 0 and 1: the sampling loop hands the scheduler step to it
void step_tail_set_mode(int m);
int step_tail_get_mode();
int launch_conv_out_tail(const StepTail& t, hipStream_t s);
// out = m ? (sa*z0 + sb*noise) : cur   (inpainting paste; m is u8 [B,1,L,L] broadcast on channels)
int launch_inpaint_paste(float* cur, const float* z0, const float* noise, const uint8_t* known, float sa, float sb,
                         int B, int C, int HW, hipStream_t s);
// add_noise / remove_noise with per-sample timesteps (ddim_scheduler.py:155-216)
int launch_add_noise(const float* x0, const float* noise, const int64_t* t_dev, const float* ac_dev, int n_train, float scale,
                     float* out, int B, size_t per, int remove, hipStream_t s);
// softmax(scale * s) over the last dim of fp32 scores [rows][n] -> probabilities in the compute dtype [rows][n]
int launch_softmax_rows(const float* s, void* p, int rows, int n, float scale, int dtype, hipStream_t st);
int launch_axpby(const float* x, float a, float b, float* y, size_t n, hipStream_t s);  // y = a*x + b
// bit codec of segment ids (ldmseg/data/coco.py:377-390)
int launch_panoptic_postprocess(const float* logits, int B, int C, int HW, int threshold_output, int threshold_mode,
                                float mask_th, int count_th, double overlap_th, int64_t ignore_label, int32_t* labels,
                                int32_t* panoptic, uint8_t* keep, int32_t* counts, int32_t* mask_counts, hipStream_t s);
// fused evaluation tail on the decoder's 4L NHWC output: x2 -> input size -> crop -> original size -> panoptic post-processing
int launch_panoptic_from_decoder(const void* x4, int B, int H4, int W4, int C, int dtype, int in_h, int in_w,
                                 const int32_t* boxes_host, const int32_t* sizes_host, const int64_t* offsets_host,
                                 int threshold_output, int threshold_mode, float mask_th, int count_th, double overlap_th,
                                 int64_t ignore_label, int32_t* labels, int32_t* panoptic, uint8_t* keep, int32_t* counts,
                                 int32_t* mask_counts, hipStream_t s, float* volume = nullptr);   // volume: test hook, [sum_b C * h_b * w_b] fp32
int launch_bit_encode(const int64_t* ids, float* out, uint8_t* ignore, int B, int n, int HW, int64_t ignore_label,
                      float fill, float mul, float add, hipStream_t s);
int launch_bit_decode(const float* x, int64_t* out, int B, int n, int HW, hipStream_t s);
// mean + exp(0.5*clamp(logvar))*noise on NCHW moments [B,8,HW] -> [B,4,HW]
int launch_posterior_sample(const float* moments, const float* noise, float* out, int B, int HW, hipStream_t s);

}  // namespace ldmseg
