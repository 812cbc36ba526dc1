// Host runtime of libldmseg_hip.so: weight repacking, workspace planning, the UNet /
// seg-VAE executors, the DDIM sampling loop and the C ABI (include/ldmseg_hip.h).
//
// Layout in HBM: activations NHWC [B, H*W, C] in the compute dtype;
// conv/linear weights [N][K] with K = (tap, channel) so one 128-B line of a weight row
// is one K tile; norm parameters, biases and the time-embedding MLP stay fp32.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ldmseg_hip.h"
#include "kernels.h"

using namespace ldmseg;

namespace {

thread_local std::string g_err;
// Bumped whenever a process-global tuning knob that the workspace plan depends on changes (ldmseg_debug_set keys 1 / 5:
// tile policy and forced instantiation decide the split-K slice counts, i.e. the size of the partial-sum scratch).
// Handles remember the epoch their cached plan was made under and re-plan when it moved.
int g_plan_epoch = 0;
int g_gnfold_mode = 1;  // debug key 22: 1 (default) = the GroupNorm in front of a 320-channel transformer as a statistics pass + a sweep inside
                        // proj_ln_qkv (tproj.hip); 0 = a GroupNorm launch of its own; n > 1 = on, with n pixel chunks per image
int g_ffp_mode = 1;     // debug key 20: 1 (default) = ff.net.2 and proj_out of the 640- / 1280-channel transformers as one chained Linear
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e__ = (expr);                                                                  \
    if (e__ != hipSuccess)                                                                    \
      return fail(LDMSEG_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));          \
  } while (0)
#define TRY(expr)                                                                             \
  do {                                                                                        \
    int r__ = (expr);                                                                         \
    if (r__ != 0) {                                                                           \
      if (g_err.empty())                                                                      \
        g_err = std::string(#expr) + " failed (" + std::to_string(r__) + ")" +                \
                (r__ == -3 ? std::string(": ") + hipGetErrorString(hipGetLastError()) : ""); \
      return r__;                                                                             \
    }                                                                                         \
  } while (0)

// Makes `device` current for the duration of a call on a handle (a process may hold handles on several GPUs; the
// kernels, workspaces and per-device launcher state of a handle all belong to cfg.device) and restores the caller's.
struct DeviceGuard {
  int prev = -1;
  bool changed = false;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) == hipSuccess && prev != device) changed = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceGuard() {
    if (changed) (void)hipSetDevice(prev);
  }
};

inline size_t rup(size_t v, size_t a) { return (v + a - 1) / a * a; }
// a handle's hand-off region of the cooperative GroupNorm kernel (norm.hip), zeroed once
int alloc_gn_sync(void** out) {
  void* p = nullptr;
  if (hipMalloc(&p, gn_sync_bytes()) != hipSuccess) return fail(LDMSEG_E_OOM, "hipMalloc of the GroupNorm hand-off region failed");
  if (gn_sync_init(p, nullptr) != 0 || hipStreamSynchronize(nullptr) != hipSuccess) { (void)hipFree(p); return fail(LDMSEG_E_HIP, "GroupNorm hand-off region init failed"); }
  *out = p;
  return 0;
}
// a handle's counter region of the in-launch split-K finish (igemm.hip, IgemmParams::cf_ctr), zeroed once
int alloc_cf_sync(void** out) {
  void* p = nullptr;
  if (hipMalloc(&p, igemm_cf_bytes()) != hipSuccess) return fail(LDMSEG_E_OOM, "hipMalloc of the split-K counter region failed");
  if (hipMemset(p, 0, igemm_cf_bytes()) != hipSuccess) { (void)hipFree(p); return fail(LDMSEG_E_HIP, "split-K counter region init failed"); }
  *out = p;
  return 0;
}
inline size_t esize(int dt) { return dt == DT_BF16 ? 2 : 4; }
inline int bke(int dt) { return dt == DT_BF16 ? 64 : 32; }  // channels per 128-B K tile

// ------------------------------------------------------------------ profiling
struct ProfFamily {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  std::vector<std::string> label;
  std::vector<double> lflops;
  int64_t launches = 0;
  double flops = 0, bytes = 0;
};
struct Profiler {
  bool on = false;
  ProfFamily fam[5];
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
  }
} g_prof;

struct ProfScope {
  int f; hipStream_t s; hipEvent_t a{}, b{}; bool on;
  ProfScope(int family, hipStream_t st, double flops, double bytes, bool dry, const std::string& label = "")
      : f(family), s(st) {
    on = g_prof.on && !dry;
    if (on) {
      g_prof.fam[f].label.push_back(label);
      g_prof.fam[f].lflops.push_back(flops);
      a = g_prof.get(); b = g_prof.get();
      (void)hipEventRecord(a, s);
      g_prof.fam[f].launches++;
      g_prof.fam[f].flops += flops;
      g_prof.fam[f].bytes += bytes;
    }
  }
  ~ProfScope() {
    if (on) { (void)hipEventRecord(b, s); g_prof.fam[f].ev.emplace_back(a, b); }
  }
};

// ------------------------------------------------------------------ device memory
struct DeviceArena {  // persistent parameters: chunked bump allocator
  std::vector<void*> chunks;
  char* cur = nullptr;
  size_t left = 0;
  size_t total = 0;
  int alloc(void** out, size_t bytes) {
    bytes = rup(bytes, 256);
    if (bytes > left) {
      const size_t chunk = bytes > ((size_t)256 << 20) ? bytes : ((size_t)256 << 20);
      void* p = nullptr;
      if (hipMalloc(&p, chunk) != hipSuccess) return fail(LDMSEG_E_OOM, "hipMalloc of parameter chunk failed");
      chunks.push_back(p);
      cur = (char*)p;
      left = chunk;
      total += chunk;
    }
    *out = cur;
    cur += bytes;
    left -= bytes;
    return 0;
  }
  void release() {
    for (void* p : chunks) (void)hipFree(p);
    chunks.clear();
    cur = nullptr;
    left = 0;
  }
};

struct Workspace {  // per-forward activations: persist (bump) + scratch (stack)
  char* base = nullptr;
  size_t cap = 0;
  size_t persist_top = 0, scratch_top = 0, scratch_base = 0;
  size_t persist_peak = 0, scratch_peak = 0;
  bool dry = false;
  bool overflow = false;   // a non-dry request went past the planned capacity (stale plan): the forward fails instead of writing out of range
  void begin(bool dry_, size_t scratch_base_) {
    dry = dry_;
    overflow = false;
    persist_top = 0;
    scratch_base = scratch_base_;
    scratch_top = 0;
    persist_peak = scratch_peak = 0;
  }
  void* persist(size_t bytes) {
    const size_t off = persist_top;
    persist_top += rup(bytes, 256);
    if (persist_top > persist_peak) persist_peak = persist_top;
    if (!dry && persist_top > (scratch_base ? scratch_base : cap)) { overflow = true; return base; }
    return dry ? (void*)(uintptr_t)(0x1000 + off) : base + off;
  }
  void* scratch(size_t bytes) {
    const size_t off = scratch_top;
    scratch_top += rup(bytes, 256);
    if (scratch_top > scratch_peak) scratch_peak = scratch_top;
    if (!dry && scratch_base + scratch_top > cap) { overflow = true; return base; }
    return dry ? (void*)(uintptr_t)(0x1000 + off) : base + scratch_base + off;
  }
  size_t mark() const { return scratch_top; }
  void reset(size_t m) { scratch_top = m; }
};

struct WeightMap {
  std::unordered_map<std::string, std::pair<const float*, int64_t>> m;
  int get(const std::string& key, int64_t numel, const float** out) const {
    auto it = m.find(key);
    if (it == m.end()) return fail(LDMSEG_E_WEIGHT, "missing state-dict key: " + key);
    if (it->second.second != numel)
      return fail(LDMSEG_E_WEIGHT, "state-dict key " + key + " has " + std::to_string(it->second.second) +
                                       " elements, expected " + std::to_string(numel));
    *out = it->second.first;
    return 0;
  }
};

struct ConvW {
  void* w = nullptr;      // [N][taps*cin_pad]
  float* bias = nullptr;  // [N]
  float* c1 = nullptr;    // [N] row sums of w when a LayerNorm is folded into this GEMM (IgemmParams::c1), else null
  int N = 0, n_valid = 0, cin_pad = 0, taps = 1, cout = 0;
  void* w_cm = nullptr;   // 3x3 layers that may run on large maps: second packing in channel-major K order (IgemmParams::cm)
  void* w_up4 = nullptr;  // upsampler convs, bf16: [4 phases][N][2x2 taps][C] with the 3x3 taps pre-summed per phase (IgemmParams::up4)
  int xt_cin = 0;         // extra-tap packings (ResnetW::conv2x): input channels of the 1x1 part behind the nine taps' columns
};
struct NormW { float* g = nullptr; float* b = nullptr; int C = 0; };

struct Builder {
  DeviceArena* arena;
  const WeightMap* wm;
  int dt;
  hipStream_t s;
  int64_t nparams = 0;
  std::vector<void*> temps;
  bool x3 = false;                                         // LDMSEG_BF16X3 handle: every GEMM weight matrix becomes hi | lo planes at finish()
  std::vector<std::pair<void*, size_t>> x3_w;              // (packed fp32 matrix, floats)
  void weights(void* w, size_t nfloats) { if (x3 && dt == DT_F32) x3_w.emplace_back(w, nfloats); }

  int f32_copy(const std::string& key, int64_t n, float** out, int64_t pad_to = 0) {
    const float* src;
    TRY(wm->get(key, n, &src));
    const int64_t tot = pad_to > n ? pad_to : n;
    void* p;
    TRY(arena->alloc(&p, tot * sizeof(float)));
    if (tot > n) HIP_TRY(hipMemsetAsync(p, 0, tot * sizeof(float), s));
    HIP_TRY(hipMemcpyAsync(p, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    *out = (float*)p;
    nparams += n;
    return 0;
  }
  int norm(const std::string& prefix, int C, NormW* out) {
    out->C = C;
    TRY(f32_copy(prefix + ".weight", C, &out->g));
    TRY(f32_copy(prefix + ".bias", C, &out->b));
    return 0;
  }
  // Conv2d / Linear: OIHW -> [Npad][k*k][cin_pad]; n_store >= Co forces explicit zero output channels
  int conv(const std::string& prefix, int Co, int Ci, int k, int cin_pad, ConvW* out, bool has_bias = true,
           int n_store = 0, int epi = EPI_STORE, bool cm = false) {
    const float* w;
    TRY(wm->get(prefix + ".weight", (int64_t)Co * Ci * k * k, &w));
    const int nreal = n_store > Co ? n_store : Co;
    const int bn = igemm_pick_bn(nreal, epi);
    const int Npad = (int)rup(nreal, bn);
    out->N = Npad;
    out->n_valid = nreal;
    out->cin_pad = cin_pad;
    out->taps = k * k;
    out->cout = Co;
    TRY(arena->alloc(&out->w, (size_t)Npad * k * k * cin_pad * esize(dt)));
    TRY(launch_repack_conv(w, out->w, Co, Ci, k, k, Npad, cin_pad, dt, s));
    weights(out->w, (size_t)Npad * k * k * cin_pad);
    if (cm && k == 3 && cin_pad % bke(dt) == 0 && dt == DT_BF16 && Npad % 160 == 0) {
      TRY(arena->alloc(&out->w_cm, (size_t)Npad * k * k * cin_pad * esize(dt)));
      TRY(launch_repack_conv(w, out->w_cm, Co, Ci, k, k, Npad, cin_pad, dt, s, bke(dt)));
    }
    nparams += (int64_t)Co * Ci * k * k;
    if (has_bias) TRY(f32_copy(prefix + ".bias", Co, &out->bias, Npad));
    else {
      void* p;
      TRY(arena->alloc(&p, Npad * sizeof(float)));
      HIP_TRY(hipMemsetAsync(p, 0, Npad * sizeof(float), s));
      out->bias = (float*)p;
    }
    return 0;
  }
  int upload_ints(const std::vector<int>& v, int** dev) {
    void* p;
    HIP_TRY(hipMalloc(&p, v.size() * sizeof(int)));
    temps.push_back(p);
    HIP_TRY(hipMemcpyAsync(p, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice, s));
    *dev = (int*)p;
    return 0;
  }
  int finish() {
    // (after every row sum / bias that is computed FROM the packed matrices: the folded-LayerNorm c1 vectors)
    for (const auto& e : x3_w) TRY(launch_split_planes(e.first, e.second, s));
    x3_w.clear();
    HIP_TRY(hipStreamSynchronize(s));
    for (void* p : temps) (void)hipFree(p);
    temps.clear();
    return 0;
  }
};

// ------------------------------------------------------------------ executor helpers
struct Act {
  void* p = nullptr;
  int C = 0, H = 0, W = 0;
};

struct Exec {
  Workspace* ws;
  int dt;
  int B;
  hipStream_t s;
  int attn_fp8_min_tokens = 0;
  void* gn_sync = nullptr;     // the handle's hand-off region of the cooperative GroupNorm (gn_sync_bytes())
  int gn_poll_us = -1;         // >= 0: poll bound of this handle's cooperative norms (backing off, see ldmseg_sample_loop)
  int x3 = 0;                  // fp32 handles in LDMSEG_BF16X3 mode: IgemmParams::x3 of every GEMM launch
  void* cf_sync = nullptr;     // the handle's counter region of the in-launch split-K finish (igemm_cf_bytes())
  bool dry() const { return ws->dry; }
  // a request beyond the planned workspace (a plan made under other tuning knobs): fail before anything is launched on it
  int ws_ok() const { return ws->overflow ? fail(LDMSEG_E_OOM, "workspace plan exceeded (stale plan): nothing was launched") : 0; }

  Act new_act(int C, int H, int W, bool persist) {
    Act a;
    a.C = C; a.H = H; a.W = W;
    const size_t bytes = (size_t)B * H * W * C * esize(dt);
    a.p = persist ? ws->persist(bytes) : ws->scratch(bytes);
    return a;
  }

  int igemm(IgemmParams& p) {
    p.cf_ctr = (unsigned long long*)cf_sync;
    if (p.splits == 1) {
      const int sp = igemm_plan_splits(p, dt);
      if (sp > 1) {
        p.splits = sp;
        p.partial = (float*)ws->scratch((size_t)sp * p.M * p.N * sizeof(float));
      }
    }
    // GEGLU: n_valid counts the OUTPUT channels, the GEMM computes value and gate columns for each
    const double ncols = (p.epi == EPI_GEGLU) ? 2.0 * p.n_valid : (double)p.n_valid;
    const int ktot = p.taps * (p.C0 + p.C1) + p.C2 + p.C3;
    const double flops = 2.0 * p.M * ncols * ktot;
    const double bytes = ((double)p.M * (p.C0 + p.C1 + p.C2 + p.C3) + (double)p.N * ktot + (double)p.M * p.n_valid) * esize(dt);
    std::string label;
    if (g_prof.on && !dry())
      label = "M=" + std::to_string(p.M) + " N=" + std::to_string((int)ncols) + " K=" + std::to_string(ktot) + (p.C2 ? "(+1x1 " + std::to_string(p.C2 + p.C3) + ")" : "") +
              " taps=" + std::to_string(p.taps) + " stride=" + std::to_string(p.stride) + " up=" + std::to_string(p.up) +
              " epi=" + std::to_string(p.epi) + " splits=" + std::to_string(p.splits);
    ProfScope ps(0, s, flops, bytes, dry(), label);
    if (dry()) return 0;
    TRY(ws_ok());
    p.x3 = (dt == DT_F32) ? ((p.w_dynamic && x3) ? 1 : x3) : 0;      // (W = an activation tensor: split in the K loop)
    return launch_igemm(p, dt, s);
  }

  // conv3x3 / 1x1 over NHWC `x` (optionally channel-concatenated with `x2`)
  int conv(const ConvW& w, const Act& x, const Act* x2, Act* out, int stride, int up, bool persist,
           const float* rowbias, int rb_stride, const Act* resid, int silu = 0, int pad = -1) {
    const int ctot = x.C + (x2 ? x2->C : 0);
    if (ctot != w.cin_pad) return fail(LDMSEG_E_SHAPE, "conv: channel mismatch");
    if (up && w.w_up4 && w.taps == 9 && stride == 1 && !x2 && !resid && !rowbias && !silu && pad < 0 &&
        igemm_up4_ok(B, x.H, x.W, x.C, w.N, dt)) {
      // conv3x3(nearest_x2(x)) as four 2x2 phase convs on the low-resolution map: 4 B H W virtual rows, K = 4 C
      *out = new_act(w.n_valid, 2 * x.H, 2 * x.W, persist);
      IgemmParams p;
      p.src0 = x.p; p.C0 = x.C;
      p.B = B; p.Hi = p.Ho = x.H; p.Wi = p.Wo = x.W;
      p.taps = 4; p.stride = 1; p.up4 = 1;
      p.M = 4 * B * x.H * x.W; p.N = w.N; p.n_valid = w.n_valid;
      p.W = w.w_up4; p.bias = w.bias;
      p.out = out->p; p.ldo = w.n_valid;
      p.epi = EPI_STORE;
      return igemm(p);
    }
    const int Hl = up ? 2 * x.H : x.H, Wl = up ? 2 * x.W : x.W;
    // stride 2: pad 1 both sides (UNet downsample_padding=1) or pad 0 + one zero row/column at the bottom/right
    const int Ho = (w.taps == 9 && stride == 2) ? (pad == 0 ? Hl / 2 : (Hl - 1) / 2 + 1) : Hl;
    const int Wo = (w.taps == 9 && stride == 2) ? (pad == 0 ? Wl / 2 : (Wl - 1) / 2 + 1) : Wl;
    *out = new_act(w.n_valid, Ho, Wo, persist);
    IgemmParams p;
    p.src0 = x.p; p.C0 = x.C;
    if (x2) { p.src1 = x2->p; p.C1 = x2->C; }
    p.B = B; p.Hi = x.H; p.Wi = x.W; p.Ho = Ho; p.Wo = Wo;
    p.taps = w.taps; p.stride = stride; p.up = up; p.pad = pad;
    p.M = B * Ho * Wo; p.N = w.N; p.n_valid = w.n_valid;
    p.W = w.w; p.bias = w.bias;
    if (w.w_cm && pad < 0 && igemm_conv_cm(x.H * x.W, ctot, w.N, 3, stride, up, dt)) { p.W = w.w_cm; p.cm = 1; }
    p.rowbias = rowbias; p.rb_stride = rb_stride;
    if (resid) { p.resid = resid->p; p.ldr = resid->C; }
    p.out = out->p; p.ldo = w.n_valid;
    p.epi = EPI_STORE; p.silu = silu;
    return igemm(p);
  }

  // resnet tail: out = conv2(h) + conv_shortcut(cat([x, x2])) as one launch (ResnetW::conv2x); false = the launch has no
  // extra-tap form here (fp32, debug key 19 off, a forced tile policy ...): the caller runs the two convs
  bool conv_xt_ok(const ConvW& w, const Act& h, const Act& x, const Act* x2) const {
    if (!w.w || x.C + (x2 ? x2->C : 0) != w.xt_cin) return false;      // (a mismatch takes the two-conv path, which reports it)
    IgemmParams p;
    p.C0 = h.C; p.taps = 9; p.N = w.N; p.src2 = x.p ? x.p : (const void*)1; p.C2 = x.C;
    if (x2) { p.src3 = x2->p ? x2->p : (const void*)1; p.C3 = x2->C; }
    return igemm_xt_ok(p, dt);
  }
  int conv_xt(const ConvW& w, const Act& h, const Act& x, const Act* x2, Act* out) {
    if (h.C != w.cin_pad) return fail(LDMSEG_E_SHAPE, "conv_xt: channel mismatch");
    // the weight rows are [9 cout | xt_cin] long: a shortcut input of another width would walk off them
    if (x.C + (x2 ? x2->C : 0) != w.xt_cin) return fail(LDMSEG_E_SHAPE, "conv_xt: shortcut channel mismatch");
    *out = new_act(w.n_valid, h.H, h.W, true);
    IgemmParams p;
    p.src0 = h.p; p.C0 = h.C;
    p.src2 = x.p; p.C2 = x.C;
    if (x2) { p.src3 = x2->p; p.C3 = x2->C; }
    p.B = B; p.Hi = p.Ho = h.H; p.Wi = p.Wo = h.W;
    p.taps = 9; p.stride = 1; p.up = 0; p.pad = -1;
    p.M = B * h.H * h.W; p.N = w.N; p.n_valid = w.n_valid;
    p.W = w.w; p.bias = w.bias;
    p.out = out->p; p.ldo = w.n_valid;
    p.epi = EPI_STORE;
    return igemm(p);
  }

  // 3x3 conv (stride 1, time-embedding row) whose only consumer is a GroupNorm (resnet conv1 -> norm2): where the conv runs
  // as K slices and the map is small, the slices' finish and the norm are one launch and the conv's own output is never
  // stored (launch_finish_groupnorm); otherwise conv and norm as usual.
  int conv_groupnorm(const ConvW& w, const Act& x, const float* rowbias, int rb_stride, const NormW& n, float eps, int silu,
                     Act* out) {
    if (x.C != w.cin_pad || n.C != w.n_valid) return fail(LDMSEG_E_SHAPE, "conv_groupnorm: channel mismatch");
    IgemmParams p;
    p.src0 = x.p; p.C0 = x.C;
    p.B = B; p.Hi = x.H; p.Wi = x.W; p.Ho = x.H; p.Wo = x.W;
    p.taps = w.taps; p.stride = 1; p.up = 0; p.pad = -1;
    p.M = B * x.H * x.W; p.N = w.N; p.n_valid = w.n_valid;
    p.W = w.w; p.bias = w.bias;
    p.rowbias = rowbias; p.rb_stride = rb_stride;
    p.epi = EPI_STORE;
    const int sp = igemm_plan_splits(p, dt);
    // round 6 (debug key 23 bit 2, default on): where the conv finishes its K slices inside the launch - bf16, 256-row tiles, i.e. the
    // 16x16 maps at B = 8 - run conv + GroupNorm (gn_group_kernel: 5 us) instead of slabs + the fused finish-GroupNorm launch (16 us):
    // measured -12 us per forward in one process (7.963 -> 7.951 ms); on the 128-row tiles of the 8x8 maps the fused launch stays
    const bool conv_then_gn = sp >= 2 && dt == DT_BF16 && (igemm_get_cf_mode() & 5) == 5 && p.M >= 2048 && p.M % 256 == 0;
    if (sp < 2 || conv_then_gn || !finish_groupnorm_ok(B, x.H * x.W, w.n_valid, dt)) {
      Act h;
      TRY(conv(w, x, nullptr, &h, 1, 0, false, rowbias, rb_stride, nullptr));
      return groupnorm(n, h, nullptr, eps, silu, out);
    }
    *out = new_act(w.n_valid, x.H, x.W, false);
    if (w.w_cm && igemm_conv_cm(x.H * x.W, x.C, w.N, 3, 1, 0, dt)) { p.W = w.w_cm; p.cm = 1; }
    p.splits = sp;
    p.partial = (float*)ws->scratch((size_t)sp * p.M * p.N * sizeof(float));
    p.no_finish = 1;
    TRY(igemm(p));
    GNParams g;
    g.src0 = nullptr; g.C0 = w.n_valid; g.B = B; g.HW = x.H * x.W; g.groups = 32;
    g.gamma = n.g; g.beta = n.b; g.eps = eps; g.silu = silu;
    g.out = out->p;
    const double bytes = ((double)sp * p.M * p.N * sizeof(float)) + (double)p.M * w.n_valid * esize(dt);
    ProfScope ps(2, s, 0, bytes, dry(), "finish+GN HW=" + std::to_string(g.HW) + " C=" + std::to_string(g.C0));
    if (dry()) return 0;
    TRY(ws_ok());
    return launch_finish_groupnorm(p, g, dt, s);
  }

  int groupnorm(const NormW& n, const Act& x, const Act* x2, float eps, int silu, Act* out) {
    const int ctot = x.C + (x2 ? x2->C : 0);
    if (ctot != n.C) return fail(LDMSEG_E_SHAPE, "groupnorm: channel mismatch");
    *out = new_act(ctot, x.H, x.W, false);
    GNParams g;
    g.src0 = x.p; g.C0 = x.C;
    if (x2) { g.src1 = x2->p; g.C1 = x2->C; }
    g.B = B; g.HW = x.H * x.W; g.groups = 32;
    g.gamma = n.g; g.beta = n.b; g.eps = eps; g.silu = silu;
    g.out = out->p;
    g.nchunk = gn_nchunk(B, g.HW);
    g.sync_region = gn_sync;
    g.poll_us = gn_poll_us;
    g.partial = (float*)ws->scratch((size_t)B * g.nchunk * 32 * 2 * sizeof(float));
    const double bytes = 3.0 * B * g.HW * ctot * esize(dt);
    ProfScope ps(2, s, 0, bytes, dry(), "HW=" + std::to_string(g.HW) + " C=" + std::to_string(ctot));
    if (dry()) return 0;
    TRY(ws_ok());
    return launch_groupnorm(g, dt, s);
  }

  // GroupNorm statistics only ({mean, M2} per image, pixel chunk and group): the consumer (tproj.hip) combines and applies them
  int groupnorm_stats(const NormW& n, const Act& x, float eps, GnFold* gf) {
    if (x.C != n.C) return fail(LDMSEG_E_SHAPE, "groupnorm: channel mismatch");
    GNParams g;
    g.src0 = x.p; g.C0 = x.C;
    g.B = B; g.HW = x.H * x.W; g.groups = 32;
    g.nchunk = g_gnfold_mode > 1 ? (g_gnfold_mode > 64 ? 64 : g_gnfold_mode) : proj_qkv_gn_chunks(B, g.HW);
    g.partial = (float*)ws->scratch((size_t)B * g.nchunk * 32 * 2 * sizeof(float));
    gf->partial = g.partial; gf->gamma = n.g; gf->beta = n.b; gf->nchunk = g.nchunk; gf->HW = g.HW; gf->eps = eps;
    const double bytes = 1.0 * B * g.HW * x.C * esize(dt);
    ProfScope ps(2, s, 0, bytes, dry(), "stats HW=" + std::to_string(g.HW) + " C=" + std::to_string(x.C));
    if (dry()) return 0;
    TRY(ws_ok());
    return launch_groupnorm_stats(g, dt, s);
  }

  // LayerNorm statistics only (the normalisation itself is folded into the consuming GEMM)
  int rowstats(const Act& x, float eps, float* stats) {
    const int M = B * x.H * x.W;
    ProfScope ps(3, s, 0, 1.0 * M * x.C * esize(dt), dry());
    if (dry()) return 0;
    TRY(ws_ok());
    return launch_rowstats(x.p, stats, M, x.C, eps, dt, s);
  }

  int layernorm(const NormW& n, const Act& x, float eps, int silu, Act* out, bool inplace = false) {
    *out = inplace ? x : new_act(x.C, x.H, x.W, false);
    const int M = B * x.H * x.W;
    ProfScope ps(3, s, 0, 2.0 * M * x.C * esize(dt), dry());
    if (dry()) return 0;
    TRY(ws_ok());
    return launch_layernorm(x.p, out->p, n.g, n.b, M, x.C, eps, silu, dt, s);
  }
};

// ------------------------------------------------------------------ UNet
constexpr int kBlockOut[4] = {320, 640, 1280, 1280};
constexpr int kTimeDim = 1280;

struct ResnetW {
  NormW norm1, norm2;
  ConvW conv1, conv2, shortcut;
  ConvW conv2x;           // bf16, has_shortcut: conv2 and conv_shortcut as ONE matrix [N][9 cout | cin], bias = b2 + bs (IgemmParams::src2)
  bool has_shortcut = false;
  int cin = 0, cout = 0, temb_off = 0;
};
struct TransformerW {
  int C = 0;
  NormW norm, ln1, ln3;
  ConvW proj_in, qkv, attn_out, ff1, ff2, proj_out;
  ConvW ffp;                    // bf16, levels without the row-local fused kernel: ff.net.2 and proj_out as ONE Linear over [g | h] (round 5)
  void* mlp_stream = nullptr;   // 320-channel level, bf16: the feed-forward's weights as one consumption-ordered stream (tfuse.hip)
  void* in_stream = nullptr;    // the same for proj_in | to_q | to_k | to_v (tproj.hip)
  float* in_bias = nullptr;     // [4][C]: proj_in bias | W_q beta | W_k beta | W_v beta
};

}  // namespace

struct ldmseg_unet {
  ldmseg_unet_cfg cfg{};
  int dt = DT_BF16;
  bool x3 = false;      // fp32 storage with split-bf16 GEMM arithmetic (LDMSEG_BF16X3)
  DeviceArena arena;
  Workspace ws;
  void* ws_mem = nullptr;
  size_t ws_cap = 0;
  int64_t nparams = 0;

  float *te1_w = nullptr, *te1_b = nullptr, *te2_w = nullptr, *te2_b = nullptr;
  float *tproj_w = nullptr, *tproj_b = nullptr;
  int temb_total = 0;
  ConvW conv_in, conv_out;
  NormW norm_out;
  ResnetW down_res[4][2], mid_res[2], up_res[4][3];
  TransformerW down_attn[3][2], mid_attn, up_attn[4][3];  // up_attn[0] unused
  ConvW down_conv[3], up_conv[3];
  // sampler state
  int plan_B = 0, plan_L = 0, plan_epoch = -1;
  size_t plan_persist = 0, plan_scratch = 0;
  int attn_fp8_min_tokens = 0;   // > 0: bf16 mode runs attention levels with N >= this many tokens on the fp8 operand path
  float* cond = nullptr;   // [B,4,L,L] self-conditioning channel
  float* eps = nullptr;
  size_t loop_elems = 0;
  // ldmseg_sample_loop: the time-embedding rows of ALL steps (they depend on the timestep only) from one pass over the
  // 100 MB of time_emb_proj weights instead of one pass per step
  void* temb_buf = nullptr;            // [steps] int64 timesteps | sinusoid | two MLP activations | [steps][temb_total] rows
  int temb_cap = 0;                    // steps the buffer holds
  const float* temb_override = nullptr;   // non-null while the loop runs a forward: this step's row (broadcast over the batch)
  void* gn_sync = nullptr;
  void* cf_sync = nullptr;
  // fused step tail (tail.hip): the sampling loop parks the scheduler step here before a forward; the forward's conv_out
  // launch carries it out (tail_done) and leaves the next forward's packed input in place (xin_ready)
  const StepTail* tail_req = nullptr;
  bool tail_done = false, xin_ready = false;
  // Cooperative GroupNorm back-off: the fallback counter of the handle's hand-off region is copied to this
  // pinned word at the end of every sampling call; a call that finds it has grown since the last look runs its norms with a
  // 2 us poll bound (partners that are not co-resident - another process or stream holding CUs - then cost the extra reads of
  // the two-launch scheme instead of a 100 us spin per norm) for the next few calls, then tries the full bound again.
  unsigned long long* gn_diag_host = nullptr;
  unsigned long long gn_diag_seen = 0;
  int gn_backoff_calls = 0;
  const void* xin_ptr = nullptr;       // where the tail wrote that input: the next forward skips its pack only if its own xin IS this buffer

  ~ldmseg_unet() {
    arena.release();
    if (ws_mem) (void)hipFree(ws_mem);
    if (cond) (void)hipFree(cond);
    if (eps) (void)hipFree(eps);
    if (temb_buf) (void)hipFree(temb_buf);
    if (gn_diag_host) (void)hipHostFree(gn_diag_host);
    if (gn_sync) (void)hipFree(gn_sync);
    if (cf_sync) (void)hipFree(cf_sync);
  }
};

namespace {

int build_resnet(Builder& b, const std::string& p, int cin, int cout, int* temb_off, ResnetW* r) {
  r->cin = cin; r->cout = cout;
  TRY(b.norm(p + "norm1", cin, &r->norm1));
  // (the 320- / 640-channel levels can run on maps of 64x64 and more: they also hold the channel-major packing, see igemm_conv_cm)
  TRY(b.conv(p + "conv1", cout, cin, 3, cin, &r->conv1, true, 0, EPI_STORE, cout <= 640));
  TRY(b.norm(p + "norm2", cout, &r->norm2));
  TRY(b.conv(p + "conv2", cout, cout, 3, cout, &r->conv2, true, 0, EPI_STORE, cout <= 640));
  r->has_shortcut = cin != cout;
  if (r->has_shortcut) TRY(b.conv(p + "conv_shortcut", cout, cin, 1, cin, &r->shortcut));
  // conv2(h) + conv_shortcut(x) is one accumulation over K = 9 cout + cin: a second packing with the shortcut's columns behind
  // conv2's, run as one launch with an extra centre tap (igemm.hip, XT) - the shortcut's own launch, its output tensor and the
  // residual read of conv2's epilogue disappear (14 resnets of the UNet)
  if (r->has_shortcut && b.dt == DT_BF16 && r->conv2.N == r->shortcut.N && r->conv2.N % 160 == 0 && cin % bke(b.dt) == 0 && cout % bke(b.dt) == 0) {
    ConvW& x = r->conv2x;
    x.N = r->conv2.N; x.n_valid = r->conv2.n_valid; x.cin_pad = cout; x.taps = 9; x.cout = cout; x.xt_cin = cin;
    TRY(b.arena->alloc(&x.w, (size_t)x.N * (9 * cout + cin) * esize(b.dt)));
    TRY(launch_concat_rows(r->conv2.w, 9 * cout, r->shortcut.w, cin, x.w, x.N, b.dt, b.s));
    void* pb;
    TRY(b.arena->alloc(&pb, x.N * sizeof(float)));
    x.bias = (float*)pb;
    TRY(launch_vec_add(r->conv2.bias, r->shortcut.bias, x.bias, x.N, b.s));
  }
  r->temb_off = *temb_off;
  *temb_off += cout;
  return 0;
}

int build_transformer(Builder& b, const std::string& p, int C, TransformerW* t) {
  t->C = C;
  const int dt = b.dt;
  TRY(b.norm(p + "norm", C, &t->norm));
  TRY(b.conv(p + "proj_in", C, C, 1, C, &t->proj_in));
  const std::string tb = p + "transformer_blocks.0.";
  TRY(b.norm(tb + "norm1", C, &t->ln1));
  // fused q|k|v projection, no bias, with norm1 (LayerNorm) folded in: W' = gamma (.) W, c1 = rowsum(W'), bias = W beta
  {
    ConvW& q = t->qkv;
    q.N = 3 * C; q.n_valid = 3 * C; q.cin_pad = C; q.taps = 1; q.cout = 3 * C;
    TRY(b.arena->alloc(&q.w, (size_t)3 * C * C * esize(dt)));
    b.weights(q.w, (size_t)3 * C * C);
    std::vector<int> ident(C);
    for (int r = 0; r < C; ++r) ident[r] = r;
    int* dident;
    TRY(b.upload_ints(ident, &dident));
    void* tmp;
    HIP_TRY(hipMalloc(&tmp, (size_t)3 * C * C * sizeof(float)));
    b.temps.push_back(tmp);
    const char* names[3] = {"attn1.to_q.weight", "attn1.to_k.weight", "attn1.to_v.weight"};
    for (int i = 0; i < 3; ++i) {
      const float* w;
      TRY(b.wm->get(tb + names[i], (int64_t)C * C, &w));
      TRY(launch_repack_rows_scaled(w, (char*)q.w + (size_t)i * C * C * esize(dt), dident, C, C, t->ln1.g, dt, b.s));
      TRY(launch_repack_rows_scaled(w, (char*)tmp + (size_t)i * C * C * sizeof(float), dident, C, C, t->ln1.b, DT_F32, b.s));
      b.nparams += (int64_t)C * C;
    }
    void *pc1, *pc2;
    TRY(b.arena->alloc(&pc1, (size_t)3 * C * sizeof(float)));
    TRY(b.arena->alloc(&pc2, (size_t)3 * C * sizeof(float)));
    TRY(launch_rowsum(q.w, nullptr, (float*)pc1, 3 * C, C, dt, b.s));
    TRY(launch_rowsum(tmp, nullptr, (float*)pc2, 3 * C, C, DT_F32, b.s));
    q.c1 = (float*)pc1;
    q.bias = (float*)pc2;
  }
  TRY(b.conv(tb + "attn1.to_out.0", C, C, 1, C, &t->attn_out));
  TRY(b.norm(tb + "norm3", C, &t->ln3));
  // GEGLU projection [8C, C]: rows interleaved in 16-row (a | gate) pairs so one lane owns both halves
  {
    ConvW& f = t->ff1;
    const int N = 8 * C;
    f.N = N; f.n_valid = 4 * C; f.cin_pad = C; f.taps = 1; f.cout = 4 * C;
    std::vector<int> map(N);
    for (int r = 0; r < N; ++r) {
      const int blk = r / 32, w = r % 32;
      map[r] = (w < 16) ? blk * 16 + w : 4 * C + blk * 16 + (w - 16);
    }
    int* dmap;
    TRY(b.upload_ints(map, &dmap));
    const float *w, *bias;
    TRY(b.wm->get(tb + "ff.net.0.proj.weight", (int64_t)N * C, &w));
    TRY(b.wm->get(tb + "ff.net.0.proj.bias", N, &bias));
    // norm3 (LayerNorm) folded in: W' = gamma (.) W in the packed row order, c1 = rowsum(W'), bias = W beta + b
    TRY(b.arena->alloc(&f.w, (size_t)N * C * esize(dt)));
    b.weights(f.w, (size_t)N * C);
    TRY(launch_repack_rows_scaled(w, f.w, dmap, N, C, t->ln3.g, dt, b.s));
    void* tmp;
    HIP_TRY(hipMalloc(&tmp, (size_t)N * C * sizeof(float)));
    b.temps.push_back(tmp);
    TRY(launch_repack_rows_scaled(w, tmp, dmap, N, C, t->ln3.b, DT_F32, b.s));
    void *pb, *pc1, *pc2;
    TRY(b.arena->alloc(&pb, N * sizeof(float)));
    TRY(launch_repack_rows(bias, pb, dmap, N, 1, DT_F32, b.s));
    TRY(b.arena->alloc(&pc1, N * sizeof(float)));
    TRY(b.arena->alloc(&pc2, N * sizeof(float)));
    TRY(launch_rowsum(f.w, nullptr, (float*)pc1, N, C, dt, b.s));
    TRY(launch_rowsum(tmp, (const float*)pb, (float*)pc2, N, C, DT_F32, b.s));
    f.c1 = (float*)pc1;
    f.bias = (float*)pc2;
    b.nparams += (int64_t)N * C + N;
  }
  TRY(b.conv(tb + "ff.net.2", C, 4 * C, 1, 4 * C, &t->ff2));
  TRY(b.conv(p + "proj_out", C, C, 1, C, &t->proj_out));
  if (const size_t sb = (dt == DT_BF16 && t->ff1.N == 8 * C && t->ff2.N == C && t->proj_out.N == C) ? mlp_fused_stream_bytes(C) : 0) {
    TRY(b.arena->alloc(&t->mlp_stream, sb));
    TRY(launch_pack_mlp_stream(t->ff1.w, t->ff2.w, t->proj_out.w, t->mlp_stream, C, b.s));
  }
  // proj_out(h + ff.net.2(g)) + x has no nonlinearity between the two Linears (the transformer block ends with the feed-forward's
  // residual, Transformer2DModel.proj_out follows): = [Wp W2 | Wp] [g | h] + (bp + Wp b2) + x.  Same FLOPs (2 M C 5C), one launch
  // and one [M, C] round trip fewer: the levels that do not run the row-local fused kernel hold the chained matrix.
  if (dt == DT_BF16 && !t->mlp_stream && C % 160 == 0 && t->ff2.N == C && t->proj_out.N == C) {
    const float *w2, *b2, *wp, *bp;
    TRY(b.wm->get(tb + "ff.net.2.weight", (int64_t)C * 4 * C, &w2));
    TRY(b.wm->get(tb + "ff.net.2.bias", C, &b2));
    TRY(b.wm->get(p + "proj_out.weight", (int64_t)C * C, &wp));
    TRY(b.wm->get(p + "proj_out.bias", C, &bp));
    void* wcat;
    HIP_TRY(hipMalloc(&wcat, (size_t)C * 5 * C * sizeof(float)));
    b.temps.push_back(wcat);
    void* bcat;
    TRY(b.arena->alloc(&bcat, C * sizeof(float)));
    TRY(launch_chain_weights(wp, w2, b2, bp, (float*)wcat, (float*)bcat, C, b.s));
    std::vector<int> ident(C);
    for (int r = 0; r < C; ++r) ident[r] = r;
    int* dident;
    TRY(b.upload_ints(ident, &dident));
    ConvW& f = t->ffp;
    f.N = C; f.n_valid = C; f.cin_pad = 5 * C; f.taps = 1; f.cout = C;
    TRY(b.arena->alloc(&f.w, (size_t)C * 5 * C * esize(dt)));
    TRY(launch_repack_rows((const float*)wcat, f.w, dident, C, 5 * C, dt, b.s));
    f.bias = (float*)bcat;
  }
  if (const size_t sb = (dt == DT_BF16 && t->proj_in.N == C && t->proj_in.cin_pad == C && t->qkv.N == 3 * C) ? proj_qkv_stream_bytes(C) : 0) {
    TRY(b.arena->alloc(&t->in_stream, sb));
    TRY(launch_pack_proj_qkv_stream(t->proj_in.w, t->qkv.w, t->in_stream, C, b.s));
    void* pb;
    TRY(b.arena->alloc(&pb, (size_t)4 * C * sizeof(float)));
    t->in_bias = (float*)pb;
    HIP_TRY(hipMemcpyAsync(t->in_bias, t->proj_in.bias, (size_t)C * sizeof(float), hipMemcpyDeviceToDevice, b.s));
    HIP_TRY(hipMemcpyAsync(t->in_bias + C, t->qkv.bias, (size_t)3 * C * sizeof(float), hipMemcpyDeviceToDevice, b.s));
  }
  return 0;
}

int unet_build(ldmseg_unet* u, const WeightMap& wm) {
  hipStream_t s = nullptr;
  Builder b{&u->arena, &wm, u->dt, s};
  b.x3 = u->x3;
  const int cp = bke(u->dt);
  if (u->cfg.in_channels > cp) return fail(LDMSEG_E_ARG, "in_channels too large");
  TRY(b.conv("conv_in", kBlockOut[0], u->cfg.in_channels, 3, cp, &u->conv_in));
  TRY(b.f32_copy("time_embedding.linear_1.weight", (int64_t)kTimeDim * 320, &u->te1_w));
  TRY(b.f32_copy("time_embedding.linear_1.bias", kTimeDim, &u->te1_b));
  TRY(b.f32_copy("time_embedding.linear_2.weight", (int64_t)kTimeDim * kTimeDim, &u->te2_w));
  TRY(b.f32_copy("time_embedding.linear_2.bias", kTimeDim, &u->te2_b));

  int temb_off = 0;
  std::vector<int> skip_ch{kBlockOut[0]};
  int c = kBlockOut[0];
  for (int i = 0; i < 4; ++i) {
    const int co = kBlockOut[i];
    for (int j = 0; j < 2; ++j) {
      const std::string p = "down_blocks." + std::to_string(i);
      TRY(build_resnet(b, p + ".resnets." + std::to_string(j) + ".", c, co, &temb_off, &u->down_res[i][j]));
      c = co;
      if (i < 3) TRY(build_transformer(b, p + ".attentions." + std::to_string(j) + ".", c, &u->down_attn[i][j]));
      skip_ch.push_back(c);
    }
    if (i < 3) {
      TRY(b.conv("down_blocks." + std::to_string(i) + ".downsamplers.0.conv", c, c, 3, c, &u->down_conv[i]));
      skip_ch.push_back(c);
    }
  }
  TRY(build_resnet(b, "mid_block.resnets.0.", c, c, &temb_off, &u->mid_res[0]));
  TRY(build_transformer(b, "mid_block.attentions.0.", c, &u->mid_attn));
  TRY(build_resnet(b, "mid_block.resnets.1.", c, c, &temb_off, &u->mid_res[1]));
  for (int i = 0; i < 4; ++i) {
    const int co = kBlockOut[3 - i];
    for (int j = 0; j < 3; ++j) {
      const std::string p = "up_blocks." + std::to_string(i);
      const int sk = skip_ch.back();
      skip_ch.pop_back();
      TRY(build_resnet(b, p + ".resnets." + std::to_string(j) + ".", c + sk, co, &temb_off, &u->up_res[i][j]));
      c = co;
      if (i > 0) TRY(build_transformer(b, p + ".attentions." + std::to_string(j) + ".", c, &u->up_attn[i][j]));
    }
    if (i < 3) {
      const std::string key = "up_blocks." + std::to_string(i) + ".upsamplers.0.conv";
      TRY(b.conv(key, c, c, 3, c, &u->up_conv[i]));
      // Upsample2D = nearest x2 then this conv: an output pixel sees only 2 x 2 distinct source pixels, so the conv runs as four
      // 2x2-tap phase convs on the low-resolution map with pre-summed weights (K = 4 C instead of 9 C; igemm.hip UP4)
      ConvW& cw = u->up_conv[i];
      if (u->dt == DT_BF16 && cw.N % 160 == 0 && c % bke(u->dt) == 0) {
        const float* wraw;
        TRY(b.wm->get(key + ".weight", (int64_t)c * c * 9, &wraw));
        TRY(b.arena->alloc(&cw.w_up4, (size_t)16 * cw.N * c * esize(u->dt)));
        TRY(launch_pack_up4(wraw, cw.w_up4, c, c, cw.N, c, u->dt, b.s));
      }
    }
  }
  TRY(b.norm("conv_norm_out", c, &u->norm_out));
  TRY(b.conv("conv_out", 4, c, 3, c, &u->conv_out));

  // all 22 time_emb_proj Linear(1280 -> cout) concatenated: one GEMV per forward
  u->temb_total = temb_off;
  {
    void *pw, *pb;
    TRY(u->arena.alloc(&pw, (size_t)temb_off * kTimeDim * sizeof(float)));
    TRY(u->arena.alloc(&pb, (size_t)temb_off * sizeof(float)));
    u->tproj_w = (float*)pw;
    u->tproj_b = (float*)pb;
    auto put = [&](const std::string& p, const ResnetW& r) -> int {
      const float *w, *bias;
      TRY(wm.get(p + "time_emb_proj.weight", (int64_t)r.cout * kTimeDim, &w));
      TRY(wm.get(p + "time_emb_proj.bias", r.cout, &bias));
      HIP_TRY(hipMemcpyAsync(u->tproj_w + (size_t)r.temb_off * kTimeDim, w, (size_t)r.cout * kTimeDim * sizeof(float),
                             hipMemcpyDeviceToDevice, s));
      HIP_TRY(hipMemcpyAsync(u->tproj_b + r.temb_off, bias, r.cout * sizeof(float), hipMemcpyDeviceToDevice, s));
      b.nparams += (int64_t)r.cout * kTimeDim + r.cout;
      return 0;
    };
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 2; ++j)
        TRY(put("down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j) + ".", u->down_res[i][j]));
    TRY(put("mid_block.resnets.0.", u->mid_res[0]));
    TRY(put("mid_block.resnets.1.", u->mid_res[1]));
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 3; ++j)
        TRY(put("up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j) + ".", u->up_res[i][j]));
  }
  TRY(b.finish());
  u->nparams = b.nparams;
  return 0;
}

int run_resnet(Exec& ex, const ResnetW& r, const Act& x, const Act* skip, const float* temb, int temb_stride, Act* out) {
  Workspace* ws = ex.ws;
  // output first (persist), temporaries on the scratch stack
  const size_t m = ws->mark();
  Act n1, n2, sc;
  TRY(ex.groupnorm(r.norm1, x, skip, 1e-5f, 1, &n1));
  TRY(ex.conv_groupnorm(r.conv1, n1, temb + r.temb_off, temb_stride, r.norm2, 1e-5f, 1, &n2));
  const Act* resid = &x;
  if (r.has_shortcut && ex.conv_xt_ok(r.conv2x, n2, x, skip)) {
    TRY(ex.conv_xt(r.conv2x, n2, x, skip, out));
    ws->reset(m);
    return 0;
  }
  if (r.has_shortcut) {
    TRY(ex.conv(r.shortcut, x, skip, &sc, 1, 0, false, nullptr, 0, nullptr));
    resid = &sc;
  } else if (skip) {
    return fail(LDMSEG_E_SHAPE, "resnet without shortcut cannot take a concat input");
  }
  TRY(ex.conv(r.conv2, n2, nullptr, out, 1, 0, true, nullptr, 0, resid));
  ws->reset(m);
  return 0;
}

int run_transformer(Exec& ex, const TransformerW& t, const Act& x, Act* out) {
  Workspace* ws = ex.ws;
  const size_t m = ws->mark();
  const int C = t.C, N = x.H * x.W, M = ex.B * N;
  Act n, h, qkv, att, ff;
  const bool entry_fused = t.in_stream && proj_qkv_fused_ok(C, M, ex.dt);
  const bool gn_fold = entry_fused && g_gnfold_mode && proj_qkv_gn_fold_ok(N);
  GnFold gf;
  if (gn_fold) TRY(ex.groupnorm_stats(t.norm, x, 1e-6f, &gf));     // the norm's apply pass runs inside the fused entry, on its LDS tile
  else TRY(ex.groupnorm(t.norm, x, nullptr, 1e-6f, 0, &n));
  if (entry_fused) {
    // 320-channel level: proj_in -> norm1 -> q|k|v in one row-local launch (tproj.hip): h is written once and not read back
    h = ex.new_act(C, x.H, x.W, false);
    qkv = ex.new_act(3 * C, x.H, x.W, false);
    const double flops = 2.0 * M * 4.0 * C * C;
    const double bytes = (double)M * C * esize(ex.dt) * 5.0 + (double)proj_qkv_stream_bytes(C);
    ProfScope ps(0, ex.s, flops, bytes, ex.dry(), "M=" + std::to_string(M) + " proj_ln_qkv C=" + std::to_string(C));
    if (!ex.dry()) {
      TRY(ex.ws_ok());
      const int r = launch_proj_qkv_fused(gn_fold ? x.p : n.p, h.p, qkv.p, t.in_stream, t.in_bias, igemm_zero_page(), M, C, 1e-5f,
                                          gn_fold ? &gf : nullptr, ex.s);
      if (r) return fail(r == -2 ? LDMSEG_E_SHAPE : LDMSEG_E_HIP, "launch_proj_qkv_fused failed");
    }
  } else {
    TRY(ex.conv(t.proj_in, n, nullptr, &h, 1, 0, false, nullptr, 0, nullptr));
    // norm1 is folded into the q|k|v GEMM: one statistics pass over h (mean, rstd per token), the GEMM reads h itself
    float* stats = (float*)ws->scratch((size_t)M * 2 * sizeof(float));
    TRY(ex.rowstats(h, 1e-5f, stats));
    qkv = ex.new_act(3 * C, x.H, x.W, false);
    IgemmParams p;
    p.src0 = h.p; p.C0 = C; p.B = ex.B; p.Hi = p.Ho = x.H; p.Wi = p.Wo = x.W;
    p.M = M; p.N = t.qkv.N; p.n_valid = 3 * C; p.W = t.qkv.w; p.bias = t.qkv.bias;
    p.rowstats = stats; p.c1 = t.qkv.c1;
    p.out = qkv.p; p.ldo = 3 * C;
    TRY(ex.igemm(p));
  }
  att = ex.new_act(C, x.H, x.W, false);
  {
    const double d = C / 8.0;
    ProfScope ps(1, ex.s, 4.0 * ex.B * 8 * (double)N * N * d, 4.0 * M * C * esize(ex.dt), ex.dry(),
                 "N=" + std::to_string(N) + " C=" + std::to_string(C));
    // (only where the fp8 path is the faster one: head dim 40 on whole 128-key tiles = the block-scaled 2x-rate MFMAs.  The
    // unscaled fp8 kernel that would serve head dim 80 / ragged lengths runs at the bf16 MFMA rate and measured SLOWER than the
    // bf16 kernel - 222 vs 205 us at d = 80, N = 4096 - so a handle asked for fp8 from 4096 tokens up keeps that level in bf16)
    const size_t kv8 = (ex.dt == DT_BF16 && ex.attn_fp8_min_tokens > 0 && N >= ex.attn_fp8_min_tokens && attention_mx_ok(N, C, 8))
                           ? attention_fp8_scratch_bytes(ex.B, N, C, 8) : 0;
    if (kv8) {                                     // long-context level on the fp8 operand path (BASELINE configs[4])
      void* scratch = ws->scratch(kv8);
      if (!ex.dry()) TRY(ex.ws_ok());
      if (!ex.dry()) TRY(launch_attention_fp8(qkv.p, scratch, att.p, ex.B, N, C, 8, ex.s));
    } else if (!ex.dry()) {
      TRY(ex.ws_ok());
      TRY(launch_attention(qkv.p, att.p, ex.B, N, C, 8, ex.x3 ? 2 : ex.dt, ex.s));   // 2 = fp32 tensors, split-bf16 products
    }
  }
  // h = to_out(att) + h  (in place on h)
  {
    IgemmParams p;
    p.src0 = att.p; p.C0 = C; p.B = ex.B; p.Hi = p.Ho = x.H; p.Wi = p.Wo = x.W;
    p.M = M; p.N = t.attn_out.N; p.n_valid = C; p.W = t.attn_out.w; p.bias = t.attn_out.bias;
    p.resid = h.p; p.ldr = C; p.out = h.p; p.ldo = C;
    TRY(ex.igemm(p));
  }
  if (t.mlp_stream && mlp_fused_ok(C, ex.dt)) {
    // 320-channel level: norm3 -> GEGLU -> ff.net.2 (+h) [-> proj_out (+x)] in one row-local launch (tfuse.hip); the [M, 4C]
    // hidden tensor and the statistics pass never exist
    const bool proj = mlp_fused_proj();
    if (proj) *out = ex.new_act(C, x.H, x.W, true);
    {
      const double flops = 2.0 * M * (8.0 * C * C + 4.0 * C * C + (proj ? 1.0 * C * C : 0.0));
      const double bytes = (double)M * C * esize(ex.dt) * (proj ? 3.0 : 2.0) + (double)mlp_fused_stream_bytes(C);
      ProfScope ps(0, ex.s, flops, bytes, ex.dry(), "M=" + std::to_string(M) + " mlp_fused C=" + std::to_string(C) + " proj=" + std::to_string((int)proj));
      if (!ex.dry()) {
        TRY(ex.ws_ok());
        const int r = launch_mlp_fused(h.p, proj ? out->p : h.p, x.p, t.mlp_stream, t.ff1.bias, t.ff2.bias, t.proj_out.bias,
                                       igemm_zero_page(), M, C, 1e-5f, proj ? 1 : 0, N, ex.s);
        if (r) return fail(r == -2 ? LDMSEG_E_SHAPE : LDMSEG_E_HIP, "launch_mlp_fused failed");
      }
    }
    if (!proj) TRY(ex.conv(t.proj_out, h, nullptr, out, 1, 0, true, nullptr, 0, &x));
    ws->reset(m);
    return 0;
  }
  float* stats = (float*)ws->scratch((size_t)M * 2 * sizeof(float));
  TRY(ex.rowstats(h, 1e-5f, stats));          // norm3, folded into the GEGLU GEMM the same way
  ff = ex.new_act(4 * C, x.H, x.W, false);
  {
    IgemmParams p;
    p.src0 = h.p; p.C0 = C; p.B = ex.B; p.Hi = p.Ho = x.H; p.Wi = p.Wo = x.W;
    p.M = M; p.N = t.ff1.N; p.n_valid = 4 * C; p.W = t.ff1.w; p.bias = t.ff1.bias;
    p.rowstats = stats; p.c1 = t.ff1.c1;
    p.out = ff.p; p.ldo = 4 * C; p.epi = EPI_GEGLU;
    TRY(ex.igemm(p));
  }
  if (t.ffp.w && g_ffp_mode) {
    // out = proj_out(h + ff.net.2(g)) + x as one Linear over torch.cat([g, h], -1) (TransformerW::ffp)
    TRY(ex.conv(t.ffp, ff, &h, out, 1, 0, true, nullptr, 0, &x));
    ws->reset(m);
    return 0;
  }
  {
    IgemmParams p;
    p.src0 = ff.p; p.C0 = 4 * C; p.B = ex.B; p.Hi = p.Ho = x.H; p.Wi = p.Wo = x.W;
    p.M = M; p.N = t.ff2.N; p.n_valid = C; p.W = t.ff2.w; p.bias = t.ff2.bias;
    p.resid = h.p; p.ldr = C; p.out = h.p; p.ldo = C;
    TRY(ex.igemm(p));
  }
  TRY(ex.conv(t.proj_out, h, nullptr, out, 1, 0, true, nullptr, 0, &x));
  ws->reset(m);
  return 0;
}

// dry = true only measures the workspace
int unet_forward_impl(ldmseg_unet* u, const float* a, int Ca, const float* b, int Cb, const float* c, int Cc,
                      const int64_t* t_dev, int t_count, int64_t t_host, int B, int L, float* out, hipStream_t s,
                      bool dry, size_t scratch_base) {
  if (B < 1 || L < 8 || L % 8 != 0) return fail(LDMSEG_E_SHAPE, "L must be a positive multiple of 8, B >= 1");
  if (Ca + Cb + Cc != u->cfg.in_channels) return fail(LDMSEG_E_SHAPE, "input channel count != in_channels");
  if (t_dev && t_count != 1 && t_count != B) return fail(LDMSEG_E_ARG, "t_count must be 1 or B");
  Workspace* ws = &u->ws;
  ws->begin(dry, scratch_base);
  Exec ex{ws, u->dt, B, s};
  ex.attn_fp8_min_tokens = u->attn_fp8_min_tokens;
  ex.gn_sync = u->gn_sync;
  ex.cf_sync = u->cf_sync;
  ex.x3 = u->x3 ? 2 : 0;          // (2: the handle's weights are hi | lo planes, Builder::finish)
  ex.gn_poll_us = u->gn_backoff_calls > 0 ? 2 : -1;
  const int dt = u->dt;

  // --- time embedding: sinusoid -> MLP -> every resnet's time_emb_proj(SiLU(emb)) ---
  // a scalar timestep (timestep.expand(B), unet.py:302-303) gives every image the same embedding:
  // run the MLP for one row and broadcast it (row stride 0) instead of B identical rows
  const bool per_sample_t = (t_dev != nullptr && t_count > 1);
  const int TB = per_sample_t ? B : 1;
  float* sinus = (float*)ws->persist((size_t)B * 320 * sizeof(float));
  float* e1 = (float*)ws->persist((size_t)B * kTimeDim * sizeof(float));
  float* emb = (float*)ws->persist((size_t)B * kTimeDim * sizeof(float));
  float* temb = (float*)ws->persist((size_t)B * u->temb_total * sizeof(float));
  if (u->temb_override && !dry) {
    temb = const_cast<float*>(u->temb_override);     // read-only from here on
  } else {
    ProfScope ps(4, s, 0, 0, dry);
    if (!dry) {
      TRY(ex.ws_ok());
      TRY(launch_time_embed(t_dev, t_count, t_host, TB, sinus, s));
      TRY(launch_small_linear(sinus, u->te1_w, u->te1_b, e1, TB, 320, kTimeDim, 0, 1, s));
      TRY(launch_small_linear(e1, u->te2_w, u->te2_b, emb, TB, kTimeDim, kTimeDim, 0, 0, s));
      TRY(launch_small_linear(emb, u->tproj_w, u->tproj_b, temb, TB, kTimeDim, u->temb_total, 1, 0, s));
    }
  }
  const int tstride = (per_sample_t && !(u->temb_override && !dry)) ? u->temb_total : 0;

  // --- conv_in on the channel-concatenated fp32 NCHW input ---
  Act xin = ex.new_act(bke(dt), L, L, true);
  // (a re-plan or a workspace reallocation between two steps moves xin: the pointer the tail wrote to must be this one)
  const bool xin_kept = !dry && u->xin_ready && u->xin_ptr == xin.p;
  if (!dry) { u->xin_ready = false; u->xin_ptr = nullptr; }
  if (xin_kept) {
    // the previous step's tail already wrote [latents | rgb | cond] here (tail.hip)
  } else {
    ProfScope ps(4, s, 0, 0, dry);
    if (!dry) TRY(ex.ws_ok());
    if (!dry) TRY(launch_pack_concat3(a, Ca, b, Cb, c, Cc, xin.p, B, L * L, bke(dt), dt, s));
  }
  Act h;
  TRY(ex.conv(u->conv_in, xin, nullptr, &h, 1, 0, true, nullptr, 0, nullptr));
  std::vector<Act> skips{h};

  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 2; ++j) {
      Act o;
      TRY(run_resnet(ex, u->down_res[i][j], h, nullptr, temb, tstride, &o));
      h = o;
      if (i < 3) {
        TRY(run_transformer(ex, u->down_attn[i][j], h, &o));
        h = o;
      }
      skips.push_back(h);
    }
    if (i < 3) {
      Act o;
      TRY(ex.conv(u->down_conv[i], h, nullptr, &o, 2, 0, true, nullptr, 0, nullptr));
      h = o;
      skips.push_back(h);
    }
  }
  {
    Act o;
    TRY(run_resnet(ex, u->mid_res[0], h, nullptr, temb, tstride, &o)); h = o;
    TRY(run_transformer(ex, u->mid_attn, h, &o)); h = o;
    TRY(run_resnet(ex, u->mid_res[1], h, nullptr, temb, tstride, &o)); h = o;
  }
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 3; ++j) {
      Act sk = skips.back();
      skips.pop_back();
      Act o;
      TRY(run_resnet(ex, u->up_res[i][j], h, &sk, temb, tstride, &o));   // torch.cat([hidden, skip], 1)
      h = o;
      if (i > 0) {
        TRY(run_transformer(ex, u->up_attn[i][j], h, &o));
        h = o;
      }
    }
    if (i < 3) {
      Act o;
      TRY(ex.conv(u->up_conv[i], h, nullptr, &o, 1, 1, true, nullptr, 0, nullptr));  // nearest x2 folded in
      h = o;
    }
  }
  // --- conv_norm_out -> SiLU -> conv_out, written straight to fp32 NCHW ---
  {
    const size_t m = ws->mark();
    Act n;
    TRY(ex.groupnorm(u->norm_out, h, nullptr, 1e-5f, 1, &n));
    if (conv_out_tail_ok(n.C, L, L, dt)) {
      // bf16: conv_out as a halo-resident stencil and - inside ldmseg_sample_loop - the rest of the step in its epilogue
      StepTail t;
      if (u->tail_req && step_tail_fused() && !dry) {
        t = *u->tail_req;
        t.xin_next = t.last ? nullptr : xin.p;
        t.eps_out = nullptr;
      } else {
        t.eps_out = out;
      }
      t.x = n.p; t.w = u->conv_out.w; t.bias = u->conv_out.bias; t.zeros = igemm_zero_page();
      t.B = B; t.H = L; t.W = L;
      ProfScope ps(0, s, 2.0 * B * L * L * 4 * 9 * n.C, (double)B * L * L * n.C * esize(dt), dry, "M=" + std::to_string(B * L * L) + " conv_out_tail ddim=" + std::to_string(t.ddim));
      if (!dry) {
        TRY(ex.ws_ok());
        const int r = launch_conv_out_tail(t, s);
        if (r) return fail(r == -2 ? LDMSEG_E_SHAPE : LDMSEG_E_HIP, "launch_conv_out_tail failed");
        if (t.ddim) { u->tail_done = true; u->xin_ready = t.xin_next != nullptr; u->xin_ptr = t.xin_next; }
      }
    } else {
      IgemmParams p;
      p.src0 = n.p; p.C0 = n.C; p.B = B; p.Hi = p.Ho = L; p.Wi = p.Wo = L;
      p.taps = 9; p.M = B * L * L; p.N = u->conv_out.N; p.n_valid = 4;
      p.W = u->conv_out.w; p.bias = u->conv_out.bias; p.out = out; p.epi = EPI_NCHW_F32;
      TRY(ex.igemm(p));
    }
    ws->reset(m);
  }
  return 0;
}

int ensure_ws(void** mem, size_t* cap, Workspace* ws, size_t need) {
  if (need <= *cap) return 0;
  if (*mem) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipFree(*mem));
    *mem = nullptr;
    *cap = 0;
  }
  if (hipMalloc(mem, need) != hipSuccess) return fail(LDMSEG_E_OOM, "workspace hipMalloc failed (" + std::to_string(need) + " bytes)");
  *cap = need;
  ws->base = (char*)*mem;
  ws->cap = need;
  return 0;
}

int unet_forward_checked(ldmseg_unet* u, const float* a, int Ca, const float* b, int Cb, const float* c, int Cc,
                         const int64_t* t_dev, int t_count, int64_t t_host, int B, int L, float* out, hipStream_t s) {
  // measure (cached per shape), (re)allocate, run
  if (u->plan_B != B || u->plan_L != L || u->plan_epoch != g_plan_epoch) {
    TRY(unet_forward_impl(u, a, Ca, b, Cb, c, Cc, t_dev, t_count, t_host, B, L, out, s, true, 0));
    u->plan_persist = rup(u->ws.persist_peak, 4096);
    u->plan_scratch = rup(u->ws.scratch_peak, 4096);
    u->plan_B = B;
    u->plan_L = L;
    u->plan_epoch = g_plan_epoch;
  }
  {
    const void* before = u->ws_mem;
    const size_t cap_before = u->ws_cap;
    TRY(ensure_ws(&u->ws_mem, &u->ws_cap, &u->ws, u->plan_persist + u->plan_scratch));
    if (u->ws_mem != before || u->ws_cap != cap_before) { u->xin_ready = false; u->xin_ptr = nullptr; }   // (a new arena may reuse the old address)
  }
  return unet_forward_impl(u, a, Ca, b, Cb, c, Cc, t_dev, t_count, t_host, B, L, out, s, false, u->plan_persist);
}

int check_arch(int device) {
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(LDMSEG_E_ARCH, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  return 0;
}

int make_weight_map(int n, const char* const* names, const void* const* ptrs, const int64_t* numels, WeightMap* wm) {
  if (!names || !ptrs || !numels) return fail(LDMSEG_E_ARG, "null weight arrays");
  for (int i = 0; i < n; ++i) wm->m[names[i]] = {(const float*)ptrs[i], numels[i]};
  return 0;
}

}  // namespace

// =================================================================== seg-VAE
struct ldmseg_vae {
  ldmseg_vae_cfg cfg{};
  int dt = DT_BF16;
  bool x3 = false;      // fp32 storage with split-bf16 GEMM arithmetic (LDMSEG_BF16X3)
  DeviceArena arena;
  Workspace ws;
  void* ws_mem = nullptr;
  size_t ws_cap = 0;
  int64_t nparams = 0;
  // encoder
  ConvW enc[10];          // encoder.{0,2,3,5,6,8,9,11,15} in order (9 used)
  NormW enc_gn;
  int enc_ch[5]{};        // storage channel counts
  // decoder
  ConvW dec_in, dec_out;
  ConvW convt[4];
  NormW ln[4], dec_gn;
  void* gn_sync = nullptr;
  void* cf_sync = nullptr;
  ~ldmseg_vae() {
    arena.release();
    if (ws_mem) (void)hipFree(ws_mem);
    if (gn_sync) (void)hipFree(gn_sync);
    if (cf_sync) (void)hipFree(cf_sync);
  }
};

namespace {

inline int cstore(int c, int dt) { return (int)rup(c, bke(dt)); }

int vae_build(ldmseg_vae* v, const WeightMap& wm) {
  hipStream_t s = nullptr;
  const ldmseg_vae_cfg& c = v->cfg;
  const int dt = v->dt;
  Builder b{&v->arena, &wm, dt, s};
  b.x3 = v->x3;
  if (c.in_channels > bke(dt) || c.latent_channels > bke(dt)) return fail(LDMSEG_E_ARG, "too many boundary channels");
  if (c.int_channels % 64 || c.upscale_channels % 64 || c.num_upscalers < 1 || c.num_upscalers > 4)
    return fail(LDMSEG_E_ARG, "unsupported seg-VAE configuration");
  // ---- encoder (vae.py:174-244) ----
  const int* boc = c.block_out_channels;
  int k = 0;
  TRY(b.conv("encoder.0", boc[0], c.in_channels, 3, bke(dt), &v->enc[k++], true, cstore(boc[0], dt)));
  int idx = 2;
  for (int i = 0; i < 3; ++i) {
    TRY(b.conv("encoder." + std::to_string(idx), boc[i], boc[i], 3, cstore(boc[i], dt), &v->enc[k++], true, cstore(boc[i], dt)));
    TRY(b.conv("encoder." + std::to_string(idx + 1), boc[i + 1], boc[i], 3, cstore(boc[i], dt), &v->enc[k++], true,
               cstore(boc[i + 1], dt)));
    idx += 3;
  }
  TRY(b.conv("encoder." + std::to_string(idx), c.int_channels, boc[3], 3, cstore(boc[3], dt), &v->enc[k++]));
  TRY(b.norm("encoder." + std::to_string(idx + 2), c.int_channels, &v->enc_gn));
  TRY(b.conv("encoder." + std::to_string(idx + 4), c.latent_channels * c.num_latents, c.int_channels, 3, c.int_channels,
             &v->enc[k++]));
  // ---- decoder (vae.py:123-172) ----
  TRY(b.conv("decoder.0", c.int_channels, c.latent_channels, 3, bke(dt), &v->dec_in));
  idx = 2;
  int cin = c.int_channels;
  for (int i = 0; i < c.num_upscalers; ++i) {
    ConvW& t = v->convt[i];
    const int Co = c.upscale_channels;
    const float *w, *bias;
    TRY(wm.get("decoder." + std::to_string(idx) + ".weight", (int64_t)cin * Co * 4, &w));
    TRY(wm.get("decoder." + std::to_string(idx) + ".bias", Co, &bias));
    t.N = 4 * Co; t.n_valid = 4 * Co; t.cin_pad = cin; t.taps = 1; t.cout = Co;
    TRY(v->arena.alloc(&t.w, (size_t)4 * Co * cin * esize(dt)));
    TRY(launch_repack_convt2(w, t.w, cin, Co, dt, s));
    b.weights(t.w, (size_t)4 * Co * cin);
    void* pb;
    TRY(v->arena.alloc(&pb, (size_t)4 * Co * sizeof(float)));
    for (int q = 0; q < 4; ++q)
      HIP_TRY(hipMemcpyAsync((float*)pb + q * Co, bias, Co * sizeof(float), hipMemcpyDeviceToDevice, s));
    t.bias = (float*)pb;
    b.nparams += (int64_t)cin * Co * 4 + Co;
    TRY(b.norm("decoder." + std::to_string(idx + 1), Co, &v->ln[i]));
    cin = Co;
    idx += 3;
  }
  TRY(b.norm("decoder." + std::to_string(idx), cin, &v->dec_gn));
  TRY(b.conv("decoder." + std::to_string(idx + 2), c.out_channels, cin, 3, cin, &v->dec_out));
  TRY(b.finish());
  v->nparams = b.nparams;
  return 0;
}

struct ArgmaxOut { int64_t* ids = nullptr; float* prob = nullptr; float mask_th = -1.f; int64_t ignore_label = 0; };
// fused evaluation tail (ldmseg_vae_decode_panoptic): host-side per-image geometry + device outputs
struct PanopticOut {
  int in_h = 0, in_w = 0;
  const int32_t* boxes = nullptr; const int32_t* sizes = nullptr; const int64_t* offsets = nullptr;
  int threshold_output = 0, threshold_mode = 0;
  float mask_th = 0.5f; int count_th = 0; double overlap_th = 0.0; int64_t ignore_label = 0;
  int32_t* labels = nullptr; int32_t* panoptic = nullptr; uint8_t* keep = nullptr; int32_t* counts = nullptr; int32_t* mask_counts = nullptr;
};

int vae_decode_impl(ldmseg_vae* v, const float* z, float z_scale, int B, int L, int interpolate, float* logits,
                    hipStream_t s, bool dry, size_t scratch_base, const ArgmaxOut* am = nullptr, const PanopticOut* po = nullptr) {
  Workspace* ws = &v->ws;
  ws->begin(dry, scratch_base);
  Exec ex{ws, v->dt, B, s};
  ex.gn_sync = v->gn_sync;
  ex.cf_sync = v->cf_sync;
  ex.x3 = v->x3 ? 2 : 0;
  const int dt = v->dt;
  const ldmseg_vae_cfg& c = v->cfg;
  Act zin = ex.new_act(bke(dt), L, L, true);
  {
    ProfScope ps(4, s, 0, 0, dry);
    if (!dry) TRY(launch_pack_nchw(z, zin.p, B, c.latent_channels, L * L, bke(dt), z_scale, 0.f, dt, s));
  }
  Act h;
  TRY(ex.conv(v->dec_in, zin, nullptr, &h, 1, 0, true, nullptr, 0, nullptr));
  for (int i = 0; i < c.num_upscalers; ++i) {
    const ConvW& t = v->convt[i];
    Act o = ex.new_act(t.cout, 2 * h.H, 2 * h.W, true);
    IgemmParams p;
    p.src0 = h.p; p.C0 = h.C; p.B = B; p.Hi = p.Ho = h.H; p.Wi = p.Wo = h.W;
    p.M = B * h.H * h.W; p.N = t.N; p.n_valid = t.N; p.W = t.w; p.bias = t.bias;
    p.out = o.p; p.ldo = t.cout; p.epi = EPI_CONVT2; p.cout = t.cout;
    TRY(ex.igemm(p));
    Act n;
    TRY(ex.layernorm(v->ln[i], o, 1e-6f, 1, &n, true));   // LayerNorm2d + SiLU, in place
    h = n;
  }
  Act g;
  TRY(ex.groupnorm(v->dec_gn, h, nullptr, 1e-5f, 1, &g));
  const int H4 = h.H, W4 = h.W;
  IgemmParams p;
  p.src0 = g.p; p.C0 = g.C; p.B = B; p.Hi = p.Ho = H4; p.Wi = p.Wo = W4;
  p.taps = 9; p.M = B * H4 * W4; p.N = v->dec_out.N; p.n_valid = c.out_channels;
  p.W = v->dec_out.w; p.bias = v->dec_out.bias;
  if (po) {   // fused evaluation tail: NHWC logits at 4L -> (x2, input size, crop, original size) resampling + panoptic post-processing
    Act lo = ex.new_act(c.out_channels, H4, W4, false);
    p.out = lo.p; p.ldo = c.out_channels; p.epi = EPI_STORE;
    TRY(ex.igemm(p));
    ProfScope ps(4, s, 0, (double)B * H4 * W4 * c.out_channels * esize(dt), dry);
    if (!dry) {
      TRY(ex.ws_ok());
      const int r = launch_panoptic_from_decoder(lo.p, B, H4, W4, c.out_channels, dt, po->in_h, po->in_w, po->boxes, po->sizes,
                                                 po->offsets, po->threshold_output, po->threshold_mode, po->mask_th, po->count_th,
                                                 po->overlap_th, po->ignore_label, po->labels, po->panoptic, po->keep, po->counts,
                                                 po->mask_counts, s);
      if (r == -2) return fail(LDMSEG_E_ARG, "decode_panoptic: bad crop box / output size / class count");
      TRY(r);
    }
  } else if (am) {   // fused tail: NHWC logits at 4L -> bilinear x2 + argmax + max-softmax, no logits tensor
    Act lo = ex.new_act(c.out_channels, H4, W4, false);
    p.out = lo.p; p.ldo = c.out_channels; p.epi = EPI_STORE;
    TRY(ex.igemm(p));
    ProfScope ps(4, s, 0, (double)B * H4 * W4 * c.out_channels * esize(dt) + (double)B * 4 * H4 * W4 * 12.0, dry);
    if (!dry) TRY(launch_bilinear2x_argmax(lo.p, am->ids, am->prob, B, H4, W4, c.out_channels, am->mask_th, am->ignore_label, dt, s));
  } else if (!interpolate) {
    p.out = logits; p.epi = EPI_NCHW_F32;
    TRY(ex.igemm(p));
  } else {
    Act lo = ex.new_act(c.out_channels, H4, W4, false);
    p.out = lo.p; p.ldo = c.out_channels; p.epi = EPI_STORE;
    TRY(ex.igemm(p));
    ProfScope ps(4, s, 0, (double)B * H4 * W4 * c.out_channels * (esize(dt) + 16.0), dry);
    if (!dry) TRY(launch_bilinear2x_nchw(lo.p, logits, B, H4, W4, c.out_channels, dt, s));
  }
  return 0;
}

int vae_encode_impl(ldmseg_vae* v, const float* x, float mul, float add, int B, int H, float* moments, hipStream_t s,
                    bool dry, size_t scratch_base) {
  Workspace* ws = &v->ws;
  ws->begin(dry, scratch_base);
  Exec ex{ws, v->dt, B, s};
  ex.gn_sync = v->gn_sync;
  ex.cf_sync = v->cf_sync;
  ex.x3 = v->x3 ? 2 : 0;
  const int dt = v->dt;
  const ldmseg_vae_cfg& c = v->cfg;
  Act xin = ex.new_act(bke(dt), H, H, true);
  {
    ProfScope ps(4, s, 0, 0, dry);
    if (!dry) TRY(launch_pack_nchw(x, xin.p, B, c.in_channels, H * H, bke(dt), mul, add, dt, s));
  }
  Act h, o;
  int k = 0;
  TRY(ex.conv(v->enc[k++], xin, nullptr, &h, 1, 0, true, nullptr, 0, nullptr, 1));  // conv + SiLU
  for (int i = 0; i < 3; ++i) {
    TRY(ex.conv(v->enc[k++], h, nullptr, &o, 1, 0, true, nullptr, 0, nullptr, 0)); h = o;
    TRY(ex.conv(v->enc[k++], h, nullptr, &o, 2, 0, true, nullptr, 0, nullptr, 1)); h = o;
  }
  TRY(ex.conv(v->enc[k++], h, nullptr, &o, 1, 0, true, nullptr, 0, nullptr, 0)); h = o;
  Act g;
  TRY(ex.groupnorm(v->enc_gn, h, nullptr, 1e-6f, 1, &g));
  const ConvW& last = v->enc[k];
  IgemmParams p;
  p.src0 = g.p; p.C0 = g.C; p.B = B; p.Hi = p.Ho = h.H; p.Wi = p.Wo = h.W;
  p.taps = 9; p.M = B * h.H * h.W; p.N = last.N; p.n_valid = c.latent_channels * c.num_latents;
  p.W = last.w; p.bias = last.bias; p.out = moments; p.epi = EPI_NCHW_F32;
  TRY(ex.igemm(p));
  return 0;
}

}  // namespace

// =================================================================== image VAE encoder (AutoencoderKL, SD-1.x)
struct ldmseg_vae_image {
  ldmseg_vae_image_cfg cfg{};
  int dt = DT_BF16;
  bool x3 = false;      // fp32 storage with split-bf16 GEMM arithmetic (LDMSEG_BF16X3)
  DeviceArena arena;
  Workspace ws;
  void* ws_mem = nullptr;
  size_t ws_cap = 0;
  int64_t nparams = 0;
  ConvW conv_in, downs[3], conv_out, quant;
  ResnetW down[4][2], mid[2];
  NormW attn_gn, norm_out;
  ConvW q, k, proj;
  void* wv = nullptr;       // value projection [512][512], used as the X operand of the V^T GEMM
  float* bv = nullptr;
  void* gn_sync = nullptr;
  void* cf_sync = nullptr;
  ~ldmseg_vae_image() {
    arena.release();
    if (ws_mem) (void)hipFree(ws_mem);
    if (gn_sync) (void)hipFree(gn_sync);
    if (cf_sync) (void)hipFree(cf_sync);
  }
};

namespace {

constexpr int kKLCh[4] = {128, 256, 512, 512};
constexpr int kKLMid = 512;

int klenc_build(ldmseg_vae_image* v, const WeightMap& wm) {
  hipStream_t s = nullptr;
  Builder b{&v->arena, &wm, v->dt, s};
  b.x3 = v->x3;
  const int cp = bke(v->dt);
  TRY(b.conv("encoder.conv_in", kKLCh[0], 3, 3, cp, &v->conv_in));
  int cin = kKLCh[0], dummy = 0;
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 2; ++j) {
      TRY(build_resnet(b, "encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j) + ".", cin, kKLCh[i],
                       &dummy, &v->down[i][j]));
      cin = kKLCh[i];
    }
    if (i < 3) TRY(b.conv("encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv", cin, cin, 3, cin, &v->downs[i]));
  }
  for (int j = 0; j < 2; ++j)
    TRY(build_resnet(b, "encoder.mid_block.resnets." + std::to_string(j) + ".", kKLMid, kKLMid, &dummy, &v->mid[j]));
  // single-head attention block; diffusers 0.16.1 names (query/key/value/proj_attn) or the later to_q/.../to_out.0
  const std::string ap = "encoder.mid_block.attentions.0.";
  const bool old_names = wm.m.count(ap + "query.weight") != 0;
  const std::string qn = old_names ? "query" : "to_q", kn = old_names ? "key" : "to_k", vn = old_names ? "value" : "to_v",
                    pn = old_names ? "proj_attn" : "to_out.0";
  TRY(b.norm(ap + "group_norm", kKLMid, &v->attn_gn));
  TRY(b.conv(ap + qn, kKLMid, kKLMid, 1, kKLMid, &v->q));
  TRY(b.conv(ap + kn, kKLMid, kKLMid, 1, kKLMid, &v->k));
  TRY(b.conv(ap + pn, kKLMid, kKLMid, 1, kKLMid, &v->proj));
  {
    const float* w;
    TRY(wm.get(ap + vn + ".weight", (int64_t)kKLMid * kKLMid, &w));
    TRY(b.arena->alloc(&v->wv, (size_t)kKLMid * kKLMid * esize(v->dt)));
    TRY(launch_repack_conv(w, v->wv, kKLMid, kKLMid, 1, 1, kKLMid, kKLMid, v->dt, s));
    b.nparams += (int64_t)kKLMid * kKLMid;
    TRY(b.f32_copy(ap + vn + ".bias", kKLMid, &v->bv));
  }
  TRY(b.norm("encoder.conv_norm_out", kKLMid, &v->norm_out));
  // conv_out writes its 8 channels into a full K tile (zero padded) so that quant_conv (1x1) can consume it
  TRY(b.conv("encoder.conv_out", 8, kKLMid, 3, kKLMid, &v->conv_out, true, cp));
  TRY(b.conv("quant_conv", 8, 8, 1, cp, &v->quant, true, 0, EPI_NCHW_F32));
  TRY(b.finish());
  v->nparams = b.nparams;
  return 0;
}

// diffusers ResnetBlock2D with temb=None, eps 1e-6 (AutoencoderKL)
int run_resnet_plain(Exec& ex, const ResnetW& r, const Act& x, Act* out) {
  Workspace* ws = ex.ws;
  const size_t m = ws->mark();
  Act n1, h1, n2, sc;
  TRY(ex.groupnorm(r.norm1, x, nullptr, 1e-6f, 1, &n1));
  TRY(ex.conv(r.conv1, n1, nullptr, &h1, 1, 0, false, nullptr, 0, nullptr));
  TRY(ex.groupnorm(r.norm2, h1, nullptr, 1e-6f, 1, &n2));
  const Act* resid = &x;
  if (r.has_shortcut) {
    TRY(ex.conv(r.shortcut, x, nullptr, &sc, 1, 0, false, nullptr, 0, nullptr));
    resid = &sc;
  }
  TRY(ex.conv(r.conv2, n2, nullptr, out, 1, 0, true, nullptr, 0, resid));
  ws->reset(m);
  return 0;
}

// AttentionBlock of the mid block: one head of width C=512 over the N=H*W tokens of each image.  Built from the
// implicit-GEMM kernel: S_b = Q_b K_b^T (fp32 rows) -> row softmax -> V_b^T = W_v X_b^T -> O_b = P_b V_b (+b_v, since
// softmax rows sum to one) -> proj + residual.  Runs once per image, not per denoising step.
int klenc_attention(Exec& ex, const ldmseg_vae_image* v, const Act& x, Act* out) {
  Workspace* ws = ex.ws;
  const size_t m = ws->mark();
  const int C = kKLMid, N = x.H * x.W;
  const size_t es = esize(ex.dt);
  if (N % 64 != 0) return fail(LDMSEG_E_SHAPE, "image VAE attention: (H/8)*(W/8) must be a multiple of 64");
  Act n, q, k, att;
  TRY(ex.groupnorm(v->attn_gn, x, nullptr, 1e-6f, 0, &n));
  TRY(ex.conv(v->q, n, nullptr, &q, 1, 0, false, nullptr, 0, nullptr));
  TRY(ex.conv(v->k, n, nullptr, &k, 1, 0, false, nullptr, 0, nullptr));
  att = ex.new_act(C, x.H, x.W, false);
  float* S = (float*)ws->scratch((size_t)N * N * sizeof(float));
  void* P = ws->scratch((size_t)N * N * es);
  void* Vt = ws->scratch((size_t)C * N * es);
  for (int b = 0; b < ex.B; ++b) {
    const size_t off = (size_t)b * N * C * es;
    {
      IgemmParams p;                                   // S = Q_b K_b^T
      p.src0 = (const char*)q.p + off; p.C0 = C; p.B = 1; p.Hi = p.Ho = N; p.Wi = p.Wo = 1;
      p.M = N; p.N = N; p.n_valid = N; p.W = (const char*)k.p + off; p.out = S; p.ldo = N; p.epi = EPI_ROWS_F32;
      p.w_dynamic = 1;
      TRY(ex.igemm(p));
    }
    {
      ProfScope ps(4, ex.s, 0, (double)N * N * (4 + es), ex.dry());
      if (!ex.dry()) TRY(launch_softmax_rows(S, P, N, N, 1.0f / std::sqrt((float)C), ex.dt, ex.s));
    }
    {
      IgemmParams p;                                   // V_b^T [C][N] = W_v X_b^T
      p.src0 = v->wv; p.C0 = C; p.B = 1; p.Hi = p.Ho = C; p.Wi = p.Wo = 1;
      p.M = C; p.N = N; p.n_valid = N; p.W = (const char*)n.p + off; p.out = Vt; p.ldo = N;
      p.w_dynamic = 1;
      TRY(ex.igemm(p));
    }
    {
      IgemmParams p;                                   // O_b = P_b V_b + b_v
      p.src0 = P; p.C0 = N; p.B = 1; p.Hi = p.Ho = N; p.Wi = p.Wo = 1;
      p.M = N; p.N = C; p.n_valid = C; p.W = Vt; p.bias = v->bv; p.out = (char*)att.p + off; p.ldo = C;
      p.w_dynamic = 1;
      TRY(ex.igemm(p));
    }
  }
  TRY(ex.conv(v->proj, att, nullptr, out, 1, 0, true, nullptr, 0, &x));
  ws->reset(m);
  return 0;
}

int klenc_encode_impl(ldmseg_vae_image* v, const float* x, float mul, float add, int B, int H, int W, float* moments,
                      hipStream_t s, bool dry, size_t scratch_base) {
  Workspace* ws = &v->ws;
  ws->begin(dry, scratch_base);
  Exec ex{ws, v->dt, B, s};
  ex.gn_sync = v->gn_sync;
  ex.cf_sync = v->cf_sync;
  ex.x3 = v->x3 ? 2 : 0;
  const int dt = v->dt;
  Act xin = ex.new_act(bke(dt), H, W, true);
  {
    ProfScope ps(4, s, 0, 0, dry);
    if (!dry) TRY(launch_pack_nchw(x, xin.p, B, 3, H * W, bke(dt), mul, add, dt, s));
  }
  Act h, o;
  TRY(ex.conv(v->conv_in, xin, nullptr, &h, 1, 0, true, nullptr, 0, nullptr));
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 2; ++j) { TRY(run_resnet_plain(ex, v->down[i][j], h, &o)); h = o; }
    if (i < 3) { TRY(ex.conv(v->downs[i], h, nullptr, &o, 2, 0, true, nullptr, 0, nullptr, 0, 0)); h = o; }
  }
  TRY(run_resnet_plain(ex, v->mid[0], h, &o)); h = o;
  TRY(klenc_attention(ex, v, h, &o)); h = o;
  TRY(run_resnet_plain(ex, v->mid[1], h, &o)); h = o;
  Act g, c8;
  TRY(ex.groupnorm(v->norm_out, h, nullptr, 1e-6f, 1, &g));
  TRY(ex.conv(v->conv_out, g, nullptr, &c8, 1, 0, true, nullptr, 0, nullptr));
  IgemmParams p;
  p.src0 = c8.p; p.C0 = c8.C; p.B = B; p.Hi = p.Ho = h.H; p.Wi = p.Wo = h.W;
  p.taps = 1; p.M = B * h.H * h.W; p.N = v->quant.N; p.n_valid = 8;
  p.W = v->quant.w; p.bias = v->quant.bias; p.out = moments; p.epi = EPI_NCHW_F32;
  TRY(ex.igemm(p));
  return 0;
}

}  // namespace

// =================================================================== C ABI
extern "C" {

const char* ldmseg_last_error(void) { return g_err.c_str(); }
const char* ldmseg_version(void) { return "ldmseg_hip 0.1 (gfx950)"; }

int ldmseg_unet_create(const ldmseg_unet_cfg* cfg, int n_weights, const char* const* names,
                       const void* const* dev_ptrs, const int64_t* numels, ldmseg_unet** out) {
  g_err.clear();
  if (!cfg || !out) return fail(LDMSEG_E_ARG, "null argument");
  if (cfg->cross_attention) return fail(LDMSEG_E_ARG, "cross-attention (encoder_hidden_states) is not supported: the reference default removes it (base.yaml:71)");
  if (cfg->compute_dtype != LDMSEG_F32 && cfg->compute_dtype != LDMSEG_BF16 && cfg->compute_dtype != LDMSEG_BF16X3) return fail(LDMSEG_E_ARG, "bad compute_dtype");
  if (cfg->in_channels != 4 && cfg->in_channels != 8 && cfg->in_channels != 12) return fail(LDMSEG_E_ARG, "in_channels must be 4, 8 or 12");
  DeviceGuard dg(cfg->device);
  TRY(check_arch(cfg->device));
  WeightMap wm;
  TRY(make_weight_map(n_weights, names, dev_ptrs, numels, &wm));
  ldmseg_unet* u = new ldmseg_unet();
  u->cfg = *cfg;
  u->dt = cfg->compute_dtype == LDMSEG_BF16 ? DT_BF16 : DT_F32;
  u->x3 = cfg->compute_dtype == LDMSEG_BF16X3;
  int r = unet_build(u, wm);
  if (r == 0) r = igemm_warm();
  if (r == 0) r = alloc_gn_sync(&u->gn_sync);
  if (r == 0) r = alloc_cf_sync(&u->cf_sync);
  if (r != 0) { delete u; return r; }
  *out = u;
  return 0;
}

void ldmseg_unet_destroy(ldmseg_unet* h) { delete h; }

int64_t ldmseg_unet_num_params(const ldmseg_unet* h) { return h ? h->nparams : 0; }

size_t ldmseg_unet_workspace_bytes(const ldmseg_unet* h, int B, int L) {
  if (!h) return 0;
  ldmseg_unet* u = const_cast<ldmseg_unet*>(h);
  DeviceGuard dg(u->cfg.device);
  const int ci = u->cfg.in_channels;
  if (unet_forward_impl(u, nullptr, ci, nullptr, 0, nullptr, 0, nullptr, 1, 0, B, L, nullptr, nullptr, true, 0) != 0) return 0;
  return rup(u->ws.persist_peak, 4096) + rup(u->ws.scratch_peak, 4096);
}

int ldmseg_unet_forward(ldmseg_unet* h, const float* x, const int64_t* t_dev, int t_count, int64_t t_host, int B, int L,
                        float* out, void* stream) {
  g_err.clear();
  if (!h || !x || !out) return fail(LDMSEG_E_ARG, "null argument");
  DeviceGuard dg(h->cfg.device);
  return unet_forward_checked(h, x, h->cfg.in_channels, nullptr, 0, nullptr, 0, t_dev, t_count, t_host, B, L, out,
                              (hipStream_t)stream);
}

int ldmseg_unet_forward_parts(ldmseg_unet* h, const float* latents, const float* rgb_latents, const float* cond,
                              const int64_t* t_dev, int t_count, int64_t t_host, int B, int L, float* out, void* stream) {
  g_err.clear();
  if (!h || !latents || !rgb_latents || !out) return fail(LDMSEG_E_ARG, "null argument");
  DeviceGuard dg(h->cfg.device);
  return unet_forward_checked(h, latents, 4, rgb_latents, 4, cond, cond ? 4 : 0, t_dev, t_count, t_host, B, L, out,
                              (hipStream_t)stream);
}

int ldmseg_vae_create(const ldmseg_vae_cfg* cfg, int n_weights, const char* const* names, const void* const* dev_ptrs,
                      const int64_t* numels, ldmseg_vae** out) {
  g_err.clear();
  if (!cfg || !out) return fail(LDMSEG_E_ARG, "null argument");
  if (cfg->compute_dtype != LDMSEG_F32 && cfg->compute_dtype != LDMSEG_BF16 && cfg->compute_dtype != LDMSEG_BF16X3) return fail(LDMSEG_E_ARG, "bad compute_dtype");
  if (cfg->num_latents != 2 || cfg->norm_num_groups != 32) return fail(LDMSEG_E_ARG, "only the gaussian parametrization with 32 groups is supported");
  DeviceGuard dg(cfg->device);
  TRY(check_arch(cfg->device));
  WeightMap wm;
  TRY(make_weight_map(n_weights, names, dev_ptrs, numels, &wm));
  ldmseg_vae* v = new ldmseg_vae();
  v->cfg = *cfg;
  v->dt = cfg->compute_dtype == LDMSEG_BF16 ? DT_BF16 : DT_F32;
  v->x3 = cfg->compute_dtype == LDMSEG_BF16X3;
  int r = vae_build(v, wm);
  if (r == 0) r = igemm_warm();
  if (r == 0) r = alloc_gn_sync(&v->gn_sync);
  if (r == 0) r = alloc_cf_sync(&v->cf_sync);
  if (r != 0) { delete v; return r; }
  *out = v;
  return 0;
}
void ldmseg_vae_destroy(ldmseg_vae* h) { delete h; }
int64_t ldmseg_vae_num_params(const ldmseg_vae* h) { return h ? h->nparams : 0; }

int ldmseg_vae_decode(ldmseg_vae* h, const float* z, float z_scale, int B, int L, int interpolate, float* logits,
                      void* stream) {
  g_err.clear();
  if (!h || !z || !logits) return fail(LDMSEG_E_ARG, "null argument");
  DeviceGuard dg(h->cfg.device);
  if (B < 1 || L < 1) return fail(LDMSEG_E_SHAPE, "bad B/L");
  TRY(vae_decode_impl(h, z, z_scale, B, L, interpolate, logits, (hipStream_t)stream, true, 0));
  const size_t persist = rup(h->ws.persist_peak, 4096), scratch = rup(h->ws.scratch_peak, 4096);
  TRY(ensure_ws(&h->ws_mem, &h->ws_cap, &h->ws, persist + scratch));
  return vae_decode_impl(h, z, z_scale, B, L, interpolate, logits, (hipStream_t)stream, false, persist);
}

int ldmseg_vae_decode_argmax(ldmseg_vae* h, const float* z, float z_scale, int B, int L, float mask_th, int64_t ignore_label,
                             int64_t* ids, float* max_prob, void* stream) {
  g_err.clear();
  if (!h || !z || !ids) return fail(LDMSEG_E_ARG, "null argument");
  DeviceGuard dg(h->cfg.device);
  if (B < 1 || L < 1) return fail(LDMSEG_E_SHAPE, "bad B/L");
  if (h->cfg.num_upscalers != 2) return fail(LDMSEG_E_ARG, "the fused tail assumes interpolation_factor 2 (num_upscalers 2)");
  ArgmaxOut am{ids, max_prob, mask_th, ignore_label};
  TRY(vae_decode_impl(h, z, z_scale, B, L, 1, nullptr, (hipStream_t)stream, true, 0, &am));
  const size_t persist = rup(h->ws.persist_peak, 4096), scratch = rup(h->ws.scratch_peak, 4096);
  TRY(ensure_ws(&h->ws_mem, &h->ws_cap, &h->ws, persist + scratch));
  return vae_decode_impl(h, z, z_scale, B, L, 1, nullptr, (hipStream_t)stream, false, persist, &am);
}

int ldmseg_vae_decode_panoptic(ldmseg_vae* h, const float* z, float z_scale, int B, int L, int in_h, int in_w,
                               const int32_t* crop_boxes, const int32_t* out_sizes, const int64_t* out_offsets,
                               int threshold_output, int threshold_mode, float mask_th, int count_th, double overlap_th,
                               int64_t ignore_label, int32_t* labels, int32_t* panoptic, uint8_t* keep, int32_t* counts,
                               int32_t* mask_counts, void* stream) {
  g_err.clear();
  if (!h || !z || !out_sizes || !out_offsets || !labels || !panoptic || !keep || !counts || !mask_counts)
    return fail(LDMSEG_E_ARG, "null argument");
  if (B < 1 || L < 1 || in_h < 1 || in_w < 1) return fail(LDMSEG_E_SHAPE, "bad B/L/input size");
  if (threshold_mode != 0 && threshold_mode != 1) return fail(LDMSEG_E_ARG, "threshold_mode must be 0 (max) or 1 (topk_diff)");
  if (h->cfg.num_upscalers != 2) return fail(LDMSEG_E_ARG, "the fused tail assumes interpolation_factor 2 (num_upscalers 2)");
  DeviceGuard dg(h->cfg.device);
  PanopticOut po;
  po.in_h = in_h; po.in_w = in_w; po.boxes = crop_boxes; po.sizes = out_sizes; po.offsets = out_offsets;
  po.threshold_output = threshold_output; po.threshold_mode = threshold_mode; po.mask_th = mask_th; po.count_th = count_th;
  po.overlap_th = overlap_th; po.ignore_label = ignore_label;
  po.labels = labels; po.panoptic = panoptic; po.keep = keep; po.counts = counts; po.mask_counts = mask_counts;
  TRY(vae_decode_impl(h, z, z_scale, B, L, 1, nullptr, (hipStream_t)stream, true, 0, nullptr, &po));
  const size_t persist = rup(h->ws.persist_peak, 4096), scratch = rup(h->ws.scratch_peak, 4096);
  TRY(ensure_ws(&h->ws_mem, &h->ws_cap, &h->ws, persist + scratch));
  return vae_decode_impl(h, z, z_scale, B, L, 1, nullptr, (hipStream_t)stream, false, persist, nullptr, &po);
}

int ldmseg_vae_encode(ldmseg_vae* h, const float* x, float in_mul, float in_add, int B, int H, float* moments,
                      void* stream) {
  g_err.clear();
  if (!h || !x || !moments) return fail(LDMSEG_E_ARG, "null argument");
  DeviceGuard dg(h->cfg.device);
  if (B < 1 || H < 8 || H % 8) return fail(LDMSEG_E_SHAPE, "H must be a multiple of 8");
  TRY(vae_encode_impl(h, x, in_mul, in_add, B, H, moments, (hipStream_t)stream, true, 0));
  const size_t persist = rup(h->ws.persist_peak, 4096), scratch = rup(h->ws.scratch_peak, 4096);
  TRY(ensure_ws(&h->ws_mem, &h->ws_cap, &h->ws, persist + scratch));
  return vae_encode_impl(h, x, in_mul, in_add, B, H, moments, (hipStream_t)stream, false, persist);
}

int ldmseg_vae_image_create(const ldmseg_vae_image_cfg* cfg, int n_weights, const char* const* names,
                            const void* const* dev_ptrs, const int64_t* numels, ldmseg_vae_image** out) {
  g_err.clear();
  if (!cfg || !out) return fail(LDMSEG_E_ARG, "null argument");
  if (cfg->compute_dtype != LDMSEG_F32 && cfg->compute_dtype != LDMSEG_BF16 && cfg->compute_dtype != LDMSEG_BF16X3) return fail(LDMSEG_E_ARG, "bad compute_dtype");
  DeviceGuard dg(cfg->device);
  TRY(check_arch(cfg->device));
  WeightMap wm;
  TRY(make_weight_map(n_weights, names, dev_ptrs, numels, &wm));
  ldmseg_vae_image* v = new ldmseg_vae_image();
  v->cfg = *cfg;
  v->dt = cfg->compute_dtype == LDMSEG_BF16 ? DT_BF16 : DT_F32;
  v->x3 = cfg->compute_dtype == LDMSEG_BF16X3;
  int r = klenc_build(v, wm);
  if (r == 0) r = igemm_warm();
  if (r == 0) r = alloc_gn_sync(&v->gn_sync);
  if (r == 0) r = alloc_cf_sync(&v->cf_sync);
  if (r != 0) { delete v; return r; }
  *out = v;
  return 0;
}
void ldmseg_vae_image_destroy(ldmseg_vae_image* h) { delete h; }
int64_t ldmseg_vae_image_num_params(const ldmseg_vae_image* h) { return h ? h->nparams : 0; }

int ldmseg_vae_image_encode(ldmseg_vae_image* h, const float* x, float in_mul, float in_add, int B, int H, int W,
                            float* moments, void* stream) {
  g_err.clear();
  if (!h || !x || !moments) return fail(LDMSEG_E_ARG, "null argument");
  DeviceGuard dg(h->cfg.device);
  if (B < 1 || H < 8 || W < 8 || H % 8 || W % 8) return fail(LDMSEG_E_SHAPE, "H and W must be multiples of 8");
  TRY(klenc_encode_impl(h, x, in_mul, in_add, B, H, W, moments, (hipStream_t)stream, true, 0));
  const size_t persist = rup(h->ws.persist_peak, 4096), scratch = rup(h->ws.scratch_peak, 4096);
  TRY(ensure_ws(&h->ws_mem, &h->ws_cap, &h->ws, persist + scratch));
  return klenc_encode_impl(h, x, in_mul, in_add, B, H, W, moments, (hipStream_t)stream, false, persist);
}

int ldmseg_vae_posterior(const float* moments, const float* noise, float out_scale, int B, int l, float* out,
                         void* stream) {
  g_err.clear();
  if (!moments || !out) return fail(LDMSEG_E_ARG, "null argument");
  TRY(launch_posterior_sample(moments, noise, out, B, l * l, (hipStream_t)stream));
  if (out_scale != 1.0f) TRY(launch_axpby(out, out_scale, 0.f, out, (size_t)B * 4 * l * l, (hipStream_t)stream));
  return 0;
}

int ldmseg_ddim_step(const float* model_output, const float* sample, float sqrt_alpha_t, float sqrt_beta_t,
                     float sqrt_alpha_prev, float sqrt_beta_prev, int prediction_type, int clip_sample,
                     float clip_sample_range, int use_clipped_model_output, float* prev_sample,
                     float* pred_original_sample, size_t n, void* stream) {
  g_err.clear();
  if (!model_output || !sample) return fail(LDMSEG_E_ARG, "null argument");
  if (prediction_type < 0 || prediction_type > 2) return fail(LDMSEG_E_ARG, "unknown prediction_type");
  DdimCoef c{sqrt_alpha_t, sqrt_beta_t, sqrt_alpha_prev, sqrt_beta_prev, prediction_type, clip_sample, clip_sample_range,
             use_clipped_model_output};
  TRY(launch_ddim_step(model_output, sample, prev_sample, pred_original_sample, n, c, (hipStream_t)stream));
  return 0;
}

static int noise_args_ok(const void* a, const void* noise, const void* t, const void* ac, const void* out, int B,
                         size_t per, int n_train) {
  if (!a || !noise || !t || !ac || !out) return fail(LDMSEG_E_ARG, "null argument");
  if (B < 1 || per < 1 || n_train < 1) return fail(LDMSEG_E_ARG, "B, per_sample and n_train_timesteps must be positive");
  return 0;
}
int ldmseg_add_noise(const float* original, const float* noise, const int64_t* timesteps_dev,
                     const float* alphas_cumprod_dev, int n_train_timesteps, float scale, float* out, int B,
                     size_t per_sample, void* stream) {
  g_err.clear();
  TRY(noise_args_ok(original, noise, timesteps_dev, alphas_cumprod_dev, out, B, per_sample, n_train_timesteps));
  TRY(launch_add_noise(original, noise, timesteps_dev, alphas_cumprod_dev, n_train_timesteps, scale, out, B, per_sample, 0,
                       (hipStream_t)stream));
  return 0;
}
int ldmseg_remove_noise(const float* noisy, const float* noise, const int64_t* timesteps_dev,
                        const float* alphas_cumprod_dev, int n_train_timesteps, float scale, float* out, int B,
                        size_t per_sample, void* stream) {
  g_err.clear();
  TRY(noise_args_ok(noisy, noise, timesteps_dev, alphas_cumprod_dev, out, B, per_sample, n_train_timesteps));
  TRY(launch_add_noise(noisy, noise, timesteps_dev, alphas_cumprod_dev, n_train_timesteps, scale, out, B, per_sample, 1,
                       (hipStream_t)stream));
  return 0;
}

// (re)allocate the sampler's eps / self-condition buffers; growth synchronises the device
// time-embedding rows of every step of a sampling loop: rows[i] = time_emb_proj_all(silu(MLP(sinusoid(timesteps[i]))))
// room for the time-embedding rows of `n_steps` steps (at least 64: 6 MB, so that a short warm-up call followed by a longer
// run does not re-allocate); ldmseg_unet_reserve sizes it up front
static int temb_reserve(ldmseg_unet* h, int n_steps) {
  if (h->temb_cap >= n_steps) return 0;
  const size_t per = sizeof(int64_t) + (320 + 2 * (size_t)kTimeDim + h->temb_total) * sizeof(float);
  int cap = 64;
  while (cap < n_steps) cap *= 2;
  HIP_TRY(hipDeviceSynchronize());
  if (h->temb_buf) (void)hipFree(h->temb_buf);
  h->temb_buf = nullptr; h->temb_cap = 0;
  HIP_TRY(hipMalloc(&h->temb_buf, per * cap));
  h->temb_cap = cap;
  return 0;
}
static int loop_time_embeddings(ldmseg_unet* h, const int64_t* timesteps, int n_steps, hipStream_t s, const float** rows) {
  TRY(temb_reserve(h, n_steps));
  int64_t* ts = (int64_t*)h->temb_buf;
  float* sinus = (float*)(ts + h->temb_cap);
  float* e1 = sinus + (size_t)h->temb_cap * 320;
  float* emb = e1 + (size_t)h->temb_cap * kTimeDim;
  float* table = emb + (size_t)h->temb_cap * kTimeDim;
  HIP_TRY(hipMemcpyAsync(ts, timesteps, (size_t)n_steps * sizeof(int64_t), hipMemcpyHostToDevice, s));
  for (int i0 = 0; i0 < n_steps; i0 += 64) {          // launch_small_linear: up to 64 rows
    const int r = n_steps - i0 < 64 ? n_steps - i0 : 64;
    TRY(launch_time_embed(ts + i0, r, 0, r, sinus + (size_t)i0 * 320, s));
    TRY(launch_small_linear(sinus + (size_t)i0 * 320, h->te1_w, h->te1_b, e1 + (size_t)i0 * kTimeDim, r, 320, kTimeDim, 0, 1, s));
    TRY(launch_small_linear(e1 + (size_t)i0 * kTimeDim, h->te2_w, h->te2_b, emb + (size_t)i0 * kTimeDim, r, kTimeDim, kTimeDim, 0, 0, s));
    TRY(launch_small_linear(emb + (size_t)i0 * kTimeDim, h->tproj_w, h->tproj_b, table + (size_t)i0 * h->temb_total, r, kTimeDim,
                            h->temb_total, 1, 0, s));
  }
  *rows = table;
  return 0;
}

static int loop_reserve(ldmseg_unet* h, size_t n) {
  if (h->loop_elems >= n) return 0;
  HIP_TRY(hipDeviceSynchronize());
  if (h->cond) (void)hipFree(h->cond);
  if (h->eps) (void)hipFree(h->eps);
  h->cond = h->eps = nullptr;
  h->loop_elems = 0;
  HIP_TRY(hipMalloc((void**)&h->cond, n * sizeof(float)));
  HIP_TRY(hipMalloc((void**)&h->eps, n * sizeof(float)));
  h->loop_elems = n;
  return 0;
}

int ldmseg_unet_set_attention_fp8(ldmseg_unet* h, int min_tokens) {
  g_err.clear();
  if (!h || min_tokens < 0) return fail(LDMSEG_E_ARG, "bad argument");
  if (min_tokens > 0 && h->dt != DT_BF16) return fail(LDMSEG_E_ARG, "the fp8 attention path belongs to the bf16 perf mode");
  h->attn_fp8_min_tokens = min_tokens;
  h->plan_B = h->plan_L = 0;          // the workspace plan depends on it
  return 0;
}

int ldmseg_unet_reserve(ldmseg_unet* h, int B, int L) {
  g_err.clear();
  if (!h) return fail(LDMSEG_E_ARG, "null argument");
  if (B < 1 || L < 8 || L % 8 != 0) return fail(LDMSEG_E_SHAPE, "L must be a positive multiple of 8, B >= 1");
  DeviceGuard dg(h->cfg.device);
  const int ci = h->cfg.in_channels;
  if (h->plan_B != B || h->plan_L != L || h->plan_epoch != g_plan_epoch) {
    TRY(unet_forward_impl(h, nullptr, ci, nullptr, 0, nullptr, 0, nullptr, 1, 0, B, L, nullptr, nullptr, true, 0));
    h->plan_persist = rup(h->ws.persist_peak, 4096);
    h->plan_scratch = rup(h->ws.scratch_peak, 4096);
    h->plan_B = B;
    h->plan_L = L;
    h->plan_epoch = g_plan_epoch;
  }
  TRY(ensure_ws(&h->ws_mem, &h->ws_cap, &h->ws, h->plan_persist + h->plan_scratch));
  TRY(loop_reserve(h, (size_t)B * 4 * L * L));
  TRY(temb_reserve(h, 64));
  TRY(igemm_warm());
  return 0;
}

int ldmseg_sample_loop(ldmseg_unet* h, const ldmseg_sample_cfg* cfg, float* latents, const float* rgb_latents, int B, int L,
                       float* all_latents, void* stream) {
  g_err.clear();
  if (!h || !cfg || !latents || !rgb_latents || !cfg->timesteps || !cfg->coef) return fail(LDMSEG_E_ARG, "null argument");
  if (cfg->n_steps < 1) return fail(LDMSEG_E_ARG, "n_steps must be >= 1");
  if (B < 1 || L < 8 || L % 8 != 0) return fail(LDMSEG_E_SHAPE, "L must be a positive multiple of 8, B >= 1");
  if (cfg->prediction_type < 0 || cfg->prediction_type > 2) return fail(LDMSEG_E_ARG, "unknown prediction_type");
  for (int i = 0; i < cfg->n_steps; ++i)
    if (cfg->timesteps[i] < 0) return fail(LDMSEG_E_ARG, "negative timestep");
  DeviceGuard dg(h->cfg.device);
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)B * 4 * L * L;
  const bool selfc = cfg->self_condition != 0;
  if ((h->cfg.in_channels == 12) != selfc) return fail(LDMSEG_E_ARG, "self_condition needs the 12-channel conv_in (and vice versa)");
  TRY(loop_reserve(h, n));      // no-op after ldmseg_unet_reserve or a previous call at this size
  const bool inpaint = cfg->known_dev != nullptr;
  if (inpaint && (!cfg->z0_dev || !cfg->noise_dev || !cfg->paste_coef)) return fail(LDMSEG_E_ARG, "inpainting needs z0, noise and paste_coef");
  if (selfc) HIP_TRY(hipMemsetAsync(h->cond, 0, n * sizeof(float), s));   // condition = zeros_like(rgb_latents)
  const float* temb_rows = nullptr;
  TRY(loop_time_embeddings(h, cfg->timesteps, cfg->n_steps, s, &temb_rows));
  if (h->gn_sync) {                  // cooperative GroupNorm back-off: did the previous calls' norms miss their partners?
    if (!h->gn_diag_host) {
      HIP_TRY(hipHostMalloc((void**)&h->gn_diag_host, sizeof(unsigned long long)));
      *h->gn_diag_host = 0;
    }
    const unsigned long long cur = *(volatile unsigned long long*)h->gn_diag_host;     // whatever the last finished copy left
    // (only launches under the full bound count - norm.hip - so the eight short-bound calls decay unconditionally and the
    // ninth probes the full bound again)
    if (cur >= h->gn_diag_seen + 256) h->gn_backoff_calls = 8;
    else if (h->gn_backoff_calls > 0) --h->gn_backoff_calls;
    h->gn_diag_seen = cur;
  }
  struct Clear { ldmseg_unet* u; ~Clear() { u->temb_override = nullptr; u->tail_req = nullptr; u->tail_done = false; u->xin_ready = false; u->xin_ptr = nullptr; } } clear{h};     // (also on the error returns below)
  for (int i = 0; i < cfg->n_steps; ++i) {
    h->temb_override = temb_rows + (size_t)i * h->temb_total;
    const float* c = cfg->coef + 4 * i;
    DdimCoef dc{c[0], c[1], c[2], c[3], cfg->prediction_type, cfg->clip_sample, cfg->clip_sample_range, 0};
    const bool last = (i == cfg->n_steps - 1);
    // the scheduler step of this iteration, offered to the forward's conv_out launch (bf16: tail.hip carries it out in its
    // epilogue together with the inpainting paste, the self-condition write and the next step's input pack)
    StepTail st;
    st.ddim = 1; st.last = last ? 1 : 0; st.c = dc; st.latents = latents; st.cond = selfc ? h->cond : nullptr; st.rgb = rgb_latents;
    if (inpaint) { st.known = cfg->known_dev; st.z0 = cfg->z0_dev; st.noise = cfg->noise_dev; st.sa = cfg->paste_coef[2 * i]; st.sb = cfg->paste_coef[2 * i + 1]; }
    h->tail_req = &st;
    h->tail_done = false;
    TRY(unet_forward_checked(h, latents, 4, rgb_latents, 4, selfc ? h->cond : nullptr, selfc ? 4 : 0, nullptr, 1,
                             cfg->timesteps[i], B, L, h->eps, s));
    h->tail_req = nullptr;
    if (!h->tail_done) {
      // condition <- pred_original_sample; latents <- prev_sample (last step: pred_original_sample)
      if (!last) {
        TRY(launch_ddim_step(h->eps, latents, latents, selfc ? h->cond : nullptr, n, dc, s));
      } else {
        TRY(launch_ddim_step(h->eps, latents, nullptr, latents, n, dc, s));
      }
      if (inpaint)
        TRY(launch_inpaint_paste(latents, cfg->z0_dev, cfg->noise_dev, cfg->known_dev, cfg->paste_coef[2 * i],
                                 cfg->paste_coef[2 * i + 1], B, 4, L * L, s));
    }
    if (all_latents)
      HIP_TRY(hipMemcpyAsync(all_latents + (size_t)i * n, latents, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  if (h->gn_sync && h->gn_diag_host)
    HIP_TRY(hipMemcpyAsync(h->gn_diag_host, gn_sync_diag_ptr(h->gn_sync), sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  return 0;
}

int ldmseg_unet_cf_fallbacks(ldmseg_unet* h, int64_t* count) {
  g_err.clear();
  if (!h || !count) return fail(LDMSEG_E_ARG, "null argument");
  DeviceGuard dg(h->cfg.device);
  unsigned long long n = 0;
  if (h->cf_sync) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&n, (const unsigned long long*)h->cf_sync + igemm_cf_bytes() / 8 - 1, sizeof n, hipMemcpyDeviceToHost));
  }
  *count = (int64_t)n;
  return 0;
}

int ldmseg_unet_gn_backoff(ldmseg_unet* h, int32_t* calls_left) {
  g_err.clear();
  if (!h || !calls_left) return fail(LDMSEG_E_ARG, "null argument");
  *calls_left = h->gn_backoff_calls;
  return 0;
}

int ldmseg_unet_gn_fallbacks(ldmseg_unet* h, int64_t* count) {
  g_err.clear();
  if (!h || !count) return fail(LDMSEG_E_ARG, "null argument");
  DeviceGuard dg(h->cfg.device);
  const long long n = h->gn_sync ? gn_coop_fallbacks(h->gn_sync) : 0;
  if (n < 0) return fail(LDMSEG_E_HIP, "reading the GroupNorm fallback counter failed");
  *count = n;
  return 0;
}

int ldmseg_bit_encode(const int64_t* ids, int B, int n_bits, int HW, int64_t ignore_label, float fill_value, float mul,
                      float add, float* bits, uint8_t* ignore_mask, void* stream) {
  g_err.clear();
  if (!ids || !bits || n_bits < 1 || n_bits > 31) return fail(LDMSEG_E_ARG, "bad bit_encode argument");
  TRY(launch_bit_encode(ids, bits, ignore_mask, B, n_bits, HW, ignore_label, fill_value, mul, add, (hipStream_t)stream));
  return 0;
}
int ldmseg_bit_decode(const float* x, int B, int n_bits, int HW, int64_t* ids, void* stream) {
  g_err.clear();
  if (!x || !ids || n_bits < 1 || n_bits > 24) return fail(LDMSEG_E_ARG, "bad bit_decode argument");
  TRY(launch_bit_decode(x, ids, B, n_bits, HW, (hipStream_t)stream));
  return 0;
}

int ldmseg_panoptic_postprocess(const float* logits, int B, int C, int H, int W, int threshold_output, int threshold_mode,
                                float mask_th, int count_th, double overlap_th, int64_t ignore_label, int32_t* labels,
                                int32_t* panoptic, uint8_t* keep, int32_t* counts, int32_t* mask_counts, void* stream) {
  g_err.clear();
  if (!logits || !labels || !panoptic || !keep || !counts || !mask_counts || B < 1 || C < 1 || C > 256 || H < 1 || W < 1 ||
      (threshold_mode != 0 && threshold_mode != 1))
    return fail(LDMSEG_E_ARG, "bad panoptic_postprocess argument");
  TRY(launch_panoptic_postprocess(logits, B, C, H * W, threshold_output, threshold_mode, mask_th, count_th, overlap_th,
                                  ignore_label, labels, panoptic, keep, counts, mask_counts, (hipStream_t)stream));
  return 0;
}

int ldmseg_profile_enable(int enable) {
  g_prof.on = enable != 0;
  return 0;
}
int ldmseg_profile_reset(void) {
  for (auto& f : g_prof.fam) {
    for (auto& e : f.ev) { g_prof.pool.push_back(e.first); g_prof.pool.push_back(e.second); }
    f.ev.clear();
    f.label.clear();
    f.lflops.clear();
    f.launches = 0;
    f.flops = f.bytes = 0;
  }
  return 0;
}
int ldmseg_profile_read(int family, int64_t* launches, double* total_ms, double* flops, double* bytes) {
  g_err.clear();
  if (family < 0 || family > 4) return fail(LDMSEG_E_ARG, "family must be 0..4");
  ProfFamily& f = g_prof.fam[family];
  double ms = 0;
  for (auto& e : f.ev) {
    HIP_TRY(hipEventSynchronize(e.second));
    float t = 0;
    HIP_TRY(hipEventElapsedTime(&t, e.first, e.second));
    ms += t;
  }
  if (launches) *launches = f.launches;
  if (total_ms) *total_ms = ms;
  if (flops) *flops = f.flops;
  if (bytes) *bytes = f.bytes;
  return 0;
}

// tuning knobs for experiments: key 0 = igemm K-loop ring depth (2, 3, 4)
int ldmseg_debug_set(int key, int value) {
  if (key == 2) { attention_set_qf1(value); return 0; }
  if (key == 1) { igemm_set_dbg(value); ++g_plan_epoch; return 0; }
  if (key == 5) { igemm_force_cfg(value); ++g_plan_epoch; return 0; }   // tools/tune_igemm.py: entry of igemm's instantiation list, -1 = off
  if (key == 8) { groupnorm_set_variant(value); ++g_plan_epoch; return 0; }   // (bit 3 changes which scratch conv_groupnorm plans)
  if (key == 9) { igemm_set_cm_mode(value); ++g_plan_epoch; return 0; }   // K order of 3x3 conv launches: -1 rule, 0 tap-major, 1 channel-major
  // cooperative GroupNorm hand-off: 10 = mode (1: every workgroup computes its partners' records itself), 11 = poll bound in us
  static int gn_mode = 0, gn_poll = 100;
  if (key == 10) { gn_mode = value; groupnorm_set_coop(gn_mode, gn_poll); return 0; }
  if (key == 11) { gn_poll = value; groupnorm_set_coop(gn_mode, gn_poll); return 0; }
  if (key == 12) { mlp_fused_set_mode(value); ++g_plan_epoch; return 0; }   // transformer feed-forward fusion: bit 0 MLP, bit 1 + proj_out
  if (key == 13) { mlp_fused_set_dbg(value); return 0; }                    // bit 8: no start-chunk rotation (bits 0-7: ablate builds)
  // 14: step tail.  bit 0: dedicated conv_out kernel (bf16); bit 1: scheduler step in its epilogue.  (The implicit-GEMM
  // conv_out it falls back to plans split-K scratch: a new plan epoch.)
  if (key == 14) { step_tail_set_mode(value); ++g_plan_epoch; return 0; }
  if (key == 16) { proj_qkv_set_mode(value); ++g_plan_epoch; return 0; }   // proj_in -> norm1 -> q|k|v in one launch (bf16, 320 channels); default 1
  if (key == 15) { attention_mx_set_mode(value); ++g_plan_epoch; return 0; }   // fp8 attention: 1 = scaled MFMAs where the shape allows (default), 0 = unscaled
  // (17 was round 5's weight-streaming kernel: measured level with igemm_kernel, now a record under tools/experiments/)
  // 23: K slices finished inside the igemm launch (round 6).  bit 0: on (256-row tiles, where it measured faster); bit 1: zero-length
  // partner poll (test: the last arriver reduces the shares of everybody who gave up); bit 2: resnet conv1 -> norm2 on the maps whose conv
  // runs on those tiles as conv (finished in-launch) + GroupNorm instead of slabs + the fused finish-GroupNorm launch; bit 3: in-launch
  // finish on every tile form that has the instantiation; bits 8-23: poll bound in microseconds (0 = 200).  Default 5.
  if (key == 23) { igemm_set_cf_mode(value); ++g_plan_epoch; return 0; }
  if (key == 24) { igemm_set_table_override(value); ++g_plan_epoch; return 0; }   // tuning: launch-table override, -1 = off (igemm_set_table_override)
  if (key == 19) { igemm_set_xt_mode(value); ++g_plan_epoch; return 0; }   // 1 (default): resnet conv2 + conv_shortcut as one launch (bf16)
  if (key == 20) { g_ffp_mode = value ? 1 : 0; ++g_plan_epoch; return 0; }
  if (key == 22) { g_gnfold_mode = value < 0 ? 0 : value; ++g_plan_epoch; return 0; }   // GroupNorm folded into the fused transformer entry: 0 off, 1 on, n > 1 on with n pixel chunks
  if (key == 21) { igemm_set_up4_mode(value); ++g_plan_epoch; return 0; }   // 1 (default): upsampler convs as four 2x2 phase convs (bf16)
  if (key == 6 || key == 7) { ops_bench_knob(key, value); return 0; }   // ldmseg_bench_igemm: 6 = weight copies rotated, 7 = folded-LN launch
  static unsigned long long ts_ptr = 0;                // keys 3/4: low/high half of a device stamp buffer (ablate builds)
  if (key == 3) { ts_ptr = (ts_ptr & 0xffffffff00000000ull) | (unsigned)value; igemm_set_tsbuf((void*)(uintptr_t)ts_ptr); return 0; }
  if (key == 4) { ts_ptr = (ts_ptr & 0xffffffffull) | ((unsigned long long)(unsigned)value << 32); igemm_set_tsbuf((void*)(uintptr_t)ts_ptr); return 0; }
  return fail(LDMSEG_E_ARG, "unknown debug key");
}

int ldmseg_debug_get(int key) {
  if (key == 1) return igemm_get_dbg();
  if (key == -1) return igemm_default_dbg();     // the shipped value of key 1
  if (key == 9) return igemm_get_cm_mode();
  if (key == 12) return mlp_fused_get_mode();
  if (key == 14) return step_tail_get_mode();
  if (key == 16) return proj_qkv_get_mode();
  if (key == 15) return attention_mx_get_mode();
  if (key == 23) return igemm_get_cf_mode();
  if (key == 19) return igemm_get_xt_mode();
  if (key == 20) return g_ffp_mode;
  if (key == 22) return g_gnfold_mode;
  if (key == 21) return igemm_get_up4_mode();
  if (key == 10) { const long long n = gn_coop_fallbacks(nullptr); return n > 0x7fffffffll ? 0x7fffffff : (int)n; }   // ring regions (ldmseg_op_* launches)
  return 0;
}

// "igemm<bf16,BM,BN,WM,WN,NST,PIPE,LDR>[/splitk] splits=S grid=G" of the most recent igemm launch
int ldmseg_igemm_last_kernel(char* buf, int n) {
  if (!buf || n < 1) return LDMSEG_E_ARG;
  const IgemmDispatch d = igemm_last_dispatch();
  std::snprintf(buf, (size_t)n, "%s splits=%d grid=%d", igemm_dispatch_name(d).c_str(), d.splits, d.grid);
  return 0;
}
int ldmseg_igemm_log(int enable) { igemm_log_enable(enable); return 0; }
int ldmseg_igemm_log_read(char* buf, int n) {
  if (!buf || n < 1) return LDMSEG_E_ARG;
  const std::string s = igemm_log_read();
  if ((int)s.size() + 1 > n) return LDMSEG_E_ARG;
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return 0;
}

// one CSV line per recorded launch: family,label,ms,flops
int ldmseg_profile_dump(const char* path) {
  g_err.clear();
  FILE* fp = std::fopen(path, "w");
  if (!fp) return fail(LDMSEG_E_ARG, "cannot open profile dump file");
  std::fprintf(fp, "family,label,ms,flops\n");
  for (int fi = 0; fi < 5; ++fi) {
    ProfFamily& f = g_prof.fam[fi];
    for (size_t i = 0; i < f.ev.size(); ++i) {
      (void)hipEventSynchronize(f.ev[i].second);
      float t = 0;
      (void)hipEventElapsedTime(&t, f.ev[i].first, f.ev[i].second);
      std::fprintf(fp, "%d,%s,%.6f,%.0f\n", fi, i < f.label.size() ? f.label[i].c_str() : "", t,
                   i < f.lflops.size() ? f.lflops[i] : 0.0);
    }
  }
  std::fclose(fp);
  return 0;
}

}  // extern "C"
