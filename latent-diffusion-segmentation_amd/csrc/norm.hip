// GroupNorm(32) [+SiLU] and LayerNorm [+SiLU] over NHWC activations (HBM-bound).
//
// GroupNorm runs as two launches over [B, HW, C]:
//   1. gn_partial: each workgroup owns a pixel chunk of one image, streams whole
//      rows with 16-B loads (every lane a fixed channel vector, so the group of
//      each element is a per-thread constant), and writes per-(chunk, group)
//      {sum, sumsq} partials in a fixed order (deterministic, no atomics).
//   2. gn_apply: reduces the partials of its image (tiny), then normalises,
//      applies gamma/beta (+SiLU) and writes the tensor the next conv gathers.
//      Two sources are read as one channel-concatenated tensor, which is how
//      torch.cat([hidden, skip], 1) disappears from the up path.
// LayerNorm: one wavefront per row, values kept in registers, two-pass mean /
// centred variance with wave shuffles (DPP) - same arithmetic as torch.
#include <mutex>

#include "common.h"
#include "kernels.h"

namespace ldmseg {
namespace {

#ifdef LDMSEG_GN_STAMP
__device__ unsigned long long g_gn_ts[8192 * 8];
#define GNSTAMP(slot) if (threadIdx.x == 0) g_gn_ts[((size_t)blockIdx.y * gridDim.x + blockIdx.x) % 8192 * 8 + (slot)] = __builtin_amdgcn_s_memtime();
#else
#define GNSTAMP(slot)
#endif
int num_cus_gn() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}
constexpr int kMaxIter = 4;   // ceil(nvec / 256) supported (C*sizeof(T)/16 <= 1024)

template <typename T>
__device__ __forceinline__ uint4 load_cat(const GNParams& p, int b, int pix, int v) {
  constexpr int PC = Chunk<T>::N;
  const int c = v * PC;
  if (c < p.C0) return *(const uint4*)((const T*)p.src0 + ((size_t)b * p.HW + pix) * p.C0 + c);
  return *(const uint4*)((const T*)p.src1 + ((size_t)b * p.HW + pix) * p.C1 + (c - p.C0));
}

// Parallel-variance combination (Chan et al.): (n, mean, M2) <- (n, mean, M2) + (nb, mb, M2b).  Used at every level of
// the GroupNorm statistics (thread -> workgroup -> image), so the variance never comes from E[x^2] - E[x]^2 and large
// channel means (real checkpoints) cannot cancel it away.
__device__ __forceinline__ void chan_add(double& n, double& mean, double& m2, double nb, double mb, double m2b) {
  if (nb <= 0.0) return;
  const double tot = n + nb;
  const double d = mb - mean;
  mean += d * (nb / tot);
  m2 += m2b + d * d * (n * nb / tot);
  n = tot;
}

// fixed-order reduction of one image's per-chunk partials {mean, M2}: 8 lanes per group (each a fixed chunk subset,
// fixed shuffle tree -> deterministic).  256 threads; out = {mean, rstd} per group.
// Closed form of the pairwise update, in two passes over the register-resident partials: with delta_i = mean_i - K
// (K = the group's first chunk mean, so the deltas are of the order of the deviation and fp32 carries them exactly
// enough even under a channel mean 1000x larger),  mean = K + sum n_i delta_i / N,  M2 = sum M2_i + n_i (delta_i - dbar)^2.
// All loads are issued before the first use.  The earlier form - a rolled loop of one dependent L2 load + one fp64
// pairwise update (two fp64 divisions) per chunk - cost 5.6 us in front of every apply workgroup's first pixel and
// 3.8 us at the end of every statistics workgroup (s_memtime stamps, tools/gn_stamps.py), as much as the streaming itself.
__device__ __forceinline__ float sum8_fixed(float v) {
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) {
    const float w = __shfl_xor(v, o);
    v = (threadIdx.x & o) ? w + v : v + w;      // same operand order on both partners
  }
  return v;
}
__device__ __forceinline__ void gn_reduce_stats(const GNParams& p, int b, int cpg, float* out) {
  const int g = threadIdx.x >> 3, sub = threadIdx.x & 7;
  const int per = p.per;
  constexpr int MAXP = 16;                     // nchunk <= 128 (gn_nchunk)
  float2 pp[MAXP];
  float cnt[MAXP];                             // exact: pixels x channels per group < 2^24
  const float2* src = (const float2*)p.partial + ((size_t)b * p.nchunk * p.groups + (g < p.groups ? g : 0));
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int ch = sub + i * 8;
    const int p0 = ch * per, p1 = min(p.HW, p0 + per);
    const bool ok = g < p.groups && ch < p.nchunk && p1 > p0;
    pp[i] = src[(size_t)(ok ? ch : 0) * p.groups];
    cnt[i] = ok ? (float)((p1 - p0) * cpg) : 0.f;
  }
  const float K = __shfl(pp[0].x, threadIdx.x & ~7);          // chunk 0 of this group (always populated)
  float n = 0.f, sd = 0.f;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) { n += cnt[i]; sd += cnt[i] * (pp[i].x - K); }
  n = sum8_fixed(n);
  sd = sum8_fixed(sd);
  const float dbar = n > 0.f ? sd / n : 0.f;
  float m2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const float d = (pp[i].x - K) - dbar;
    m2 += cnt[i] > 0.f ? pp[i].y + cnt[i] * d * d : 0.f;
  }
  m2 = sum8_fixed(m2);
  if (g < p.groups && sub == 0) {
    const float var = n > 0.f ? m2 / n : 0.f;
    out[g * 2 + 0] = K + dbar;
    out[g * 2 + 1] = 1.0f / sqrtf(var + p.eps);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const GNParams p) {
  constexpr int PC = Chunk<T>::N;
  const int C = p.C0 + p.C1;
  const int cpg = p.cpg;                       // (host-prepared, with the divisors: see GNParams)
  const int nvec = C / PC;
  const int VX = p.vx;
  const int TY = p.ty;
  const int tid = threadIdx.x;
  const int ty = fd_div(tid, p.fd_vx), tx = tid - ty * VX;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int per = p.per;
  const int p0 = chunk * per;
  const int p1 = min(p.HW, p0 + per);

  __shared__ float red[kMaxIter][256][4];   // per thread: {mean, M2} of its low / high group part
  GNSTAMP(0)

  const int niter = (nvec + VX - 1) / VX;
  for (int it = 0; it < niter; ++it) {
    const int v = tx + it * VX;
    // shifted sums: the shift K is the thread's first sample of each part, so the fp32 accumulators only ever see
    // deviations of the order of the standard deviation (no cancellation against a large mean)
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f, k0 = 0.f, k1 = 0.f;
    int npix = 0, split = PC;
    if (ty < TY && v < nvec) {
      const int c0 = v * PC;
      const int g0 = fd_div(c0, p.fd_cpg);
      split = min(PC, (g0 + 1) * cpg - c0);  // elements [0,split) -> g0, rest -> g0+1
      int pix = p0 + ty;
      // Every trip requests all of its pixels before the first use (eight, then four, then the last one to three together): a thread's
      // ~11 pixels of a 64-pixel chunk are two round trips to the L2.  (Round 5: the earlier form - one load for the shift, trips of
      // four, then a rolled one-load-per-trip tail - was six dependent trips.)  The shift K is the first sample, taken inside the
      // first trip; the accumulation order is the pixel order, as before.
      bool have_k = false;
      auto take = [&](const uint4& raw) __attribute__((always_inline)) {
        float f[PC];
        Chunk<T>::unpack(raw, f);
        if (!have_k) {
          k0 = f[0];
#pragma unroll
          for (int e = 1; e < PC; ++e) if (e == split) k1 = f[e];
          have_k = true;
        }
#pragma unroll
        for (int e = 0; e < PC; ++e) {
          if (e < split) { const float d = f[e] - k0; s0 += d; q0 += d * d; }
          else { const float d = f[e] - k1; s1 += d; q1 += d * d; }
        }
      };
      for (; pix + 7 * TY < p1; pix += 8 * TY) {
        uint4 raw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) raw[u] = load_cat<T>(p, b, pix + u * TY, v);
#pragma unroll
        for (int u = 0; u < 8; ++u) take(raw[u]);
        npix += 8;
      }
      for (; pix + 3 * TY < p1; pix += 4 * TY) {
        uint4 raw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) raw[u] = load_cat<T>(p, b, pix + u * TY, v);
#pragma unroll
        for (int u = 0; u < 4; ++u) take(raw[u]);
        npix += 4;
      }
      if (pix < p1) {
        uint4 raw[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) raw[u] = load_cat<T>(p, b, min(pix + u * TY, p1 - 1), v);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          if (pix + u * TY < p1) {
            take(raw[u]);
            ++npix;
          }
        }
      }
    }
    const float n0 = (float)(npix * split), n1 = (float)(npix * (PC - split));
    const float ms0 = n0 > 0.f ? s0 / n0 : 0.f, ms1 = n1 > 0.f ? s1 / n1 : 0.f;
    red[it][tid][0] = k0 + ms0; red[it][tid][1] = fmaxf(q0 - s0 * ms0, 0.f);
    red[it][tid][2] = k1 + ms1; red[it][tid][3] = fmaxf(q1 - s1 * ms1, 0.f);
  }
  GNSTAMP(1)
  __syncthreads();
  GNSTAMP(2)
  // fixed-order combination, 8 lanes per group (lane `sub` takes the thread rows y = sub, sub+8, ...), no integer division
  // in the loops (a runtime divide is a ~40-instruction sequence: the one-thread-per-group form of this block, with three
  // of them per pair, took 3.8 us - as long as the loads).  Deltas against one pair's mean as in gn_reduce_stats.
  {
    const int g = tid >> 3, sub = tid & 7;
    const bool gv = g < p.groups;
    const int c_lo = (gv ? g : 0) * cpg;
    const int v_first = max(0, c_lo / PC - 1);                    // (PC is a compile-time power of two)
    const int v_last = min(nvec - 1, (c_lo + cpg - 1) / PC);
    const int npx = p1 - p0;
    const int np_base = npx / TY, np_rem = npx - np_base * TY;   // (one runtime division per workgroup, off the load path)    // thread row y holds np_base + (y < np_rem) pixels
    auto locate = [&](int v, int& it, int& x, int& part, int& cnt) __attribute__((always_inline)) {
      const int c0 = v * PC;
      it = 0; x = v;
      while (x >= VX) { x -= VX; ++it; }
      if (c0 >= c_lo && c0 < c_lo + cpg) { part = 0; cnt = min(PC, c_lo + cpg - c0); }
      else if (c0 < c_lo && c0 + PC > c_lo) { part = 2; cnt = c0 + PC - c_lo; }
      else { part = 0; cnt = 0; }
    };
    float K;
    {
      int it, x, part, cnt;
      locate(c_lo / PC, it, x, part, cnt);                         // the vector that holds channel c_lo, thread row 0
      K = red[it][x][part];
    }
    float n = 0.f, sd = 0.f, m2 = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
      float dbar = 0.f;
      if (pass) { n = sum8_fixed(n); sd = sum8_fixed(sd); dbar = n > 0.f ? sd / n : 0.f; }
      for (int v = v_first; v <= v_last; ++v) {
        int it, x, part, cnt;
        locate(v, it, x, part, cnt);
        if (cnt == 0) continue;
        for (int y = sub; y < TY; y += 8) {
          const int np = np_base + (y < np_rem ? 1 : 0);
          if (np == 0) continue;
          const float2 e = *(const float2*)&red[it][y * VX + x][part];
          const float c = (float)(np * cnt);
          if (!pass) { n += c; sd += c * (e.x - K); }
          else { const float d = (e.x - K) - dbar; m2 += e.y + c * d * d; }
        }
      }
      if (pass) {
        m2 = sum8_fixed(m2);
        if (gv && sub == 0) {
          float* dst = p.partial + (((size_t)b * p.nchunk + chunk) * p.groups + g) * 2;
          dst[0] = K + dbar;
          dst[1] = m2;
        }
      }
    }
  }
  GNSTAMP(3)
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const GNParams p) {
  constexpr int PC = Chunk<T>::N;
  const int C = p.C0 + p.C1;
  const int cpg = p.cpg;
  const int nvec = C / PC;
  const int b = blockIdx.y;
  __shared__ float s_stat[64][2];
  GNSTAMP(4)
  // every thread owns fixed channel vectors, so the per-channel affine (x*a + b with
  // a = rstd*gamma, b = beta - mean*a) is computed once and the pixel loop is one FMA per element
  const int VX = p.vx;
  const int TY = p.ty;
  const int tid = threadIdx.x;
  const int ty = fd_div(tid, p.fd_vx), tx = tid - ty * VX;
  const bool act = ty < TY;
  const int per = (p.HW + gridDim.x - 1) / gridDim.x;    // (wave-uniform: one scalar-side division)
  const int p0 = blockIdx.x * per;
  const int p1 = min(p.HW, p0 + per);
  const T* src = nullptr;
  T* dst = nullptr;
  int cs = 0;
  auto locate = [&](int v) __attribute__((always_inline)) {
    const int c0 = v * PC;
    int coff;
    if (c0 < p.C0) { src = (const T*)p.src0; cs = p.C0; coff = c0; }
    else { src = (const T*)p.src1; cs = p.C1; coff = c0 - p.C0; }
    src += (size_t)b * p.HW * cs + coff;
    dst = (T*)p.out + (size_t)b * p.HW * C + c0;
  };
  // Everything that does not depend on the statistics - the first four pixels of this thread's vector, its gamma / beta -
  // is requested BEFORE the reduction of the partials, so that the reduction's own memory round trip and shuffle chain
  // (about 4 us per workgroup by s_memtime stamps, as long as streaming the workgroup's ~11 pixels per thread) overlap
  // with the first data round trip instead of preceding it.
  uint4 raw0[4];
  float gam[PC], bet[PC];
  if (act) {
    locate(tx);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pix = p0 + ty + u * TY;
      raw0[u] = pix < p1 ? *(const uint4*)(src + (size_t)pix * cs) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < PC; ++e) { gam[e] = p.gamma[tx * PC + e]; bet[e] = p.beta[tx * PC + e]; }
  }
  // every apply workgroup reduces the image's partials itself.  (Having the last-finishing partial workgroup finalise
  // (mean, rstd) behind a ticket needs agent-scope fences, which on this 8-XCD part write back / invalidate whole
  // L2s: measured 2.4x slower for the GroupNorm family.)
  gn_reduce_stats(p, b, cpg, &s_stat[0][0]);
  __syncthreads();
  GNSTAMP(5)
  if (!act) return;
  for (int v = tx; v < nvec; v += VX) {
    const int c0 = v * PC;
    if (v != tx) {                                     // (only tensors with more than 256 vectors per pixel come here)
      locate(v);
#pragma unroll
      for (int e = 0; e < PC; ++e) { gam[e] = p.gamma[c0 + e]; bet[e] = p.beta[c0 + e]; }
    }
    // a vector spans at most two groups (run_gn checks it): one runtime division per vector, not one per element
    const int g0 = fd_div(c0, p.fd_cpg);
    const int split = min(PC, (g0 + 1) * cpg - c0);
    const float m_lo = s_stat[g0][0], r_lo = s_stat[g0][1];
    const float m_hi = s_stat[split < PC ? g0 + 1 : g0][0], r_hi = s_stat[split < PC ? g0 + 1 : g0][1];
    float a[PC], bb[PC];
#pragma unroll
    for (int e = 0; e < PC; ++e) {
      a[e] = (e < split ? r_lo : r_hi) * gam[e];
      bb[e] = bet[e] - (e < split ? m_lo : m_hi) * a[e];
    }
    auto emit = [&](const uint4& raw, int pix) __attribute__((always_inline)) {
      float f[PC];
      Chunk<T>::unpack(raw, f);
#pragma unroll
      for (int e = 0; e < PC; ++e) {
        float y = f[e] * a[e] + bb[e];
        if (p.silu) y = silu_f(y);
        f[e] = y;
      }
      *(uint4*)(dst + (size_t)pix * C) = Chunk<T>::pack(f);
    };
    int pix = p0 + ty;
    if (v == tx) {                                     // the trip that was requested up front
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (pix + u * TY < p1) emit(raw0[u], pix + u * TY);
      pix += 4 * TY;
    }
    for (; pix + 3 * TY < p1; pix += 4 * TY) {          // four independent 16-B loads in flight per thread
      uint4 raw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) raw[u] = *(const uint4*)(src + (size_t)(pix + u * TY) * cs);
#pragma unroll
      for (int u = 0; u < 4; ++u) emit(raw[u], pix + u * TY);
    }
    for (; pix < p1; pix += TY) emit(*(const uint4*)(src + (size_t)pix * cs), pix);
  }
  GNSTAMP(6)
}

__device__ __forceinline__ float wave_sum(float v) { return wave64_sum(v); }

// Each wave keeps gamma/beta of its channel vectors in registers and walks rows in batches of RPW
// (all RPW row loads are issued before the first reduction, so several KB per wave are in flight);
// C*sizeof(T)/16 <= 64*VPL chunks per row.
// STATS: only the row statistics are wanted (LayerNorm folded into the consuming GEMM, see IgemmParams::rowstats):
// y, gamma, beta are unused and (mean, rstd) go to stats[row] - half the traffic of the normalising form.
template <typename T, int VPL, int RPW, bool STATS = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* x, T* y, const float* gamma, const float* beta,
                                                        int M, int C, float eps, int silu, float* stats = nullptr) {
  constexpr int PC = Chunk<T>::N;
  const int lane = threadIdx.x & 63;
  const int nvec = C / PC;
  float g[VPL][PC], bt[VPL][PC];
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int v = lane + k * 64;
#pragma unroll
    for (int e = 0; e < PC; ++e) {
      g[k][e] = (!STATS && v < nvec) ? gamma[v * PC + e] : 0.f;
      bt[k][e] = (!STATS && v < nvec) ? beta[v * PC + e] : 0.f;
    }
  }
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  const float invC = 1.0f / (float)C;
  for (int row0 = wave_global * RPW; row0 < M; row0 += nwaves * RPW) {
    uint4 raw[RPW][VPL];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
      for (int k = 0; k < VPL; ++k) {
        const int v = lane + k * 64;
        raw[r][k] = (v < nvec && row0 + r < M) ? *(const uint4*)(x + (size_t)(row0 + r) * C + v * PC) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      if (row0 + r >= M) break;
      float f[VPL][PC];
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < VPL; ++k) {
        Chunk<T>::unpack(raw[r][k], f[k]);
#pragma unroll
        for (int e = 0; e < PC; ++e) s += f[k][e];      // padding vectors are zero
      }
      const float mean = wave_sum(s) * invC;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < VPL; ++k) {
        if (lane + k * 64 < nvec) {
#pragma unroll
          for (int e = 0; e < PC; ++e) { const float d = f[k][e] - mean; q += d * d; }
        }
      }
      const float rstd = 1.0f / sqrtf(wave_sum(q) * invC + eps);
      if constexpr (STATS) {
        if (lane == 0) *(float2*)(stats + (size_t)(row0 + r) * 2) = make_float2(mean, rstd);
        continue;
      }
#pragma unroll
      for (int k = 0; k < VPL; ++k) {
        const int v = lane + k * 64;
        if (v < nvec) {
#pragma unroll
          for (int e = 0; e < PC; ++e) {
            float o = (f[k][e] - mean) * rstd * g[k][e] + bt[k][e];
            if (silu) o = silu_f(o);
            f[k][e] = o;
          }
          *(uint4*)(y + (size_t)(row0 + r) * C + v * PC) = Chunk<T>::pack(f[k]);
        }
      }
    }
  }
}

// LayerNorm statistics of the folded norms (transformer norm1 / norm3, C = 320 / 640 / 1280): LPR = 8 / 16 / 32 lanes share a
// row (5 vectors per lane), so a wave reduces 8 / 4 / 2 rows with ONE butterfly instead of one row with a full 64-lane one, no
// lane idles on the 40-vector rows, and 2 batches of rows are in flight per wave.  Two passes over the registers (mean, then
// centred second moment), like layernorm_kernel.  Measured at B = 8: 187 -> 172 us per forward over the 32 launches.
template <int LPR>
__device__ __forceinline__ float lanes_sum(float v) {
  v += dpp_f<0xB1>(v);          // xor 1
  v += dpp_f<0x4E>(v);          // xor 2
  v += dpp_f<0x141>(v);         // mirror within 8
  if constexpr (LPR >= 16) v += dpp_f<0x140>(v);   // mirror within 16
  if constexpr (LPR >= 32) v = xor16_sum(v);
  if constexpr (LPR >= 64) v = xor32_sum(v);
  return v;
}
template <typename T, int LPR, int RPW>
__global__ __launch_bounds__(256) void rowstats_kernel(const T* __restrict__ x, float* __restrict__ stats, int M, int C, float eps) {
  constexpr int PC = Chunk<T>::N, VPL = 5, RW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int sub = lane & (LPR - 1), rsel = lane / LPR;
  const int nvec = C / PC;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  const float invC = 1.0f / (float)C;
  for (int row0 = wave_global * (RW * RPW); row0 < M; row0 += nwaves * (RW * RPW)) {
    uint4 raw[RPW][VPL];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int row = row0 + r * RW + rsel;
#pragma unroll
      for (int k = 0; k < VPL; ++k) {
        const int v = sub + k * LPR;
        raw[r][k] = (v < nvec && row < M) ? *(const uint4*)(x + (size_t)row * C + v * PC) : make_uint4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int row = row0 + r * RW + rsel;
      float f[VPL][PC];
      float sm = 0.f;
#pragma unroll
      for (int k = 0; k < VPL; ++k) {
        Chunk<T>::unpack(raw[r][k], f[k]);
#pragma unroll
        for (int e = 0; e < PC; ++e) sm += f[k][e];        // absent vectors are zero
      }
      const float mean = lanes_sum<LPR>(sm) * invC;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < VPL; ++k) {
        if (sub + k * LPR < nvec) {
#pragma unroll
          for (int e = 0; e < PC; ++e) { const float d = f[k][e] - mean; q += d * d; }
        }
      }
      const float rstd = 1.0f / sqrtf(lanes_sum<LPR>(q) * invC + eps);
      if (sub == 0 && row < M) *(float2*)(stats + (size_t)row * 2) = make_float2(mean, rstd);
    }
  }
}

// Small feature maps (8x8 .. 32x32): one workgroup per (image, group) keeps the whole group in
// registers - statistics, normalisation and the store in ONE launch instead of two dependent ones
// (these tensors are a few hundred KB; the two-kernel path is pure launch latency there).
// Elements are handled in 4-byte units U (2 bf16 / 1 f32); cpg is even so a group starts unit-aligned.
template <typename T, int MAXU>
__global__ __launch_bounds__(256) void gn_small_kernel(const GNParams p) {
  constexpr int EPU = 4 / (int)sizeof(T);          // elements per unit
  const int C = p.C0 + p.C1;
  const int cpg = p.cpg;
  const int upg = cpg / EPU;                        // units per pixel of this group (EPU is a power of two)
  const int g = blockIdx.x, b = blockIdx.y;
  const int nunits = p.HW * upg;
  const int c_first = g * cpg;
  float v[MAXU][EPU];
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int k = 0; k < MAXU; ++k) {
    const int u = threadIdx.x + k * 256;
    if (u < nunits) {
      const int pix = fd_div(u, p.fd_aux), j = u - pix * upg;
      const int c = c_first + j * EPU;
      const T* src = (c < p.C0) ? (const T*)p.src0 + ((size_t)b * p.HW + pix) * p.C0 + c
                                : (const T*)p.src1 + ((size_t)b * p.HW + pix) * p.C1 + (c - p.C0);
      const uint32_t raw = *(const uint32_t*)src;
      if constexpr (EPU == 2) { v[k][0] = bits_f32(raw << 16); v[k][1] = bits_f32(raw & 0xffff0000u); }
      else { v[k][0] = bits_f32(raw); }
#pragma unroll
      for (int e = 0; e < EPU; ++e) s += v[k][e];
    }
  }
  // two passes over the register-resident values: mean first, then the centred second moment (torch's arithmetic)
  __shared__ float red[2][4];
  s = wave64_sum(s);
  if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = s;
  __syncthreads();
  const float mean = (float)(((double)red[0][0] + (double)red[0][1] + (double)red[0][2] + (double)red[0][3]) * p.inv_n);
  q = 0.f;
#pragma unroll
  for (int k = 0; k < MAXU; ++k) {
    if (threadIdx.x + k * 256 < nunits) {
#pragma unroll
      for (int e = 0; e < EPU; ++e) { const float d = v[k][e] - mean; q += d * d; }
    }
  }
  q = wave64_sum(q);
  if ((threadIdx.x & 63) == 0) red[1][threadIdx.x >> 6] = q;
  __syncthreads();
  const double var = ((double)red[1][0] + (double)red[1][1] + (double)red[1][2] + (double)red[1][3]) * p.inv_n;
  const float rstd = 1.0f / sqrtf((float)var + p.eps);
#pragma unroll
  for (int k = 0; k < MAXU; ++k) {
    const int u = threadIdx.x + k * 256;
    if (u < nunits) {
      const int pix = fd_div(u, p.fd_aux), j = u - pix * upg;
      const int c = c_first + j * EPU;
      float y[EPU];
#pragma unroll
      for (int e = 0; e < EPU; ++e) {
        const float a = rstd * p.gamma[c + e];
        y[e] = v[k][e] * a + (p.beta[c + e] - mean * a);
        if (p.silu) y[e] = silu_f(y[e]);
      }
      T* dst = (T*)p.out + ((size_t)b * p.HW + pix) * C + c;
      if constexpr (EPU == 2) *(uint32_t*)dst = pack_bf16x2(y[0], y[1]);
      else *(float*)dst = y[0];
    }
  }
}

// Feature maps up to 32x32: one workgroup per (image, block of GB = 1 or 2 groups whose channels make whole 16-byte
// vectors), the block's HW x GB*cpg elements held in registers as 16-byte vectors: statistics, normalisation and the
// store in ONE launch that reads the tensor once (the two-launch path reads it twice and pays two dependent launches;
// at these sizes - 0.3 to 5 MB - that is most of its time).  Thread t owns vector j = t % vpp of pixels t / vpp + PPI*k,
// so its channels, groups and gamma/beta are fixed and all MAXV loads are issued before the first use.
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void gn_fused_kernel(const GNParams p, int GB) {
  constexpr int PC = Chunk<T>::N;
  const int C = p.C0 + p.C1;
  const int cpg = p.cpg;
  const int vpp = GB * cpg / PC;                    // vectors per pixel of this group block (PC is a power of two)
  const int ppi = p.ty;                             // pixels per trip = 256 / vpp (host)
  const int b = blockIdx.y;
  const int cfirst = blockIdx.x * GB * cpg;
  const int tid = threadIdx.x;
  const int pr = fd_div(tid, p.fd_aux), j = tid - pr * vpp;
  const bool active = pr < ppi;
  const int c0 = cfirst + j * PC;                   // first channel of this thread's vector
  const int glo = (j * PC) >= cpg ? 1 : 0;          // block-local group of its first element (a block holds 1 or 2 groups)
  const int split = min(PC, (glo + 1) * cpg - j * PC);   // elements [0,split) -> glo, the rest -> glo+1
  const T* src;
  int cs;
  if (c0 < p.C0) { src = (const T*)p.src0 + (size_t)b * p.HW * p.C0 + c0; cs = p.C0; }
  else { src = (const T*)p.src1 + (size_t)b * p.HW * p.C1 + (c0 - p.C0); cs = p.C1; }
  u32x4 raw[MAXV];
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int pix = pr + ppi * k;
    raw[k] = (active && pix < p.HW) ? *(const u32x4*)(src + (size_t)pix * cs) : u32x4{0u, 0u, 0u, 0u};
  }
  // pass 1: group sums -> means; pass 2: centred second moments of the register-resident values
  float slo = 0.f, shi = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
#pragma unroll
    for (int e = 0; e < PC; ++e) {
      const float f = to_f32<T>(chunk_elem<T>(raw[k], e));     // out-of-range vectors are zero: they add nothing
      if (e < split) slo += f;
      else shi += f;
    }
  }
  // block-local group sums: group 0 gets the low part of glo == 0 vectors, group 1 the rest
  __shared__ float red[4][4];
  {
    float v0 = wave64_sum(glo == 0 ? slo : 0.f), v1 = wave64_sum(glo == 0 ? shi : slo);
    if ((tid & 63) == 0) { red[0][tid >> 6] = v0; red[1][tid >> 6] = v1; }
  }
  __syncthreads();
  float mean[2], rstd[2];
#pragma unroll
  for (int g = 0; g < 2; ++g)
    mean[g] = (float)(((double)red[g][0] + (double)red[g][1] + (double)red[g][2] + (double)red[g][3]) * p.inv_n);
  {
    const float mlo = glo == 0 ? mean[0] : mean[1], mhi = mean[1];
    float qlo = 0.f, qhi = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      if (active && pr + ppi * k < p.HW) {
#pragma unroll
        for (int e = 0; e < PC; ++e) {
          const float f = to_f32<T>(chunk_elem<T>(raw[k], e));
          if (e < split) { const float d = f - mlo; qlo += d * d; }
          else { const float d = f - mhi; qhi += d * d; }
        }
      }
    }
    float v0 = wave64_sum(glo == 0 ? qlo : 0.f), v1 = wave64_sum(glo == 0 ? qhi : qlo);
    if ((tid & 63) == 0) { red[2][tid >> 6] = v0; red[3][tid >> 6] = v1; }
  }
  __syncthreads();
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const double var = ((double)red[2 + g][0] + (double)red[2 + g][1] + (double)red[2 + g][2] + (double)red[2 + g][3]) * p.inv_n;
    rstd[g] = 1.0f / sqrtf((float)var + p.eps);
  }
  if (!active) return;
  float a[PC], bb[PC];
#pragma unroll
  for (int e = 0; e < PC; ++e) {
    const int g = (e < split) ? glo : glo + 1;
    a[e] = (g == 0 ? rstd[0] : rstd[1]) * p.gamma[c0 + e];
    bb[e] = p.beta[c0 + e] - (g == 0 ? mean[0] : mean[1]) * a[e];
  }
  T* dst = (T*)p.out + (size_t)b * p.HW * C + c0;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int pix = pr + ppi * k;
    if (pix < p.HW) {
      float f[PC];
#pragma unroll
      for (int e = 0; e < PC; ++e) {
        float y = to_f32<T>(chunk_elem<T>(raw[k], e)) * a[e] + bb[e];
        if (p.silu) y = silu_f(y);
        f[e] = y;
      }
      *(uint4*)(dst + (size_t)pix * C) = Chunk<T>::pack(f);
    }
  }
}

// gn_fused_kernel with the dependent chain cut down - at these sizes (1 .. 10 MB) the kernel is a latency chain, not a
// stream: [loads] -> reduce -> barrier -> reduce -> barrier -> [gamma / beta loads] -> store became
// [loads of data, gamma, beta and the groups' shift samples, all issued together] -> ONE reduction -> barrier -> store.
// The statistics are shifted sums around a sample of the group itself (its first element): sum(x - K) and sum((x - K)^2)
// in one pass over the registers; with K a member of the group |mean - K| is a few standard deviations at most, so
// var = E[(x-K)^2] - E[x-K]^2 loses nothing to cancellation (a mean of 1000 with deviation 1 is a test case).  MAXV is a
// template parameter chosen by the launcher (2 / 6 / 12 / 22): the register loops carry no dead iterations.
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void gn_one_kernel(const GNParams p, int GB) {
  constexpr int PC = Chunk<T>::N;
  const int C = p.C0 + p.C1;
  const int cpg = p.cpg;
  const int vpp = GB * cpg / PC;
  const int ppi = p.ty;
  // consecutive workgroup ids go to consecutive XCDs: image = id % B keeps the group blocks of one image - whose 80 .. 320-byte
  // runs share 128-byte lines with their neighbours' - on one XCD (one L2) when B is a multiple of 8
  const int b = (int)(blockIdx.x % (unsigned)p.B);
  const int cfirst = (int)(blockIdx.x / (unsigned)p.B) * GB * cpg;
  const int tid = threadIdx.x;
  const int pr = fd_div(tid, p.fd_aux), j = tid - pr * vpp;
  const bool active = pr < ppi;
  const int c0 = cfirst + j * PC;
  const int glo = (j * PC) >= cpg ? 1 : 0;
  const int split = min(PC, (glo + 1) * cpg - j * PC);
  const T* src;
  int cs;
  if (c0 < p.C0) { src = (const T*)p.src0 + (size_t)b * p.HW * p.C0 + c0; cs = p.C0; }
  else { src = (const T*)p.src1 + (size_t)b * p.HW * p.C1 + (c0 - p.C0); cs = p.C1; }
  u32x4 raw[MAXV];
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int pix = pr + ppi * k;
    raw[k] = (active && pix < p.HW) ? *(const u32x4*)(src + (size_t)pix * cs) : u32x4{0u, 0u, 0u, 0u};
  }
  // shift samples: pixel 0 of the first channel of the block's group(s) - the same two addresses for every thread
  float K[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int c = cfirst + (g < GB ? g : 0) * cpg;
    const T* sp = c < p.C0 ? (const T*)p.src0 + (size_t)b * p.HW * p.C0 + c : (const T*)p.src1 + (size_t)b * p.HW * p.C1 + (c - p.C0);
    K[g] = to_f32<T>(*sp);
  }
  // affine parameters of this thread's channels: requested now, used after the barrier
  float ga[PC], be[PC];
  if (active) {
#pragma unroll
    for (int e = 0; e < PC; e += 4) {
      const f32x4 g4 = *(const f32x4*)(p.gamma + c0 + e), b4 = *(const f32x4*)(p.beta + c0 + e);
#pragma unroll
      for (int r = 0; r < 4; ++r) { ga[e + r] = g4[r]; be[e + r] = b4[r]; }
    }
  }
  const float klo = glo == 0 ? K[0] : K[1], khi = K[1];
  float slo = 0.f, shi = 0.f, qlo = 0.f, qhi = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    if (active && pr + ppi * k < p.HW) {
#pragma unroll
      for (int e = 0; e < PC; ++e) {
        const float f = to_f32<T>(chunk_elem<T>(raw[k], e));
        if (e < split) { const float d = f - klo; slo += d; qlo += d * d; }
        else { const float d = f - khi; shi += d; qhi += d * d; }
      }
    }
  }
  __shared__ float red[4][4];     // [sum g0, sum g1, sq g0, sq g1][wave]
  {
    const float s0 = wave64_sum(glo == 0 ? slo : 0.f), s1 = wave64_sum(glo == 0 ? shi : slo);
    const float q0 = wave64_sum(glo == 0 ? qlo : 0.f), q1 = wave64_sum(glo == 0 ? qhi : qlo);
    if ((tid & 63) == 0) { red[0][tid >> 6] = s0; red[1][tid >> 6] = s1; red[2][tid >> 6] = q0; red[3][tid >> 6] = q1; }
  }
  __syncthreads();
  float mean[2], var[2], rstd[2];
  const float inv_n = (float)p.inv_n;
  bool far = false;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const float ms = (((red[g][0] + red[g][1]) + red[g][2]) + red[g][3]) * inv_n;           // E[x - K]
    const float m2 = (((red[2 + g][0] + red[2 + g][1]) + red[2 + g][2]) + red[2 + g][3]) * inv_n;   // E[(x - K)^2]
    mean[g] = K[g] + ms;
    var[g] = fmaxf(m2 - ms * ms, 0.f);
    far |= ms * ms > 64.f * var[g];      // the sample was an outlier of its group (same verdict in every thread)
  }
  if (far) {                             // rare: second moment again, centred on the mean that is known now
    const float mlo = glo == 0 ? mean[0] : mean[1], mhi = mean[1];
    qlo = 0.f; qhi = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      if (active && pr + ppi * k < p.HW) {
#pragma unroll
        for (int e = 0; e < PC; ++e) {
          const float f = to_f32<T>(chunk_elem<T>(raw[k], e));
          if (e < split) { const float d = f - mlo; qlo += d * d; }
          else { const float d = f - mhi; qhi += d * d; }
        }
      }
    }
    const float q0 = wave64_sum(glo == 0 ? qlo : 0.f), q1 = wave64_sum(glo == 0 ? qhi : qlo);
    __syncthreads();                     // everybody has read the first round's records
    if ((tid & 63) == 0) { red[2][tid >> 6] = q0; red[3][tid >> 6] = q1; }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 2; ++g) var[g] = (((red[2 + g][0] + red[2 + g][1]) + red[2 + g][2]) + red[2 + g][3]) * inv_n;
  }
  if (!active) return;
#pragma unroll
  for (int g = 0; g < 2; ++g) rstd[g] = 1.0f / sqrtf(var[g] + p.eps);
  float a[PC], bb[PC];
#pragma unroll
  for (int e = 0; e < PC; ++e) {
    const int g = (e < split) ? glo : glo + 1;
    a[e] = (g == 0 ? rstd[0] : rstd[1]) * ga[e];
    bb[e] = be[e] - (g == 0 ? mean[0] : mean[1]) * a[e];
  }
  T* dst = (T*)p.out + (size_t)b * p.HW * C + c0;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int pix = pr + ppi * k;
    if (pix < p.HW) {
      float f[PC];
#pragma unroll
      for (int e = 0; e < PC; ++e) {
        float y = to_f32<T>(chunk_elem<T>(raw[k], e)) * a[e] + bb[e];
        if (p.silu) y = silu_f(y);
        f[e] = y;
      }
      *(uint4*)(dst + (size_t)pix * C) = Chunk<T>::pack(f);
    }
  }
}

// Split-K finish + GroupNorm (+SiLU) in one launch (launch_finish_groupnorm): gn_one_kernel whose "load" is the sum of the
// K-slice slabs of the producing conv plus its bias and time-embedding row.  On the 8x8 / 16x16 maps the finish kernel and the
// norm were two latency chains of 6 - 9 us each around a tensor of 1 - 5 MB that nobody else reads.  Values stay fp32 from the
// accumulators to the normalisation (the unfused path rounds them to the storage type in between).  The shift sample of the
// one-pass sums is bias + time-embedding value of the group's first channel (no group member is known before the sums);
// the same outlier fallback as gn_one_kernel covers the case where that is far from the group's mean.
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void finish_gn_kernel(const GNParams p, int GB, const float* __restrict__ partial, int splits,
                                                        int N, long slab, const float* __restrict__ bias,
                                                        const float* __restrict__ rowbias, int rb_stride) {
  constexpr int PC = Chunk<T>::N;
  const int C = p.C0;
  const int cpg = p.cpg;
  const int vpp = GB * cpg / PC;
  const int ppi = p.ty;
  // consecutive workgroup ids go to consecutive XCDs: image = id % B keeps the group blocks of one image - whose 80 .. 320-byte
  // runs share 128-byte lines with their neighbours' - on one XCD (one L2) when B is a multiple of 8
  const int b = (int)(blockIdx.x % (unsigned)p.B);
  const int cfirst = (int)(blockIdx.x / (unsigned)p.B) * GB * cpg;
  const int tid = threadIdx.x;
  const int pr = fd_div(tid, p.fd_aux), j = tid - pr * vpp;
  const bool active = pr < ppi;
  const int c0 = cfirst + j * PC;
  const int glo = (j * PC) >= cpg ? 1 : 0;
  const int split = min(PC, (glo + 1) * cpg - j * PC);
  float v[MAXV][PC];
  const float* src = partial + (size_t)b * p.HW * N + c0;
  // the slices of ZU K slices are requested together: one round trip per ZU slices instead of one per slice (a plain
  // `for z: acc += load` waits for every load before it issues the next - 8 slices were 8 dependent trips to the MALL)
  constexpr int ZU = MAXV <= 2 ? 4 : 2;
#pragma unroll
  for (int k = 0; k < MAXV; ++k)
#pragma unroll
    for (int e = 0; e < PC; ++e) v[k][e] = 0.f;
  for (int z0 = 0; z0 < splits; z0 += ZU) {
    f32x4 t[ZU][MAXV][PC / 4];
#pragma unroll
    for (int zi = 0; zi < ZU; ++zi) {
      const bool zok = z0 + zi < splits;                 // wave-uniform
#pragma unroll
      for (int k = 0; k < MAXV; ++k) {
        const int pix = pr + ppi * k;
        const bool ok = zok && active && pix < p.HW;
#pragma unroll
        for (int e = 0; e < PC; e += 4)
          t[zi][k][e / 4] = ok ? *(const f32x4*)(src + (size_t)(z0 + zi) * slab + (size_t)pix * N + e) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int zi = 0; zi < ZU; ++zi)
#pragma unroll
      for (int k = 0; k < MAXV; ++k)
#pragma unroll
        for (int e = 0; e < PC; ++e) v[k][e] += t[zi][k][e / 4][e & 3];
  }
  const float* rbp = rowbias ? rowbias + (size_t)b * rb_stride : nullptr;
  float K[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int c = cfirst + (g < GB ? g : 0) * cpg;
    K[g] = bias[c] + (rbp ? rbp[c] : 0.f);
  }
  float ga[PC], be[PC], add[PC];
#pragma unroll
  for (int e = 0; e < PC; ++e) { ga[e] = 0.f; be[e] = 0.f; add[e] = 0.f; }
  if (active) {
#pragma unroll
    for (int e = 0; e < PC; e += 4) {
      const f32x4 g4 = *(const f32x4*)(p.gamma + c0 + e), b4 = *(const f32x4*)(p.beta + c0 + e);
      f32x4 a4 = *(const f32x4*)(bias + c0 + e);
      if (rbp) a4 += *(const f32x4*)(rbp + c0 + e);
#pragma unroll
      for (int r = 0; r < 4; ++r) { ga[e + r] = g4[r]; be[e + r] = b4[r]; add[e + r] = a4[r]; }
    }
  }
  const float klo = glo == 0 ? K[0] : K[1], khi = K[1];
  float slo = 0.f, shi = 0.f, qlo = 0.f, qhi = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const bool ok = active && pr + ppi * k < p.HW;
#pragma unroll
    for (int e = 0; e < PC; ++e) {
      v[k][e] += add[e];
      if (ok) {
        if (e < split) { const float d = v[k][e] - klo; slo += d; qlo += d * d; }
        else { const float d = v[k][e] - khi; shi += d; qhi += d * d; }
      }
    }
  }
  __shared__ float red[4][4];
  {
    const float s0 = wave64_sum(glo == 0 ? slo : 0.f), s1 = wave64_sum(glo == 0 ? shi : slo);
    const float q0 = wave64_sum(glo == 0 ? qlo : 0.f), q1 = wave64_sum(glo == 0 ? qhi : qlo);
    if ((tid & 63) == 0) { red[0][tid >> 6] = s0; red[1][tid >> 6] = s1; red[2][tid >> 6] = q0; red[3][tid >> 6] = q1; }
  }
  __syncthreads();
  float mean[2], var[2], rstd[2];
  const float inv_n = (float)p.inv_n;
  bool far = false;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const float ms = (((red[g][0] + red[g][1]) + red[g][2]) + red[g][3]) * inv_n;
    const float m2 = (((red[2 + g][0] + red[2 + g][1]) + red[2 + g][2]) + red[2 + g][3]) * inv_n;
    mean[g] = K[g] + ms;
    var[g] = fmaxf(m2 - ms * ms, 0.f);
    far |= ms * ms > 64.f * var[g];
  }
  if (far) {
    const float mlo = glo == 0 ? mean[0] : mean[1], mhi = mean[1];
    qlo = 0.f; qhi = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      if (active && pr + ppi * k < p.HW) {
#pragma unroll
        for (int e = 0; e < PC; ++e) {
          if (e < split) { const float d = v[k][e] - mlo; qlo += d * d; }
          else { const float d = v[k][e] - mhi; qhi += d * d; }
        }
      }
    }
    const float q0 = wave64_sum(glo == 0 ? qlo : 0.f), q1 = wave64_sum(glo == 0 ? qhi : qlo);
    __syncthreads();
    if ((tid & 63) == 0) { red[2][tid >> 6] = q0; red[3][tid >> 6] = q1; }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 2; ++g) var[g] = (((red[2 + g][0] + red[2 + g][1]) + red[2 + g][2]) + red[2 + g][3]) * inv_n;
  }
  if (!active) return;
#pragma unroll
  for (int g = 0; g < 2; ++g) rstd[g] = 1.0f / sqrtf(var[g] + p.eps);
  float a[PC], bb[PC];
#pragma unroll
  for (int e = 0; e < PC; ++e) {
    const int g = (e < split) ? glo : glo + 1;
    a[e] = (g == 0 ? rstd[0] : rstd[1]) * ga[e];
    bb[e] = be[e] - (g == 0 ? mean[0] : mean[1]) * a[e];
  }
  T* dst = (T*)p.out + (size_t)b * p.HW * C + c0;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int pix = pr + ppi * k;
    if (pix < p.HW) {
      float f[PC];
#pragma unroll
      for (int e = 0; e < PC; ++e) {
        float y = v[k][e] * a[e] + bb[e];
        if (p.silu) y = silu_f(y);
        f[e] = y;
      }
      *(uint4*)(dst + (size_t)pix * C) = Chunk<T>::pack(f);
    }
  }
}

// The 64x64 maps in ONE pass with every CU streaming: S workgroups share one (image, block of GB groups whose channels make
// whole 16-byte vectors), each keeps its HW/S pixels x GB*cpg channels in registers (10 .. 20 vectors per thread), reduces
// them to per-group (mean, M2) - two passes over registers, torch's arithmetic -, hands the 2*GB numbers to its S-1 partners
// and, once theirs have arrived, combines all S records in the fixed order 0..S-1 (Chan's update: exact, deterministic),
// normalises and stores.  The tensor is read once and written once; the two-launch scheme (statistics pass + apply pass)
// moved three times the tensor.  A register-resident slab per (image, group block) WITHOUT the split was measured first and
// lost (tools/experiments/r03_groupnorm_register_slab_kernel.patch): 64 .. 128 workgroups cannot pull 42 MB fast enough, a
// CU gets ~27 GB/s from beyond its L2 whatever it keeps in flight.
// Hand-off (MI355X_MICROARCH.md, "data-tagged granules"): a record entry is ONE naturally aligned 8-byte {value, tag} written
// by a relaxed agent-scope atomic store (global_store_dwordx2 sc1: write-through, visible across XCDs) and polled with
// relaxed agent-scope loads until the tag equals this launch's generation - no flag, no fence, nothing to reset between
// launches.
// Generation tag (round 4): drawn ON THE DEVICE from a per-slab ticket counter - every workgroup of a launch adds 8 / S, so
// the S workgroups of one launch see tickets inside one block of eight and agree on ticket >> 3 without talking to each
// other, and the next launch on the same region (the next norm of the stream, or the next replay of a captured graph)
// draws the next block.  The round-3 form passed a host-side counter as a kernel argument: a replayed graph re-ran with
// the tag it was captured with and met the previous replay's records.
// Progress without co-residency (round 4): the poll is bounded (GNParams::poll_ticks of the 100 MHz wall clock, ~100 us)
// and a workgroup whose partners did not show up computes THEIR records itself - it loads each missing split's pixels the
// way that split's own threads would, runs the same statistics code on them (bit-identical numbers), then reloads its own
// pixels and carries on.  Co-residency of the grid (at most one workgroup per CU) is therefore an expectation that makes
// the common case fast, not a requirement: a second stream or process holding CUs, or two such grids holding half the
// chip each, costs the extra reads of the two-launch scheme instead of dead-locking or poisoning the output (round 3
// wrote NaNs after a multi-second spin and still returned 0).  GNParams::coop_mode 1 forces that path (tests).
template <typename T, int MAXV, int NT>
__global__ __launch_bounds__(NT) void gn_coop_kernel(const GNParams p, int GB) {
  constexpr int PC = Chunk<T>::N;
  constexpr int NWV = NT / 64, MAXG = 4, MAXS = 8;
  const int C = p.C0 + p.C1;
  const int cpg = p.cpg;
  const int S = p.splits;
  const int vpp = GB * cpg / PC;                    // vectors per pixel of this group block
  const int ppi = p.ty;                             // pixels per trip = NT / vpp (host)
  int logical;
  {
    const int G = gridDim.x, bid = blockIdx.x;
    const int q = G >> 3, r = G & 7, xcd = bid & 7, idx = bid >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;      // an image's workgroups share an XCD
  }
  const int slab = fd_div(logical, p.fd_vx);        // (fd_vx: divisor S)  slab = image * nblk + group block
  const int split = logical - slab * S;
  const int nblk = p.groups / GB;
  const int b = fd_div(slab, p.fd_cpg);             // (fd_cpg: divisor nblk, set by the host for this kernel)
  const int cfirst = (slab - b * nblk) * GB * cpg;
  const int per = p.per;                            // pixels per split
  const int pix0 = split * per, pix1 = min(p.HW, pix0 + per);
  const int tid = threadIdx.x;
  const int pr = fd_div(tid, p.fd_aux), j = tid - pr * vpp;
  const bool active = pr < ppi;
  const int c0 = cfirst + j * PC;
  int glo = 0;
#pragma unroll
  for (int g = 1; g < MAXG; ++g) glo += (j * PC >= g * cpg) ? 1 : 0;
  const int split_e = min(PC, (glo + 1) * cpg - j * PC);   // elements [0,split_e) -> group glo, the rest -> glo+1
  const T* src;
  int cs;
  if (c0 < p.C0) { src = (const T*)p.src0 + (size_t)b * p.HW * p.C0 + c0; cs = p.C0; }
  else { src = (const T*)p.src1 + (size_t)b * p.HW * p.C1 + (c0 - p.C0); cs = p.C1; }
  u32x4 raw[MAXV];
  // pixels of split `sp` as that split's own threads hold them (thread -> (pixel row, vector) does not depend on the split)
  auto load_raw = [&](int sp) __attribute__((always_inline)) {
    const int q0 = sp * per, q1 = min(p.HW, q0 + per);
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int pix = q0 + pr + ppi * k;
      raw[k] = (active && pix < q1) ? *(const u32x4*)(src + (size_t)pix * cs) : u32x4{0u, 0u, 0u, 0u};
    }
  };
  load_raw(split);
  // ---- this launch's generation: one returning agent-scope add per workgroup, requested behind the pixel loads and needed
  // only when the record is published
  unsigned long long ticket = 0;
  if (tid == 0)
    ticket = __hip_atomic_fetch_add(p.sync_ctr + slab, (unsigned long long)(8 / S), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __shared__ float red[2][MAXG][NWV];
  __shared__ float s_stat[MAXG][2];
  __shared__ float s_rec[MAXS][2 * MAXG];           // (mean, M2) x group of every split, in split order
  __shared__ unsigned s_missing;
  // ---- statistics of the split whose pixels sit in raw[]: one accumulator per element position (dword in bf16), combined
  // per group afterwards: every value is consumed once, next to where it is produced.  bf16: v_dot2_f32_bf16 against (1,1)
  // sums a dword's pair without unpacking it (group boundaries fall on dwords: cpg is even) - hipcc would otherwise keep the
  // unpacked fp32 copies of pass 1 alive for passes 2 and 3 (8 registers per vector instead of 4; 227 VGPRs) and the grid
  // could not be co-resident.  Leaves the split's record value of lane `tid` (< 2 * GB: even = mean, odd = M2) in `recv`.
  auto split_stats = [&](int sp, float& recv) __attribute__((always_inline)) {
    const int q0 = sp * per, q1 = min(p.HW, q0 + per);
    constexpr int NACC = sizeof(T) == 2 ? 4 : PC;
    constexpr int EPA = PC / NACC;
    float acc1[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc1[i] = 0.f;
    if constexpr (sizeof(T) == 2) {
      typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
      const bf2 ones = __builtin_bit_cast(bf2, 0x3f803f80u);
#pragma unroll
      for (int k = 0; k < MAXV; ++k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned w = raw[k][i];                     // absent vectors are zero: they add nothing
          acc1[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, w), ones, acc1[i], false);
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < MAXV; ++k) {
#pragma unroll
        for (int e = 0; e < PC; ++e) acc1[e] += to_f32<T>(chunk_elem<T>(raw[k], e));
      }
    }
    float slo = 0.f, shi = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (i * EPA < split_e) slo += acc1[i];
      else shi += acc1[i];
    }
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
      const float v = wave64_sum(glo == g ? slo : (glo + 1 == g ? shi : 0.f));
      if ((tid & 63) == 0) red[0][g][tid >> 6] = v;
    }
    __syncthreads();
    const float n_loc = (float)((q1 - q0) * cpg);          // samples per group in this split (exact: < 2^24)
    float mlo = 0.f, mhi = 0.f, mloc[MAXG];
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
      double a = 0.0;
#pragma unroll
      for (int w = 0; w < NWV; ++w) a += (double)red[0][g][w];
      mloc[g] = n_loc > 0.f ? (float)(a / (double)n_loc) : 0.f;
      if (g == glo) mlo = mloc[g];
      if (g == glo + 1) mhi = mloc[g];
    }
    {
      float acc2[PC];
#pragma unroll
      for (int e = 0; e < PC; ++e) acc2[e] = 0.f;
#pragma unroll
      for (int k = 0; k < MAXV; ++k) {
        const bool ok = active && q0 + pr + ppi * k < q1;
#pragma unroll
        for (int e = 0; e < PC; ++e) {
          const float d = ok ? to_f32<T>(chunk_elem<T>(raw[k], e)) - (e < split_e ? mlo : mhi) : 0.f;
          acc2[e] += d * d;
        }
      }
      float qlo = 0.f, qhi = 0.f;
#pragma unroll
      for (int e = 0; e < PC; ++e) {
        if (e < split_e) qlo += acc2[e];
        else qhi += acc2[e];
      }
#pragma unroll
      for (int g = 0; g < MAXG; ++g) {
        const float v = wave64_sum(glo == g ? qlo : (glo + 1 == g ? qhi : 0.f));
        if ((tid & 63) == 0) red[1][g][tid >> 6] = v;
      }
    }
    __syncthreads();
    recv = 0.f;
    if (tid < 2 * GB) {
      const int g = tid >> 1;
      if (tid & 1) {
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) a += (double)red[1][g][w];
        recv = (float)a;                                                          // M2 of this split
      } else {
        recv = mloc[g];
      }
    }
  };
  float myrec;
  split_stats(split, myrec);
  // ---- publish this split's record, collect the partners', combine in split order
  unsigned long long* rec = p.sync + (size_t)slab * (MAXS * 2 * MAXG);      // [S][2 * MAXG] granules of this slab
  if (tid < 64) {                                     // one wave publishes and polls: lane = (split, entry)
    const unsigned tk = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ticket >> 3));
    const unsigned gen = (tk & 0x7fffffffu) + 1u;     // tag 0 = never written
    if (tid < 2 * GB) {
      const unsigned long long gran = ((unsigned long long)gen << 32) | (unsigned long long)f32_bits(myrec);
      __hip_atomic_store(rec + split * (2 * MAXG) + tid, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int sidx = tid / (2 * MAXG), ent = tid - sidx * (2 * MAXG);
    const bool mine = sidx < S && ent < 2 * GB;
    const bool own = sidx == split;
    unsigned long long gran = ((unsigned long long)gen << 32) | (unsigned long long)f32_bits(myrec);   // (own entries: lane == entry)
    bool done = !mine || own;
    if (p.coop_mode != 1) {
      const unsigned long long t0 = wall_clock64();
      while (true) {
        if (!done) {
          gran = __hip_atomic_load(rec + sidx * (2 * MAXG) + ent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          done = (unsigned)(gran >> 32) == gen;
        }
        if (__all(done) || wall_clock64() - t0 > (unsigned long long)p.poll_ticks) break;
        __builtin_amdgcn_s_sleep(2);
      }
    }
    // own entries come from this workgroup's registers (lane tid < 2*GB holds entry tid of split `split`)
    const float ownv = __shfl(myrec, ent, 64);
    if (mine && done) s_rec[sidx][ent] = own ? ownv : bits_f32((unsigned)gran);
    const unsigned long long late = __ballot(mine && !done);
    if (tid == 0) {
      unsigned m = 0;
#pragma unroll
      for (int sx = 0; sx < MAXS; ++sx) m |= ((late >> (sx * 2 * MAXG)) & 0xffull) ? (1u << sx) : 0u;
      s_missing = m;
    }
  }
  __syncthreads();
  const unsigned missing = s_missing;                 // workgroup-uniform
  if (missing) {
    // ---- cold path: partners that are not running (yet).  Their records are a pure function of their pixels: compute them.
    for (int sx = 0; sx < S; ++sx) {
      if (!((missing >> sx) & 1u)) continue;
      load_raw(sx);
      float v;
      split_stats(sx, v);
      if (tid < 2 * GB) s_rec[sx][tid] = v;
    }
    load_raw(split);
    if (tid == 0 && p.sync_diag) __hip_atomic_fetch_add(p.sync_diag, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }
  if (tid < 2 * GB && !(tid & 1)) {                   // lane g*2 takes group g (fixed split order -> deterministic)
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int sx = 0; sx < S; ++sx) {
      const int q0 = sx * per, q1 = min(p.HW, q0 + per);
      chan_add(n, mean, m2, (double)max(0, q1 - q0) * cpg, (double)s_rec[sx][tid], (double)s_rec[sx][tid + 1]);
    }
    const float var = n > 0.0 ? (float)(m2 / n) : 0.f;
    s_stat[tid >> 1][0] = (float)mean;
    s_stat[tid >> 1][1] = 1.0f / sqrtf(var + p.eps);
  }
  __syncthreads();
  if (!active) return;
  float a[PC], bb[PC];
  {
    const float m0 = s_stat[glo][0], r0 = s_stat[glo][1];
    const float m1 = s_stat[glo + 1 < MAXG ? glo + 1 : glo][0], r1 = s_stat[glo + 1 < MAXG ? glo + 1 : glo][1];
#pragma unroll
    for (int e = 0; e < PC; ++e) {
      a[e] = (e < split_e ? r0 : r1) * p.gamma[c0 + e];
      bb[e] = p.beta[c0 + e] - (e < split_e ? m0 : m1) * a[e];
    }
  }
  T* dst = (T*)p.out + (size_t)b * p.HW * C + c0;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int pix = pix0 + pr + ppi * k;
    if (pix < pix1) {
      float f[PC];
      if constexpr (sizeof(T) == 2) {
        // (unpacked through volatile asm so that these are not the values of pass 2 kept alive - see pass 1)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned w = raw[k][i];
          asm volatile("v_lshlrev_b32 %0, 16, %2\n\tv_and_b32 %1, 0xffff0000, %2" : "=&v"(f[2 * i]), "=&v"(f[2 * i + 1]) : "v"(w));
        }
      } else {
#pragma unroll
        for (int e = 0; e < PC; ++e) f[e] = to_f32<T>(chunk_elem<T>(raw[k], e));
      }
#pragma unroll
      for (int e = 0; e < PC; ++e) {
        float y = f[e] * a[e] + bb[e];
        if (p.silu) y = silu_f(y);
        f[e] = y;
      }
      *(uint4*)(dst + (size_t)pix * C) = Chunk<T>::pack(f);
    }
  }
}

// Hand-off state of the cooperative kernel: [kGnMaxSlabs] ticket counters | [kGnMaxSlabs][8 splits][8 entries] granules |
// one diagnostic counter (workgroups that took the cold path).  Zeroed once; nothing is ever reset (tags only grow).
// Every handle (UNet, seg-VAE, image VAE) owns one region, allocated at *_create: its launches are ordered by its stream,
// so consecutive norms - and replays of a captured graph - draw consecutive generations on it.  Launches that do not come
// from a handle (the ldmseg_op_* test surface) take a region of a small per-device ring under a mutex.
constexpr int kGnMaxSlabs = 2048, kGnRing = 16;
constexpr size_t kGnCtrWords = kGnMaxSlabs, kGnRecWords = (size_t)kGnMaxSlabs * 8 * 8;
constexpr size_t kGnRegionWords = kGnCtrWords + kGnRecWords + 8;
void gn_bind_region(GNParams& p, unsigned long long* base) {
  p.sync_ctr = base;
  p.sync = base + kGnCtrWords;
  p.sync_diag = base + kGnCtrWords + kGnRecWords;
}
std::mutex g_gn_mu;
unsigned long long* g_gn_ring[64] = {};
hipStream_t g_gn_ring_stream[64][kGnRing] = {};
int g_gn_ring_used[64] = {};
// One region per (device, stream): launches of one stream are ordered, so consecutive norms draw consecutive generations
// on their region exactly like a handle's do; two streams never share one (round 4 handed the regions out round-robin,
// so that two cooperative launches in flight on different streams could meet on one region after the ring wrapped and
// read each other's records - ADVICE r04).  More distinct streams than regions: null, the caller takes the two-launch path.
unsigned long long* gn_ring_region(hipStream_t s, int* dev_out = nullptr) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (dev_out) *dev_out = dev;
  std::lock_guard<std::mutex> lk(g_gn_mu);
  if (!g_gn_ring[dev]) {
    void* q = nullptr;
    if (hipMalloc(&q, kGnRegionWords * kGnRing * sizeof(unsigned long long)) != hipSuccess) return nullptr;
    (void)hipMemset(q, 0, kGnRegionWords * kGnRing * sizeof(unsigned long long));
    g_gn_ring[dev] = (unsigned long long*)q;
  }
  int idx = -1;
  for (int i = 0; i < g_gn_ring_used[dev]; ++i)
    if (g_gn_ring_stream[dev][i] == s) { idx = i; break; }
  if (idx < 0) {
    if (g_gn_ring_used[dev] >= kGnRing) return nullptr;
    idx = g_gn_ring_used[dev]++;
    g_gn_ring_stream[dev][idx] = s;
  }
  return g_gn_ring[dev] + (size_t)idx * kGnRegionWords;
}

int g_gn_coop_mode = 0;     // 1: every cooperative workgroup computes its partners' records itself (tests of the cold path)
int g_gn_poll_us = 100;     // bound of the partner poll

int g_gn_variant = 0;   // tuning knob (ldmseg_debug_set key 8): bit0 = no cooperative kernel (64x64 maps on the two-launch path)

// One workgroup per (image, group), the whole slice in the registers of 512 threads (round 6; bf16, one source): no hand-off between
// workgroups at all - gn_coop_kernel's S splits of an (image, group block) exchange partial statistics through memory (~3 us of a
// 12 us launch at 32x32 x 640) - and the tensor is still read once and written once.  The price is that a group's channels are not
// 16-byte aligned, so the slice moves in 8-byte pieces (cpg % 4 == 0) of partial cache lines.  The 4-byte prototype
// (tools/ubench/gn_group.hip, B = 8): 8.8 against 12.0 us at 32x32 x 640 (20 dwords per thread), but 20.5 against 16.6 us at 64x64 x 320
// (40 dwords per thread of 4-byte accesses: TA-bound) and far worse beyond - hence the rule in run_gn; the 8-byte form shipped here
// measures 7.5 us on the first shape.  torch.cat([x0, x1], 1) inputs: an access lies in one source (C0 % 4 == 0), chosen per access.
// Block b -> image b % B, group b / B: with block b on XCD b % 8 (observed, speed only) all groups of an image share one L2, so every
// line is fetched into and written back from one L2.  Statistics: two passes over the registers (mean, then centred squares), fixed
// reduction order (deterministic).  W = dwords per access (1 | 2), NV = accesses per thread.
template <int NV, int W>
__global__ __launch_bounds__(512) void gn_group_kernel(const GNParams p, int upp, int stepq, int stepr) {
  __shared__ float red[8];
  __shared__ float gb[2][128];
  const int tid = threadIdx.x;
  const int b = blockIdx.x % p.B, g = blockIdx.x / p.B;
  const int cpg = p.cpg, C = p.C0 + p.C1;
  const int total = p.HW * upp;                       // accesses of this (image, group) slice; upp = accesses per pixel
  typedef unsigned uw __attribute__((ext_vector_type(W)));
  // torch.cat([x0, x1], 1): access j of a pixel covers channels g cpg + 2 W j ..; C0 is a multiple of 2 W, so an access lies in ONE source.
  // jsplit = first access of this group that comes from the second source (0: the whole group does, upp: none of it)
  const int c_lo = g * cpg;
  const int jsplit = c_lo >= p.C0 ? 0 : (c_lo + cpg <= p.C0 ? upp : (p.C0 - c_lo) / (2 * W));
  const unsigned* x0 = (const unsigned*)p.src0 + (size_t)b * p.HW * (p.C0 / 2) + (size_t)(c_lo / 2);
  const unsigned* x1 = (const unsigned*)p.src1 + (ptrdiff_t)b * p.HW * (p.C1 / 2) + (ptrdiff_t)((c_lo - p.C0) / 2);   // (only dereferenced for j >= jsplit; may point below src1 when the group straddles)
  unsigned* yout = (unsigned*)p.out + (size_t)b * p.HW * (C / 2) + (size_t)g * (cpg / 2);
  if (tid < cpg) { gb[0][tid] = p.gamma[g * cpg + tid]; gb[1][tid] = p.beta[g * cpg + tid]; }
  uw v[NV];
  int pix = tid / upp, j = tid - pix * upp;            // (one division per thread, off the critical path of nothing: 16 loads follow)
  const int pix0 = pix, j0 = j;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int d = tid + i * 512;
    const int pc = d < total ? pix : 0, jc = d < total ? j : 0;      // (clamped: a guarded load is a branch and a wait each)
    v[i] = jc < jsplit ? *(const uw*)(x0 + (size_t)pc * (p.C0 / 2) + jc * W) : *(const uw*)(x1 + (ptrdiff_t)pc * (p.C1 / 2) + jc * W);
    pix += stepq; j += stepr;
    if (j >= upp) { j -= upp; ++pix; }
  }
  auto wg_sum = [&](float a) __attribute__((always_inline)) -> float {
    a = wave64_sum(a);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = a;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t += red[w];
    return t;
  };
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (tid + i * 512 < total) {
#pragma unroll
      for (int e = 0; e < W; ++e) { const unsigned u = v[i][e]; s += bits_f32(u << 16) + bits_f32(u & 0xffff0000u); }
    }
  }
  const float inv_n = (float)p.inv_n;
  const float mean = wg_sum(s) * inv_n;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (tid + i * 512 < total) {
#pragma unroll
      for (int e = 0; e < W; ++e) {
        const unsigned u = v[i][e];
        const float a = bits_f32(u << 16) - mean, c = bits_f32(u & 0xffff0000u) - mean;
        q += a * a + c * c;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(wg_sum(q) * inv_n + p.eps);
  pix = pix0; j = j0;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (tid + i * 512 < total) {
      uw o;
#pragma unroll
      for (int e = 0; e < W; ++e) {
        const unsigned u = v[i][e];
        const int c = 2 * (j * W + e);
        float a = (bits_f32(u << 16) - mean) * rstd * gb[0][c] + gb[1][c];
        float d = (bits_f32(u & 0xffff0000u) - mean) * rstd * gb[0][c + 1] + gb[1][c + 1];
        if (p.silu) { a = silu_f(a); d = silu_f(d); }
        o[e] = pack_bf16x2(a, d);
      }
      *(uw*)(yout + (size_t)pix * (C / 2) + j * W) = o;
    }
    pix += stepq; j += stepr;
    if (j >= upp) { j -= upp; ++pix; }
  }
}

template <typename T>
int run_gn(const GNParams& pin, hipStream_t s) {
  constexpr int PC = Chunk<T>::N;
  GNParams p = pin;
  const int C = p.C0 + p.C1;
  if (p.groups < 1 || C % p.groups != 0 || C % PC != 0 || p.C0 % PC != 0 || p.groups > 64) return -2;
  p.cpg = C / p.groups;
  p.fd_cpg = fastdiv_make(p.cpg);
  p.inv_n = 1.0 / ((double)p.HW * p.cpg);
  // a 16-B vector may span at most two groups: cpg >= PC, or exactly two whole groups per vector
  // (128 channels in bf16: cpg 4, PC 8 - the image VAE's first level)
  if (C / p.groups < PC && 2 * (C / p.groups) != PC) return -2;
  if (C / PC > 256 * kMaxIter) return -2;
  auto try_coop = [&]() -> int {   // returns 1 when the shape is not eligible
    // cooperative one-pass kernel: the maps the register-resident kernel above cannot hold (64x64 and up)
    constexpr int NTC = 512, MAXVC = 21;
    const int cpg = C / p.groups;
    for (int GB = 1; GB <= 4; GB *= 2) {
      if ((GB * cpg) % PC != 0 || p.groups % GB != 0 || cpg < PC || cpg % 2 != 0) continue;
      const int vpp = GB * cpg / PC;
      if (vpp > 64) continue;
      const int ppi = NTC / vpp;
      const int slabs = p.B * (p.groups / GB);
      if (slabs > kGnMaxSlabs) continue;
      // splits: enough workgroups for every CU, as few as the registers allow
      int S = 1;
      while (S < 8 && ((p.HW + S - 1) / S + ppi - 1) / ppi > MAXVC) S *= 2;
      while (S < 8 && slabs * S * 2 <= num_cus_gn() && p.HW / (2 * S) >= ppi) S *= 2;
      if (((p.HW + S - 1) / S + ppi - 1) / ppi > MAXVC) continue;
      if (slabs * S > num_cus_gn()) continue;                     // every workgroup must be resident: one per CU (8 waves at ~200 registers)
      unsigned long long* region = (unsigned long long*)p.sync_region;
      if (!region) {
        region = gn_ring_region(s);
        if (!region) break;                                         // no region for this stream: the two-launch path below
      }
      gn_bind_region(p, region);
      p.splits = S;
      p.coop_mode = g_gn_coop_mode;
      p.poll_ticks = (p.poll_us >= 0 ? p.poll_us : g_gn_poll_us) * 100;   // wall_clock64: 100 MHz
      // a handle that is backing off (its own short bound) does not count what that bound makes it miss: the counter then only
      // says what FULL-bound launches met, which is what the back-off decision wants to know (with every fallback
      // counted, 256 of them under the 2 us bound re-armed the back-off on every call and the full bound was never tried again)
      if (p.poll_us >= 0) p.sync_diag = nullptr;
      p.per = (p.HW + S - 1) / S;
      p.ty = ppi;
      p.fd_aux = fastdiv_make(vpp);
      p.fd_vx = fastdiv_make(S);
      p.fd_cpg = fastdiv_make(p.groups / GB);
      hipLaunchKernelGGL((gn_coop_kernel<T, MAXVC, NTC>), dim3(slabs * S), dim3(NTC), 0, s, p, GB);
      return hipGetLastError() == hipSuccess ? 0 : -3;
    }
    p.fd_cpg = fastdiv_make(p.cpg);
    return 1;
  };
  // round 6: one workgroup per (image, group) with the slice in registers where that slice is small enough for 8-byte pieces to win
  // (<= 40 dwords per thread and channels per group a multiple of 4; measured at B = 8, kbench gn: 32x32 x 640 12.0 -> 7.5 us (11 norms of a
  // forward), 32x32 x (640+640) 15.5 -> 12.0, 16x16 x 640 6.1 -> 4.4, 16x16 x 1280 6.4 -> 5.2, 16x16 x (1280+1280) 9.3 -> 6.9, 16x16 x
  // (1280+640) 10.4 -> 7.5; 8x8 maps level with gn_one_kernel and left there; whole forward -0.07 ms; tuning bit 5 switches it off)
  if constexpr (sizeof(T) == 2) {
    const long acc8 = (long)p.HW * (p.cpg / 4);                     // 8-byte accesses per (image, group) slice
    constexpr int nv_max = 20;
    if (!(g_gn_variant & 33) && p.HW >= 256 && p.cpg % 4 == 0 && p.cpg <= 128 && p.C0 % 4 == 0 && p.C1 % 4 == 0 && acc8 <= 512L * nv_max &&
        (long)p.B * p.groups >= 128) {
      const int upp = p.cpg / 4;                                    // 8-byte accesses per pixel
      if (acc8 <= 512 * 10) hipLaunchKernelGGL((gn_group_kernel<10, 2>), dim3(p.B * p.groups), dim3(512), 0, s, p, upp, 512 / upp, 512 % upp);
      else hipLaunchKernelGGL((gn_group_kernel<20, 2>), dim3(p.B * p.groups), dim3(512), 0, s, p, upp, 512 / upp, 512 % upp);
      return hipGetLastError() == hipSuccess ? 0 : -3;
    }
  }
  // the cooperative kernel first from 32x32 maps up (measured at B = 8: 32x32 x 640 13.9 -> 11.7 us, 32x32 x 1920 29.1 -> 18.8 us,
  // 64x64 x 320 22.9 -> 16.3 us against the paths below); tuning bit1 extends it to the 16x16 maps
  const int coop_min_hw = (g_gn_variant & 2) ? 256 : 1024;
  if (!(g_gn_variant & 1) && p.HW >= coop_min_hw) {
    const int r = try_coop();
    if (r <= 0) return r;
  }
  {
    // single-launch register-resident kernel: GB = 1 or 2 groups per workgroup forming whole 16-byte vectors
    constexpr int MAXV = 22;
    const int cpg = C / p.groups;
    for (int GB = 1; GB <= 2; ++GB) {
      if ((GB * cpg) % PC != 0 || p.groups % GB != 0) continue;
      const int vpp = GB * cpg / PC;
      if (vpp > 64) continue;
      const int ppi = 256 / vpp;
      if ((p.HW + ppi - 1) / ppi > MAXV) continue;
      if ((long)p.B * (p.groups / GB) < 96) continue;          // too few workgroups to fill the chip
      p.ty = ppi;
      p.fd_aux = fastdiv_make(vpp);
      const dim3 grid(p.groups / GB, p.B), grid1((p.groups / GB) * p.B);
      const int nv = (p.HW + ppi - 1) / ppi;
      if (!(g_gn_variant & 4) && nv <= 12) {                   // shipped: one reduction, one barrier (gn_one_kernel); measured at
        // B = 8: 8x8 x 1280 6.2 -> 4.1 us, 16x16 x 1280 7.9 -> 6.9 us; with more than 12 vectors per thread the two-pass kernel
        // below is the faster one (16x16 x 1920: 10.2 against 11.4 us)
        if (nv <= 2) hipLaunchKernelGGL((gn_one_kernel<T, 2>), grid1, dim3(256), 0, s, p, GB);
        else if (nv <= 6) hipLaunchKernelGGL((gn_one_kernel<T, 6>), grid1, dim3(256), 0, s, p, GB);
        else hipLaunchKernelGGL((gn_one_kernel<T, 12>), grid1, dim3(256), 0, s, p, GB);
        return hipGetLastError() == hipSuccess ? 0 : -3;
      }
      if (p.HW <= 64 && vpp < 10) continue;                    // 8x8 maps with short runs: gn_small measured faster (8.0 vs 9.7 us)
      hipLaunchKernelGGL((gn_fused_kernel<T, MAXV>), grid, dim3(256), 0, s, p, GB);
      return hipGetLastError() == hipSuccess ? 0 : -3;
    }
  }
  if (!(g_gn_variant & 1) && p.HW >= 2048) {
    const int r = try_coop();
    if (r <= 0) return r;
  }
  {
    constexpr int EPU = 4 / (int)sizeof(T);
    const int cpg = C / p.groups;
    const long nunits = (long)p.HW * (cpg / EPU);
    // measured: wins for the 8x8 maps (16 -> 9 us), loses from 16x16 up (its 4-byte strided loads)
    if (cpg % EPU == 0 && p.C0 % EPU == 0 && nunits <= 256 * 12 && (long)p.B * p.groups >= 128) {
      p.fd_aux = fastdiv_make(cpg / EPU);
      hipLaunchKernelGGL((gn_small_kernel<T, 12>), dim3(p.groups, p.B), dim3(256), 0, s, p);
      return hipGetLastError() == hipSuccess ? 0 : -3;
    }
  }
  const int nvec = C / PC;
  p.vx = nvec < 256 ? nvec : 256;
  p.ty = 256 / p.vx;
  p.fd_vx = fastdiv_make(p.vx);
  // the two-launch path combines statistics with 8 lanes per group in a 256-thread workgroup (32 groups) and keeps
  // at most 128 chunk partials per lane set (gn_reduce_stats: MAXP = 16 x 8 lanes)
  if (p.nchunk < 1 || p.nchunk > 128 || p.groups > 32) return -2;
  p.per = (p.HW + p.nchunk - 1) / p.nchunk;
  hipLaunchKernelGGL(gn_partial_kernel<T>, dim3(p.nchunk, p.B), dim3(256), 0, s, p);
  // pixel chunks: >= 4 pixels per thread row, ~2048 workgroups in total
  const int ty = p.ty;
  // 16 pixels per thread row (4 unrolled trips) on the big maps, down to 4 when that would leave fewer than ~512
  // workgroups on the chip (every workgroup re-reduces the image's partials first, so fewer and fatter is better)
  int blocks = (p.HW + 16 * ty - 1) / (16 * ty);
  const int want = max(1, 512 / p.B);
  if (blocks < want) blocks = min(want, (p.HW + 4 * ty - 1) / (4 * ty));
  const int cap = max(1, 1024 / p.B);
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(gn_apply_kernel<T>, dim3(blocks, p.B), dim3(256), 0, s, p);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <typename T>
int run_ln(const void* x, void* y, const float* g, const float* b, int M, int C, float eps, int silu, hipStream_t s) {
  constexpr int PC = Chunk<T>::N;
  if (C % PC != 0) return -2;
  const int nvec = C / PC;
  const int vpl = (nvec + 63) / 64;
  dim3 block(256);
  auto grid_for_rpw = [&](int rpw) {
    int blocks = (M + 4 * rpw - 1) / (4 * rpw);
    if (blocks > 2048) blocks = 2048;      // grid-stride over row batches beyond that
    return dim3(blocks < 1 ? 1 : blocks);
  };
  switch (vpl) {
    case 1: hipLaunchKernelGGL((layernorm_kernel<T, 1, 4>), grid_for_rpw(4), block, 0, s, (const T*)x, (T*)y, g, b, M, C, eps, silu); break;
    case 2: hipLaunchKernelGGL((layernorm_kernel<T, 2, 4>), grid_for_rpw(4), block, 0, s, (const T*)x, (T*)y, g, b, M, C, eps, silu); break;
    case 3: hipLaunchKernelGGL((layernorm_kernel<T, 3, 2>), grid_for_rpw(2), block, 0, s, (const T*)x, (T*)y, g, b, M, C, eps, silu); break;
    case 4: case 5: hipLaunchKernelGGL((layernorm_kernel<T, 5, 1>), grid_for_rpw(1), block, 0, s, (const T*)x, (T*)y, g, b, M, C, eps, silu); break;
    default: return -2;
  }
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <typename T>
int run_rowstats(const void* x, float* stats, int M, int C, float eps, hipStream_t s) {
  constexpr int PC = Chunk<T>::N;
  if (C % PC != 0) return -2;
  const int vpl = (C / PC + 63) / 64;
  dim3 block(256);
  auto grid_for_rpw = [&](int rpw) {
    int blocks = (M + 4 * rpw - 1) / (4 * rpw);
    if (blocks > 2048) blocks = 2048;
    return dim3(blocks < 1 ? 1 : blocks);
  };
  const T* xp = (const T*)x;
  {
    const int nvec = C / PC;
    constexpr int RPW = 2;
    auto grid_rows = [&](int rows_per_wave) {
      int blocks = (M + 4 * rows_per_wave - 1) / (4 * rows_per_wave);
      if (blocks > 4096) blocks = 4096;
      return dim3(blocks < 1 ? 1 : blocks);
    };
    if (!(g_gn_variant & 16)) {       // (tuning bit 4: the one-row-per-wave kernel below)
      if (nvec <= 40) { hipLaunchKernelGGL((rowstats_kernel<T, 8, RPW>), grid_rows(8 * RPW), block, 0, s, xp, stats, M, C, eps); return hipGetLastError() == hipSuccess ? 0 : -3; }
      if (nvec <= 80) { hipLaunchKernelGGL((rowstats_kernel<T, 16, RPW>), grid_rows(4 * RPW), block, 0, s, xp, stats, M, C, eps); return hipGetLastError() == hipSuccess ? 0 : -3; }
      if (nvec <= 160) { hipLaunchKernelGGL((rowstats_kernel<T, 32, RPW>), grid_rows(2 * RPW), block, 0, s, xp, stats, M, C, eps); return hipGetLastError() == hipSuccess ? 0 : -3; }
      if (nvec <= 320) { hipLaunchKernelGGL((rowstats_kernel<T, 64, RPW>), grid_rows(RPW), block, 0, s, xp, stats, M, C, eps); return hipGetLastError() == hipSuccess ? 0 : -3; }
    }
  }
  switch (vpl) {
    case 1: hipLaunchKernelGGL((layernorm_kernel<T, 1, 4, true>), grid_for_rpw(4), block, 0, s, xp, (T*)nullptr, nullptr, nullptr, M, C, eps, 0, stats); break;
    case 2: hipLaunchKernelGGL((layernorm_kernel<T, 2, 4, true>), grid_for_rpw(4), block, 0, s, xp, (T*)nullptr, nullptr, nullptr, M, C, eps, 0, stats); break;
    case 3: hipLaunchKernelGGL((layernorm_kernel<T, 3, 2, true>), grid_for_rpw(2), block, 0, s, xp, (T*)nullptr, nullptr, nullptr, M, C, eps, 0, stats); break;
    case 4: case 5: hipLaunchKernelGGL((layernorm_kernel<T, 5, 2, true>), grid_for_rpw(2), block, 0, s, xp, (T*)nullptr, nullptr, nullptr, M, C, eps, 0, stats); break;
    default: return -2;
  }
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

int launch_rowstats(const void* x, float* stats, int M, int C, float eps, int dtype, hipStream_t s) {
  return dtype == DT_BF16 ? run_rowstats<bf16_t>(x, stats, M, C, eps, s) : run_rowstats<float>(x, stats, M, C, eps, s);
}

int gn_nchunk(int B, int HW) {
  // ~512-1024 workgroups in total, at least 8 pixels per chunk
  int n = 768 / (B > 0 ? B : 1);
  if (n < 1) n = 1;
  if (n > 128) n = 128;
  while (n > 1 && HW / n < 8) n >>= 1;
  return n;
}

void groupnorm_set_variant(int v) { g_gn_variant = v; }
void groupnorm_set_coop(int mode, int poll_us) {
  g_gn_coop_mode = mode == 1 ? 1 : 0;
  g_gn_poll_us = poll_us < 0 ? 100 : (poll_us > 1000000 ? 1000000 : poll_us);
}
size_t gn_sync_bytes() { return kGnRegionWords * sizeof(unsigned long long); }
int gn_sync_init(void* region, hipStream_t s) {
  return hipMemsetAsync(region, 0, gn_sync_bytes(), s) == hipSuccess ? 0 : -3;
}
int gn_warm() {
  std::lock_guard<std::mutex> lk(g_gn_mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!g_gn_ring[dev]) {
    void* q = nullptr;
    if (hipMalloc(&q, kGnRegionWords * kGnRing * sizeof(unsigned long long)) != hipSuccess) return -3;
    (void)hipMemset(q, 0, kGnRegionWords * kGnRing * sizeof(unsigned long long));
    g_gn_ring[dev] = (unsigned long long*)q;
  }
  return 0;
}
const void* gn_sync_diag_ptr(const void* region) {
  return region ? (const void*)((const unsigned long long*)region + kGnCtrWords + kGnRecWords) : nullptr;
}
long long gn_coop_fallbacks(const void* region) {
  unsigned long long v = 0, tot = 0;
  if (region) {
    if (hipMemcpy(&v, (const unsigned long long*)region + kGnCtrWords + kGnRecWords, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (long long)v;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  const unsigned long long* ring;
  { std::lock_guard<std::mutex> lk(g_gn_mu); ring = g_gn_ring[dev]; }
  if (!ring) return 0;
  for (int i = 0; i < kGnRing; ++i) {
    if (hipMemcpy(&v, ring + (size_t)i * kGnRegionWords + kGnCtrWords + kGnRecWords, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    tot += v;
  }
  return (long long)tot;
}

int launch_groupnorm(const GNParams& p, int dtype, hipStream_t s) {
  return dtype == DT_BF16 ? run_gn<bf16_t>(p, s) : run_gn<float>(p, s);
}

namespace {
// group block / pixels per trip / vectors per thread of the finish + GroupNorm kernel; false: no instantiation
template <typename T>
bool finish_gn_plan(int B, int HW, int C, int* GBo, int* vppo, int* ppio, int* nvo) {
  constexpr int PC = Chunk<T>::N;
  if (C % 32 != 0) return false;
  const int cpg = C / 32;
  for (int GB = 1; GB <= 2; ++GB) {
    if ((GB * cpg) % PC != 0 || 32 % GB != 0) continue;
    const int vpp = GB * cpg / PC;
    if (vpp > 64) continue;
    const int ppi = 256 / vpp;
    const int nv = (HW + ppi - 1) / ppi;
    if (nv > 12) continue;
    if ((long)B * (32 / GB) < 96) continue;
    *GBo = GB; *vppo = vpp; *ppio = ppi; *nvo = nv;
    return true;
  }
  return false;
}
template <typename T>
int run_finish_gn(const IgemmParams& ip, GNParams p, hipStream_t s) {
  int GB, vpp, ppi, nv;
  if (!finish_gn_plan<T>(p.B, p.HW, p.C0, &GB, &vpp, &ppi, &nv)) return -2;
  if (ip.splits < 2 || !ip.partial || !ip.bias || ip.n_valid != p.C0 || ip.M != p.B * p.HW || p.C1 != 0) return -2;
  p.groups = 32;
  p.cpg = p.C0 / 32;
  p.inv_n = 1.0 / ((double)p.HW * p.cpg);
  p.ty = ppi;
  p.fd_aux = fastdiv_make(vpp);
  const dim3 grid((32 / GB) * p.B);
  const long slab = (long)ip.M * ip.N;
#define LDMSEG_FGN(MV) hipLaunchKernelGGL((finish_gn_kernel<T, MV>), grid, dim3(256), 0, s, p, GB, ip.partial, ip.splits, ip.N, slab, \
                                          ip.bias, ip.rowbias, ip.rb_stride)
  if (nv <= 2) LDMSEG_FGN(2);
  else if (nv <= 6) LDMSEG_FGN(6);
  else LDMSEG_FGN(12);
#undef LDMSEG_FGN
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
}  // namespace

bool finish_groupnorm_ok(int B, int HW, int C, int dtype) {
  int a, b, c, d;
  if (g_gn_variant & 8) return false;       // tuning bit 3: keep finish and norm apart
  return dtype == DT_BF16 ? finish_gn_plan<bf16_t>(B, HW, C, &a, &b, &c, &d) : finish_gn_plan<float>(B, HW, C, &a, &b, &c, &d);
}
// Statistics half of the two-launch path on its own (round 5): partial[b][chunk][group] = {mean, M2} over `per` = ceil(HW / nchunk)
// pixels per chunk, nothing else is written.  The consumer (tproj.hip: the transformer's norm folded into the fused entry) combines
// the chunks and applies the affine to its own rows.
template <typename T>
int run_gn_stats(const GNParams& pin, hipStream_t s) {
  constexpr int PC = Chunk<T>::N;
  GNParams p = pin;
  const int C = p.C0 + p.C1;
  if (p.groups < 1 || C % p.groups != 0 || C % PC != 0 || p.C0 % PC != 0 || p.groups > 32 || !p.partial) return -2;
  p.cpg = C / p.groups;
  p.fd_cpg = fastdiv_make(p.cpg);
  if (p.cpg < PC && 2 * p.cpg != PC) return -2;
  if (C / PC > 256 * kMaxIter) return -2;
  const int nvec = C / PC;
  p.vx = nvec < 256 ? nvec : 256;
  p.ty = 256 / p.vx;
  p.fd_vx = fastdiv_make(p.vx);
  if (p.nchunk < 1 || p.nchunk > 128) return -2;
  p.per = (p.HW + p.nchunk - 1) / p.nchunk;
  hipLaunchKernelGGL(gn_partial_kernel<T>, dim3(p.nchunk, p.B), dim3(256), 0, s, p);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
int launch_groupnorm_stats(const GNParams& p, int dtype, hipStream_t s) {
  return dtype == DT_BF16 ? run_gn_stats<bf16_t>(p, s) : run_gn_stats<float>(p, s);
}

int launch_finish_groupnorm(const IgemmParams& ip, const GNParams& g, int dtype, hipStream_t s) {
  return dtype == DT_BF16 ? run_finish_gn<bf16_t>(ip, g, s) : run_finish_gn<float>(ip, g, s);
}

int launch_layernorm(const void* x, void* y, const float* gamma, const float* beta, int M, int C, float eps,
                     int silu, int dtype, hipStream_t s) {
  return dtype == DT_BF16 ? run_ln<bf16_t>(x, y, gamma, beta, M, C, eps, silu, s)
                          : run_ln<float>(x, y, gamma, beta, M, C, eps, silu, s);
}

}  // namespace ldmseg

#ifdef LDMSEG_GN_STAMP
extern "C" int ldmseg_debug_gn_stamps(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(ldmseg::g_gn_ts), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
