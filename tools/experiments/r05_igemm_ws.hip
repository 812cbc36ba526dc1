// Weight-streaming implicit GEMM for the small feature maps (round 5): the 8x8 / 16x16 levels of the UNet, where a launch
// is M = B*H*W <= ~2048 output rows against 30 - 60 MB of weights.
//
//   partial[z][m][n] = sum_{k in slice z} X[m][k] * W[n][k]        m = (b, oy, ox), k = (tap, channel)
//
// Why a second kernel.  igemm_kernel brings BOTH operands into the LDS with LDS-DMA, (BM + BN) * 128 B per K tile; on these
// maps the output is so small (512 x 1280) that the launch is a 256-way K split, every workgroup re-stages the X rows of
// its m tile for every n tile and the W rows of its n tile for every m tile, and the launch is bound by what the L2 -> LDS
// path delivers per CU (207 MB staged for 15 GFLOP at M = 512: 23 us, 0.26 of the matrix peak) - whatever tile the launch
// table picks, bytes per MAC go like 1/BM + 1/BN.  Here only X goes through the LDS (a 128-row tile shared by all waves,
// gathered by LDS-DMA exactly as in igemm.hip: tap-major K, zero page for padding taps, nearest-x2 / concat as pointer
// arithmetic).  Every wave owns a 128 x 32 output tile, i.e. its 32 weight rows are needed by NO other wave of the
// workgroup: they go global -> VGPR directly (global_load_dwordx4 in MFMA A-fragment layout, lane = (row & 15, 16-byte
// chunk)), four K tiles ahead, through the vector L1 instead of the DMA path.  Per K tile a workgroup of 8 waves then
// moves 16 KB by DMA + 32 KB by plain loads for 128 x 256 x 64 MACs (igemm's 128 x 160 tile: 36 KB by DMA for 128 x 160 x 64).
//
// Synchronisation: vmcnt counts LDS-DMA and register loads of a wave in ONE in-order queue, so every VMEM instruction of the
// K loop is inline asm with a fixed count per K tile (XG DMA pieces, 2 + 2 fragment loads) and ONE counted s_waitcnt per
// tile covers "X tile t+1 has landed" and "my W fragments of tile t+1 have landed" together; the registers a wait covers
// are operands of the wait, which is what keeps hipcc from moving their uses above it.
#include <cstdio>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace ldmseg {

namespace {

constexpr int kRowBytes = 128;
constexpr int kBM = 128;          // rows of the X tile (all waves share it)
constexpr int kStageBytes = kBM * kRowBytes;

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off"
      :
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt_tied(v4i& a, v4i& b, v4i& c, v4i& d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(N) : "memory");
}

template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (N > 0) {
    sfor<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

struct RowInfo {
  int pix_base;     // b * Hi * Wi
  int yx;           // (iy0 << 16) | (ix0 & 0xffff): top-left input coordinate of the 3x3 window (rows past M: iy0 = -30000)
};

// NWN waves side by side along N (BN = 32 * NWN).  One work item per workgroup: (K slice z, n tile, m tile), m fastest, so
// that the m tiles that stream the same weight slice are neighbours (same XCD, same moment: one HBM read, L2 hits for the rest).
// kNST: LDS ring stages = W register ring slots = unroll factor of the K loop.  PIPE 0: X fragments in halves of four
// through two 16-register buffers (reads one half-step = 8 MFMAs ahead); PIPE 1: whole k-groups of eight fragments through
// two 32-register buffers (reads 16 MFMAs ahead; with the 4-slot W ring that does not fit 256 registers, hence kNST = 3).
template <int NWN, int kNST, int PIPE>
__global__ __launch_bounds__(NWN * 64, 2) void igemm_ws_kernel(const IgemmParams p) {
  constexpr int BN = 32 * NWN;
  constexpr int MF = kBM / 16, NF = 2;
  constexpr int XG = kBM / 8 / NWN;            // 8-row LDS-DMA groups per wave per K tile
  constexpr int BKE = 64;                      // bf16 elements per K tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int MT = (p.M + kBM - 1) / kBM;
  const int ntiles = MT * (p.N / BN);
  int item;
  {
    const int G = gridDim.x, bid = blockIdx.x;
    const int q = G >> 3, r = G & 7, xcd = bid & 7, idx = bid >> 3;
    item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int z = fd_div(item, p.fd_ntiles);
  const int tile = item - z * ntiles;
  const int nt = fd_div(tile, p.fd_nt);          // fd_nt: divisor MT here (m fastest)
  const int m0 = (tile - nt * MT) * kBM, n0 = nt * BN;
  const int Ctot = p.C0 + p.C1;
  const int K = p.taps * Ctot;
  const int nk_total = K / BKE;
  const int kb = fd_div(nk_total * z, p.fd_nsplit);
  const int nk = fd_div(nk_total * (z + 1), p.fd_nsplit) - kb;
  const int Hlog = p.up ? 2 * p.Hi : p.Hi, Wlog = p.up ? 2 * p.Wi : p.Wi;
  const int pad = (p.taps == 9) ? (p.pad >= 0 ? p.pad : 1) : 0;
  const int HWo = p.Ho * p.Wo;

  // ---------------- X gather stream (LDS-DMA, as igemm.hip) ----------------
  const int ld_r = lane >> 3, ld_j = (lane & 7) ^ ld_r;
  const unsigned char* zpage = (const unsigned char*)p.zeros;
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane(
      (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem) + (unsigned)wave * 1024u);
  RowInfo ri[XG];
  const unsigned char* rowptr[XG];
  unsigned rowinc[XG];
#pragma unroll
  for (int i = 0; i < XG; ++i) {
    const int m = m0 + (i * NWN + wave) * 8 + ld_r;
    if (m < p.M) {
      const int b = fd_div(m, p.fd_hwo), rem = m - b * HWo;
      const int oy = fd_div(rem, p.fd_wo), ox = rem - oy * p.Wo;
      ri[i].pix_base = b * p.Hi * p.Wi;
      ri[i].yx = ((oy * p.stride - pad) << 16) | ((ox * p.stride - pad) & 0xffff);
    } else {
      ri[i].pix_base = 0;
      ri[i].yx = (-30000) << 16;
    }
  }
  int f_tap, f_cc;
  {
    const int tiles_per_tap = Ctot / BKE;
    f_tap = fd_div(kb, p.fd_tpt);
    f_cc = (kb - f_tap * tiles_per_tap) * BKE;
  }
  bool need_setup = true;
  auto seg_setup = [&]() __attribute__((always_inline)) {
    const int ky = (p.taps == 9) ? f_tap / 3 : 0;
    const int kx = (p.taps == 9) ? f_tap - ky * 3 : 0;
    const unsigned char* sbase;
    int cs, coff;
    if (f_cc < p.C0) { sbase = (const unsigned char*)p.src0; cs = p.C0; coff = f_cc; }
    else { sbase = (const unsigned char*)p.src1; cs = p.C1; coff = f_cc - p.C0; }
#pragma unroll
    for (int i = 0; i < XG; ++i) {
      const int uy = (ri[i].yx >> 16) + ky, ux = (int)(short)(ri[i].yx & 0xffff) + kx;
      const bool inb = (uy >= 0) & (uy < Hlog) & (ux >= 0) & (ux < Wlog);
      const int iy = p.up ? (uy >> 1) : uy, ix = p.up ? (ux >> 1) : ux;
      const size_t off = ((size_t)(ri[i].pix_base + iy * p.Wi + ix) * cs + coff) * 2 + ld_j * 16;
      rowptr[i] = inb ? sbase + off : zpage;
      rowinc[i] = inb ? (unsigned)kRowBytes : 0u;
    }
  };
  // (valid == false: a stream position past the end of the slice - the piece still goes out, from the page of zeros, so
  // that every K tile carries the same number of VMEM instructions and the counted waits below stay exact)
  auto issue_x = [&](int stage, bool valid) __attribute__((always_inline)) {
    if (valid && need_setup) seg_setup();
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_wave + (unsigned)stage * (unsigned)kStageBytes);
#pragma unroll
    for (int i = 0; i < XG; ++i) {
      glds16(valid ? rowptr[i] : zpage, dst + i * (NWN * 1024));
      if (valid) rowptr[i] += rowinc[i];
    }
    if (valid) {
      f_cc += BKE;
      if (f_cc == Ctot) { f_cc = 0; ++f_tap; }
      need_setup = (f_cc == 0) | (f_cc == p.C0);
    }
  };

  // ---------------- W fragment stream (global -> VGPR) ----------------
  // The weights are held fragment-major (launch_pack_ws): the 16 bytes lane l of fragment a wants for K tile t, k-group kg
  // sit at ((n16 * nk_total + t) * 2 + kg) * 1024 + 16 l - one wave instruction reads 1 KiB of consecutive bytes (eight full
  // lines; the same rows out of the [N][K] matrix are sixteen half lines per instruction, which made the vector L1's tag rate
  // the bound: 1.6 us per K tile measured, three times the matrix time), and a fragment's whole K range is one sequential
  // stream.  Buffer loads: a position past the end of the slice gets an out-of-range offset, which returns zeros WITHOUT a
  // memory access - the instruction count per K tile never changes (exact vmcnt waits, no control flow around registers
  // with a load in flight) and the tail costs no bandwidth.  The K-tile offset is stepped in the VGPR offset (an SGPR
  // soffset overwritten soon after issue is a hazard on gfx950).
  v4i srd;
  {
    const unsigned long long b = (unsigned long long)(uintptr_t)p.Wf;
    srd[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    srd[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    srd[2] = __builtin_amdgcn_readfirstlane((int)((unsigned)p.N * (unsigned)K * 2u));
    srd[3] = 0x00020000;
  }
  constexpr unsigned kTileStep = 2048;           // two k-groups of 1 KiB per K tile
  unsigned wv0 = (unsigned)((((size_t)((n0 >> 4) + wave * 2) * nk_total + kb) * 2) * 1024) + (unsigned)lane * 16u;
  unsigned wv1 = wv0 + (unsigned)((size_t)nk_total * 2048);
  v4i w[kNST][2][NF];            // [slot][k-group][fragment]
#define WS_LOADW(SLOT, KG, VALID)                                                                                  \
  {                                                                                                                \
    const unsigned e0_ = (VALID) ? wv0 : 0x80000000u, e1_ = (VALID) ? wv1 : 0x80000000u;                            \
    asm volatile("buffer_load_dwordx4 %0, %2, %4, 0 offen offset:%5\n\tbuffer_load_dwordx4 %1, %3, %4, 0 offen offset:%5" \
                 : "=&v"(w[SLOT][KG][0]), "=&v"(w[SLOT][KG][1])                                                    \
                 : "v"(e0_), "v"(e1_), "s"(srd), "i"((KG) * 1024)                                                  \
                 : "memory");                                                                                      \
  }

  f32x4 acc[NF][MF];
#pragma unroll
  for (int a = 0; a < NF; ++a)
#pragma unroll
    for (int b = 0; b < MF; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr_row = (lane & 15) * kRowBytes;
  const int fr_c0 = (((lane >> 4)) ^ (lane & 7)) * 16;
  const int fr_c1 = (((lane >> 4) + 4) ^ (lane & 7)) * 16;
  const int lg = lane >> 4;

  // ---------------- prologue: X tiles 0..2 and W tiles 0..3 in flight ----------------
#pragma unroll
  for (int j = 0; j < kNST - 1; ++j) issue_x(j, j < nk);
  sfor<kNST>([&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    WS_LOADW(j, 0, j < nk)
    WS_LOADW(j, 1, j < nk)
    wv0 += kTileStep;
    wv1 += kTileStep;
  });
  wait_vmcnt_tied<4 * (kNST - 2)>(w[0][0][0], w[0][0][1], w[0][1][0], w[0][1][1]);       // every X piece, W 0 and W 1 have landed
  __syncthreads();

  // in flight behind X(t+1) at the wait of tile t: W(t+2) .. and X(t+2) .. up to X(t+kNST-1), W(t+kNST).kg0
  constexpr int kWaitN = 4 + (kNST - 3) * (XG + 4) + XG + 2;
  constexpr int HF = MF / 2;
#define WS_READ(XF, STAGE, CO, ROW0, CNT)                                                 \
  {                                                                                       \
    const unsigned char* xs_ = smem + (STAGE) * kStageBytes + (ROW0) * (16 * kRowBytes) + fr_row + (CO); \
    _Pragma("unroll") for (int b = 0; b < (CNT); ++b) XF[b] = *(const uint4*)(xs_ + b * 16 * kRowBytes); \
  }
#define WS_MMA(SLOT, KG, XF, ROW0, CNT)                                                   \
  _Pragma("unroll") for (int a = 0; a < NF; ++a)                                          \
    _Pragma("unroll") for (int b = 0; b < (CNT); ++b)                                     \
      acc[a][(ROW0) + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[SLOT][KG][a]), \
                                                          __builtin_bit_cast(bf16x8, XF[b]), acc[a][(ROW0) + b], 0, 0, 0);
  // one ds_read_b128 in front of every two MFMAs (igemm.hip's issue order)
#define WS_INTERLEAVE(CNT)                                                                \
  sfor<(CNT)>([&](auto) __attribute__((always_inline)) {                                  \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                    \
  });                                                                                     \
  __builtin_amdgcn_sched_barrier(0);

  // ---------------- K loop, unrolled over the ring slots; no control flow inside a group of kNST tiles: positions past
  // the end of the slice (the slice length rounded up to a multiple of kNST) multiply zeros by zeros ----------------
  const int ngrp = (nk + kNST - 1) / kNST;
  if constexpr (PIPE == 0) {
    uint4 xP[HF], xQ[HF];
    WS_READ(xP, 0, fr_c0, 0, HF)
    for (int g = 0; g < ngrp; ++g) {
      sfor<kNST>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr int jn = (j + 1) % kNST;
        const int t = g * kNST + j;
        issue_x((j + kNST - 1) % kNST, t + kNST - 1 < nk);
        const bool wvalid = t + kNST < nk;
        WS_READ(xQ, j, fr_c0, HF, HF)
        WS_MMA(j, 0, xP, 0, HF)
        WS_INTERLEAVE(HF)
        WS_READ(xP, j, fr_c1, 0, HF)
        WS_MMA(j, 0, xQ, HF, HF)
        WS_INTERLEAVE(HF)
        WS_LOADW(j, 0, wvalid)
        WS_READ(xQ, j, fr_c1, HF, HF)
        WS_MMA(j, 1, xP, 0, HF)
        WS_INTERLEAVE(HF)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // nobody reads this stage any more
        wait_vmcnt_tied<kWaitN>(w[jn][0][0], w[jn][0][1], w[jn][1][0], w[jn][1][1]);
        __syncthreads();
        WS_READ(xP, jn, fr_c0, 0, HF)
        WS_MMA(j, 1, xQ, HF, HF)
        WS_INTERLEAVE(HF)
        WS_LOADW(j, 1, wvalid)
        wv0 += kTileStep;
        wv1 += kTileStep;
      });
    }
  } else {
    uint4 xA[MF], xB[MF];
    WS_READ(xA, 0, fr_c0, 0, MF)
    for (int g = 0; g < ngrp; ++g) {
      sfor<kNST>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr int jn = (j + 1) % kNST;
        const int t = g * kNST + j;
        issue_x((j + kNST - 1) % kNST, t + kNST - 1 < nk);
        const bool wvalid = t + kNST < nk;
        WS_READ(xB, j, fr_c1, 0, MF)
        WS_MMA(j, 0, xA, 0, MF)
        WS_INTERLEAVE(MF)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // nobody reads this stage any more
        WS_LOADW(j, 0, wvalid)
        wait_vmcnt_tied<kWaitN>(w[jn][0][0], w[jn][0][1], w[jn][1][0], w[jn][1][1]);
        __syncthreads();
        WS_READ(xA, jn, fr_c0, 0, MF)
        WS_MMA(j, 1, xB, 0, MF)
        WS_INTERLEAVE(MF)
        WS_LOADW(j, 1, wvalid)
        wv0 += kTileStep;
        wv1 += kTileStep;
      });
    }
  }
#undef WS_READ
#undef WS_MMA
#undef WS_INTERLEAVE
#undef WS_LOADW
  // every fragment register may still have a (zero-returning) load in flight: they stay allocated until it has landed
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#define WS_KEEP(J) asm volatile("" : "+v"(w[J][0][0]), "+v"(w[J][0][1]), "+v"(w[J][1][0]), "+v"(w[J][1][1]) : : "memory");
  WS_KEEP(0) WS_KEEP(1) WS_KEEP(2)
  if constexpr (kNST > 3) { WS_KEEP(kNST - 1) }
#undef WS_KEEP

  // ---------------- epilogue: the raw fp32 accumulators go to this slice's slab ----------------
#pragma unroll
  for (int b = 0; b < MF; ++b) {
    const int m = m0 + b * 16 + (lane & 15);
    if (m < p.M) {
      float* dst = p.partial + ((size_t)z * p.M + m) * p.N + n0 + wave * 32 + lg * 4;
#pragma unroll
      for (int a = 0; a < NF; ++a) *(f32x4*)(dst + a * 16) = acc[a][b];
    }
  }
}

int g_ws_mode = 0;        // bit 0: on (default OFF: measured level with igemm_kernel); bit 1: 4-wave workgroups (BN = 128, two per CU) instead of 8-wave (BN = 256);
                          // bit 2: whole-k-group X buffers with a 3-slot ring instead of half-step buffers with a 4-slot ring
int g_ws_max_m = 1024;    // launches with more output rows stay on igemm_kernel
int g_ws_min_nk = 40;     // ... and so do launches with fewer K tiles (K = 64 nk)

template <int NWN, int kNST, int PIPE>
int run_ws(const IgemmParams& pin, hipStream_t s) {
  constexpr int BN = 32 * NWN;
  IgemmParams p = pin;
  p.zeros = igemm_zero_page();
  if (!p.zeros) return -3;
  const int mt = (p.M + kBM - 1) / kBM, nt = p.N / BN;
  p.fd_hwo = fastdiv_make(p.Ho * p.Wo);
  p.fd_wo = fastdiv_make(p.Wo);
  p.fd_nt = fastdiv_make(mt);
  p.fd_ntiles = fastdiv_make(mt * nt);
  p.fd_nsplit = fastdiv_make(p.splits);
  p.fd_tpt = fastdiv_make((p.C0 + p.C1) / 64);
  const int nwork = mt * nt * p.splits;
  const size_t lds = (size_t)kNST * kStageBytes;
  auto kern = igemm_ws_kernel<NWN, kNST, PIPE>;
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(nwork), dim3(NWN * 64), lds, s, p);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

void igemm_ws_set_mode(int mode, int max_m, int min_nk) {
  g_ws_mode = mode;
  if (max_m > 0) g_ws_max_m = max_m;
  if (min_nk > 0) g_ws_min_nk = min_nk;
}
int igemm_ws_get_mode() { return g_ws_mode; }

// The launch shapes the kernel takes: bf16, plain-store epilogue (the split-K finish applies bias / residual / SiLU), tap-major
// K, N a multiple of the workgroup's 256 (128) columns, few rows, long K.
bool igemm_ws_ok(const IgemmParams& p, int dtype) {
  if (!(g_ws_mode & 1) || dtype != DT_BF16 || p.epi != EPI_STORE || p.rowstats || p.cm || !p.Wf) return false;
  const int bn = (g_ws_mode & 2) ? 128 : 256;
  if (p.N % bn != 0 || p.M > g_ws_max_m) return false;
  const long nk = (long)p.taps * (p.C0 + p.C1) / 64;
  if (nk < g_ws_min_nk) return false;
  // byte offsets into the fragment-major matrix are 32-bit, bit 31 marks "out of range"
  if ((long)p.N * p.taps * (p.C0 + p.C1) * 2 >= (1l << 31) || p.N % 16 != 0) return false;
  return igemm_ws_splits(p) >= 2;
}

// K slices: one work item per CU (8-wave form) or two (4-wave form), at least 8 K tiles per slice
int igemm_ws_splits(const IgemmParams& p) {
  int ncu = 256;
  {
    int dev = 0;
    hipDeviceProp_t prop;
    static int cached[64] = {};
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!cached[dev]) cached[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    ncu = cached[dev];
  }
  const int bn = (g_ws_mode & 2) ? 128 : 256;
  const int slots = (g_ws_mode & 2) ? 2 * ncu : ncu;
  const long tiles = (long)((p.M + kBM - 1) / kBM) * (p.N / bn);
  const long nk = (long)p.taps * (p.C0 + p.C1) / 64;
  long sp = slots / tiles;
  if (sp > nk / 8) sp = nk / 8;
  if (sp > 32) sp = 32;
  return sp < 1 ? 1 : (int)sp;
}

namespace {
__global__ void pack_ws_kernel(const uint4* __restrict__ w, uint4* __restrict__ out, int nk, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const size_t blk = i >> 6;
    const int kg = (int)(blk & 1);
    const size_t r = blk >> 1;
    const int kt = (int)(r % nk);
    const size_t n16 = r / nk;
    out[i] = w[(n16 * 16 + (lane & 15)) * ((size_t)nk * 8) + (size_t)kt * 8 + kg * 4 + (lane >> 4)];
  }
}
}  // namespace
int launch_pack_ws(const void* w_nk, void* out, int N, int K, hipStream_t s) {
  if (!w_nk || !out || N % 16 != 0 || K % 64 != 0) return -2;
  const size_t total = (size_t)N * (K / 8);
  const size_t blocks = (total + 255) / 256;
  hipLaunchKernelGGL(pack_ws_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, (const uint4*)w_nk, (uint4*)out, K / 64,
                     total);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
// layers that can meet igemm_ws_ok at some map size: wide outputs in whole 256-column tiles, long K
bool igemm_ws_wants(int N, int K, int epi, int dtype) {
  return (g_ws_mode & 1) && dtype == DT_BF16 && epi == EPI_STORE && N >= 1024 && N % 256 == 0 && K % 64 == 0 && K / 64 >= 16 && (long)N * K * 2 < (1l << 31);
}

int launch_igemm_ws(const IgemmParams& p, hipStream_t s) {
  if (p.splits < 2 || !p.partial || !p.Wf) return -2;
  if ((g_ws_mode & 6) == 4) return run_ws<8, 3, 1>(p, s);       // (the 4-wave form of it spills: not built)
  return (g_ws_mode & 2) ? run_ws<4, 4, 0>(p, s) : run_ws<8, 4, 0>(p, s);
}
int igemm_ws_waves() { return (g_ws_mode & 2) ? 4 : 8; }

}  // namespace ldmseg
