// Implicit-GEMM convolution / GEMM for gfx950 (CDNA4), the kernel ~84 % of the
// UNet's FLOPs run through: conv3x3 (stride 1/2, optional folded nearest-x2
// upsample, optional channel-concat of two sources), conv1x1 and Linear, in
// NHWC, bf16 (v_mfma_f32_16x16x32_bf16) or exact f32 (v_mfma_f32_16x16x4_f32).
//
//   out[m][n] = sum_k X[m][k] * W[n][k]      m = (b, oy, ox), k = (tap, channel)
//
// Design (see DESIGN.md "igemm"):
//  * X rows are gathered on the fly: a K tile is 128 B of one tap (64 bf16 /
//    32 f32 channels), so every 16-B chunk a lane fetches is contiguous in HBM
//    and 8 lanes cover one full 128-B line.  Padding / out-of-image taps are
//    predicated to zero; the upsample is an index shift; torch.cat([h,skip])
//    is a pointer switch at a K-tile boundary (all channel counts are
//    multiples of 64).
//  * W is pre-packed [N][K] so both operands are K-contiguous: the same
//    ds_read_b128 pattern feeds either MFMA operand.  Operands are swapped
//    (MFMA A = W, B = X) so a lane ends up holding 4 consecutive n of one m:
//    8-/16-byte stores along the NHWC channel axis and in-lane GEGLU.
//  * LDS tiles are [rows][128 B] with the 16-B chunk index XOR (row & 7):
//    conflict-free for ds_read_b128 over the 16-lane groups of gfx950.
//  * a ring of NST LDS stages (2..4 full K tiles) filled by LDS-DMA (global_load_lds_dwordx4: no VGPR
//    round trip, no ds_write pass); the DMA stream runs NST-1 tiles ahead of the MFMA stream, also
//    across work-item boundaries (persistent grid); counted vmcnt + one barrier per K tile.
//    Out-of-image taps read a 16-B page of zeros.
//  * workgroup id -> (m tile, n tile) is remapped so that each XCD (own L2)
//    owns a contiguous range of tiles.
#include <cstdio>
#include <set>
#include <string>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace ldmseg {

namespace {

constexpr int kThreads = 256;
constexpr int kRowBytes = 128;

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA: 64 lanes x 16 B from per-lane global addresses into LDS at (wave-uniform) lds_dst +
// lane*16.  Issued through inline asm so that hipcc does not track it: with the builtin the
// compiler waits vmcnt(0) before the next ds_read (it cannot prove the DMA target and the tile
// being read are different stages), which serialises DMA and MFMA.  The caller owns the wait:
// s_waitcnt vmcnt(0) + barrier before any wave reads the stage.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off"
      :
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
// same, address = wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit offset: no VALU per tile
__device__ __forceinline__ void glds16_sbase(unsigned voff, const void* sbase, unsigned lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 3\n\t"   // M0 write -> LDS-DMA needs 1 wait state; an SGPR base written by SALU / v_readfirstlane needs 5
                       // before a VMEM instruction reads it, and the hazard recognizer does not look inside inline asm
                       // (a register-load variant of this helper went to wild addresses without them)
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

struct RowInfo {
  int pix_base;  // b * Hi * Wi
  int iy0, ix0;  // top-left input coordinate (logical, before upsample shift)
};

// Ablation switches (skip DMA / MFMA / epilogue phases ...) exist only in builds with
// -DLDMSEG_IGEMM_ABLATE: every one of them is a branch in code that runs cold once per launch.
#ifdef LDMSEG_IGEMM_ABLATE
#define DBG(p, bit) ((p).dbg & (bit))
#define STAMP(p, slot)                                                                              \
  if ((p).ts && threadIdx.x == 0) (p).ts[(size_t)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memtime();
#elif defined(LDMSEG_IGEMM_STAMP)     // stamps only (no phase-skipping branches: the 12-wave tiles keep their register allocation)
#define DBG(p, bit) 0
#define STAMP(p, slot)                                                                              \
  if ((p).ts && threadIdx.x == 0) (p).ts[(size_t)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memtime();
#else
#define DBG(p, bit) 0
#define STAMP(p, slot)
#endif
// stamp builds: slot 14 / 15 = the 100 MHz wall clock (s_memrealtime: one counter for the whole device, comparable ACROSS XCDs, unlike
// s_memtime) at workgroup start / end, slot 5 = the XCD the workgroup ran on (tools/launch_boundary.py)
#if defined(LDMSEG_IGEMM_ABLATE) || defined(LDMSEG_IGEMM_STAMP)
#define STAMP_WALL(p, slot)                                                                         \
  if ((p).ts && threadIdx.x == 0) (p).ts[(size_t)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memrealtime();
#define STAMP_XCC(p)                                                                                \
  if ((p).ts && threadIdx.x == 0) {                                                                 \
    unsigned xcc_;                                                                                  \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                             \
    (p).ts[(size_t)blockIdx.x * 16 + 5] = (xcc_ & 0xfu) + 1u;                                        \
  }
#else
#define STAMP_WALL(p, slot)
#define STAMP_XCC(p)
#endif

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// Bytes of the row-major epilogue staging area (one 16-row x WTN fp32 block per wave).  It normally
// borrows the ring stage that is free at the end of an item; a tile whose stage is smaller than
// that gets a dedicated area behind the ring.
template <int BM, int BN, int WAVES_M, int WAVES_N>
constexpr int epi_stage_bytes() { return WAVES_M * WAVES_N * 16 * ((BN / WAVES_N) * 4 + 16); }
template <int BM, int BN, int WAVES_M, int WAVES_N>
constexpr bool epi_stage_dedicated() { return epi_stage_bytes<BM, BN, WAVES_M, WAVES_N>() > (BM + BN) * kRowBytes; }

// Persistent implicit-GEMM.  A fixed grid of workgroups (two per CU) walks the work items
// (output tile, K slice); NST LDS stages (full K tiles) form a ring and the LDS-DMA stream runs NST-1
// tiles ahead of the MFMA stream ACROSS item boundaries, so the DMA round trip at the start of an
// item and the epilogue at its end overlap with neighbouring items instead of idling the CU.
// Per K tile every wave: issues the DMA of stream position +NST-1 into the stage read one tile ago,
// computes the current tile, waits with a COUNTED vmcnt until the next tile has landed (later ones
// stay in flight) and passes the single barrier.
// LNF: the launch carries a folded LayerNorm (IgemmParams::rowstats / c1).  A separate instantiation, because the extra
// epilogue code is not free for the launches that do not use it: these kernels are ~55 KB of code each, short launches
// run their epilogue cold out of HBM (1.6 GB of weights stream through L2 per forward), and the 64-row-tile 1x1 GEMMs
// measured +10 us per launch with the LayerNorm branch merely present.
// CM: channel-major K order of a 3x3 conv (IgemmParams::cm).  A separate instantiation as well: carried as a run-time
// branch it cost every launch of the family 2-5 % (measured; scalar registers and code in the K loop's DMA step).
// XT: the launch carries an extra centre tap (IgemmParams::src2 / C2 / src3 / C3).  A separate instantiation for the same reason.
// UP4: conv3x3(nearest_x2(x)) as four 2x2-tap phase convs on the low-resolution map (IgemmParams::up4).  Separate as well.
// CF: K slices are reduced inside the launch (IgemmParams::cf_ctr; the block behind the item loop).  A separate instantiation like
// the others: with the finish code merely present every instantiation's register allocation moved by a few VGPRs (three crossed an
// occupancy step) and whole forwards measured +0.6 %.
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int NST, bool PIPE, int LDR, bool LNF, bool CM = false, bool XT = false,
          bool UP4 = false, bool CF = false>
__global__ __launch_bounds__((WAVES_M * WAVES_N + LDR) * 64, (WAVES_M * WAVES_N + LDR) > 8 ? 3 : 2) void igemm_kernel(const IgemmParams p) {   // <=256 regs at 2 waves/SIMD; 12-wave workgroups (8 compute + 4 loader waves) need 3 per SIMD: <=168
  constexpr int NW = WAVES_M * WAVES_N;   // 4 waves (128-row tiles, 2 workgroups/CU) or 8 (256-row tiles)
  static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
  // LDR > 0: producer / consumer split.  Waves NW..NW+LDR-1 only run the LDS-DMA stream (they sit blocked in the
  // 64 B/clk DMA queue most of the time), waves 0..NW-1 only read fragments and issue MFMAs - so the matrix pipe of a
  // SIMD is fed by a wave that never stalls on a DMA issue.  LDR == 0: every wave loads its share and computes.
  constexpr int NLW = LDR ? LDR : NW;     // waves that run the DMA stream
  static_assert(LDR == 0 || PIPE, "loader waves exist only in the pipelined loop");
  constexpr int BKE = kRowBytes / (int)sizeof(T);  // K elements per tile
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MF = WTM / 16, NF = WTN / 16;
  constexpr int kStageBytes = (BM + BN) * kRowBytes;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_ldr = LDR ? wave_id >= NW : true;      // runs the DMA stream
  const bool is_cmp = LDR ? wave_id < NW : true;       // reads fragments, MFMAs, epilogue
  const int wave = LDR ? (wave_id < NW ? wave_id : wave_id - NW) : wave_id;   // index within its role
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // ---- work items: (m tile, n tile, K slice), n fastest.  Round j of this workgroup is item
  // j*G + b', b' = XCD-aware bijective remap of blockIdx within the grid: every XCD (own L2) works
  // on a contiguous run of tiles in each round.
  const int NT = p.N / BN;
  const int MT = (p.M + BM - 1) / BM;
  const int ntiles = MT * NT;
  const int nsplit = p.splits > 1 ? p.splits : 1;
  const int nwork = ntiles * nsplit;
  const int G = gridDim.x;
  int first_item;
  {
    const int bid = blockIdx.x;
    const int q = G >> 3, r = G & 7, xcd = bid & 7, idx = bid >> 3;
    first_item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  if (first_item >= nwork) return;

  const int Ctot = p.C0 + p.C1;
  const int K = p.taps * Ctot + (XT ? p.C2 + p.C3 : 0);
  const int nk_total = K / BKE;
  const int nk_main = p.taps * (Ctot / BKE);       // K tiles of the 3x3 taps; XT: the extra tap's tiles follow
  const int Hlog = p.up ? 2 * p.Hi : p.Hi;
  const int Wlog = p.up ? 2 * p.Wi : p.Wi;
  const int pad = (p.taps == 9) ? (p.pad >= 0 ? p.pad : 1) : 0;
  const int HWo = p.Ho * p.Wo;
  // UP4: virtual row (phase, image, i, j) -> row (image, 2i + py, 2j + px) of the upsampled output
  auto out_row = [&](int m) __attribute__((always_inline)) -> size_t {
    if constexpr (UP4) {
      const int ph = fd_div(m, p.fd_mphase), mr = m - ph * p.fd_mphase.d;
      const int b = fd_div(mr, p.fd_hwo), rem = mr - b * HWo;
      const int oy = fd_div(rem, p.fd_wo), ox = rem - oy * p.Wo;
      return ((size_t)b * 2 * p.Ho + 2 * oy + (ph >> 1)) * (size_t)(2 * p.Wo) + 2 * ox + (ph & 1);
    } else {
      return (size_t)m;
    }
  };

  // ---- direct-to-LDS staging (global_load_lds_dwordx4): one wave instruction fills one 8-row
  // group (1 KiB, lane l -> row l>>3, physical chunk l&7).  The XOR swizzle therefore lives on the
  // SOURCE side: the lane fetches logical chunk (l&7)^(row&7).  Rows 0..BM-1 of a stage are X,
  // rows BM.. are W; group g = i*NW + wave belongs to wave (g % NW).
  constexpr int XG = BM / 8 / NLW;             // X groups per loading wave per tile
  constexpr int WGN = BN / 8;                  // W groups per tile
  constexpr int WG = (WGN + NLW - 1) / NLW;    // W rounds; the last one may cover only waves < WREM
  constexpr int WREM = WGN % NLW;
  static_assert(BM % (8 * NLW) == 0 && BN % 8 == 0, "tile rows / loader mismatch");
  const int ld_r = lane >> 3;                  // row within the 8-row group (== row & 7)
  const int ld_j = (lane & 7) ^ ld_r;          // logical 16-B chunk this lane fetches
  const bool w_last = (WREM == 0) || (wave < WREM);   // owns a group in the last W round (wave-uniform)
  const unsigned char* zpage = (const unsigned char*)p.zeros;
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane(
      (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem) + (unsigned)wave * 1024u);
  unsigned woff[WG];                           // per-lane W row offsets relative to the item's W tile
#pragma unroll
  for (int i = 0; i < WG; ++i)
    woff[i] = (unsigned)(((size_t)((i * NLW + wave) * 8 + ld_r) * K) * sizeof(T)) + (unsigned)ld_j * 16u;

  auto item_range = [&](int item, int& m0, int& n0, int& z, int& kb, int& ke) __attribute__((always_inline)) {
    z = fd_div(item, p.fd_ntiles);                       // (all divisions here and in item_setup: see FastDiv, kernels.h)
    const int tile = item - z * ntiles;
    const int mt = fd_div(tile, p.fd_nt), nt = tile - mt * NT;
    m0 = mt * BM;
    n0 = nt * BN;
    kb = fd_div(nk_total * z, p.fd_nsplit);              // nk_total * splits < 2^31 (launch_igemm checks the shape)
    ke = fd_div(nk_total * (z + 1), p.fd_nsplit);
  };

  // ================= fetch stream state (runs NST-1 K tiles ahead of the compute stream) =================
  int f_item = first_item, f_kt = 0, f_kend = 0;   // item / next K tile / end of its slice
  bool f_done = false;
  RowInfo ri[XG];
  const unsigned char* wtile0 = nullptr;
  int f_tap = 0, f_cc = 0, f_pixb = 0;
  const unsigned char* rowptr[XG];
  unsigned rowinc[XG];
  bool need_setup = true;

  auto item_setup = [&]() __attribute__((always_inline)) {
    int m0, n0, z;
    item_range(f_item, m0, n0, z, f_kt, f_kend);
    int ph = 0;                                           // UP4: phase (py, px) of this tile (tiles never straddle phases)
    if constexpr (UP4) ph = fd_div(m0, p.fd_mphase);
#pragma unroll
    for (int i = 0; i < XG; ++i) {
      const int m = m0 + (i * NLW + wave) * 8 + ld_r;
      if (m < p.M) {
        const int mr = UP4 ? m - ph * p.fd_mphase.d : m;
        const int b = fd_div(mr, p.fd_hwo), rem = mr - b * HWo;
        const int oy = fd_div(rem, p.fd_wo), ox = rem - oy * p.Wo;
        ri[i].pix_base = b * p.Hi * p.Wi;
        if constexpr (UP4) {                              // window rows {i - 1, i} for py = 0, {i, i + 1} for py = 1
          ri[i].iy0 = oy - 1 + (ph >> 1);
          ri[i].ix0 = ox - 1 + (ph & 1);
        } else {
          ri[i].iy0 = oy * p.stride - pad;
          ri[i].ix0 = ox * p.stride - pad;
        }
      } else {
        ri[i].pix_base = 0;
        ri[i].iy0 = -(1 << 28);
        ri[i].ix0 = 0;
      }
    }
    wtile0 = (const unsigned char*)p.W + ((size_t)n0 + (UP4 ? (size_t)ph * p.N : 0)) * K * sizeof(T);
    if constexpr (CM) {                                  // K order [channel tile][tap]: fd_tpt divides by 9
      const int ct = fd_div(f_kt, p.fd_tpt);
      f_tap = f_kt - ct * 9;
      f_cc = ct * BKE;
#pragma unroll
      for (int i = 0; i < XG; ++i) {                     // bit t: tap t of this row lies inside the image (rows past M: none)
        unsigned mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int uy = ri[i].iy0 + t / 3, ux = ri[i].ix0 + t % 3;
          mk |= ((uy >= 0) & (uy < p.Hi) & (ux >= 0) & (ux < p.Wi)) ? (1u << t) : 0u;
        }
        rowinc[i] = mk;
      }
    } else {
      const int tiles_per_tap = Ctot / BKE;
      if (XT && f_kt >= nk_main) {          // the slice starts inside the extra tap
        f_tap = 9;
        f_cc = (f_kt - nk_main) * BKE;
      } else {
        f_tap = fd_div(f_kt, p.fd_tpt);
        f_cc = (f_kt - f_tap * tiles_per_tap) * BKE;
      }
    }
    need_setup = true;
  };
  // Channel-major K order (IgemmParams::cm; 3x3, stride 1, no upsample): the nine taps of one 64-channel tile are
  // consecutive K tiles, so a row's nine gathers hit lines that are at most two image rows apart - they come out of the
  // L2 instead of making nine passes over the whole map.  rowptr[] then holds the (possibly out-of-image) address of the
  // current tap, stepped by a wave-uniform stride per tile; rowinc[] holds the row's tap validity mask.
  auto seg_setup_cm = [&]() __attribute__((always_inline)) {
    const int ky = f_tap / 3, kx = f_tap - ky * 3;
    const unsigned char* sbase;
    int cs, coff;
    if (f_cc < p.C0) { sbase = (const unsigned char*)p.src0; cs = p.C0; coff = f_cc; }
    else { sbase = (const unsigned char*)p.src1; cs = p.C1; coff = f_cc - p.C0; }
    f_pixb = cs * (int)sizeof(T);
#pragma unroll
    for (int i = 0; i < XG; ++i) {
      const long long pix = (long long)ri[i].pix_base + (long long)(ri[i].iy0 + ky) * p.Wi + (ri[i].ix0 + kx);   // (rows past M: iy0 = -2^28)
      rowptr[i] = sbase + (pix * cs + coff) * (long long)sizeof(T) + ld_j * 16;
    }
  };
  // A "segment" is a run of K tiles inside one (tap, source tensor): there the gather address of a
  // row only advances by 128 B per tile.  Row pointers are set up once per segment; rows whose tap
  // falls outside the image sit on the zero page with stride 0.
  auto seg_setup = [&]() __attribute__((always_inline)) {
    const bool xtap = XT && f_tap == 9;           // the extra tap reads the output pixel itself (window offset = pad)
    const int ky = UP4 ? (f_tap >> 1) : (xtap ? pad : ((p.taps == 9) ? f_tap / 3 : 0));
    const int kx = UP4 ? (f_tap & 1) : (xtap ? pad : ((p.taps == 9) ? f_tap - ky * 3 : 0));
    const unsigned char* sbase;
    int cs, coff;
    if (xtap) {
      if (f_cc < p.C2) { sbase = (const unsigned char*)p.src2; cs = p.C2; coff = f_cc; }
      else { sbase = (const unsigned char*)p.src3; cs = p.C3; coff = f_cc - p.C2; }
    } else if (f_cc < p.C0) { sbase = (const unsigned char*)p.src0; cs = p.C0; coff = f_cc; }
    else { sbase = (const unsigned char*)p.src1; cs = p.C1; coff = f_cc - p.C0; }
#pragma unroll
    for (int i = 0; i < XG; ++i) {
      const int uy = ri[i].iy0 + ky, ux = ri[i].ix0 + kx;
      const bool inb = (uy >= 0) & (uy < Hlog) & (ux >= 0) & (ux < Wlog);
      const int iy = p.up ? (uy >> 1) : uy, ix = p.up ? (ux >> 1) : ux;
      const size_t off = ((size_t)(ri[i].pix_base + iy * p.Wi + ix) * cs + coff) * sizeof(T) + ld_j * 16;
      rowptr[i] = inb ? sbase + off : zpage;
      rowinc[i] = inb ? (unsigned)kRowBytes : 0u;
    }
  };
  // issue the DMA of the next stream position into `stage`; returns false when the stream is exhausted
  auto fetch_next = [&](int stage) __attribute__((always_inline)) -> bool {
    if (f_done) return false;
    if (f_kt == f_kend) {                      // slice finished: move to this workgroup's next item
      f_item += G;
      if (f_item >= nwork) { f_done = true; return false; }
      item_setup();
    }
    // (values below are wave-uniform; readfirstlane makes that provable so they can live in SGPRs)
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_wave + (unsigned)stage * (unsigned)kStageBytes);
#ifdef LDMSEG_IGEMM_SKIPX     // probe: X bytes of a halo-resident conv (same DMA instruction count, 1 lane instead of 64 on 5 of 6 tiles)
    const bool xfull = (p.taps != 9) || (f_kt % LDMSEG_IGEMM_SKIPX) == 0;
#else
    constexpr bool xfull = true;
#endif
    if constexpr (CM) {
      if (need_setup) seg_setup_cm();
      const int tap = __builtin_amdgcn_readfirstlane(f_tap);
      // stride to the next stream position: next tap in the row, first tap of the next image row, or back to tap 0 of
      // the next channel tile
      const int kx = tap - (tap / 3) * 3;
      const long long step = __builtin_amdgcn_readfirstlane(
          tap == 8 ? kRowBytes - (2 * p.Wi + 2) * f_pixb : (kx == 2 ? (p.Wi - 2) * f_pixb : f_pixb));
#pragma unroll
      for (int i = 0; i < XG; ++i) {
        const unsigned char* src = ((rowinc[i] >> tap) & 1u) ? rowptr[i] : zpage;
        if (xfull || lane == 0) glds16(src, dst + i * (NLW * 1024));
        rowptr[i] += step;
      }
    } else {
      if (need_setup) seg_setup();
#pragma unroll
      for (int i = 0; i < XG; ++i) {
        if (xfull || lane == 0) glds16(rowptr[i], dst + i * (NLW * 1024));
        rowptr[i] += rowinc[i];
      }
    }
    const unsigned long long wt_u = (unsigned long long)(uintptr_t)(wtile0 + (size_t)f_kt * kRowBytes);
    const unsigned wt_lo = __builtin_amdgcn_readfirstlane((unsigned)wt_u);
    const unsigned wt_hi = __builtin_amdgcn_readfirstlane((unsigned)(wt_u >> 32));
    const unsigned char* wt = (const unsigned char*)(uintptr_t)(((unsigned long long)wt_hi << 32) | wt_lo);
    const unsigned wdst = dst + BM * kRowBytes;
#pragma unroll
    for (int i = 0; i < WG; ++i)
      if ((i + 1 < WG || w_last) && !DBG(p, 2)) glds16_sbase(woff[i], wt, wdst + i * (NLW * 1024));
    ++f_kt;
    if constexpr (CM) {
      need_setup = false;
      if (++f_tap == 9) { f_tap = 0; f_cc += BKE; need_setup = (f_cc == p.C0); }
    } else {
      f_cc += BKE;
      if (XT && f_tap == 9) {
        need_setup = (f_cc == p.C2);             // switch to the second tensor of the extra tap
      } else {
        if (f_cc == Ctot) { f_cc = 0; ++f_tap; }
        need_setup = (f_cc == 0) | (f_cc == p.C0);
      }
    }
    return true;
  };
  // wait until at most AHEAD of this wave's per-tile DMA batches are still in flight
  auto wait_dma = [&](auto ahead_tag) __attribute__((always_inline)) {
    constexpr int AHEAD = decltype(ahead_tag)::value;
    if constexpr (AHEAD == 0) {
      wait_vmcnt<0>();
    } else if constexpr (WREM == 0) {
      wait_vmcnt<AHEAD*(XG + WG)>();
    } else {
      if (w_last) wait_vmcnt<AHEAD*(XG + WG)>();
      else wait_vmcnt<AHEAD*(XG + WG - 1)>();
    }
  };

  f32x4 acc[NF][MF];
#pragma unroll
  for (int a = 0; a < NF; ++a)
#pragma unroll
    for (int b = 0; b < MF; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row = frag*16 + (lane&15); chunk = kg*4 + (lane>>4), XOR (row&7)
  const int fr_row = (lane & 15) * kRowBytes;
  const int fr_c0 = (((lane >> 4)) ^ (lane & 7)) * 16;
  const int fr_c1 = (((lane >> 4) + 4) ^ (lane & 7)) * 16;
  const int lg = lane >> 4;

  // Row-major epilogue for the plain store (EPI_STORE): the MFMA layout gives a lane 4 channels of one
  // row, i.e. 8-byte stores in 32-byte runs - measured 1.4 TB/s and most of a short-K GEMM's time.
  // Instead each wave parks a 16-row x WTN block (fp32, after bias) in the LDS stage that is free at
  // the end of an item (the one the next DMA will target), then every lane takes 16 output bytes of
  // one row: 16-byte residual loads and 16-byte stores covering whole rows of the wave's tile.
  auto epilogue_rows = [&](int m0, int n0, int stage_free) __attribute__((always_inline)) {
    constexpr int E = 16 / (int)sizeof(T);              // output elements per 16-byte chunk
    constexpr int SROW = WTN * 4 + 16;                  // staged row stride (bytes), padded
    constexpr int CPR = WTN / E;                        // chunks per row
    constexpr int NCH = 16 * CPR;                       // chunks per 16-row block
    unsigned char* stg = (epi_stage_dedicated<BM, BN, WAVES_M, WAVES_N>() ? smem + NST * kStageBytes
                                                                          : smem + stage_free * kStageBytes) +
                         wave * (16 * SROW);
    const int lq = lane & 15;
    // bias of this lane's NF channel quads: loaded ONCE per item, unconditionally and back to back
    // (a load per fragment behind its own branch costs one exposed L2 round trip each)
    const float* biasp = p.bias ? p.bias : (const float*)p.zeros;
    const int nl = n0 + wn * WTN + lg * 4;
    f32x4 bv[NF];
#pragma unroll
    for (int a = 0; a < NF; ++a) bv[a] = *(const f32x4*)(biasp + nl + a * 16);
    f32x4 c1v[LNF ? NF : 1];
    if constexpr (LNF) {
#pragma unroll
      for (int a = 0; a < NF; ++a) c1v[a] = *(const f32x4*)(p.c1 + nl + a * 16);
    }
    // Everything a 16-row block needs from global memory is requested one block ahead (the first block's requests
    // go out here, before any staging): residual chunks, LayerNorm row statistics.  Issued at their point of use they
    // cost one exposed L2 / HBM round trip per block, MF times per item, in a tail nothing else overlaps.
    constexpr int NI = (NCH + 63) / 64;
    const int m_wave = m0 + wm * WTM;
    const bool has_res = p.resid != nullptr && !DBG(p, 128);
    auto load_res = [&](int b, uint4 (&r)[NI]) __attribute__((always_inline)) {
      if (!has_res) return;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = lane + i * 64;
        const int row = c / CPR, cc = c - row * CPR;
        const int m = m_wave + b * 16 + row, n = n0 + wn * WTN + cc * E;
        if (c < NCH && m < p.M && n < p.n_valid) r[i] = *(const uint4*)((const T*)p.resid + (size_t)m * p.ldr + n);
      }
    };
    auto load_st = [&](int b) __attribute__((always_inline)) -> float2 {
      const int m = m_wave + b * 16 + lq;
      return *(const float2*)(p.rowstats + (size_t)(m < p.M ? m : p.M - 1) * 2);
    };
    uint4 rcur[NI], rnext[NI];
    float2 st_cur = make_float2(0.f, 1.f), st_next = st_cur;
    load_res(0, rcur);
    if constexpr (LNF) st_cur = load_st(0);
    // time-embedding row: one image per tile (every map of 16x16 and up) -> it is the same for all rows, add it to the bias
    // quads once; tiles that span images (8x8 maps) fetch it per block
    bool rb_rows = false;
    if (!LNF && p.rowbias) {
      const int m_last = (m0 + BM <= p.M ? m0 + BM : p.M) - 1;
      const int img0 = fd_div(m0, p.fd_hwo);
      if (img0 == fd_div(m_last, p.fd_hwo)) {
        const float* rbp = p.rowbias + (size_t)img0 * p.rb_stride + nl;
#pragma unroll
        for (int a = 0; a < NF; ++a) bv[a] += *(const f32x4*)(rbp + a * 16);
      } else {
        rb_rows = true;
      }
    }
    // One 16-row block of the wave's tile.  BI = accumulator slot that holds it.
    auto row_block = [&](int b, auto bidx) __attribute__((always_inline)) {
      constexpr int BI = decltype(bidx)::value;
      const int mb = m_wave + b * 16;
      STAMP(p, 6 + 2 * b)
      {
        if constexpr (LNF) {                               // folded LayerNorm: rstd * (acc - mean * c1) + c2
#pragma unroll
          for (int a = 0; a < NF; ++a)
            *(f32x4*)(stg + lq * SROW + (a * 16 + lg * 4) * 4) = (acc[a][BI] - c1v[a] * st_cur.x) * st_cur.y + bv[a];
        } else if (rb_rows) {                              // time-embedding row of this pixel's image
          const int m = mb + lq;
          const int bimg = fd_div(m < p.M ? m : p.M - 1, p.fd_hwo);
          const float* rbp = p.rowbias + (size_t)bimg * p.rb_stride + nl;
          f32x4 rb[NF];
#pragma unroll
          for (int a = 0; a < NF; ++a) rb[a] = *(const f32x4*)(rbp + a * 16);
#pragma unroll
          for (int a = 0; a < NF; ++a)
            *(f32x4*)(stg + lq * SROW + (a * 16 + lg * 4) * 4) = acc[a][BI] + bv[a] + rb[a];
        } else {
#pragma unroll
          for (int a = 0; a < NF; ++a)
            *(f32x4*)(stg + lq * SROW + (a * 16 + lg * 4) * 4) = acc[a][BI] + bv[a];
        }
      }
      // (requested after the staging writes: this block's accumulators are dead by now, which is what makes room for them
      // in the 168-register budget of the 12-wave tiles)
      if (b + 1 < MF) {
        load_res(b + 1, rnext);
        if constexpr (LNF) st_next = load_st(b + 1);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's block is in LDS (per-wave region)
      STAMP(p, 7 + 2 * b)
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = lane + i * 64;
        if (c < NCH) {
          const int row = c / CPR, cc = c - row * CPR;
          const int m = mb + row;
          const int n = n0 + wn * WTN + cc * E;
          if (m < p.M && n < p.n_valid) {
            float v[E];
            const unsigned char* sp = stg + row * SROW + cc * E * 4;
#pragma unroll
            for (int q = 0; q < E / 4; ++q) {
              const f32x4 t = *(const f32x4*)(sp + q * 16);
              v[q * 4 + 0] = t[0]; v[q * 4 + 1] = t[1]; v[q * 4 + 2] = t[2]; v[q * 4 + 3] = t[3];
            }
            if (has_res) {
              float r[E];
              Chunk<T>::unpack(rcur[i], r);
#pragma unroll
              for (int e = 0; e < E; ++e) v[e] += r[e];
            }
            if (p.silu) {
#pragma unroll
              for (int e = 0; e < E; ++e) v[e] = silu_f(v[e]);
            }
            if (!DBG(p, 64)) *(uint4*)((T*)p.out + out_row(m) * p.ldo + n) = Chunk<T>::pack(v);
            else asm volatile("" ::"v"(v[0]), "v"(v[E - 1]));
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next block overwrites
#pragma unroll
      for (int i = 0; i < NI; ++i) rcur[i] = rnext[i];
      st_cur = st_next;
    };
    // Two-block tiles keep the loop ROLLED (the current block is always slot 0, the other rotates down):
    // measured -5..10 % on the 128-row conv launches, whose single item per workgroup runs this code cold.
    // Four-block tiles run several items per workgroup; there the rotation costs more than it saves.
    if constexpr (MF <= 2) {
#pragma unroll 1
      for (int b = 0; b < MF; ++b) {
        row_block(b, std::integral_constant<int, 0>{});
#pragma unroll
        for (int a = 0; a < NF; ++a)
#pragma unroll
          for (int j = 0; j + 1 < MF; ++j) acc[a][j] = acc[a][j + 1];
      }
    } else {
      static_for<MF>([&](auto bi) __attribute__((always_inline)) { row_block(decltype(bi)::value, bi); });
    }
  };

  // GEGLU through the same staging: value (fragment a) and gate (a+1) columns of 16 output channels sit in
  // the same lane (16-row interleaved weight packing), so the product is formed in registers, parked
  // as a 16-row x WTN/2 fp32 block and leaves as whole 16-byte chunks of output rows.
  auto epilogue_geglu_rows = [&](int m0, int n0, int stage_free) __attribute__((always_inline)) {
    if constexpr (NF % 2 == 0) {
      constexpr int E = 16 / (int)sizeof(T);
      constexpr int WO = WTN / 2;                       // output channels of the wave's block
      constexpr int SROW = WO * 4 + 16;
      constexpr int CPR = WO / E;
      constexpr int NCH = 16 * CPR;
      static_assert(WO % E == 0, "GEGLU block / chunk mismatch");
      unsigned char* stg = (epi_stage_dedicated<BM, BN, WAVES_M, WAVES_N>() ? smem + NST * kStageBytes
                                                                            : smem + stage_free * kStageBytes) +
                           wave * (16 * SROW);
      const int lq = lane & 15;
      const float* biasp = p.bias ? p.bias : (const float*)p.zeros;
      const int nl = n0 + wn * WTN + lg * 4;
      f32x4 bv[NF];
#pragma unroll
      for (int a = 0; a < NF; ++a) bv[a] = *(const f32x4*)(biasp + nl + a * 16);
      f32x4 c1v[LNF ? NF : 1];
      if constexpr (LNF) {
#pragma unroll
        for (int a = 0; a < NF; ++a) c1v[a] = *(const f32x4*)(p.c1 + nl + a * 16);
      }
      const int oc0 = (n0 + wn * WTN) >> 1;
      // folded LayerNorm: rstd * (acc - mean * c1) + c2; the row statistics of a block are requested one block ahead
      auto load_st = [&](int b) __attribute__((always_inline)) -> float2 {
        const int m = m0 + wm * WTM + b * 16 + lq;
        return *(const float2*)(p.rowstats + (size_t)(m < p.M ? m : p.M - 1) * 2);
      };
      float2 st = make_float2(0.f, 1.f), st_next = st;
      if constexpr (LNF) st = load_st(0);
      auto row_block = [&](int b, auto bidx) __attribute__((always_inline)) {
        constexpr int BI = decltype(bidx)::value;
        const int mb = m0 + wm * WTM + b * 16;
        if constexpr (LNF) { if (b + 1 < MF) st_next = load_st(b + 1); }
#pragma unroll
        for (int a = 0; a < NF; a += 2) {
          f32x4 av = acc[a][BI], gv = acc[a + 1][BI];
          if constexpr (LNF) {
            av = (av - c1v[a] * st.x) * st.y;
            gv = (gv - c1v[a + 1] * st.x) * st.y;
          }
          av += bv[a];
          gv += bv[a + 1];
          f32x4 o;
          if constexpr (sizeof(T) == 2) {
            o = DBG(p, 128) ? av * gv : av * gelu_erf_bf16_f4(gv);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = av[r] * gelu_erf_f(gv[r]);
          }
          *(f32x4*)(stg + lq * SROW + ((a >> 1) * 16 + lg * 4) * 4) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < (NCH + 63) / 64; ++i) {
          const int c = lane + i * 64;
          if (c < NCH) {
            const int row = c / CPR, cc = c - row * CPR;
            const int m = mb + row;
            if (m < p.M) {
              float v[E];
              const unsigned char* sp = stg + row * SROW + cc * E * 4;
#pragma unroll
              for (int q = 0; q < E / 4; ++q) {
                const f32x4 t = *(const f32x4*)(sp + q * 16);
                v[q * 4 + 0] = t[0]; v[q * 4 + 1] = t[1]; v[q * 4 + 2] = t[2]; v[q * 4 + 3] = t[3];
              }
              if (!DBG(p, 64)) *(uint4*)((T*)p.out + (size_t)m * p.ldo + oc0 + cc * E) = Chunk<T>::pack(v);
              else asm volatile("" ::"v"(v[0]), "v"(v[E - 1]));
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        st = st_next;
      };
      static_for<MF>([&](auto bi) __attribute__((always_inline)) { row_block(decltype(bi)::value, bi); });
    }
  };

  auto epilogue = [&](int m0, int n0, int zc) __attribute__((always_inline)) {
  f32x4 gbias[NF];   // GEGLU: bias of the lane's value / gate quads, loaded once per item
  if (p.epi == EPI_GEGLU) {
    const float* biasp = p.bias ? p.bias : (const float*)p.zeros;
#pragma unroll
    for (int a = 0; a < NF; ++a) gbias[a] = *(const f32x4*)(biasp + n0 + wn * WTN + a * 16 + lg * 4);
  }
#pragma unroll
  for (int b = 0; b < MF; ++b) {
    const int m = m0 + wm * WTM + b * 16 + (lane & 15);
    if (m >= p.M) continue;
    const int bimg = fd_div(m, p.fd_hwo);
    float2 st = make_float2(0.f, 1.f);                     // folded LayerNorm: rstd * (acc - mean * c1) + c2
    if constexpr (LNF) st = *(const float2*)(p.rowstats + (size_t)m * 2);
    if (p.epi == EPI_GEGLU) {
      if constexpr (NF % 2 == 0) {
#pragma unroll
        for (int a = 0; a < NF; a += 2) {
          const int n = n0 + wn * WTN + a * 16 + lg * 4;
          const int oc = ((n0 + wn * WTN + a * 16) >> 1) + lg * 4;
          f32x4 c1a = f32x4{0.f, 0.f, 0.f, 0.f}, c1g = c1a;
          if constexpr (LNF) { c1a = *(const f32x4*)(p.c1 + n); c1g = *(const f32x4*)(p.c1 + n + 16); }
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float av = (acc[a][b][r] - c1a[r] * st.x) * st.y + gbias[a][r];
            const float gv = (acc[a + 1][b][r] - c1g[r] * st.x) * st.y + gbias[a + 1][r];
            v[r] = av * gelu_erf_f(gv);
          }
          T* o = (T*)p.out + (size_t)m * p.ldo + oc;
          if constexpr (sizeof(T) == 2) {
            *(uint2*)o = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
          } else {
            *(f32x4*)o = f32x4{v[0], v[1], v[2], v[3]};
          }
        }
      }
      continue;
    }
#pragma unroll
    for (int a = 0; a < NF; ++a) {
      const int n = n0 + wn * WTN + a * 16 + lg * 4;
      if (n >= p.n_valid) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[a][b][r];
      if constexpr (LNF) {
        const f32x4 c1q = *(const f32x4*)(p.c1 + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (v[r] - c1q[r] * st.x) * st.y;
      }
      if (p.bias) {
        const f32x4 bv = *(const f32x4*)(p.bias + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bv[r];
      }
      if (p.rowbias) {
        const f32x4 bv = *(const f32x4*)(p.rowbias + (size_t)bimg * p.rb_stride + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bv[r];
      }
      if (p.epi == EPI_STORE) {
        if (p.resid) {
          // 4 consecutive channels of one row: a single 8-/16-byte load
          const T* rp = (const T*)p.resid + (size_t)m * p.ldr + n;
          if constexpr (sizeof(T) == 2) {
            const uint2 rr = *(const uint2*)rp;
            v[0] += bits_f32(rr.x << 16); v[1] += bits_f32(rr.x & 0xffff0000u);
            v[2] += bits_f32(rr.y << 16); v[3] += bits_f32(rr.y & 0xffff0000u);
          } else {
            const f32x4 rr = *(const f32x4*)rp;
            v[0] += rr[0]; v[1] += rr[1]; v[2] += rr[2]; v[3] += rr[3];
          }
        }
        if (p.silu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
        }
        T* o = (T*)p.out + (size_t)m * p.ldo + n;
        if constexpr (sizeof(T) == 2) {
          *(uint2*)o = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        } else {
          *(f32x4*)o = f32x4{v[0], v[1], v[2], v[3]};
        }
      } else if (p.epi == EPI_ROWS_F32) {      // fp32 row-major regardless of the compute dtype (attention scores)
        *(f32x4*)((float*)p.out + (size_t)m * p.ldo + n) = f32x4{v[0], v[1], v[2], v[3]};
      } else if (p.epi == EPI_NCHW_F32) {
        const int pix = m - bimg * HWo;
        float* o = (float*)p.out + ((size_t)bimg * p.n_valid + n) * HWo + pix;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < p.n_valid) o[(size_t)r * HWo] = p.silu ? silu_f(v[r]) : v[r];
      } else {  // EPI_CONVT2
        const int tap = n / p.cout, co = n - tap * p.cout;
        const int dy = tap >> 1, dx = tap & 1;
        const int rem = m - bimg * HWo;
        const int y = fd_div(rem, p.fd_wo), x = rem - y * p.Wo;
        const size_t orow = ((size_t)bimg * 2 * p.Ho + 2 * y + dy) * (2 * p.Wo) + 2 * x + dx;
        T* o = (T*)p.out + orow * p.ldo + co;
        if constexpr (sizeof(T) == 2) {
          *(uint2*)o = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        } else {
          *(f32x4*)o = f32x4{v[0], v[1], v[2], v[3]};
        }
      }
    }
  }
  };

  // split-K item: the raw fp32 accumulators go to this slice's slab
  auto epilogue_partial = [&](int m0, int n0, int zc) __attribute__((always_inline)) {
    if constexpr (CF) {
      // cooperative finish: the slabs are read by OTHER workgroups of this launch, so they are published write-through
      // (sc1: past the non-coherent per-XCD L2s; 16-byte sc1 stores cost what plain ones do) - see cf_finish below
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.partial, 0, (int)p.cf_bytes, 0x00020000);
      auto row_block = [&](int b, auto bidx) __attribute__((always_inline)) {
        constexpr int BI = decltype(bidx)::value;
        const int m = m0 + wm * WTM + b * 16 + (lane & 15);
        if (m < p.M) {
          const unsigned off = (unsigned)((((size_t)zc * p.M + m) * p.N + n0 + wn * WTN + lg * 4) * sizeof(float));
#pragma unroll
          for (int a = 0; a < NF; ++a)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[a][BI]), rs, off + a * 64, 0, 16);
        }
      };
      static_for<MF>([&](auto bi) __attribute__((always_inline)) { row_block(decltype(bi)::value, bi); });
      return;
    }
    auto row_block = [&](int b, auto bidx) __attribute__((always_inline)) {
      constexpr int BI = decltype(bidx)::value;
      const int m = m0 + wm * WTM + b * 16 + (lane & 15);
      if (m < p.M) {
        float* dst = p.partial + ((size_t)zc * p.M + m) * p.N + n0 + wn * WTN + lg * 4;
#pragma unroll
        for (int a = 0; a < NF; ++a) *(f32x4*)(dst + a * 16) = acc[a][BI];
      }
    };
    static_for<MF>([&](auto bi) __attribute__((always_inline)) { row_block(decltype(bi)::value, bi); });
  };

  if constexpr (LDR > 0) {
    if (!is_cmp) {
      // ---- loader wave: nothing but the DMA stream, in step with the compute waves' barriers ----
      item_setup();
      int issued = 0;
#pragma unroll
      for (int j = 0; j < NST - 1; ++j)
        if (fetch_next(j)) ++issued;
      if (issued == NST - 1) wait_dma(std::integral_constant<int, NST - 2>{});
      else wait_dma(std::integral_constant<int, 0>{});
      __syncthreads();
      int fstl = NST - 1;
      for (int c_item = first_item; c_item < nwork; c_item += G) {
        int m0c, n0c, zc, kb, ke;
        item_range(c_item, m0c, n0c, zc, kb, ke);
        for (int kt = kb; kt < ke; ++kt) {
          const bool more = fetch_next(fstl);
          if (more) wait_dma(std::integral_constant<int, NST - 2>{});
          else wait_dma(std::integral_constant<int, 0>{});
          __syncthreads();
          fstl = (fstl + 1 == NST) ? 0 : fstl + 1;
        }
        __syncthreads();   // pairs with the barrier after the compute waves' epilogue
      }
      return;
    }
  }
  // ---- prologue: NST-1 stream positions in flight, the first one landed ----
  STAMP(p, 0)
  STAMP_WALL(p, 14)
  STAMP_XCC(p)
  if (is_ldr) {
    item_setup();
    int issued = 0;
#pragma unroll
    for (int j = 0; j < NST - 1; ++j)
      if (fetch_next(j)) ++issued;
    if (issued == NST - 1) wait_dma(std::integral_constant<int, NST - 2>{});
    else wait_dma(std::integral_constant<int, 0>{});
  }
  __syncthreads();

  STAMP(p, 1)
  int cur = 0;            // stage of the tile being computed
  int fst = NST - 1;      // stage the next DMA goes to
  for (int c_item = first_item; c_item < nwork; c_item += G) {
    int m0c, n0c, zc, kb, ke;
    item_range(c_item, m0c, n0c, zc, kb, ke);
    if constexpr (PIPE) {
      // One workgroup per CU (8-wave tiles): nobody else hides this wave's ds_read latency, so the two
      // 16-byte K groups of a tile live in two fragment sets (A: chunks 0-3, B: 4-7) and the loop is
      // rotated - every LDS read batch is issued one MFMA batch ahead of its use:
      //   read B(t) | MFMA A(t) | wait B, DMA(t+1) landed, barrier | read A(t+1) | MFMA B(t)
      // (with two workgroups per CU the co-resident one fills those gaps and this form measured slower)
      uint4 wfA[NF], xfA[MF], wfB[NF], xfB[MF];
#define IGEMM_READ(WF, XF, STAGE, CO)                                                                  \
  if (!DBG(p, 8)) {                                                                                    \
    const unsigned char* xs_ = smem + (STAGE) * kStageBytes + (wm * WTM) * kRowBytes + fr_row + (CO);  \
    const unsigned char* ws_ = smem + (STAGE) * kStageBytes + (BM + wn * WTN) * kRowBytes + fr_row + (CO); \
    /* W0, every X, then W1.. : the order the W-major MFMA batch first needs them */                     \
    WF[0] = *(const uint4*)(ws_);                                                                        \
    _Pragma("unroll") for (int b = 0; b < MF; ++b) XF[b] = *(const uint4*)(xs_ + b * 16 * kRowBytes);   \
    _Pragma("unroll") for (int a = 1; a < NF; ++a) WF[a] = *(const uint4*)(ws_ + a * 16 * kRowBytes);   \
  }
#define IGEMM_MMA(WF, XF)                                                                              \
  if (!DBG(p, 4)) {                                                                                  \
    _Pragma("unroll") for (int a = 0; a < NF; ++a)                                                      \
      _Pragma("unroll") for (int b = 0; b < MF; ++b) mma_kgroup<T>(WF[a], XF[b], acc[a][b]);            \
  } else {                                                                                             \
    _Pragma("unroll") for (int a = 0; a < NF; ++a) asm volatile("" ::"v"(WF[a].x), "v"(WF[a].w));       \
    _Pragma("unroll") for (int b = 0; b < MF; ++b) asm volatile("" ::"v"(XF[b].x), "v"(XF[b].w));       \
  }
      // Issue order of a half tile: one ds_read_b128 in front of every MPR MFMAs instead of all reads, then all MFMAs - the
      // read burst of eight waves otherwise meets the loader waves' LDS-DMA writes at the same moment.  Model of this loop
      // in tools/ubench/mfma_lds.hip: 908 -> 817 ns per K tile with L2-resident operands, 1306 -> 1158 ns from the MALL;
      // the kernel: -6 % over the layer shapes, -8..10 % on the 3x3 convs.
      constexpr int NMMA = NF * MF * (sizeof(T) == 2 ? 1 : 4);   // MFMA instructions per half tile (fp32: four 16x16x4 per k-group)
      constexpr int MPR = NMMA / (NF + MF);
#define IGEMM_INTERLEAVE                                                                                \
  static_for<NF + MF>([&](auto) __attribute__((always_inline)) {                                        \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                  \
    __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);                                                \
  });                                                                                                   \
  __builtin_amdgcn_sched_group_barrier(0x008, NMMA - MPR * (NF + MF), 0);                               \
  __builtin_amdgcn_sched_barrier(0);
      if (is_cmp) IGEMM_READ(wfA, xfA, cur, fr_c0)
      for (int kt = kb; kt < ke; ++kt) {
        bool more = false;
        if (is_ldr) more = DBG(p, 1) ? false : fetch_next(fst);
        if (is_cmp) {
          IGEMM_READ(wfB, xfB, cur, fr_c1)
          IGEMM_MMA(wfA, xfA)
          IGEMM_INTERLEAVE
          // B is in registers, so nobody still reads this stage
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (is_ldr) {   // the next stream position must have landed before anyone reads it (later ones may stay in flight)
          if (more) wait_dma(std::integral_constant<int, NST - 2>{});
          else wait_dma(std::integral_constant<int, 0>{});
        }
        __syncthreads();
        cur = (cur + 1 == NST) ? 0 : cur + 1;
        fst = (fst + 1 == NST) ? 0 : fst + 1;
        if (is_cmp) {
          // (after an item's last tile this reads a stage that may not have landed: harmless, the next item re-reads its
          // first A fragments after the epilogue - a conditional read would split the block and undo the interleaving)
          IGEMM_READ(wfA, xfA, cur, fr_c0)
          IGEMM_MMA(wfB, xfB)
          IGEMM_INTERLEAVE
        }
      }
#undef IGEMM_READ
#undef IGEMM_MMA
#undef IGEMM_INTERLEAVE
    } else {
    for (int kt = kb; kt < ke; ++kt) {
      const bool more = DBG(p, 1) ? false : fetch_next(fst);
      const unsigned char* xs = smem + cur * kStageBytes + (wm * WTM) * kRowBytes + fr_row;
      const unsigned char* ws = smem + cur * kStageBytes + (BM + wn * WTN) * kRowBytes + fr_row;
      bool x3_done = false;
      if constexpr (sizeof(T) == 4) {
        // Split-bf16 arithmetic on fp32 operands (IgemmParams::x3, compute_dtype "bf16x3"): every fp32 value is hi + lo with
        // hi = bf16(x) and lo = bf16(x - hi) (x - hi is exact in fp32), and X.W ~ Wl.Xh + Wh.Xl + Wh.Xh on v_mfma_f32_16x16x32_bf16
        // with fp32 accumulation - the dropped Wl.Xl term and the rounding of lo are below 2^-16 relative per product.  One
        // K tile (32 floats per row) is exactly one K = 32 MFMA step: a lane's two 16-byte chunks (k = 4 lg .. and 16 + 4 lg ..)
        // are its eight contraction elements, the same lane -> k map for both operands.  3 MFMAs of 16 cycles per 32 k
        // instead of 8 v_mfma_f32_16x16x4_f32 of 32 cycles: 5.3 x the matrix rate of the exact fp32 path; ~24 VALU
        // instructions per fragment for the split.
        if (p.x3) {
          x3_done = true;
          auto split = [&](const uint4& c0, const uint4& c1, uint4& hi, uint4& lo) __attribute__((always_inline)) {
            const float f[8] = {bits_f32(c0.x), bits_f32(c0.y), bits_f32(c0.z), bits_f32(c0.w),
                                bits_f32(c1.x), bits_f32(c1.y), bits_f32(c1.z), bits_f32(c1.w)};
            unsigned h[4], l[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              h[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
              const float r0 = f[2 * i] - bits_f32(h[i] << 16), r1 = f[2 * i + 1] - bits_f32(h[i] & 0xffff0000u);
              l[i] = pack_bf16x2(r0, r1);
            }
            hi = make_uint4(h[0], h[1], h[2], h[3]);
            lo = make_uint4(l[0], l[1], l[2], l[3]);
          };
          // X fragments are split once per tile and kept (MF x 8 registers); W fragments go through one at a time (the 64 x 80
          // wave tiles have 80 accumulator registers: holding every split fragment at once spilled 60-90 VGPRs)
          uint4 xh[MF], xl[MF];
#pragma unroll
          for (int b = 0; b < MF; ++b)
            split(*(const uint4*)(xs + b * 16 * kRowBytes + fr_c0), *(const uint4*)(xs + b * 16 * kRowBytes + fr_c1), xh[b], xl[b]);
#pragma unroll
          for (int a = 0; a < NF; ++a) {
            uint4 wh, wl;
            if (p.x3 == 2) {            // round 6: the weights were split at create time (launch_split_planes): chunk lg = hi, chunk lg + 4 = lo
              wh = *(const uint4*)(ws + a * 16 * kRowBytes + fr_c0);
              wl = *(const uint4*)(ws + a * 16 * kRowBytes + fr_c1);
            } else {
              split(*(const uint4*)(ws + a * 16 * kRowBytes + fr_c0), *(const uint4*)(ws + a * 16 * kRowBytes + fr_c1), wh, wl);
            }
            // small terms first; consecutive MFMAs go to different accumulators
#pragma unroll
            for (int b = 0; b < MF; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wl), __builtin_bit_cast(bf16x8, xh[b]), acc[a][b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < MF; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh), __builtin_bit_cast(bf16x8, xl[b]), acc[a][b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < MF; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh), __builtin_bit_cast(bf16x8, xh[b]), acc[a][b], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        if (x3_done) break;
        if (DBG(p, 8)) break;          // ablation: no LDS reads, no MFMA
        const int co = kg ? fr_c1 : fr_c0;
        uint4 wf[NF], xf[MF];
#pragma unroll
        for (int a = 0; a < NF; ++a) wf[a] = *(const uint4*)(ws + a * 16 * kRowBytes + co);
#pragma unroll
        for (int b = 0; b < MF; ++b) xf[b] = *(const uint4*)(xs + b * 16 * kRowBytes + co);
        if (!DBG(p, 4)) {
#pragma unroll
          for (int a = 0; a < NF; ++a)
#pragma unroll
            for (int b = 0; b < MF; ++b) mma_kgroup<T>(wf[a], xf[b], acc[a][b]);
        } else {
#pragma unroll
          for (int a = 0; a < NF; ++a) asm volatile("" ::"v"(wf[a].x), "v"(wf[a].w));
#pragma unroll
          for (int b = 0; b < MF; ++b) asm volatile("" ::"v"(xf[b].x), "v"(xf[b].w));
        }
      }
      // the next stream position must have landed before anyone reads it; later ones may stay in flight
      if (more) wait_dma(std::integral_constant<int, NST - 2>{});
      else wait_dma(std::integral_constant<int, 0>{});
      __syncthreads();
      cur = (cur + 1 == NST) ? 0 : cur + 1;
      fst = (fst + 1 == NST) ? 0 : fst + 1;
    }
    }
    // the DMA of the next item's first tiles is already in flight while this epilogue runs
    if (c_item == first_item) { STAMP(p, 2) }
    if (!DBG(p, 16) && is_cmp) {
      constexpr int E = 16 / (int)sizeof(T);
      const bool rows_ok = p.epi == EPI_STORE && p.splits <= 1 && !DBG(p, 32) && (p.n_valid % E == 0) &&
                           (p.ldo % E == 0) && (!p.resid || p.ldr % E == 0);
      const bool geglu_ok = p.epi == EPI_GEGLU && (NF % 2 == 0) && !DBG(p, 32) && (p.ldo % E == 0);
      if (rows_ok) epilogue_rows(m0c, n0c, fst);
      else if (geglu_ok) epilogue_geglu_rows(m0c, n0c, fst);
      else if (p.splits > 1) epilogue_partial(m0c, n0c, zc);
      else epilogue(m0c, n0c, zc);
    }
    if (c_item == first_item) { STAMP(p, 3) }
    __syncthreads();   // the staging stage is handed back to the DMA ring
    if (c_item == first_item) { STAMP(p, 4) }
    if (c_item + G >= nwork) { STAMP_WALL(p, 15) }
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
      for (int b = 0; b < MF; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // ================= in-launch cooperative split-K finish (IgemmParams::cf; round 6) =================
  // The launcher sets cf only when every work item has its own co-resident workgroup (nwork <= grid): this workgroup ran
  // exactly one item (tile, K slice z) and its slab went out write-through.  Instead of a second launch reading all S slabs
  // back, the S workgroups of a tile each reduce 1/S of it - rows [z RS, (z+1) RS) - in the finish kernel's slice order (bit
  // identical to the two-launch path), apply bias / time-embedding row / residual / SiLU and store the output once.
  // Hand-off (MI355X guide, G16 form R1): sc1 slab stores -> every wave s_waitcnt vmcnt(0) -> barrier -> ONE relaxed agent-scope
  // ticket add per workgroup; readers poll the ticket word (relaxed, s_sleep) and read the slabs with sc1 loads.  Placement
  // independent: nothing here assumes which XCD a workgroup runs on.
  // Progress without co-residency: the poll is bounded; a workgroup that gives up sets its bit in the ticket word's mask
  // (one fetch_or: if everybody had arrived meanwhile it does its share after all) and leaves; the LAST arriver sees the
  // mask in its ticket's return value and reduces those shares as well.  Nobody waits for a workgroup that has not started.
  // Counters are self-cleaning: the last of the S workgroups to leave the tile (second word) zeroes both words.
  if constexpr (CF) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's slab stores have left the chip's caches
    __syncthreads();                                          // (loader waves have returned: only compute waves are left)
    int m0, n0, z, kb_, ke_;
    item_range(first_item, m0, n0, z, kb_, ke_);
    const int tile = first_item - z * ntiles;
    const int S = p.splits;
    unsigned long long* ctr = p.cf_ctr + 2 * (size_t)tile;
    volatile unsigned* sflag = (volatile unsigned*)smem;      // the ring is idle now
    if (tid == 0) {
      unsigned shares = 0;
      const unsigned long long old = __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((int)(old & 0xffull) + 1 == S) {
        shares = (1u << z) | (unsigned)(old >> 8);            // last arriver: own share + the shares of those who gave up
      } else {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (true) {
          const unsigned long long v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((int)(v & 0xffull) == S) { shares = 1u << z; break; }
          if (__builtin_amdgcn_s_memrealtime() - t0 >= (unsigned long long)p.cf_poll_ticks) {
            const unsigned long long o2 = __hip_atomic_fetch_or(ctr, 1ull << (8 + z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)(o2 & 0xffull) == S) shares = 1u << z;   // everybody arrived in the meantime: the last arriver did not see the bit
            else __hip_atomic_fetch_add(p.cf_ctr + p.cf_diag, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // diagnostics: last word of the region
            break;
          }
          __builtin_amdgcn_s_sleep(4);
        }
      }
      // this workgroup will not look at the ticket word again
      const unsigned long long d = __hip_atomic_fetch_add(ctr + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((int)d + 1 == S) {
        __hip_atomic_store(ctr, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ctr + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      sflag[0] = shares;
    }
    __syncthreads();
    const unsigned shares = sflag[0];
    if (shares) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.partial, 0, (int)p.cf_bytes, 0x00020000);
      const int RS = (BM + S - 1) / S;                        // rows per share
      constexpr int QPR = BN / 4;                             // channel quads per tile row
      const unsigned slab = (unsigned)((size_t)p.M * p.N * sizeof(float));
      const int nq4 = p.n_valid >> 2;
      // sum of the S slabs of one quad in splitk_finish_kernel's association; SS = compile-time slice count (0: run-time loop)
      auto reduce_quad = [&](unsigned off, auto ss_tag) __attribute__((always_inline)) -> f32x4 {
        constexpr int SS = decltype(ss_tag)::value;
        auto ld = [&](int zz) __attribute__((always_inline)) -> f32x4 {
          return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + (unsigned)zz * slab, 0, 16));
        };
        if constexpr (SS > 0) {
          f32x4 L[SS];
#pragma unroll
          for (int zz = 0; zz < SS; ++zz) L[zz] = ld(zz);     // all slices in flight together
          f32x4 v = L[0];
          int zz = 1;
#pragma unroll
          for (; zz + 3 < SS; zz += 4) v += (L[zz] + L[zz + 1]) + (L[zz + 2] + L[zz + 3]);
          if (zz + 1 < SS) { v += L[zz] + L[zz + 1]; zz += 2; }
          if (zz < SS) v += L[zz];
          return v;
        } else {
          f32x4 v = ld(0);
          int zz = 1;
          for (; zz + 3 < S; zz += 4) {
            const f32x4 t0 = ld(zz), t1 = ld(zz + 1), t2 = ld(zz + 2), t3 = ld(zz + 3);
            v += (t0 + t1) + (t2 + t3);
          }
          if (zz + 1 < S) { const f32x4 t0 = ld(zz), t1 = ld(zz + 1); v += t0 + t1; zz += 2; }
          if (zz < S) v += ld(zz);
          return v;
        }
      };
      auto do_share = [&](int sh, auto ss_tag) __attribute__((always_inline)) {
        const int r0 = sh * RS;
        const int r1 = r0 + RS < BM ? r0 + RS : BM;
        const int nq = (r1 - r0) * QPR;
        for (int q = tid; q < nq; q += NW * 64) {
          const int rr = q / QPR, qc = q - rr * QPR;
          const int m = m0 + r0 + rr, n = n0 + qc * 4;
          if (m >= p.M || (n >> 2) >= nq4) continue;
          f32x4 v = reduce_quad((unsigned)(((size_t)m * p.N + n) * sizeof(float)), ss_tag);
          f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
          if (p.bias) bias = *(const f32x4*)(p.bias + n);
          v += bias;
          if (p.rowbias) v += *(const f32x4*)(p.rowbias + (size_t)fd_div(m, p.fd_hwo) * p.rb_stride + n);
          if (p.resid) {
            const T* rp = (const T*)p.resid + (size_t)m * p.ldr + n;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += to_f32<T>(rp[r]);
          }
          if (p.silu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
          }
          T* o = (T*)p.out + out_row(m) * p.ldo + n;
          if constexpr (sizeof(T) == 2) {
            *(uint2*)o = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
          } else {
            *(f32x4*)o = v;
          }
        }
      };
      for (int sh = 0; sh < S; ++sh) {
        if (!((shares >> sh) & 1u)) continue;
        switch (S) {
          case 2: do_share(sh, std::integral_constant<int, 2>{}); break;
          case 4: do_share(sh, std::integral_constant<int, 4>{}); break;
          case 8: do_share(sh, std::integral_constant<int, 8>{}); break;
          default: do_share(sh, std::integral_constant<int, 0>{}); break;
        }
      }
    }
  }
}

// split-K finish: out = sum_z partial[z] + bias (+rowbias)(+resid), optional SiLU  (EPI_STORE only)
template <typename T>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const IgemmParams p) {
  // thread -> (row, channel quad) without a division: 64 quads x 4 rows per workgroup trip, rows strided over the grid
  const int nq = p.n_valid >> 2;
  const int q0 = blockIdx.x * 64 + (threadIdx.x & 63);
  if (q0 >= nq) return;
  const int n = q0 * 4;
  f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias = *(const f32x4*)(p.bias + n);
  for (int m = blockIdx.y * 4 + (threadIdx.x >> 6); m < p.M; m += gridDim.y * 4) {
    // four slices are requested together (a plain `v += load` loop waits for each load before it issues the next: one
    // dependent trip to the MALL per K slice)
    const float* q = p.partial + (size_t)m * p.N + n;
    const size_t slab = (size_t)p.M * p.N;
    f32x4 v = *(const f32x4*)q;
    int z = 1;
    for (; z + 3 < p.splits; z += 4) {
      const f32x4 t0 = *(const f32x4*)(q + z * slab), t1 = *(const f32x4*)(q + (z + 1) * slab);
      const f32x4 t2 = *(const f32x4*)(q + (z + 2) * slab), t3 = *(const f32x4*)(q + (z + 3) * slab);
      v += (t0 + t1) + (t2 + t3);
    }
    if (z + 1 < p.splits) {
      const f32x4 t0 = *(const f32x4*)(q + z * slab), t1 = *(const f32x4*)(q + (z + 1) * slab);
      v += t0 + t1;
      z += 2;
    }
    if (z < p.splits) v += *(const f32x4*)(q + z * slab);
    v += bias;
    if (p.rowbias) v += *(const f32x4*)(p.rowbias + (size_t)fd_div(m, p.fd_hwo) * p.rb_stride + n);
    if (p.resid) {
      const T* rp = (const T*)p.resid + (size_t)m * p.ldr + n;
      for (int r = 0; r < 4; ++r) v[r] += to_f32<T>(rp[r]);
    }
    if (p.silu) for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
    size_t orow = (size_t)m;
    if (p.up4) {                                           // virtual row (phase, image, i, j) -> upsampled output row
      const int ph = fd_div(m, p.fd_mphase), mr = m - ph * p.fd_mphase.d;
      const int b = fd_div(mr, p.fd_hwo), rem = mr - b * (p.Ho * p.Wo);
      const int oy = fd_div(rem, p.fd_wo), ox = rem - oy * p.Wo;
      orow = ((size_t)b * 2 * p.Ho + 2 * oy + (ph >> 1)) * (size_t)(2 * p.Wo) + 2 * ox + (ph & 1);
    }
    T* o = (T*)p.out + orow * p.ldo + n;
    if constexpr (sizeof(T) == 2) {
      *(uint2*)o = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    } else {
      *(f32x4*)o = v;
    }
  }
}

// Per-device state (a process may hold handles on several devices; kernels of a handle always run
// on that handle's device, which the C ABI makes current before launching).
constexpr int kMaxDev = 64;
int cur_dev() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) dev = 0;
  return dev;
}

int num_cus() {
  static int n[kMaxDev] = {};
  const int dev = cur_dev();
  if (!n[dev]) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n[dev] = prop.multiProcessorCount;
    if (n[dev] <= 0) n[dev] = 256;
  }
  return n[dev];
}

const void* zero_page() {
  static void* z[kMaxDev] = {};
  const int dev = cur_dev();
  if (!z[dev]) {
    if (hipMalloc(&z[dev], 64 << 10) != hipSuccess) return nullptr;     // also serves as an all-zero bias vector
    (void)hipMemset(z[dev], 0, 64 << 10);
  }
  return z[dev];
}

constexpr int kDefaultPolicy = 61;
}  // namespace
std::string igemm_dispatch_name(const IgemmDispatch& d);
namespace {
bool g_log_on = false;
std::set<std::string> g_log;   // distinct instantiations launched while logging is on
IgemmDispatch g_last{};   // what the last launch_igemm on this thread's process ran (parity tests assert on it)
int g_dbg = 0;       // ablation flags (profiling experiments only)
void* g_tsbuf = nullptr;   // s_memtime stamp buffer (LDMSEG_IGEMM_ABLATE builds)
int g_big = kDefaultPolicy;      // bit0: 8-wave 256-row tiles with a 3-stage ring (-0.15 ms per forward, on);
                     // bit1: 4-stage ring, one workgroup per CU, for mid-size grids (+0.5 ms, off)
                     // bit2: lone 64-row 4-stage tiles; bit3: pipelined K loop on the 256-row tiles; bit4: 8-wave 128-row
                     // tiles (+ loader waves on long K); bit5: loader waves on the 256-row tiles (long K / GEGLU)

int g_force_cfg = -1;            // tools/tune_igemm.py: run every launch with this entry of the instantiation list
int g_cf_mode = 5;               // igemm_set_cf_mode (debug key 23)
int g_cm_mode = -1;              // igemm_set_cm_mode

// Launch table measured on the MI355X (tools/tune_igemm.py): launch shape -> entry of the instantiation list in run_cfg()
// + number of K slices.  Shapes that are not listed (other batch sizes, other models) use the rules in dispatch() /
// igemm_plan_splits(); so does every launch while a non-default tile policy is set (tests, ablations).
struct TunedEntry { int dtype, M, N, K, taps, stride, up, epi, lnf, cfg, splits; };
const TunedEntry kTuned[] = {
#include "igemm_tuned.inc"
    {-1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};

int g_table_override = -1;       // igemm_set_table_override (debug key 24)
const TunedEntry* tuned_lookup(const IgemmParams& p, int dtype) {
  if (g_big != kDefaultPolicy || g_force_cfg >= 0) return nullptr;
  if (g_table_override >= 0 && p.epi == EPI_STORE && !p.rowstats) {     // tuning: every plain-store launch as if the table held this entry
    static thread_local TunedEntry e;
    e = TunedEntry{dtype, p.M, p.N, 0, p.taps, p.stride, p.up, p.epi, 0, g_table_override & 0xff, (g_table_override >> 8) & 0xff};
    if (e.splits < 1) e.splits = 1;
    return &e;
  }
  const int K = p.taps * (p.C0 + p.C1), lnf = p.rowstats ? 1 : 0;
  for (const TunedEntry* e = kTuned; e->dtype >= 0; ++e)
    if (e->M == p.M && e->N == p.N && e->K == K && e->dtype == dtype && e->taps == p.taps && e->stride == p.stride &&
        e->up == p.up && e->epi == p.epi && e->lnf == lnf)
      return e;
  return nullptr;
}

// the instantiations that exist with the in-launch finish: bf16, the four tile forms K-sliced launches of the UNet run on
template <typename T, int BM, int BN, int WM, int WN, int NST, bool PIPE, int LDR, bool LNF, bool CM, bool XT, bool UP4>
constexpr bool cf_instantiated() {
  if (sizeof(T) != 2 || BN != 160 || LNF || CM) return false;
  const bool c0 = BM == 256 && WM == 4 && WN == 2 && NST == 3 && PIPE && LDR == 4;      // entry 0 of the instantiation list
  const bool c3 = BM == 128 && WM == 2 && WN == 2 && NST == 4 && PIPE && LDR == 4;      // entry 3
  const bool c4 = BM == 128 && WM == 4 && WN == 2 && NST == 3 && PIPE && LDR == 0;      // entry 4
  const bool c10 = BM == 64 && WM == 2 && WN == 2 && NST == 4 && PIPE && LDR == 4;      // entry 10
  if (UP4) return c0 && !XT;
  if (XT) return c0 || c3 || c4;
  return c0 || c3 || c4 || c10;
}
template <typename T, int BM, int BN, int WM, int WN, int NST = 2, bool PIPE = false, int LDR = 0, bool LNF = false, bool CM = false, bool XT = false,
          bool UP4 = false>
int run(const IgemmParams& pin, hipStream_t s) {
  IgemmParams p = pin;
  p.dbg = g_dbg;
  p.ts = (unsigned long long*)g_tsbuf;
  p.zeros = zero_page();
  if (!p.zeros) return -3;
  const int mt = (p.M + BM - 1) / BM, nt = p.N / BN;
  p.fd_hwo = fastdiv_make(p.Ho * p.Wo);
  p.fd_wo = fastdiv_make(p.Wo);
  p.fd_nt = fastdiv_make(nt);
  p.fd_ntiles = fastdiv_make(mt * nt);
  p.fd_nsplit = fastdiv_make(p.splits > 1 ? p.splits : 1);
  p.fd_mphase = fastdiv_make(p.up4 ? p.M / 4 : 1);
  p.fd_tpt = fastdiv_make(CM ? 9 : (p.C0 + p.C1) / (int)(kRowBytes / sizeof(T)));   // CM: K tiles per channel tile
  const int nwork = mt * nt * (p.splits > 1 ? p.splits : 1);
  // persistent grid: as many workgroups as fit on the chip at once (2 per CU for the 4-wave tiles,
  // 1 per CU for the 8-wave ones); each walks nwork / grid items
  int resident = num_cus() * ((WM * WN == 4 && NST == 2 && LDR == 0) ? 2 : 1);
  const int grid_x = nwork < resident ? nwork : resident;
  const size_t lds = (size_t)NST * (BM + BN) * kRowBytes +
                     (epi_stage_dedicated<BM, BN, WM, WN>() ? epi_stage_bytes<BM, BN, WM, WN>() : 0);
  auto kern = igemm_kernel<T, BM, BN, WM, WN, NST, PIPE, LDR, LNF, CM, XT, UP4>;
  constexpr bool kHasCf = cf_instantiated<T, BM, BN, WM, WN, NST, PIPE, LDR, LNF, CM, XT, UP4>();
  static bool attr_set[kMaxDev] = {};
  const int dev = cur_dev();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if constexpr (kHasCf)
      (void)hipFuncSetAttribute((const void*)igemm_kernel<T, BM, BN, WM, WN, NST, PIPE, LDR, LNF, CM, XT, UP4, true>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set[dev] = true;
  }
  // Cooperative finish inside the launch: every (tile, K slice) item has a workgroup of its own and all of them fit on the chip
  // together (one per CU is what every instantiation can hold), the slab set is addressable through one buffer descriptor,
  // the tile's counters fit the caller's region.  Otherwise: slabs + the finish kernel.
  p.cf = 0;
  // (measured per launch shape, tools/cf_bench.py: the in-launch finish wins 2-4 % on the 256-row tiles with 2-4 slices and loses
  // 2-20 % on the 128-row tiles with 8 - the serial chain drain -> ticket -> poll -> read costs what boundary + finish launch do;
  // mode bit 3 takes it on every tile form that has the instantiation)
  if (kHasCf && (BM == 256 || (g_cf_mode & 8)) && p.splits > 1 && p.splits <= 32 && !p.no_finish && p.cf_ctr && (g_cf_mode & 1) && nwork == grid_x && nwork <= num_cus() &&
      (size_t)p.splits * p.M * p.N * sizeof(float) < ((size_t)1 << 31) && (size_t)(2 * mt * nt + 2) * 8 <= igemm_cf_bytes()) {
    p.cf = 1;
    p.cf_diag = (int)(igemm_cf_bytes() / 8) - 1;
    p.cf_bytes = (unsigned)((size_t)p.splits * p.M * p.N * sizeof(float));
    const int us = (g_cf_mode >> 8) & 0xffff;
    p.cf_poll_ticks = (g_cf_mode & 2) ? 0 : (us ? us : 200) * 100;
  }
  g_last = IgemmDispatch{(int)sizeof(T) == 2 ? DT_BF16 : DT_F32, BM, BN, WM, WN, NST, PIPE ? 1 : 0, LDR,
                         p.splits > 1 ? p.splits : 1, grid_x, LNF ? 1 : 0, CM ? 1 : 0, p.cf, XT ? 1 : 0, UP4 ? 1 : 0};
  if (g_log_on) g_log.insert(igemm_dispatch_name(g_last));
  if constexpr (kHasCf) {
    if (p.cf) {
      hipLaunchKernelGGL((igemm_kernel<T, BM, BN, WM, WN, NST, PIPE, LDR, LNF, CM, XT, UP4, true>), dim3(grid_x), dim3((WM * WN + LDR) * 64), lds, s, p);
      return hipGetLastError() == hipSuccess ? 0 : -3;
    }
  }
  hipLaunchKernelGGL(kern, dim3(grid_x), dim3((WM * WN + LDR) * 64), lds, s, p);
  if (p.splits > 1 && !p.no_finish) {
    const int nq = p.n_valid >> 2;
    const int gx = (nq + 63) / 64;
    int gy = (p.M + 3) / 4;
    if (gy > 2048 / gx) gy = 2048 / gx > 0 ? 2048 / gx : 1;
    hipLaunchKernelGGL(splitk_finish_kernel<T>, dim3(gx, gy), dim3(256), 0, s, p);
  }
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// 8-wave 128-row tile, one workgroup per CU (3-stage ring, pipelined K loop): the mid-size grids
// (16x16 / 32x32 feature maps) where 128-row tiles x K slices give about one work item per CU.  Versus
// two co-resident 64-row workgroups it stages 36% fewer operand bytes per FLOP through the LDS DMA.
inline bool mid8_ok(long t128, int splits) {
  const long items = t128 * (splits > 1 ? splits : 1);
  return (g_big & 16) && items >= 160 && items <= 2 * (long)num_cus();
}

// The instantiation list the launch table indexes (BN and the LayerNorm fold follow from the launch): -2 = not in the list.
//   0  256 rows, 8 compute + 4 loader waves      3  128 rows, 4 compute + 4 loader waves, 4-stage ring
//   1  256 rows, 8 waves, pipelined K loop       4  128 rows, 8 waves, pipelined K loop
//   2  256 rows, 8 waves, plain K loop           5  128 rows, 4 waves, 4-stage ring (one workgroup per CU)
//   6  64 rows, 4 waves, 4-stage ring            7  64 rows, 4 waves, two workgroups per CU
//   8  128 rows, 4 waves, two workgroups per CU  9  64 rows, 8 waves (16 x 80 wave tiles), 4-stage ring
//   10 / 11  64 rows, 4 compute (32 x 80 wave tiles) + 4 loader waves, 4- / 3-stage ring: the M = 512 .. 2048 1x1 launches.
//      (entry 9 re-reads the W fragments in each of its four wave rows - 98 KB of fragment reads per K tile against 56 KB here -
//      and its compute waves issue the LDS-DMA themselves: phase ablation of M = 2048, N = K = 1280 showed DMA issue, fragment
//      reads + MFMA, epilogue and launch skeleton ADDING UP, 4 + 5 + 2 + 5 us; with loader waves 17.0 -> 13.6 us)
constexpr int kNumCfg = 12;
template <typename T, int BN, bool LNF>
int run_cfg(int cfg, const IgemmParams& p, hipStream_t s) {
  switch (cfg) {
    case 0: return run<T, 256, BN, 4, 2, 3, true, 4, LNF>(p, s);
    case 1: return run<T, 256, BN, 4, 2, 3, true, 0, LNF>(p, s);
    case 4: return run<T, 128, BN, 4, 2, 3, true, 0, LNF>(p, s);
    case 6: return run<T, 64, BN, 2, 2, 4, false, 0, LNF>(p, s);
    case 7: return run<T, 64, BN, 2, 2, 2, false, 0, LNF>(p, s);
    case 8: return run<T, 128, BN, 2, 2, 2, false, 0, LNF>(p, s);
    default: break;
  }
  if constexpr (!LNF) {
    if (cfg == 9) return run<T, 64, BN, 4, 2, 4>(p, s);
    if (cfg == 10) return run<T, 64, BN, 2, 2, 4, true, 4>(p, s);
    if (cfg == 11) return run<T, 64, BN, 2, 2, 3, true, 4>(p, s);
    switch (cfg) {
      case 2: return run<T, 256, BN, 4, 2, 3, false>(p, s);
      case 3: return run<T, 128, BN, 2, 2, 4, true, 4>(p, s);
      case 5: return run<T, 128, BN, 2, 2, 4>(p, s);
      default: break;
    }
  }
  return -2;
}
template <typename T>
int run_cfg_any(int cfg, const IgemmParams& p, hipStream_t s) {
  const bool geglu = p.epi == EPI_GEGLU;
  const int bn = geglu ? 128 : (p.N % 160 == 0 ? 160 : (p.N % 128 == 0 ? 128 : 0));
  if (p.rowstats) return bn == 160 ? run_cfg<T, 160, true>(cfg, p, s) : bn == 128 ? run_cfg<T, 128, true>(cfg, p, s) : -2;
  return bn == 160 ? run_cfg<T, 160, false>(cfg, p, s) : bn == 128 ? run_cfg<T, 128, false>(cfg, p, s) : -2;
}

// Launches with a folded LayerNorm (norm1 -> q|k|v, norm3 -> GEGLU: K = C <= 1280, N a multiple of 160 or GEGLU's 128):
// the same tile rules as dispatch() below, restricted to the instantiations those shapes can reach.
template <typename T>
int dispatch_ln(const IgemmParams& p, hipStream_t s) {
  const bool geglu = p.epi == EPI_GEGLU;
  if (!geglu && p.N % 160 != 0) return -2;
  const int bn = geglu ? 128 : 160;
  const long t256 = (long)((p.M + 255) / 256) * (p.N / bn);
  const long t128 = (long)((p.M + 127) / 128) * (p.N / bn);
  const long t64 = (long)((p.M + 63) / 64) * (p.N / bn);
  if (t256 >= 240) {
    if (geglu) return (g_big & 32) && sizeof(T) == 2 ? run<T, 256, 128, 4, 2, 3, true, 4, true>(p, s) : run<T, 256, 128, 4, 2, 3, true, 0, true>(p, s);
    return run<T, 256, 160, 4, 2, 3, true, 0, true>(p, s);
  }
  if (!geglu && mid8_ok(t128, 1)) return run<T, 128, 160, 4, 2, 3, true, 0, true>(p, s);
  if (t64 <= num_cus()) return geglu ? run<T, 64, 128, 2, 2, 4, false, 0, true>(p, s) : run<T, 64, 160, 2, 2, 4, false, 0, true>(p, s);
  return geglu ? run<T, 64, 128, 2, 2, 2, false, 0, true>(p, s) : run<T, 64, 160, 2, 2, 2, false, 0, true>(p, s);
}

int g_xt_mode = 1;               // igemm_set_xt_mode (debug key 19): 0 = engines keep conv_shortcut as a launch of its own

// conv2 + conv_shortcut of a resnet as one launch (IgemmParams::src2 ...): the three tile forms the UNet's resnet convs use.
// The launch table is consulted with the 3x3 part's K (tuned_lookup ignores the extra tap), so the launch inherits the entry
// of the conv it extends.
template <typename T>
int dispatch_xt(const IgemmParams& p, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
    if (p.N % 160 != 0) return -2;
    int cfg = -1;
    if (const TunedEntry* e = tuned_lookup(p, DT_BF16)) cfg = e->cfg;
    if (cfg != 0 && cfg != 3 && cfg != 4) {
      const int sp = p.splits > 1 ? p.splits : 1;
      const long t256 = (long)((p.M + 255) / 256) * (p.N / 160), t128 = (long)((p.M + 127) / 128) * (p.N / 160);
      const int nk_slice = ((p.taps * (p.C0 + p.C1) + p.C2 + p.C3) / 64) / sp;
      if (t256 >= 240) cfg = 0;
      else if (mid8_ok(t128, sp)) cfg = nk_slice >= 40 ? 3 : 4;
      else cfg = t256 * sp >= 160 ? 0 : 3;
    }
    if (cfg == 0) return run<T, 256, 160, 4, 2, 3, true, 4, false, false, true>(p, s);
    if (cfg == 3) return run<T, 128, 160, 2, 2, 4, true, 4, false, false, true>(p, s);
    return run<T, 128, 160, 4, 2, 3, true, 0, false, false, true>(p, s);
  }
  return -2;
}

// the split-K finish every kernel of the family shares (p.splits > 1, slabs written)
template <typename T>
int launch_finish(const IgemmParams& pin, hipStream_t s) {
  IgemmParams p = pin;
  p.fd_hwo = fastdiv_make(p.Ho * p.Wo);
  p.fd_wo = fastdiv_make(p.Wo);
  p.fd_mphase = fastdiv_make(p.up4 ? p.M / 4 : 1);
  const int nq = p.n_valid >> 2;
  const int gx = (nq + 63) / 64;
  int gy = (p.M + 3) / 4;
  if (gy > 2048 / gx) gy = 2048 / gx > 0 ? 2048 / gx : 1;
  hipLaunchKernelGGL(splitk_finish_kernel<T>, dim3(gx, gy), dim3(256), 0, s, p);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <typename T>
int dispatch(const IgemmParams& p, hipStream_t s) {
  if (p.up4) {                 // one tile form: 256 x 160 with loader waves (every up4 launch is >= 240 work items, K >= 20 tiles)
    if constexpr (sizeof(T) == 2) return run<T, 256, 160, 4, 2, 3, true, 4, false, false, false, true>(p, s);
    return -2;
  }
  if (p.C2 > 0) return dispatch_xt<T>(p, s);
  if (p.cm) {   // channel-major 3x3 conv: one instantiation (the 256-row loader-wave tile the large maps use anyway), bf16 only
    if constexpr (sizeof(T) == 2) {
      if (!p.rowstats && p.epi == EPI_STORE && p.N % 160 == 0) return run<T, 256, 160, 4, 2, 3, true, 4, false, true>(p, s);
    }
    return -2;
  }
  {
    int cfg = g_force_cfg;
    if (cfg < 0 && !p.x3) {
      if (const TunedEntry* e = tuned_lookup(p, sizeof(T) == 2 ? DT_BF16 : DT_F32)) cfg = e->cfg;
    }
    if (cfg >= 0 && !p.x3) {
      const int r = run_cfg_any<T>(cfg, p, s);
      if (r != -2 || g_force_cfg >= 0) return r;     // a forced entry that does not exist for this launch is an error
    }
  }
  if constexpr (sizeof(T) == 4) {
    // split-bf16 mode: the arithmetic lives in the plain K loop only, so these launches take the plain-loop instantiations
    if (p.x3) {
      // Tile choice: fp32 operands double the bytes staged per MAC, so these launches are bound by the L2 -> LDS path before the
      // split's VALU work: the largest tile wins even where its 64 x 80 wave tiles spill 17-19 VGPRs in this loop (measured:
      // 36.8 ms per B = 8 / L = 64 forward with the large tiles against 41.0 ms with spill-free 64-row tiles)
      const int bnx = (p.epi == EPI_GEGLU) ? 128 : (p.N % 160 == 0 ? 160 : (p.N % 128 == 0 ? 128 : (p.N % 64 == 0 ? 64 : 32)));
      const int sp = p.splits > 1 ? p.splits : 1;
      const long t256 = (long)((p.M + 255) / 256) * (p.N / bnx) * sp, t128 = (long)((p.M + 127) / 128) * (p.N / bnx) * sp;
      const long t64 = (long)((p.M + 63) / 64) * (p.N / bnx) * sp;
      if (bnx == 64) return run<T, 128, 64, 4, 1>(p, s);
      if (bnx == 32) return run<T, 128, 32, 4, 1>(p, s);
      if (p.rowstats) {
        if (t128 >= 400) return bnx == 160 ? run<T, 128, 160, 2, 2, 2, false, 0, true>(p, s) : run<T, 128, 128, 2, 2, 2, false, 0, true>(p, s);
        if (t64 <= num_cus()) return bnx == 160 ? run<T, 64, 160, 2, 2, 4, false, 0, true>(p, s) : run<T, 64, 128, 2, 2, 4, false, 0, true>(p, s);
        return bnx == 160 ? run<T, 64, 160, 2, 2, 2, false, 0, true>(p, s) : run<T, 64, 128, 2, 2, 2, false, 0, true>(p, s);
      }
      if (t256 >= 240) return bnx == 160 ? run<T, 256, 160, 4, 2, 3, false>(p, s) : run<T, 256, 128, 4, 2, 3, false>(p, s);
      if (t128 >= 400) return bnx == 160 ? run<T, 128, 160, 2, 2>(p, s) : run<T, 128, 128, 2, 2>(p, s);
      if (t64 <= num_cus()) return bnx == 160 ? run<T, 64, 160, 2, 2, 4>(p, s) : run<T, 64, 128, 2, 2, 4>(p, s);
      return bnx == 160 ? run<T, 64, 160, 2, 2>(p, s) : run<T, 64, 128, 2, 2>(p, s);
    }
  }
  if (p.rowstats) return dispatch_ln<T>(p, s);
  const int bn = (p.epi == EPI_GEGLU) ? 128 : (p.N % 160 == 0 ? 160 : (p.N % 128 == 0 ? 128 : (p.N % 64 == 0 ? 64 : 32)));
  // fewer than ~1.5 workgroups per CU with 128-row tiles: halve the M tile (2 co-resident
  // workgroups per CU are what hides the per-K-tile barrier)
  const bool small = (long)((p.M + 127) / 128) * (p.N / bn) < 400;
  // plenty of rows: 256-row tiles / 8 waves cut the operand bytes staged per FLOP (the K loop is
  // bound by global->LDS traffic, not by MFMA issue) as long as every CU still gets a workgroup
  // (the global->LDS path tops out near 12 TB/s chip-wide, i.e. needs ~70 KB in flight per CU).
  // One workgroup per CU then has to keep two tiles in flight itself: a 3-stage ring.
  const long t256 = (long)((p.M + 255) / 256) * (p.N / bn);
  const long t128 = (long)((p.M + 127) / 128) * (p.N / bn);
  const bool big = (g_big & 1) && t256 >= 240;
  // 128-row tiles that cannot put two workgroups on every CU: deeper ring, one workgroup per CU
  const bool deep = (g_big & 2) && !big && t128 >= 200 && t128 < 400;
  // at most one workgroup per CU anyway: the DMA round trip (~1.1 us) is then hidden only by the
  // workgroup's own ring, so run it four stages deep instead of two
  const bool mid8 = !big && bn >= 128 && p.epi != EPI_GEGLU && mid8_ok(t128, p.splits);
  // long K slices: 4 compute waves (64x80 each) + 4 loader waves, 4-stage ring - measured 3-15 % faster than the 8-wave
  // form on the >= 45-tile conv launches of the 16x16 / 32x32 maps and slower on short K (policy bit 1 turns it off)
  if (mid8 && !(g_big & 2) && (p.taps * (p.C0 + p.C1) / (int)(kRowBytes / sizeof(T))) / (p.splits > 1 ? p.splits : 1) >= 40)
    return bn == 160 ? run<T, 128, 160, 2, 2, 4, true, 4>(p, s) : run<T, 128, 128, 2, 2, 4, true, 4>(p, s);
  if (mid8) return bn == 160 ? run<T, 128, 160, 4, 2, 3, true>(p, s) : run<T, 128, 128, 4, 2, 3, true>(p, s);
  // 256-row tiles with four extra loader waves (12-wave workgroups): issuing an LDS-DMA instruction parks the issuing wave
  // for 60-185 cycles (MI355X_MICROARCH.md), which in the 8-wave form comes straight out of the MFMA stream.  Measured on the
  // B = 8, L = 64 layer shapes (tools/kbench.py): 3x3 convs with K >= 2880 5-17 % faster, GEGLU 5-10 %, K <= 960 GEMMs 5-13 %
  // slower (prologue / epilogue bound: the loader waves only add barrier participants) -> long K slices and GEGLU only.
  const int nk_slice = (p.taps * (p.C0 + p.C1) / (int)(kRowBytes / sizeof(T))) / (p.splits > 1 ? p.splits : 1);
  const bool ldr_big = (g_big & 32) && sizeof(T) == 2 && (nk_slice >= 24 || p.epi == EPI_GEGLU);   // (the fp32 instantiation spills)
  const long t64 = (long)((p.M + 63) / 64) * (p.N / bn) * (p.splits > 1 ? p.splits : 1);
  const bool lone = (g_big & 4) && small && !big && !deep && t64 <= num_cus();
  switch (bn) {
    case 160:
      if (big && ldr_big) return run<T, 256, 160, 4, 2, 3, true, 4>(p, s);
      return big ? ((g_big & 8) ? run<T, 256, 160, 4, 2, 3, true>(p, s) : run<T, 256, 160, 4, 2, 3, false>(p, s)) : deep ? run<T, 128, 160, 2, 2, 4>(p, s)
                 : lone ? run<T, 64, 160, 2, 2, 4>(p, s)
                 : small ? run<T, 64, 160, 2, 2>(p, s) : run<T, 128, 160, 2, 2>(p, s);
    case 128:
      if (big && ldr_big) return run<T, 256, 128, 4, 2, 3, true, 4>(p, s);
      return big ? ((g_big & 8) ? run<T, 256, 128, 4, 2, 3, true>(p, s) : run<T, 256, 128, 4, 2, 3, false>(p, s)) : deep ? run<T, 128, 128, 2, 2, 4>(p, s)
                 : lone ? run<T, 64, 128, 2, 2, 4>(p, s)
                 : small ? run<T, 64, 128, 2, 2>(p, s) : run<T, 128, 128, 2, 2>(p, s);
    case 64: return run<T, 128, 64, 4, 1>(p, s);
    default: return run<T, 128, 32, 4, 1>(p, s);
  }
}

}  // namespace

// allocates + clears this device's zero page now (otherwise the first launch does it: a hipMalloc + a synchronous memset,
// which a forward that promised not to allocate - ldmseg_unet_reserve - must not hit)
int igemm_warm() { return zero_page() ? 0 : -3; }
const void* igemm_zero_page() { return zero_page(); }
void igemm_set_tsbuf(void* b) { g_tsbuf = b; }
void igemm_force_cfg(int cfg) { g_force_cfg = cfg; }
void igemm_set_dbg(int f) { g_dbg = f & 0xff; g_big = (f >> 8) & 63; }   // bits 8-13 select the tile policy
int igemm_get_dbg() { return (g_big << 8) | g_dbg; }
int igemm_default_dbg() { return kDefaultPolicy << 8; }
IgemmDispatch igemm_last_dispatch() { return g_last; }
// "igemm<dtype,BM,BN,WM,WN,NST,PIPE,LDR>" + "/splitk" when the launch ran K slices (partial epilogue + finish kernel)
std::string igemm_dispatch_name(const IgemmDispatch& d) {
  char buf[96];
  std::snprintf(buf, sizeof buf, "igemm<%s,%d,%d,%d,%d,%d,%d,%d%s%s>%s", d.dtype == DT_BF16 ? "bf16" : "f32", d.bm, d.bn, d.wm, d.wn,
                d.nst, d.pipe, d.ldr, d.lnf ? ",ln" : "", d.cm ? ",cm" : (d.xt ? ",xt" : (d.up4 ? ",up4" : "")),
                d.splits > 1 ? (d.cf ? "/splitk-cf" : "/splitk") : "");
  return buf;
}
void igemm_log_enable(int on) { g_log_on = on != 0; if (on) g_log.clear(); }
void igemm_log_note(const char* name) { if (g_log_on) g_log.insert(name); }   // kernels of the GEMM family that are not igemm_kernel (tfuse.hip)
std::string igemm_log_read() {
  std::string out;
  for (const auto& e : g_log) { out += e; out += '\n'; }
  return out;
}

int igemm_pick_bn(int n_real, int epi) {
  if (epi == EPI_GEGLU) return 128;
  if (n_real % 160 == 0) return 160;
  if (n_real >= 128) return 128;
  if (n_real > 32) return 64;
  return 32;
}

// Split-K plan for grids that would leave most of the 256 CUs idle (the 8x8 / 16x16 feature maps):
// returns the number of K slices (1 = no split).  Mirrors dispatch()'s tile choice.
int igemm_plan_splits(const IgemmParams& p, int dtype) {
  if (p.epi != EPI_STORE || p.rowstats) return 1;
  if (p.up4) {                 // 256-row tiles: K slices until every CU has a work item, at least 10 K tiles per slice
    const long t256 = (long)(p.M / 256) * (p.N / 160);
    const int nk0 = 4 * (p.C0 + p.C1) / 64;
    int sp = t256 >= 240 ? 1 : (int)((num_cus() + t256 / 2) / t256);
    if (sp > nk0 / 10) sp = nk0 / 10;
    if (sp > 8) sp = 8;
    return sp < 2 ? 1 : sp;
  }
  if (const TunedEntry* e = tuned_lookup(p, dtype)) return e->splits;
  const int bke = dtype == DT_BF16 ? 64 : 32;
  const int bn = p.N % 160 == 0 ? 160 : (p.N % 128 == 0 ? 128 : (p.N % 64 == 0 ? 64 : 32));
  const long tiles128 = (long)((p.M + 127) / 128) * (p.N / bn);
  const int nk0 = p.taps * (p.C0 + p.C1) / bke;
  if ((g_big & 16) && bn >= 128 && tiles128 < 200 && ((long)((p.M + 255) / 256) * (p.N / bn)) < 240) {
    // 128-row 8-wave tiles: aim at one work item per CU, at least 10 K tiles per slice (a 20-tile
    // K=1280 GEMM measured faster unsplit on 64-row tiles than split in two plus the finish pass)
    int sp = (int)((num_cus() + tiles128 / 2) / tiles128);
    if (sp > nk0 / 10) sp = nk0 / 10;
    if (nk0 < 32) sp = 1;
    if (sp > 16) sp = 16;
    if (sp >= 2 && mid8_ok(tiles128, sp)) return sp;
  }
  const int bm = (bn >= 128 && tiles128 < 400) ? 64 : 128;
  const long blocks = (long)((p.M + bm - 1) / bm) * (p.N / bn);
  const int nk = p.taps * (p.C0 + p.C1) / bke;
  if (blocks >= 400 || nk < 24) return 1;
  int splits = (int)((512 + blocks - 1) / blocks);
  if (splits > nk / 12) splits = nk / 12;
  if (splits > 16) splits = 16;
  return splits < 2 ? 1 : splits;
}

size_t igemm_cf_bytes() { return 64 << 10; }
void igemm_set_cf_mode(int mode) { g_cf_mode = mode; }
void igemm_set_table_override(int v) { g_table_override = v; }
int igemm_get_cf_mode() { return g_cf_mode; }
size_t igemm_partial_bytes(const IgemmParams& p) {
  return p.splits > 1 ? (size_t)p.splits * p.M * p.N * sizeof(float) : 0;
}

int g_up4_mode = 1;
void igemm_set_up4_mode(int on) { g_up4_mode = on ? 1 : 0; }
int igemm_get_up4_mode() { return g_up4_mode; }
bool igemm_up4_ok(int B, int H, int W, int C, int N, int dtype) {
  return g_up4_mode && dtype == DT_BF16 && C % 64 == 0 && N % 160 == 0 && ((long)B * H * W) % 256 == 0 && g_big == kDefaultPolicy && g_force_cfg < 0;
}
namespace {
// OIHW fp32 3x3 -> [phase = 2 py + px][Npad][tap = 2 a + b][Cipad]: the taps of the 3x3 window that fall on the same source pixel of
// the low-resolution map, summed in fp32.  Row group a of phase py: py = 0 -> {ky 0}, {ky 1, 2}; py = 1 -> {ky 0, 1}, {ky 2}.
template <typename T>
__global__ void pack_up4_kernel(const float* __restrict__ w, T* __restrict__ out, int Co, int Ci, int Npad, int Cipad) {
  const size_t total = (size_t)4 * Npad * 4 * Cipad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cipad);
    size_t r = i / Cipad;
    const int tap = (int)(r & 3);
    r >>= 2;
    const int n = (int)(r % Npad), ph = (int)(r / Npad);
    float acc = 0.f;
    if (n < Co && c < Ci) {
      const int py = ph >> 1, px = ph & 1, a = tap >> 1, b = tap & 1;
      const int ky0 = py == 0 ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), ky1 = py == 0 ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
      const int kx0 = px == 0 ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), kx1 = px == 0 ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
      const float* wp = w + ((size_t)n * Ci + c) * 9;
      for (int ky = ky0; ky <= ky1; ++ky)
        for (int kx = kx0; kx <= kx1; ++kx) acc += wp[ky * 3 + kx];
    }
    out[i] = from_f32<T>(acc);
  }
}
}  // namespace
namespace {
// fp32 [rows][K] (K a multiple of 32) -> the same bytes as split-bf16 planes, in place: every 128-byte K tile (32 floats) becomes
// [hi: 4 chunks of 8 bf16 | lo: 4 chunks of 8 bf16] with hi = bf16(w), lo = bf16(w - hi) - exactly what the x3 K loop computes per
// fragment and tile - and chunk c holding k = 4c .. 4c+3, 16+4c .. 16+4c+3: the k set lane group c reads from an fp32 X tile (its
// chunks c and c + 4), so both MFMA operands see the same contraction order.  One thread per tile.
__global__ __launch_bounds__(256) void split_planes_kernel(float* __restrict__ w, size_t ntiles) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= ntiles) return;
  f32x4* p = (f32x4*)(w + t * 32);
  f32x4 f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = p[i];
  u32x4 hi[4], lo[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float v[8] = {f[c][0], f[c][1], f[c][2], f[c][3], f[c + 4][0], f[c + 4][1], f[c + 4][2], f[c + 4][3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned h = pack_bf16x2(v[2 * i], v[2 * i + 1]);
      const float r0 = v[2 * i] - bits_f32(h << 16), r1 = v[2 * i + 1] - bits_f32(h & 0xffff0000u);
      hi[c][i] = h;
      lo[c][i] = pack_bf16x2(r0, r1);
    }
  }
  u32x4* q = (u32x4*)p;
#pragma unroll
  for (int c = 0; c < 4; ++c) { q[c] = hi[c]; q[c + 4] = lo[c]; }
}
}  // namespace
int launch_split_planes(void* w, size_t nfloats, hipStream_t s) {
  if (nfloats % 32 != 0) return -2;
  const size_t ntiles = nfloats / 32;
  if (!ntiles) return 0;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, s, (float*)w, ntiles);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
int launch_pack_up4(const float* w, void* out, int Co, int Ci, int Npad, int Cipad, int dtype, hipStream_t s) {
  const size_t total = (size_t)16 * Npad * Cipad;
  const size_t blocks = (total + 255) / 256;
  const unsigned g = (unsigned)(blocks < 16384 ? blocks : 16384);
  if (dtype == DT_BF16) hipLaunchKernelGGL(pack_up4_kernel<bf16_t>, dim3(g), dim3(256), 0, s, w, (bf16_t*)out, Co, Ci, Npad, Cipad);
  else hipLaunchKernelGGL(pack_up4_kernel<float>, dim3(g), dim3(256), 0, s, w, (float*)out, Co, Ci, Npad, Cipad);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
void igemm_set_xt_mode(int on) { g_xt_mode = on ? 1 : 0; }
int igemm_get_xt_mode() { return g_xt_mode; }
bool igemm_xt_ok(const IgemmParams& p, int dtype) {
  return g_xt_mode && dtype == DT_BF16 && p.C2 > 0 && p.src2 && p.C2 % 64 == 0 && p.C3 % 64 == 0 && (p.C3 == 0 || p.src3) && p.taps == 9 && p.stride == 1 &&
         !p.up && !p.cm && (p.pad < 0 || p.pad == 1) && p.N % 160 == 0 && p.epi == EPI_STORE && !p.rowstats && g_big == kDefaultPolicy &&
         g_force_cfg < 0;
}

namespace {
template <typename T>
__global__ void concat_rows_kernel(const T* a, int K1, const T* b, int K2, T* out, size_t total) {
  const int K = K1 + K2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / K;
    const int k = (int)(i - n * K);
    out[i] = k < K1 ? a[n * K1 + k] : b[n * K2 + (k - K1)];
  }
}
}  // namespace
namespace {
__global__ void vec_add_kernel(const float* a, const float* b, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}
}  // namespace
namespace {
__global__ void chain_weights_kernel(const float* __restrict__ wp, const float* __restrict__ w2, const float* __restrict__ b2,
                                     const float* __restrict__ bp, float* __restrict__ wcat, float* __restrict__ bcat, int C) {
  const int n = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x, K4 = 4 * C;
  if (k < K4) {
    float acc = 0.f;
    for (int j = 0; j < C; ++j) acc = fmaf(wp[(size_t)n * C + j], w2[(size_t)j * K4 + k], acc);
    wcat[(size_t)n * (5 * C) + k] = acc;
  } else if (k < 5 * C) {
    wcat[(size_t)n * (5 * C) + k] = wp[(size_t)n * C + (k - K4)];
  }
  if (k == 0) {
    float acc = bp ? bp[n] : 0.f;
    if (b2)
      for (int j = 0; j < C; ++j) acc = fmaf(wp[(size_t)n * C + j], b2[j], acc);
    bcat[n] = acc;
  }
}
}  // namespace
int launch_chain_weights(const float* wp, const float* w2, const float* b2, const float* bp, float* wcat, float* bcat, int C, hipStream_t s) {
  hipLaunchKernelGGL(chain_weights_kernel, dim3((5 * C + 255) / 256, C), dim3(256), 0, s, wp, w2, b2, bp, wcat, bcat, C);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
int launch_vec_add(const float* a, const float* b, float* out, int n, hipStream_t s) {
  hipLaunchKernelGGL(vec_add_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, b, out, n);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
int launch_concat_rows(const void* a, int K1, const void* b, int K2, void* out, int N, int dtype, hipStream_t s) {
  const size_t total = (size_t)N * (K1 + K2);
  const size_t blocks = (total + 255) / 256;
  const unsigned g = (unsigned)(blocks < 16384 ? blocks : 16384);
  if (dtype == DT_BF16) hipLaunchKernelGGL(concat_rows_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)a, K1, (const bf16_t*)b, K2, (bf16_t*)out, total);
  else hipLaunchKernelGGL(concat_rows_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)a, K1, (const float*)b, K2, (float*)out, total);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

void igemm_set_cm_mode(int mode) { g_cm_mode = mode < -1 || mode > 1 ? -1 : mode; }
int igemm_get_cm_mode() { return g_cm_mode; }
bool igemm_conv_cm(int hw, int ctot, int n, int k, int stride, int up, int dtype) {
  if (k != 3 || stride != 1 || up || g_cm_mode == 0 || dtype != DT_BF16 || n % 160 != 0) return false;   // (what dispatch() can run)
  return g_cm_mode == 1 || (hw >= 4096 && ctot >= 640);
}

int launch_igemm(const IgemmParams& p, int dtype, hipStream_t s) {
  const int bke = dtype == DT_BF16 ? 64 : 32;
  if (p.M <= 0 || p.N <= 0 || p.N % 32 != 0) return -2;
  if (p.C0 % bke != 0 || p.C1 % bke != 0 || (p.C0 + p.C1) == 0) return -2;
  if (p.up4) {
    if (dtype != DT_BF16 || p.taps != 4 || p.stride != 1 || p.up || p.cm || p.C2 || p.epi != EPI_STORE || p.rowstats || p.resid || p.rowbias ||
        p.N % 160 != 0 || p.M % 1024 != 0 || p.M != 4 * p.B * p.Ho * p.Wo || p.Hi != p.Ho || p.Wi != p.Wo)
      return -2;
  } else if (p.taps != 1 && p.taps != 9) return -2;
  if (p.cm && (p.taps != 9 || p.stride != 1 || p.up || (p.pad >= 0 && p.pad != 1))) return -2;
  if (p.n_valid % 4 != 0 && p.epi != EPI_NCHW_F32) return -2;
  if (p.epi == EPI_GEGLU && p.N % 128 != 0) return -2;
  if (p.splits > 1 && (p.epi != EPI_STORE || p.partial == nullptr)) return -2;
  if (p.rowstats && (!p.c1 || p.rowbias || p.splits > 1 || (p.epi != EPI_STORE && p.epi != EPI_GEGLU))) return -2;
  if ((p.C2 > 0 || p.C3 > 0) && !igemm_xt_ok(p, dtype)) return -2;
  return dtype == DT_BF16 ? dispatch<bf16_t>(p, s) : dispatch<float>(p, s);
}

}  // namespace ldmseg
