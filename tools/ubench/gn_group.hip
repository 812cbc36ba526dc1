// Prototype (round 6): GroupNorm with ONE workgroup per (image, group) - the whole slice (HW pixels x C/32 channels, 20-240 bytes per
// pixel) lives in the registers of 512 threads, so there is no hand-off between workgroups (gn_coop_kernel's S splits of an (image,
// 4-group block) exchange partial statistics through memory: ~3 us) and the tensor is still read once and written once.  The price is
// 4-byte accesses (a group's channels are not 16-byte aligned) and partial cache lines; all groups of an image run on one XCD (block b
// -> XCD b % 8, image = b % B) so that every line is fetched into / written back from ONE L2.
//   hipcc --offload-arch=gfx950 -O3 -o bin/gn_group gn_group.hip && ./bin/gn_group
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>

template <int NV>
__global__ __launch_bounds__(512) void gn_group_k(const unsigned* __restrict__ x, unsigned* __restrict__ y, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, int B, int HW, int C, int dpp, int stepq, int stepr, float eps) {
  __shared__ float red[16];
  __shared__ float gb[2][128];
  const int tid = threadIdx.x;
  const int b = blockIdx.x % B, g = blockIdx.x / B;
  const int cpg = dpp * 2;
  const int total = HW * dpp;                       // dwords of this (image, group) slice
  const size_t base = (size_t)b * HW * (C / 2) + (size_t)g * dpp;   // in dwords
  if (tid < cpg) { gb[0][tid] = gamma[g * cpg + tid]; gb[1][tid] = beta[g * cpg + tid]; }
  unsigned v[NV];
  int pix = tid / dpp, j = tid - pix * dpp;         // (one division per thread)
  const int pix0 = pix, j0 = j;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int d = tid + i * 512;
    v[i] = d < total ? x[base + (size_t)pix * (C / 2) + j] : 0u;
    pix += stepq; j += stepr;
    if (j >= dpp) { j -= dpp; ++pix; }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int d = tid + i * 512;
    if (d < total) s += __builtin_bit_cast(float, v[i] << 16) + __builtin_bit_cast(float, v[i] & 0xffff0000u);
  }
  auto wg_sum = [&](float a) -> float {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = a;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t += red[w];
    return t;
  };
  const float inv_n = 1.0f / (float)(HW * cpg);
  const float mean = wg_sum(s) * inv_n;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int d = tid + i * 512;
    if (d < total) {
      const float a = __builtin_bit_cast(float, v[i] << 16) - mean, c = __builtin_bit_cast(float, v[i] & 0xffff0000u) - mean;
      q += a * a + c * c;
    }
  }
  const float rstd = 1.0f / sqrtf(wg_sum(q) * inv_n + eps);
  pix = pix0; j = j0;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int d = tid + i * 512;
    if (d < total) {
      const float a = (__builtin_bit_cast(float, v[i] << 16) - mean) * rstd * gb[0][2 * j] + gb[1][2 * j];
      const float c = (__builtin_bit_cast(float, v[i] & 0xffff0000u) - mean) * rstd * gb[0][2 * j + 1] + gb[1][2 * j + 1];
      typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
      typedef float f2 __attribute__((ext_vector_type(2)));
      const f2 p = {a, c};
      y[base + (size_t)pix * (C / 2) + j] = __builtin_bit_cast(unsigned, __builtin_convertvector(p, bf2));
    }
    pix += stepq; j += stepr;
    if (j >= dpp) { j -= dpp; ++pix; }
  }
}


// Timing probe: the 64x64 x 320 map as 16 slices of TWO groups (20 channels = 40 bytes per pixel, 8-byte accesses), NT threads per
// workgroup, B * 16 workgroups.  Statistics are taken over the pair (wrong as a GroupNorm, right as a cost model).
template <int NV, int NT>
__global__ __launch_bounds__(NT) void gn_pair_k(const uint2* __restrict__ x, uint2* __restrict__ y, int B, int HW, int C, int upp, int stepq, int stepr) {
  __shared__ float red[32];
  const int tid = threadIdx.x;
  const int b = blockIdx.x % B, g = blockIdx.x / B;
  const int total = HW * upp;
  const size_t base = (size_t)b * HW * (C / 4) + (size_t)g * upp;     // in uint2
  uint2 v[NV];
  int pix = tid / upp, j = tid - pix * upp;
  const int pix0 = pix, j0 = j;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int d = tid + i * NT;
    v[i] = x[base + (size_t)(d < total ? pix : 0) * (C / 4) + (d < total ? j : 0)];
    pix += stepq; j += stepr;
    if (j >= upp) { j -= upp; ++pix; }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += __builtin_bit_cast(float, v[i].x << 16) + __builtin_bit_cast(float, v[i].y & 0xffff0000u);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < NT / 64; ++w) t += red[w];
  const float mean = t / (float)(total * 4);
  pix = pix0; j = j0;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int d = tid + i * NT;
    uint2 o = v[i];
    o.x ^= (unsigned)(mean > 1e30f);
    if (d < total) y[base + (size_t)pix * (C / 4) + j] = o;
    pix += stepq; j += stepr;
    if (j >= upp) { j -= upp; ++pix; }
  }
}

static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
  struct Shape { int B, HW, C; };
  const Shape shapes[] = {{8, 4096, 320}, {8, 4096, 640}, {8, 4096, 960}, {8, 1024, 640}, {8, 1024, 1280}, {8, 1024, 1920}, {16, 4096, 320}};
  const int NB = 8;
  const size_t maxb = (size_t)16 * 4096 * 960 * 2;
  char* buf; float *gam, *bet;
  (void)hipMalloc(&buf, maxb * 2 * NB); (void)hipMalloc(&gam, 4096 * 4); (void)hipMalloc(&bet, 4096 * 4);
  std::vector<unsigned short> h(maxb / 2);
  for (size_t i = 0; i < h.size(); ++i) { float f = (float)((i * 2654435761u >> 16) & 1023) / 512.0f - 1.0f + (float)(i % 7) * 0.1f; unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
  for (int k = 0; k < NB; ++k) (void)hipMemcpy(buf + (size_t)k * 2 * maxb, h.data(), maxb, hipMemcpyHostToDevice);
  std::vector<float> gh(4096), bh(4096);
  for (int i = 0; i < 4096; ++i) { gh[i] = 1.0f + 0.01f * (i % 13); bh[i] = 0.02f * (i % 5); }
  (void)hipMemcpy(gam, gh.data(), 4096 * 4, hipMemcpyHostToDevice); (void)hipMemcpy(bet, bh.data(), 4096 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (const Shape& sh : shapes) {
    const int dpp = sh.C / 32 / 2, total = sh.HW * dpp, nv = (total + 511) / 512;
    const int stepq = 512 / dpp, stepr = 512 % dpp;
    const size_t bytes = (size_t)sh.B * sh.HW * sh.C * 2;
    auto launch = [&](const void* s, void* d) {
      const dim3 grid(sh.B * 32), block(512);
#define GO(N) hipLaunchKernelGGL(gn_group_k<N>, grid, block, 0, 0, (const unsigned*)s, (unsigned*)d, gam, bet, sh.B, sh.HW, sh.C, dpp, stepq, stepr, 1e-5f)
      if (nv <= 20) GO(20); else if (nv <= 40) GO(40); else if (nv <= 60) GO(60); else if (nv <= 80) GO(80); else if (nv <= 120) GO(120); else printf("nv %d too large\n", nv);
#undef GO
    };
    float ms[2];
    for (int cold = 0; cold < 2; ++cold) {
      const int iters = 64;
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        for (int it = 0; it < iters; ++it) {
          char* s = buf + (cold ? (size_t)(it % NB) * 2 * maxb : 0);
          launch(s, s + maxb);
        }
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms[cold], e0, e1);
        ms[cold] = ms[cold] * 1e3f / iters;
      }
    }
    // check image 1, group 3 against the host
    std::vector<unsigned short> out(bytes / 2);
    launch(buf, buf + maxb);
    (void)hipMemcpy(out.data(), buf + maxb, bytes, hipMemcpyDeviceToHost);
    const int cpg = sh.C / 32, b = 1, g = 3;
    double sm = 0, sq = 0;
    for (int p = 0; p < sh.HW; ++p) for (int c = 0; c < cpg; ++c) sm += bf2f(h[((size_t)b * sh.HW + p) * sh.C + g * cpg + c]);
    const double mean = sm / (sh.HW * cpg);
    for (int p = 0; p < sh.HW; ++p) for (int c = 0; c < cpg; ++c) { const double d = bf2f(h[((size_t)b * sh.HW + p) * sh.C + g * cpg + c]) - mean; sq += d * d; }
    const double rstd = 1.0 / sqrt(sq / (sh.HW * cpg) + 1e-5);
    double maxerr = 0;
    for (int p = 0; p < sh.HW; ++p) for (int c = 0; c < cpg; ++c) {
      const size_t idx = ((size_t)b * sh.HW + p) * sh.C + g * cpg + c;
      const double want = (bf2f(h[idx]) - mean) * rstd * gh[g * cpg + c] + bh[g * cpg + c];
      maxerr = fmax(maxerr, fabs(want - bf2f(out[idx])));
    }
    printf("B=%2d HW=%4d C=%4d (%5.1f MB, %3d dwords per thread): %6.1f us warm (%4.2f TB/s r+w)  %6.1f us cold   max |err| vs host %.3g\n", sh.B, sh.HW, sh.C,
           bytes / 1048576.0, nv, ms[0], 2.0 * bytes / ms[0] * 1e-6, ms[1], maxerr);
  }

  {  // pair probe
    const int B = 8, HW = 4096, C = 320, upp = 5;
    for (int nt : {512, 1024}) {
      float msb[2];
      for (int cold = 0; cold < 2; ++cold) {
        const int iters = 64;
        for (int rep = 0; rep < 2; ++rep) {
          (void)hipEventRecord(e0);
          for (int it = 0; it < iters; ++it) {
            char* sp = buf + (cold ? (size_t)(it % NB) * 2 * maxb : 0);
            if (nt == 512) hipLaunchKernelGGL((gn_pair_k<40, 512>), dim3(B * 16), dim3(512), 0, 0, (const uint2*)sp, (uint2*)(sp + maxb), B, HW, C, upp, 512 / upp, 512 % upp);
            else hipLaunchKernelGGL((gn_pair_k<20, 1024>), dim3(B * 16), dim3(1024), 0, 0, (const uint2*)sp, (uint2*)(sp + maxb), B, HW, C, upp, 1024 / upp, 1024 % upp);
          }
          (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
          (void)hipEventElapsedTime(&msb[cold], e0, e1);
          msb[cold] = msb[cold] * 1e3f / iters;
        }
      }
      printf("pair probe 64x64 x 320 as 16 two-group slices, %4d threads x 128 workgroups: %6.1f us warm  %6.1f us cold\n", nt, msb[0], msb[1]);
    }
  }
  return 0;
}
