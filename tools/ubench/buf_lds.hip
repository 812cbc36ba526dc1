// Probe: buffer_load_dwordx4 ... offen lds on gfx950 - lane -> LDS mapping, the scalar offset operand, and what an
// out-of-range offset writes (igemm's padding taps rely on it writing zeros).
//   hipcc --offload-arch=gfx950 -O3 -o bin/buf_lds buf_lds.hip && ./bin/buf_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void blds16(unsigned voff, i32x4 srd, unsigned soff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(srd), "s"(soff), "s"(lds_dst) : "memory");
}

__global__ void k(const unsigned* src, unsigned nbytes, unsigned* out, unsigned soff, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4 * 2];
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const unsigned long long base = (unsigned long long)(uintptr_t)src;
  i32x4 srd;
  srd[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
  srd[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32));
  srd[2] = __builtin_amdgcn_readfirstlane((int)nbytes);
  srd[3] = 0x00020000;
  unsigned voff = threadIdx.x * 16;                       // mode 0: plain
  if (mode == 1 && (threadIdx.x & 3) == 1) voff |= 0x80000000u;          // some lanes out of range (bit 31)
  if (mode == 2 && (threadIdx.x & 3) == 1) voff = nbytes + threadIdx.x * 16;   // just past the end
  if (mode == 3) voff = (63 - threadIdx.x) * 16;          // permuted sources: does LDS placement follow the lane?
  if (mode == 4 && (threadIdx.x & 3) == 1) voff = nbytes - 8;             // straddles the end
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned*)lds));
  blds16(voff, srd, __builtin_amdgcn_readfirstlane(soff), dst);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}

int main() {
  const unsigned n = 4096;   // dwords
  std::vector<unsigned> h(n);
  for (unsigned i = 0; i < n; ++i) h[i] = i;
  unsigned *d, *o;
  (void)hipMalloc(&d, n * 4 * 2); (void)hipMalloc(&o, 512 * 4);
  (void)hipMemset(d, 0x77, n * 8);
  (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  std::vector<unsigned> r(512);
  for (int mode = 0; mode <= 4; ++mode)
    for (unsigned soff : {0u, 1024u, 8192u}) {
      const unsigned nbytes = 8192;    // records: the first 2048 dwords
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, nbytes, o, soff, mode);
      (void)hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
      printf("mode %d soff %5u: lane0 %08x %08x %08x %08x | lane1 %08x %08x %08x %08x | lane2 %08x .. | lane63 %08x ; dword 256: %08x\n", mode, soff,
             r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[252], r[256]);
    }
  return 0;
}
