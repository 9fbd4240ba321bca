// Does a counted s_waitcnt vmcnt(N) retire LDS-DMA loads in issue order, and is the data readable right after the wait?
//   mode 0: DMA A (cold: a line never touched before, misses to HBM), DMA B (hot: the same 1 KB every time, L2 hit),
//           s_waitcnt vmcnt(1), ds_read A at once.  A still holds the sentinel => B retired before the older A.
//   mode 1: DMA A (cold), s_waitcnt vmcnt(0), ds_read A at once.  Sentinel => vmcnt retired before the data was readable.
//   mode 2: as mode 0 with four waves per block, wave w reading wave (w + 1) % 4's A after vmcnt(1) + s_barrier.
// Build: hipcc --offload-arch=gfx950 -O2 -o ldsdma_order_probe ldsdma_order_probe.hip ; run: ./ldsdma_order_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void glds16_at(const void* gsrc, uint32_t lds_byte_addr) {
  const uint32_t base = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(base) : "memory", "m0");
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t*)p;
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(const uint4* __restrict__ cold, size_t cold_lines, const uint4* __restrict__ hot,
                                             int iters, unsigned long long* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint4 lds[4][2][64];  // [wave][A, B][lane]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long stale = 0, total = 0;
  for (int it = 0; it < iters; ++it) {
    lds[wave][0][lane] = make_uint4(0xdeadbeefu, 0, 0, 0);
    lds[wave][1][lane] = make_uint4(0xdeadbeefu, 0, 0, 0);
    __syncthreads();
    const size_t line = ((size_t)(blockIdx.x * 4 + wave) * iters + it) * 2654435761ull % cold_lines;
    const uint4* a = cold + line * 64 + lane;
    if (MODE != 1) {
      glds16_at(a, lds_addr(&lds[wave][0][0]));
      glds16_at(hot + lane, lds_addr(&lds[wave][1][0]));
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    } else {
      glds16_at(a, lds_addr(&lds[wave][0][0]));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    int rw = wave;
    if (MODE == 2) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      rw = (wave + 1) & 3;
    }
    const uint32_t vx = *reinterpret_cast<volatile uint32_t*>(&lds[rw][0][lane]);
    stale += (vx == 0xdeadbeefu);
    ++total;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  atomicAdd(out, stale);
  atomicAdd(out + 1, total);
}

int main() {
  const size_t cold_lines = (size_t)1 << 21;  // 2 Mi lines x 1 KiB = 2 GiB
  uint4 *cold, *hot;
  unsigned long long* out;
  if (hipMalloc(&cold, cold_lines * 1024) != hipSuccess || hipMalloc(&hot, 1024) != hipSuccess ||
      hipMalloc(&out, 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(cold, 1, cold_lines * 1024);   // (never the sentinel pattern)
  hipMemset(hot, 2, 1024);
  for (int mode = 0; mode < 3; ++mode) {
    for (int blocks : {256, 2048}) {
      hipMemset(out, 0, 16);
      if (mode == 0) probe<0><<<blocks, 256>>>(cold, cold_lines, hot, 400, out);
      if (mode == 1) probe<1><<<blocks, 256>>>(cold, cold_lines, hot, 400, out);
      if (mode == 2) probe<2><<<blocks, 256>>>(cold, cold_lines, hot, 400, out);
      unsigned long long h[2];
      hipDeviceSynchronize();
      hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
      printf("mode %d blocks %4d: stale lane-reads %llu of %llu\n", mode, blocks, h[0], h[1]);
    }
  }
  return 0;
}
