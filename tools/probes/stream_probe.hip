// Stand-alone probe (GPU box): achievable read bandwidth of (a) plain 16-B/lane loads to registers and
// (b) global_load_lds into an LDS ring, for a buffer of `mb` megabytes.  Build & run:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_probe tools/probes/stream_probe.hip && /tmp/stream_probe 208
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

__global__ __launch_bounds__(256) void k_regs(const uint4* __restrict__ p, size_t n16, uint4* out) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
    acc.x ^= a.x ^ b.x ^ c.x ^ d.x; acc.y ^= a.y ^ b.y ^ c.y ^ d.y;
    acc.z ^= a.z ^ b.z ^ c.z ^ d.z; acc.w ^= a.w ^ b.w ^ c.w ^ d.w;
  }
  for (; i < n16; i += stride) { const uint4 a = p[i]; acc.x ^= a.x; acc.y ^= a.y; acc.z ^= a.z; acc.w ^= a.w; }
  if (acc.x == 0x12345678 && acc.y == 1) out[0] = acc;
}

__device__ __forceinline__ void glds16(const void* g, uint32_t lds_base, bool nt) {
  uint32_t keep;
  const uint32_t base = __builtin_amdgcn_readfirstlane(lds_base);
  if (nt)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(base) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(base) : "memory");
}

// each wave streams its own interleaved 8-KB pieces through a private 2- or 4-slot LDS ring; no barrier
template <int SLOTS, bool NT>
__global__ __launch_bounds__(256) void k_glds(const uint8_t* __restrict__ p, size_t bytes, uint32_t* out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint8_t* ring = smem + wave * SLOTS * 8192;
  const size_t npieces = bytes / 8192;
  const size_t gw = (size_t)blockIdx.x * 4 + wave, nw = (size_t)gridDim.x * 4;
  const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)ring;
  auto issue = [&](size_t piece, int slot) {
    const uint8_t* src = p + piece * 8192 + lane * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) glds16(src + i * 1024, lbase + slot * 8192 + i * 1024, NT);
  };
  size_t piece = gw;
  int issued = 0;
  for (int s = 0; s < SLOTS - 1 && piece + (size_t)s * nw < npieces; ++s) { issue(piece + (size_t)s * nw, s); ++issued; }
  uint32_t acc = 0;
  int it = 0;
  for (; piece < npieces; piece += nw, ++it) {
    const size_t nxt = piece + (size_t)(SLOTS - 1) * nw;
    if (nxt < npieces) {
      if constexpr (SLOTS == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (nxt < npieces) issue(nxt, (it + SLOTS - 1) % SLOTS);
    acc ^= *reinterpret_cast<const uint32_t*>(ring + (it % SLOTS) * 8192 + lane * 4);
  }
  if (acc == 0x12345678) out[0] = acc;
}

int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? atoi(argv[1]) : 208;
  const size_t bytes = mb << 20;
  uint8_t* p; uint4* out;
  hipMalloc(&p, bytes); hipMalloc(&out, 64); hipMemset(p, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %6zu MB  %8.1f us  %7.1f GB/s\n", name, mb, ms / reps * 1e3, bytes / (ms / reps) / 1e6);
  };
  for (int blocks : {256, 512, 1024, 2048}) {
    char nm[64];
    snprintf(nm, 64, "regs x4 grid=%d", blocks);
    timeit(nm, [&] { k_regs<<<blocks, 256>>>((const uint4*)p, bytes / 16, out); });
  }
  hipFuncSetAttribute((const void*)k_glds<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)k_glds<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)k_glds<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int blocks : {256, 512}) {
    char nm[64];
    snprintf(nm, 64, "glds ring4 grid=%d", blocks);
    timeit(nm, [&] { k_glds<4, false><<<blocks, 256, 4 * 4 * 8192>>>(p, bytes, (uint32_t*)out); });
    snprintf(nm, 64, "glds ring4 nt grid=%d", blocks);
    timeit(nm, [&] { k_glds<4, true><<<blocks, 256, 4 * 4 * 8192>>>(p, bytes, (uint32_t*)out); });
  }
  for (int blocks : {512, 1024}) {
    char nm[64];
    snprintf(nm, 64, "glds ring2 grid=%d", blocks);
    timeit(nm, [&] { k_glds<2, false><<<blocks, 256, 2 * 4 * 8192>>>(p, bytes, (uint32_t*)out); });
  }
  return 0;
}
