// Shared device/host helpers for the wafer_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/wafer_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

#define WM_WAVE 64

// Launch-error check used by every entry point: returns the hipError_t (positive) to the caller.
#define WM_LAUNCH_CHECK()                              \
  do {                                                 \
    hipError_t _e = hipGetLastError();                 \
    if (_e != hipSuccess) return (int)_e;              \
  } while (0)

#define WM_REQUIRE(cond, code) \
  do {                         \
    if (!(cond)) return (code); \
  } while (0)

// f32 -> bf16 bits, round-to-nearest-even (a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and
// keeps NaNs NaN; see MI355X_MICROARCH "Correctness boundaries").
__device__ __forceinline__ uint16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float bf2f(uint16_t u) {
  return __builtin_bit_cast(float, ((uint32_t)u) << 16);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

// Cross-lane reductions on the DPP network and the gfx950 lane-swap instructions instead of ds_bpermute round
// trips (__shfl_xor): an all-reduce over aligned groups of W lanes is, in butterfly order 1, 2, 4, 8, 16, 32:
//   1, 2  quad_perm [1,0,3,2] / [2,3,0,1] (exact xor exchanges);
//   4, 8  row_half_mirror / row_mirror: lane i pairs with 7-i / 15-i, which after the earlier steps holds the same
//         value as lane i^4 / i^8 -- so the result is bit-identical to the xor butterfly in this order;
//   16    v_permlane16_swap (odd 16-lane rows of one operand <-> even rows of the other);
//   32    v_permlane32_swap (upper 32 lanes <-> lower 32 lanes).
// Six VALU-rate steps, no LDS traffic, no lgkmcnt waits.
template <int CTRL>
__device__ __forceinline__ float wm_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// (own, partner) of the lane 16 / 32 away, in an order that depends on the lane: use with commutative operations
// Inline asm, not __builtin_amdgcn_permlane{16,32}_swap: hipcc (ROCm 7.2) returns the FIRST result register for both
// halves of the builtin's result pair (tools/probes/dpp_reduce_probe.hip shows it; the disassembly stores v1 twice).
// The s_nop pair covers the VALU-write -> lane-swap-read wait states the compiler inserts for the builtin form.
__device__ __forceinline__ void wm_pair16(float v, float& a, float& b) {
  a = v;
  b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void wm_pair32(float v, float& a, float& b) {
  a = v;
  b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float wm_xor16_sum(float v) { float a, b; wm_pair16(v, a, b); return a + b; }
__device__ __forceinline__ float wm_xor32_sum(float v) { float a, b; wm_pair32(v, a, b); return a + b; }
__device__ __forceinline__ float wm_xor16_max(float v) { float a, b; wm_pair16(v, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float wm_xor32_max(float v) { float a, b; wm_pair32(v, a, b); return fmaxf(a, b); }

template <int W>
__device__ __forceinline__ float group_sum(float v) {
  static_assert(W >= 1 && W <= 64 && (W & (W - 1)) == 0, "group width");
  if constexpr (W >= 2) v += wm_dpp<0xB1>(v);
  if constexpr (W >= 4) v += wm_dpp<0x4E>(v);
  if constexpr (W >= 8) v += wm_dpp<0x141>(v);
  if constexpr (W >= 16) v += wm_dpp<0x140>(v);
  if constexpr (W >= 32) v = wm_xor16_sum(v);
  if constexpr (W >= 64) v = wm_xor32_sum(v);
  return v;
}
template <int W>
__device__ __forceinline__ float group_max(float v) {
  static_assert(W >= 1 && W <= 64 && (W & (W - 1)) == 0, "group width");
  if constexpr (W >= 2) v = fmaxf(v, wm_dpp<0xB1>(v));
  if constexpr (W >= 4) v = fmaxf(v, wm_dpp<0x4E>(v));
  if constexpr (W >= 8) v = fmaxf(v, wm_dpp<0x141>(v));
  if constexpr (W >= 16) v = fmaxf(v, wm_dpp<0x140>(v));
  if constexpr (W >= 32) v = wm_xor16_max(v);
  if constexpr (W >= 64) v = wm_xor32_max(v);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }
__device__ __forceinline__ float wave_max(float v) { return group_max<64>(v); }

// s_barrier that the COMPILER may not move memory operations across.  The intrinsic alone is IntrNoMem: in the k-loops
// (`s_waitcnt vmcnt(N)` asm; barrier; issue next stage asm; fragment reads) the issue asm pins the reads below the
// barrier, but in the steps that issue nothing (the last k-step; taps 7 and 8 of conv3x3_patch, which is fully unrolled)
// hipcc hoisted the fragment reads of the step ABOVE the barrier -- a wave then read weight rows other waves' DMA had
// not landed yet.  Found in round 3 as run-to-run differences in ~0.02 % of the outputs of the 64-channel layers at
// batch 512 (tools/probes/conv_determinism.py: 18-24 k of 103 M elements per launch; none at batch 64, where the
// loads land sooner).
//
// And it retires the wave's LDS reads first (`s_waitcnt lgkmcnt(0)`): the DMA ring kernels restage a buffer right
// behind the barrier that follows its last use, and hipcc software-pipelines the loop -- the last fragment reads of a
// step are ISSUED before that barrier and consumed after it.  A read still queued in a busy LDS (three workgroups of
// conv3x3_patch per CU) could then be overtaken by another wave's DMA into the same stage: one wave of one workgroup
// multiplied a whole tile with the NEXT slice's weight rows -- 8 or 16 output channels x 64 pixels grossly wrong, in
// 1 of ~10^3 .. 10^5 workgroups, only once the launch's traffic had filled L2 (tools/probes/conv_determinism*.py;
// cdna_hip_programming.md: "restage a buffer ... 1 phase after when an lgkmcnt before the reading phase's first barrier
// retired those reads").
__device__ __forceinline__ void wm_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// XCD-aware launch order.  Workgroups are dealt round-robin over the 8 XCDs (workgroups b and b + 8 share an L2), so
// neighbouring tiles -- which share operand rows, halos or whole operand panels -- miss in each other's L2 and every one
// fetches its own copy from the fabric.  Remap the hardware's linear workgroup id so that each XCD walks a CONTIGUOUS
// range of logical ids (bijective for any workgroup count: cdna_hip_programming.md, "XCD swizzle must be bijective").
// Placement is a speed matter only; nothing depends on it for correctness.
__device__ __forceinline__ uint32_t wm_xcd_swizzle(uint32_t pid, uint32_t nwg) {
  const uint32_t q = nwg >> 3, r = nwg & 7u, xcd = pid & 7u;
  const uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (pid >> 3);
}

// LDS byte address of a __shared__ pointer (what M0 / ds instructions take).
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t*)p;
}

// One global_load_lds_dwordx4: every lane fetches 16 bytes from its own `gsrc`; the wave's 1 KiB
// lands at LDS byte address `lds_base` (wave-uniform) + 16 * lane.  Issued from inline asm so that
// hipcc neither counts it nor drains it with an s_waitcnt vmcnt(0) before the next ds_read: the
// caller orders it with its own counted `s_waitcnt vmcnt(N)` followed by a barrier
// (cdna_hip_programming.md §5.7: M0 is written and restored inside the same statement).
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_base) {
  uint32_t keep;
  const uint32_t base = __builtin_amdgcn_readfirstlane(lds_base);
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(base)
      : "memory");
}

// Same, non-temporal (streamed-once data: does not displace re-used lines from L2 / Infinity Cache).
__device__ __forceinline__ void glds16_nt(const void* gsrc, uint32_t lds_base) {
  uint32_t keep;
  const uint32_t base = __builtin_amdgcn_readfirstlane(lds_base);
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(base)
      : "memory");
}

// Same with an LDS byte address the caller already has as an integer, M0 declared clobbered instead
// of saved and restored (the k-loops issue 6-10 of these per step: the save/restore pair and the
// generic->LDS pointer cast's null check were a third of the scalar instructions of the issue phase).
__device__ __forceinline__ void glds16_at(const void* gsrc, uint32_t lds_byte_addr) {
  const uint32_t base = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(base) : "memory", "m0");
}

__device__ __forceinline__ void glds16_nt_at(const void* gsrc, uint32_t lds_byte_addr) {
  const uint32_t base = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" : : "v"(gsrc), "s"(base) : "memory", "m0");
}

// Exact-form GELU x Phi(x) and its derivative Phi(x) + x phi(x), with erf by Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7, i.e. below float32 resolution of the result for |x| < 4 and four orders below the bf16
// rounding of every tensor these feed): one v_exp, one v_rcp and a degree-5 polynomial instead of libm's erff
// (~45 instructions, which made the GELU epilogue of the fc1 GEMM cost as much as its MFMA loop).  The exponential
// exp(-x^2 / 2) is shared between erf(x / sqrt 2) and the density phi(x).
__device__ __forceinline__ void wm_gelu_parts(float v, float& cdf, float& pdf) {
  const float ax = fabsf(v) * 0.70710678118654752f;           // |x| / sqrt 2
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
  const float e = __expf(-ax * ax);                            // exp(-x^2 / 2)
  float poly = fmaf(t, 1.061405429f, -1.453152027f);
  poly = fmaf(t, poly, 1.421413741f);
  poly = fmaf(t, poly, -0.284496736f);
  poly = fmaf(t, poly, 0.254829592f);
  const float erf_abs = fmaf(-poly * t, e, 1.0f);              // erf(|x| / sqrt 2)
  cdf = 0.5f * (1.0f + copysignf(erf_abs, v));
  pdf = 0.3989422804014327f * e;
}
__device__ __forceinline__ float wm_gelu(float v) {
  float c, p;
  wm_gelu_parts(v, c, p);
  return v * c;
}
__device__ __forceinline__ float wm_gelu_grad(float v) {
  float c, p;
  wm_gelu_parts(v, c, p);
  return fmaf(v, p, c);
}


// ---- bit-reproducible reductions across workgroups: NO floating-point atomics anywhere on the training path.
// f32 atomics add in arrival order, so two runs of one launch differ in the last bits and bf16 roundings downstream
// flip (round 2: 4-10 % run-to-run gradient noise from the BatchNorm statistics and the split-K weight gradients).
// Every cross-workgroup sum is now "partials as plain stores into per-workgroup slots, summed later in slot order":
// the convolution epilogues store a tile's per-channel sums into the tile's slot, the weight-gradient kernels store
// split z's partial sums into slab z, and the BatchNorm finalize / weight-gradient fold kernels add slots in a fixed
// order.  (Two fixed-point integer-atomic forms were built first -- exact, but 64-bit atomics into a few hundred
// addresses cost the 64-channel layers 50 - 75 us per launch: profiles/r03_experiments.md.)

static inline int wm_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Division of a 31-bit index by a run-time constant (image sides, channels / 8, images per group) as a multiply-high,
// an add and a shift instead of the ~40-instruction expansion of an integer division: with l = ceil(log2 d) and
// m = ceil(2^(32+l) / d) - 2^32, floor(x / d) = (umulhi(x, m) + x) >> l for every x < 2^31 (the error term
// x (m d - 2^(32+l)) stays below 2^(32+l)).  The index decode of the per-element kernels and the prologues of the
// tile kernels were a measurable share of their instruction streams (stem / layer1 patch kernels: -20 .. -40 % time).
struct WmDiv {
  uint32_t m, l, d;
};
static inline WmDiv wm_div_make(uint32_t d) {
  WmDiv v;
  if (d == 0) d = 1;  // (never a valid divisor here; keeps the host side total)
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  v.l = l;
  v.d = d;
  v.m = (uint32_t)((((1ull << (32 + l)) + d - 1) / d) - (1ull << 32));
  return v;
}
__device__ __forceinline__ uint32_t wm_div(uint32_t x, const WmDiv& v) { return (__umulhi(x, v.m) + x) >> v.l; }
// quotient and remainder
__device__ __forceinline__ uint32_t wm_divmod(uint32_t x, const WmDiv& v, uint32_t& rem) {
  const uint32_t q = wm_div(x, v);
  rem = x - q * v.d;
  return q;
}

// Zero `n_words` 32-bit words with a KERNEL.  Not hipMemsetAsync: captured as a memset NODE inside the ~400-node
// hipGraph of a training step, the clear of the NT-Xent gradient buffer was not reliably applied before the
// kernel that adds into it (first non-finite tensor of the bs-256 run: that buffer, with finite inputs; a fill
// kernel in its place: 400 replays finite).  The library therefore issues no memset at all
// (tests/test_abi.py checks the sources); profiles/r02_nan_root_cause.md has the bisection.
__global__ __launch_bounds__(256) static void wm_zero_words_kernel(uint32_t* __restrict__ p, long long n_words) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (long long)gridDim.x * 256) p[i] = 0u;
}
static inline hipError_t wm_zero_async(void* p, size_t bytes, hipStream_t st) {
  const long long n_words = (long long)(bytes / 4);
  if (n_words == 0) return hipSuccess;
  const long long blocks = (n_words + 255) / 256;
  wm_zero_words_kernel<<<(int)(blocks < 2048 ? blocks : 2048), 256, 0, st>>>(static_cast<uint32_t*>(p), n_words);
  return hipGetLastError();
}
