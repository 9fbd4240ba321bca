// LayerNorm on token rows that already sit in registers as MFMA B-operand fragments (panel.hip, mlp.hip): a lane holds,
// for each of its two rows (row = fr of the lane), the 8-column pieces ks * 32 + fg * 8 .. + 7 of the KS k-steps; the
// four lanes fr, fr + 16, fr + 32, fr + 48 hold one row between them.  Same arithmetic as ln_fwd (transformer.hip): two
// passes (mean, then centred squares), y = fma((x - mean) * rstd, gamma, beta), rounded to bf16 -- so the fragments
// afterwards are what a separate LayerNorm launch would have written and the GEMM would have read back.
#pragma once
#include "common.h"

template <int KS>
__device__ __forceinline__ void wm_ln_fragments(bf16x8_t (&xf)[2][KS], const float* __restrict__ gamma,
                                                const float* __restrict__ beta, float eps, int fg) {
  constexpr int C = KS * 32;
  const float inv_c = 1.f / (float)C;
  // (the bf16 pieces are converted again in each pass instead of being kept as 48 floats per row: registers)
  auto val = [](const bf16x8_t& p, int e) -> float {
    const s16x8_t b = __builtin_bit_cast(s16x8_t, p);
    return bf2f((uint16_t)b[e]);
  };
  float mu[2], r[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) s += val(xf[i][ks], e);
    mu[i] = wm_xor32_sum(wm_xor16_sum(s)) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = val(xf[i][ks], e) - mu[i];
        q = fmaf(d, d, q);
      }
    r[i] = rsqrtf(wm_xor32_sum(wm_xor16_sum(q)) * inv_c + eps);
  }
  // gamma / beta: one k-step's 16 values at a time, shared by the two rows; the compiler fence keeps hipcc from issuing
  // all the loads up front (48 float4 loads = 192 registers when every (row, k-step) pair fetched its own: spills)
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    asm volatile("" ::: "memory");
    const float4 g0 = *reinterpret_cast<const float4*>(gamma + ks * 32 + fg * 8);
    const float4 g1 = *reinterpret_cast<const float4*>(gamma + ks * 32 + fg * 8 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(beta + ks * 32 + fg * 8);
    const float4 b1 = *reinterpret_cast<const float4*>(beta + ks * 32 + fg * 8 + 4);
    const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      s16x8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(fmaf((val(xf[i][ks], e) - mu[i]) * r[i], gg[e], bb[e]));
      xf[i][ks] = __builtin_bit_cast(bf16x8_t, o);
    }
  }
  asm volatile("" ::: "memory");
}
