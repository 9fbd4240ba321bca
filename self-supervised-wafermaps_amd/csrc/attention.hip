// Multi-head self-attention for short sequences (ViT-S/16: 197 / 37 tokens, ViT-B/32: 50 / 13),
// head dim 64, on MFMA.
//
// Replaces the Attention.forward of facebookresearch/dino's vision_transformer (q k^T * scale ->
// softmax -> @ v, as run by scripts/WM811k_benchmark.py:566-576) and torch.nn.MultiheadAttention
// inside torchvision's vit_b_32 encoder blocks (MAE, :903-947), and their autograd backward.
//
// One block (8 waves) per (image, head).  K, V (and Q, dO in the backward) of the head sit in LDS as
// [token][64] bf16 rows padded to 144 B: a ds_read_b128 fragment read (16 rows x one 16-byte column)
// then touches 16 distinct 4-bank groups.  A wave owns 16-row strips.
//
//   forward, strip of 16 queries:  S^T tile = mfma(A = K rows, B = Q rows): lane (fr, fg) holds the
//   scores of query fr against keys 16 t + 4 fg + e.  Softmax: lane-local over (t, e), then across
//   the four lanes fr + 16 fg.  The probabilities are already in the layout of an MFMA B operand
//   whose k-slot (fg, e8) means key 32 kk + 4 fg + e8 (e8 < 4) / 32 kk + 16 + 4 fg + e8 - 4: the A
//   operand V^T is read with ds_read_b64_tr_b16 using that same slot <-> key map, so P never goes
//   through LDS.  O^T tile = mfma(A = V^T, B = P): lane holds 4 consecutive d of query fr.
//
//   backward: pass A (query strips) recomputes P and dP = dO V^T the same way, forms
//   dS = P (dP - delta) scale in registers and dQ^T = mfma(A = K^T (tr read), B = dS).
//   Pass B (key strips) recomputes the transposed tiles S[q][key] = mfma(A = Q rows, B = K rows), so
//   that P^T / dS^T are B operands with k = query, and dV^T = mfma(A = dO^T (tr), B = P^T),
//   dK^T = mfma(A = Q^T (tr), B = dS^T).  No atomics, no transposes through LDS; S and dP are
//   computed twice (cheap at these lengths).
//
// Roofline: MFMA-issue/LDS bound, but ~8 % of a ViT-S block's FLOPs (4 S^2 64 per head vs the
// 24 S 384^2 of its four Linear layers at S = 197): correctness first, tuning later.
#include "common.h"
#include <stdlib.h>

namespace {

// forward: 8 waves (two per SIMD: one wave's softmax / LDS latency hides under the other's MFMAs);
// backward: 4 waves (it needs > 128 registers per lane; measured 1.4x slower with 8)
constexpr int AT_FWD_THREADS = 512;
constexpr int AT_BWD_THREADS = 256;
// LDS bytes per token row: HD bf16 + 16 B pad (144 for head dim 64, 80 for 32: in both, 16
// consecutive rows start on 16 distinct 4-bank groups)
#define AT_ROWB (HD * 2 + 16)

template <int HD>
__device__ __forceinline__ bf16x8_t frag_rows(const uint8_t* base, int row, int ks, int fg) {
  return *reinterpret_cast<const bf16x8_t*>(base + row * AT_ROWB + ks * 64 + fg * 16);
}

// Operand with k running along the ROWS of the LDS image: 32 rows from row0, the 16 columns of
// block blk.  k-slot (fg, e) <-> row0 + 4 fg + e (e < 4) / row0 + 16 + 4 fg + (e - 4).
template <int HD>
__device__ __forceinline__ bf16x8_t frag_cols(const uint8_t* base, int row0, int blk, int lane) {
  const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  const uint8_t* p = base + (row0 + 4 * tg + tq) * AT_ROWB + blk * 32 + 8 * tp;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
  const s16x4_t hi =
      __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 16 * AT_ROWB));
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ bf16x8_t pack_slots(const f32x4_t a, const f32x4_t b) {
  const s16x8_t v = {(short)f2bf(a[0]), (short)f2bf(a[1]), (short)f2bf(a[2]), (short)f2bf(a[3]),
                     (short)f2bf(b[0]), (short)f2bf(b[1]), (short)f2bf(b[2]), (short)f2bf(b[3])};
  return __builtin_bit_cast(bf16x8_t, v);
}

// One fragment (row, 32-wide k-slab ks, 16-byte piece fg) of a [token][HD] operand straight from global memory:
// what frag_rows reads from an LDS image.  Rows >= S read as zeros.  The forward kernel takes a strip's own query rows
// this way (only that wave uses them): the request is in flight under the K / V staging, and Q needs no LDS.
__device__ __forceinline__ bf16x8_t frag_global(const uint16_t* base, size_t row_stride, int row, int S, int ks, int fg) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (row < S) v = *reinterpret_cast<const uint4*>(base + (size_t)row * row_stride + ks * 32 + fg * 8);
  return __builtin_bit_cast(bf16x8_t, v);
}

// rows [0, S) of one [token][64] operand of (image b, head h) -> LDS; rows [S, SP) zero
template <int HD>
__device__ __forceinline__ void stage_rows(const uint16_t* src, size_t row_stride, int S, int SP, uint8_t* dst) {
  constexpr int CPR = HD / 8;  // 16-byte chunks per row
  for (int i = threadIdx.x; i < SP * CPR; i += blockDim.x) {
    const int r = i / CPR, c = i % CPR;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < S) v = *reinterpret_cast<const uint4*>(src + (size_t)r * row_stride + c * 8);
    *reinterpret_cast<uint4*>(dst + r * AT_ROWB + c * 16) = v;
  }
}

template <int NT, int HD>
__global__ __launch_bounds__(AT_FWD_THREADS) void attn_fwd(const uint16_t* __restrict__ qkv, int S, int H, float scale,
                                                       uint16_t* __restrict__ out, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) uint8_t at_smem[];
  constexpr int SP = NT * 16;
  uint8_t* sk = at_smem;
  uint8_t* sv = sk + SP * AT_ROWB;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  constexpr int KS = HD / 32, DJ = HD / 16;
  const size_t rs = (size_t)3 * H * HD;
  const uint16_t* base = qkv + (size_t)b * S * rs + h * HD;
  // this wave's first query strip: requested before the K / V staging so that it is in flight under it
  bf16x8_t qf[KS];
  // gridDim.y blocks share the query strips of one (image, head) (each stages K and V itself): 197 tokens are 13 strips
  // for 8 waves, i.e. two serial strips per block -- with two blocks per head every wave has one
  const int qs0 = wave + (int)blockIdx.y * (AT_FWD_THREADS / 64);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = frag_global(base, rs, qs0 * 16 + fr, S, ks, fg);
  stage_rows<HD>(base + (size_t)H * HD, rs, S, SP, sk);
  stage_rows<HD>(base + (size_t)2 * H * HD, rs, S, SP, sv);
  __syncthreads();

  for (int qs = qs0; qs < NT; qs += (AT_FWD_THREADS / 64) * (int)gridDim.y) {
    const int q = qs * 16 + fr;
    if (qs != qs0) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) qf[ks] = frag_global(base, rs, q, S, ks, fg);
    }
    f32x4_t sc[NT];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4_t a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<HD>(sk, t * 16 + fr, ks, fg), qf[ks], a, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int key = t * 16 + 4 * fg + e;
        a[e] = key < S ? a[e] * scale : -INFINITY;
        m = fmaxf(m, a[e]);
      }
      sc[t] = a;
    }
    m = wm_xor32_max(wm_xor16_max(m));  // lane swaps, no LDS round trip (common.h)
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sc[t][e] = expf(sc[t][e] - m);
        l += sc[t][e];
      }
    l = wm_xor32_sum(wm_xor16_sum(l));
    f32x4_t o[DJ];
#pragma unroll
    for (int j = 0; j < DJ; ++j) o[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < NT / 2; ++kk) {
      const bf16x8_t pf = pack_slots(sc[2 * kk], sc[2 * kk + 1]);
#pragma unroll
      for (int j = 0; j < DJ; ++j)
        o[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols<HD>(sv, kk * 32, j, lane), pf, o[j], 0, 0, 0);
    }
    if (q < S) {
      const float inv = 1.f / l;
      uint16_t* dst = out + ((size_t)(b * S + q) * H + h) * HD + 4 * fg;
#pragma unroll
      for (int j = 0; j < DJ; ++j)
        *reinterpret_cast<uint2*>(dst + j * 16) =
            make_uint2(pack_bf2(o[j][0] * inv, o[j][1] * inv), pack_bf2(o[j][2] * inv, o[j][3] * inv));
      if (fg == 0) lse[((size_t)b * H + h) * S + q] = m + logf(l);
    }
  }
}

template <int NT, int HD>
__global__ __launch_bounds__(AT_BWD_THREADS) void attn_bwd(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out,
                                                       const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                                                       int S, int H, float scale, uint16_t* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) uint8_t at_smem[];
  constexpr int SP = NT * 16;
  uint8_t* sq = at_smem;
  uint8_t* sk = sq + SP * AT_ROWB;
  uint8_t* sv = sk + SP * AT_ROWB;
  uint8_t* sdo = sv + SP * AT_ROWB;
  float* s_lse = reinterpret_cast<float*>(sdo + SP * AT_ROWB);
  float* s_delta = s_lse + SP;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  constexpr int KS = HD / 32, DJ = HD / 16, CPR = HD / 8;
  const size_t rs = (size_t)3 * H * HD, os = (size_t)H * HD;
  const uint16_t* base = qkv + (size_t)b * S * rs + h * HD;
  const uint16_t* obase = out + (size_t)b * S * os + h * HD;
  const uint16_t* dobase = dout + (size_t)b * S * os + h * HD;
  stage_rows<HD>(base, rs, S, SP, sq);
  stage_rows<HD>(base + (size_t)H * HD, rs, S, SP, sk);
  stage_rows<HD>(base + (size_t)2 * H * HD, rs, S, SP, sv);
  // dO -> LDS and delta[q] = sum_d dO[q][d] O[q][d] (CPR adjacent lanes per row)
  for (int i = threadIdx.x; i < SP * CPR; i += AT_BWD_THREADS) {
    const int r = i / CPR, c = i % CPR;
    uint4 dv = make_uint4(0, 0, 0, 0), ov = make_uint4(0, 0, 0, 0);
    if (r < S) {
      dv = *reinterpret_cast<const uint4*>(dobase + (size_t)r * os + c * 8);
      ov = *reinterpret_cast<const uint4*>(obase + (size_t)r * os + c * 8);
    }
    *reinterpret_cast<uint4*>(sdo + r * AT_ROWB + c * 16) = dv;
    const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w}, ow[4] = {ov.x, ov.y, ov.z, ov.w};
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      d = fmaf(bf2f((uint16_t)(dw[e] & 0xffff)), bf2f((uint16_t)(ow[e] & 0xffff)), d);
      d = fmaf(bf2f((uint16_t)(dw[e] >> 16)), bf2f((uint16_t)(ow[e] >> 16)), d);
    }
    d = group_sum<CPR>(d);
    if (c == 0) {
      s_delta[r] = d;
      s_lse[r] = r < S ? lse[((size_t)b * H + h) * S + r] : 0.f;
    }
  }
  __syncthreads();

  uint16_t* dq_base = dqkv + (size_t)b * S * rs + h * HD;
  // ---- pass A: dQ, by query strips
  for (int qs = wave; qs < NT; qs += AT_BWD_THREADS / 64) {
    const int q = qs * 16 + fr;
    bf16x8_t qf[KS], df[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = frag_rows<HD>(sq, q, ks, fg);
      df[ks] = frag_rows<HD>(sdo, q, ks, fg);
    }
    const float lq = s_lse[q], dl = s_delta[q];
    bf16x8_t dsf[NT / 2];
#pragma unroll
    for (int kk = 0; kk < NT / 2; ++kk) {
      f32x4_t ds2[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * kk + u;
        f32x4_t a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<HD>(sk, t * 16 + fr, ks, fg), qf[ks], a, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<HD>(sv, t * 16 + fr, ks, fg), df[ks], dp, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int key = t * 16 + 4 * fg + e;
          const float p = key < S ? expf(a[e] * scale - lq) : 0.f;
          ds2[u][e] = p * (dp[e] - dl) * scale;
        }
      }
      dsf[kk] = pack_slots(ds2[0], ds2[1]);
    }
    f32x4_t dq[DJ];
#pragma unroll
    for (int j = 0; j < DJ; ++j) dq[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < NT / 2; ++kk)
#pragma unroll
      for (int j = 0; j < DJ; ++j)
        dq[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols<HD>(sk, kk * 32, j, lane), dsf[kk], dq[j], 0, 0, 0);
    if (q < S) {
      uint16_t* dst = dq_base + (size_t)q * rs + 4 * fg;
#pragma unroll
      for (int j = 0; j < DJ; ++j)
        *reinterpret_cast<uint2*>(dst + j * 16) =
            make_uint2(pack_bf2(dq[j][0], dq[j][1]), pack_bf2(dq[j][2], dq[j][3]));
    }
  }
  // ---- pass B: dK, dV, by key strips (transposed tiles: lane = key fr, registers = queries)
  for (int ksn = wave; ksn < NT; ksn += AT_BWD_THREADS / 64) {
    const int key = ksn * 16 + fr;
    const bool keyok = key < S;
    bf16x8_t kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = frag_rows<HD>(sk, key, ks, fg);
      vf[ks] = frag_rows<HD>(sv, key, ks, fg);
    }
    bf16x8_t pf[NT / 2], dsf[NT / 2];
#pragma unroll
    for (int kk = 0; kk < NT / 2; ++kk) {
      f32x4_t p2[2], ds2[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * kk + u;
        f32x4_t a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<HD>(sq, t * 16 + fr, ks, fg), kf[ks], a, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<HD>(sdo, t * 16 + fr, ks, fg), vf[ks], dp, 0, 0, 0);
        }
        const int q0 = t * 16 + 4 * fg;
        const f32x4_t lq = *reinterpret_cast<const f32x4_t*>(s_lse + q0);
        const f32x4_t dl = *reinterpret_cast<const f32x4_t*>(s_delta + q0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = (keyok && q0 + e < S) ? expf(a[e] * scale - lq[e]) : 0.f;
          p2[u][e] = p;
          ds2[u][e] = p * (dp[e] - dl[e]) * scale;
        }
      }
      pf[kk] = pack_slots(p2[0], p2[1]);
      dsf[kk] = pack_slots(ds2[0], ds2[1]);
    }
    f32x4_t dv[DJ], dk[DJ];
#pragma unroll
    for (int j = 0; j < DJ; ++j) dv[j] = dk[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < NT / 2; ++kk)
#pragma unroll
      for (int j = 0; j < DJ; ++j) {
        dv[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols<HD>(sdo, kk * 32, j, lane), pf[kk], dv[j], 0, 0, 0);
        dk[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols<HD>(sq, kk * 32, j, lane), dsf[kk], dk[j], 0, 0, 0);
      }
    if (keyok) {
      uint16_t* dkp = dq_base + (size_t)key * rs + (size_t)H * HD + 4 * fg;
      uint16_t* dvp = dkp + (size_t)H * HD;
#pragma unroll
      for (int j = 0; j < DJ; ++j) {
        *reinterpret_cast<uint2*>(dkp + j * 16) = make_uint2(pack_bf2(dk[j][0], dk[j][1]), pack_bf2(dk[j][2], dk[j][3]));
        *reinterpret_cast<uint2*>(dvp + j * 16) = make_uint2(pack_bf2(dv[j][0], dv[j][1]), pack_bf2(dv[j][2], dv[j][3]));
      }
    }
  }
}

// Backward for the LONG sequences (197 tokens: NT = 14), split by ROLE: blockIdx.y = 0 computes dQ (pass A above),
// blockIdx.y = 1 computes dK / dV (pass B).  attn_bwd holds all four operands of a head in LDS -- 129 KB at 224 padded
// rows, ONE block per CU, so ViT-Tiny's 384 (image, head) pairs run as two rounds on 256 CUs with the second half
// empty (101 us per launch).  A role needs only TWO operands resident (dQ: K and V; dK / dV: Q and dO -- its own strips
// come straight from global memory as MFMA fragments, prefetched one strip ahead): 66 KB, two blocks per CU, twice
// the blocks.  Same MFMA work in total (both passes recompute S and dP anyway), same staging traffic.
template <int NT, int HD>
__global__ __launch_bounds__(AT_BWD_THREADS, 2) void attn_bwd_roles(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out,
                                                             const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                                                             int S, int H, float scale, uint16_t* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) uint8_t at_smem[];
  constexpr int SP = NT * 16;
  uint8_t* s0 = at_smem;                       // role 0: K      role 1: Q
  uint8_t* s1 = s0 + SP * AT_ROWB;             // role 0: V      role 1: dO
  float* s_lse = reinterpret_cast<float*>(s1 + SP * AT_ROWB);
  float* s_delta = s_lse + SP;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int role = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  constexpr int KS = HD / 32, DJ = HD / 16, CPR = HD / 8, NW = AT_BWD_THREADS / 64;
  const size_t rs = (size_t)3 * H * HD, os = (size_t)H * HD;
  const uint16_t* base = qkv + (size_t)b * S * rs + h * HD;
  const uint16_t* obase = out + (size_t)b * S * os + h * HD;
  const uint16_t* dobase = dout + (size_t)b * S * os + h * HD;
  const float* lse_bh = lse + ((size_t)b * H + h) * S;
  uint16_t* dq_base = dqkv + (size_t)b * S * rs + h * HD;

  if (role == 0) {
    // ---- dQ by query strips; K, V resident
    bf16x8_t qn[KS], dn[KS], on[KS];   // the NEXT strip's fragments (requested before the current strip's MFMAs)
    auto fetch = [&](int qs) {
      const int q = qs * 16 + fr;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        qn[ks] = frag_global(base, rs, q, S, ks, fg);
        dn[ks] = frag_global(dobase, os, q, S, ks, fg);
        on[ks] = frag_global(obase, os, q, S, ks, fg);
      }
    };
    fetch(wave);
    stage_rows<HD>(base + (size_t)H * HD, rs, S, SP, s0);
    stage_rows<HD>(base + (size_t)2 * H * HD, rs, S, SP, s1);
    __syncthreads();
    for (int qs = wave; qs < NT; qs += NW) {
      const int q = qs * 16 + fr;
      bf16x8_t qf[KS], df[KS];
      float dl = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = qn[ks];
        df[ks] = dn[ks];
        // delta[q] = sum_d dO[q][d] O[q][d]: this lane's 8 elements per k-slab, then the four lanes of the row
#pragma unroll
        for (int e = 0; e < 8; ++e) dl = fmaf((float)dn[ks][e], (float)on[ks][e], dl);
      }
      dl = wm_xor32_sum(wm_xor16_sum(dl));
      const float lq = q < S ? lse_bh[q] : 0.f;
      if (qs + NW < NT) fetch(qs + NW);
      bf16x8_t dsf[NT / 2];
#pragma unroll
      for (int kk = 0; kk < NT / 2; ++kk) {
        f32x4_t ds2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = 2 * kk + u;
          f32x4_t a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<HD>(s0, t * 16 + fr, ks, fg), qf[ks], a, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<HD>(s1, t * 16 + fr, ks, fg), df[ks], dp, 0, 0, 0);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int key = t * 16 + 4 * fg + e;
            const float p = key < S ? expf(a[e] * scale - lq) : 0.f;
            ds2[u][e] = p * (dp[e] - dl) * scale;
          }
        }
        dsf[kk] = pack_slots(ds2[0], ds2[1]);
        __builtin_amdgcn_sched_barrier(0);   // keep the next tile pair's fragment reads behind this one's MFMAs (registers)
      }
      f32x4_t dq[DJ];
#pragma unroll
      for (int j = 0; j < DJ; ++j) dq[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < NT / 2; ++kk)
#pragma unroll
        for (int j = 0; j < DJ; ++j)
          dq[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols<HD>(s0, kk * 32, j, lane), dsf[kk], dq[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (q < S) {
        uint16_t* dst = dq_base + (size_t)q * rs + 4 * fg;
#pragma unroll
        for (int j = 0; j < DJ; ++j)
          *reinterpret_cast<uint2*>(dst + j * 16) = make_uint2(pack_bf2(dq[j][0], dq[j][1]), pack_bf2(dq[j][2], dq[j][3]));
      }
    }
    return;
  }

  // ---- dK, dV by key strips; Q, dO (and lse, delta of every query) resident
  bf16x8_t kn[KS], vn[KS];
  auto fetch_kv = [&](int ksn) {
    const int key = ksn * 16 + fr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kn[ks] = frag_global(base + (size_t)H * HD, rs, key, S, ks, fg);
      vn[ks] = frag_global(base + (size_t)2 * H * HD, rs, key, S, ks, fg);
    }
  };
  fetch_kv(wave);
  stage_rows<HD>(base, rs, S, SP, s0);
  for (int i = threadIdx.x; i < SP * CPR; i += AT_BWD_THREADS) {
    const int r = i / CPR, c = i % CPR;
    uint4 dv = make_uint4(0, 0, 0, 0), ov = make_uint4(0, 0, 0, 0);
    if (r < S) {
      dv = *reinterpret_cast<const uint4*>(dobase + (size_t)r * os + c * 8);
      ov = *reinterpret_cast<const uint4*>(obase + (size_t)r * os + c * 8);
    }
    *reinterpret_cast<uint4*>(s1 + r * AT_ROWB + c * 16) = dv;
    const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w}, ow[4] = {ov.x, ov.y, ov.z, ov.w};
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      d = fmaf(bf2f((uint16_t)(dw[e] & 0xffff)), bf2f((uint16_t)(ow[e] & 0xffff)), d);
      d = fmaf(bf2f((uint16_t)(dw[e] >> 16)), bf2f((uint16_t)(ow[e] >> 16)), d);
    }
    d = group_sum<CPR>(d);
    if (c == 0) {
      s_delta[r] = d;
      s_lse[r] = r < S ? lse_bh[r] : 0.f;
    }
  }
  __syncthreads();
  for (int ksn = wave; ksn < NT; ksn += NW) {
    const int key = ksn * 16 + fr;
    const bool keyok = key < S;
    bf16x8_t kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = kn[ks];
      vf[ks] = vn[ks];
    }
    if (ksn + NW < NT) fetch_kv(ksn + NW);
    bf16x8_t pf[NT / 2], dsf[NT / 2];
#pragma unroll
    for (int kk = 0; kk < NT / 2; ++kk) {
      f32x4_t p2[2], ds2[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * kk + u;
        f32x4_t a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<HD>(s0, t * 16 + fr, ks, fg), kf[ks], a, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<HD>(s1, t * 16 + fr, ks, fg), vf[ks], dp, 0, 0, 0);
        }
        const int q0 = t * 16 + 4 * fg;
        const f32x4_t lq = *reinterpret_cast<const f32x4_t*>(s_lse + q0);
        const f32x4_t dl = *reinterpret_cast<const f32x4_t*>(s_delta + q0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = (keyok && q0 + e < S) ? expf(a[e] * scale - lq[e]) : 0.f;
          p2[u][e] = p;
          ds2[u][e] = p * (dp[e] - dl[e]) * scale;
        }
      }
      pf[kk] = pack_slots(p2[0], p2[1]);
      dsf[kk] = pack_slots(ds2[0], ds2[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    f32x4_t dv[DJ], dk[DJ];
#pragma unroll
    for (int j = 0; j < DJ; ++j) dv[j] = dk[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < NT / 2; ++kk)
#pragma unroll
      for (int j = 0; j < DJ; ++j) {
        dv[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols<HD>(s1, kk * 32, j, lane), pf[kk], dv[j], 0, 0, 0);
        dk[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols<HD>(s0, kk * 32, j, lane), dsf[kk], dk[j], 0, 0, 0);
      }
    if (keyok) {
      uint16_t* dkp = dq_base + (size_t)key * rs + (size_t)H * HD + 4 * fg;
      uint16_t* dvp = dkp + (size_t)H * HD;
#pragma unroll
      for (int j = 0; j < DJ; ++j) {
        *reinterpret_cast<uint2*>(dkp + j * 16) = make_uint2(pack_bf2(dk[j][0], dk[j][1]), pack_bf2(dk[j][2], dk[j][3]));
        *reinterpret_cast<uint2*>(dvp + j * 16) = make_uint2(pack_bf2(dv[j][0], dv[j][1]), pack_bf2(dv[j][2], dv[j][3]));
      }
    }
  }
}

template <typename K>
int at_set_lds(K kernel, int bytes) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     bytes);
  return e == hipSuccess ? WM_OK : (int)e;
}

template <int NT, int HD>
int launch_fwd(const void* qkv, int B, int S, int H, float scale, void* out, float* lse, hipStream_t st) {
  constexpr int lds = 2 * NT * 16 * AT_ROWB;
  static bool attr = false;
  if (!attr) {
    const int rc = at_set_lds(&attn_fwd<NT, HD>, lds);
    if (rc != WM_OK) return rc;
    attr = true;
  }
  static int qsplit = -1;  // WM_ATTN_FWD_SPLIT (experiment switch)
  if (qsplit < 0) {
    const char* e = getenv("WM_ATTN_FWD_SPLIT");
    qsplit = e ? atoi(e) : 0;
  }
  // default: two blocks per head beyond 8 strips while the (image, head) pairs alone do not fill the chip's 512 block
  // slots -- 128 x 197 x 3 heads: 26.6 -> 23.0 us; with 6 heads (768 pairs) the doubled K / V staging loses: 38.6 -> 43.7
  const int ys = qsplit > 0 ? qsplit : ((NT > 8 && (long long)B * H <= 512) ? 2 : 1);
  attn_fwd<NT, HD><<<dim3(B * H, ys), AT_FWD_THREADS, lds, st>>>(static_cast<const uint16_t*>(qkv), S, H, scale,
                                                             static_cast<uint16_t*>(out), lse);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

template <int NT, int HD>
int launch_bwd(const void* qkv, const void* out, const void* dout, const float* lse, int B, int S, int H, float scale,
               void* dqkv, hipStream_t st) {
  constexpr int lds = 4 * NT * 16 * AT_ROWB + 2 * NT * 16 * 4;
  static bool attr = false;
  if (!attr) {
    const int rc = at_set_lds(&attn_bwd<NT, HD>, lds);
    if (rc != WM_OK) return rc;
    attr = true;
  }
  attn_bwd<NT, HD><<<B * H, AT_BWD_THREADS, lds, st>>>(static_cast<const uint16_t*>(qkv), static_cast<const uint16_t*>(out),
                                               static_cast<const uint16_t*>(dout), lse, S, H, scale,
                                               static_cast<uint16_t*>(dqkv));
  WM_LAUNCH_CHECK();
  return WM_OK;
}

inline bool at_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int NT, int HD>
int launch_bwd_roles(const void* qkv, const void* out, const void* dout, const float* lse, int B, int S, int H, float scale,
                     void* dqkv, hipStream_t st) {
  constexpr int lds = 2 * NT * 16 * AT_ROWB + 2 * NT * 16 * 4;
  static bool attr = false;
  if (!attr) {
    const int rc = at_set_lds(&attn_bwd_roles<NT, HD>, lds);
    if (rc != WM_OK) return rc;
    attr = true;
  }
  attn_bwd_roles<NT, HD><<<dim3(B * H, 2), AT_BWD_THREADS, lds, st>>>(
      static_cast<const uint16_t*>(qkv), static_cast<const uint16_t*>(out), static_cast<const uint16_t*>(dout), lse, S, H,
      scale, static_cast<uint16_t*>(dqkv));
  WM_LAUNCH_CHECK();
  return WM_OK;
}

inline bool at_roles() {  // WM_ATTN_BWD_ROLES=0: the one-block-per-head kernel for every length (A/B switch)
  const char* e = getenv("WM_ATTN_BWD_ROLES");
  return !(e != nullptr && atoi(e) == 0);
}

template <int HD>
int dispatch_fwd(const void* qkv, int B, int S, int H, float scale, void* out, float* lse, hipStream_t st) {
  if (S <= 32) return launch_fwd<2, HD>(qkv, B, S, H, scale, out, lse, st);
  if (S <= 64) return launch_fwd<4, HD>(qkv, B, S, H, scale, out, lse, st);
  if (S <= 128) return launch_fwd<8, HD>(qkv, B, S, H, scale, out, lse, st);
  if (S <= 224) return launch_fwd<14, HD>(qkv, B, S, H, scale, out, lse, st);
  return launch_fwd<16, HD>(qkv, B, S, H, scale, out, lse, st);
}

template <int HD>
int dispatch_bwd(const void* qkv, const void* out, const void* dout, const float* lse, int B, int S, int H, float scale,
                 void* dqkv, hipStream_t st) {
  if (S <= 32) return launch_bwd<2, HD>(qkv, out, dout, lse, B, S, H, scale, dqkv, st);
  if (S <= 64) return launch_bwd<4, HD>(qkv, out, dout, lse, B, S, H, scale, dqkv, st);
  if (S <= 128) return launch_bwd<8, HD>(qkv, out, dout, lse, B, S, H, scale, dqkv, st);
  // > 128 tokens: the four operands of a head no longer leave room for two blocks per CU -> split by role
  if (at_roles()) {
    if (S <= 224) return launch_bwd_roles<14, HD>(qkv, out, dout, lse, B, S, H, scale, dqkv, st);
    return launch_bwd_roles<16, HD>(qkv, out, dout, lse, B, S, H, scale, dqkv, st);
  }
  if (S <= 224) return launch_bwd<14, HD>(qkv, out, dout, lse, B, S, H, scale, dqkv, st);
  return launch_bwd<16, HD>(qkv, out, dout, lse, B, S, H, scale, dqkv, st);
}

}  // namespace

extern "C" int wm_attention_fwd(const void* qkv, int B, int S, int H, int head_dim, float scale, void* out,
                                float* lse, void* stream) {
  WM_REQUIRE(qkv && out && lse, WM_EINVAL);
  WM_REQUIRE(B > 0 && S > 0 && H > 0 && (long long)B * H < (1ll << 31), WM_EINVAL);
  WM_REQUIRE(S <= 256 && (head_dim == 64 || head_dim == 32), WM_EUNSUPPORTED);
  WM_REQUIRE(at_al16(qkv) && at_al16(out), WM_EALIGN);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (head_dim == 64) return dispatch_fwd<64>(qkv, B, S, H, scale, out, lse, st);
  return dispatch_fwd<32>(qkv, B, S, H, scale, out, lse, st);
}

extern "C" int wm_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, int B, int S,
                                int H, int head_dim, float scale, void* dqkv, void* stream) {
  WM_REQUIRE(qkv && out && dout && lse && dqkv, WM_EINVAL);
  WM_REQUIRE(B > 0 && S > 0 && H > 0 && (long long)B * H < (1ll << 31), WM_EINVAL);
  WM_REQUIRE(S <= 256 && (head_dim == 64 || head_dim == 32), WM_EUNSUPPORTED);
  WM_REQUIRE(at_al16(qkv) && at_al16(out) && at_al16(dout) && at_al16(dqkv), WM_EALIGN);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (head_dim == 64) return dispatch_bwd<64>(qkv, out, dout, lse, B, S, H, scale, dqkv, st);
  return dispatch_bwd<32>(qkv, out, dout, lse, B, S, H, scale, dqkv, st);
}
