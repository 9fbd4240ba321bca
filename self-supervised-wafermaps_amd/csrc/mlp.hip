// Transformer MLP as ONE kernel: y = fc2(gelu(fc1(x))) (+ residual), the hidden activation never leaves the chip.
//
// Replaces, for forward passes that need no saved activations (the DINO teacher, kNN / embedding inference,
// validation), the pair of GEMM launches behind dino's Mlp / torchvision's MLPBlock
// (reference call sites scripts/WM811k_benchmark.py:548-551 student + teacher backbones, :578-588 the DINO step).
// Unfused, a d = 192 layer moves rows x (192 + 768 + 768 + 768 + 192) x 2 bytes through HBM for 2 x rows x 192 x 768 x 2
// FLOP: 154 FLOP per byte against a ridge of 312 -- its HBM roofline is below half of the MFMA peak.  Fused, only x and
// y move (and the two weight matrices, 590 KB, stay in L2): 1 536 FLOP per byte.
//
// Block = 128 token rows, 512 threads (8 waves: 4 row groups of 32 x 2 column halves).  LDS (128 KB, one block per CU,
// two waves per SIMD):
//   XS   [C/64 panels][128 rows][128 B]   the x tile, resident for the whole block            (48 KB at C = 192)
//   HS   [2 panels][128 rows][128 B]      one 128-wide chunk of the hidden activation, bf16    (32 KB)
//   ring 2 stages x 24 KB                 weight slices by global_load_lds: fc1 [128 hidden][64 c] or fc2 [C out][64 hidden]
// Per hidden chunk: C/64 k-steps of fc1 (A = weight fragment, B = x fragment: a lane owns 4 consecutive hidden units of
// one token), bias + GELU on the accumulators -> HS, then 2 k-steps of fc2 with HS as the token operand.  One
// `s_waitcnt vmcnt(0); s_barrier; issue(next); compute(this)` step per slice, as conv_igemm.  All rows are 128 B with
// the 16-byte chunk index XOR-swizzled by (row & 7) (conflict-free ds_read_b128 fragments).
// The arithmetic mirrors the two-launch path exactly (k order, MFMA shapes, bf16 rounding of the pre-activation and of
// the staged outputs before bias / residual), so the result is BIT-IDENTICAL to wm_linear_bias_gelu_fwd followed by
// wm_conv2d_fwd_bias_res (tests/test_gpu_vit.py).
#include "common.h"

namespace {

constexpr int ML_THREADS = 512;
constexpr int ML_ROWB = 128;                 // bytes per LDS row (64 bf16)
constexpr int ML_PANEL = 128 * ML_ROWB;      // one [128 rows][64 k] panel
constexpr int ML_HC = 128;                   // hidden units per chunk

struct MlpArgs {
  const uint16_t* x;    // [rows][C]
  const uint16_t* w1;   // [H][C]   fc1 weight, forward layout
  const float* b1;      // [H]
  const uint16_t* w2;   // [C][H]   fc2 weight, forward layout
  const float* b2;      // [C]
  const uint16_t* res;  // [rows][C] or NULL
  uint16_t* y;          // [rows][C]
  int rows, H;
};

__device__ __attribute__((aligned(256))) uint16_t mlp_zero_page[128];

template <int C>
__global__ __launch_bounds__(ML_THREADS) void mlp_fused_fwd(const MlpArgs a) {
  static_assert(C % 64 == 0 && C <= 192, "x tile + ring must fit 160 KB of LDS");
  constexpr int XP = C / 64;                       // x panels = fc1 k-steps per chunk
  constexpr int STEPS = XP + 2;                    // slices per hidden chunk
  constexpr int STAGE = C * ML_ROWB;               // fc2 slice [C rows][64 k] (>= the 16 KB fc1 slice)
  constexpr int OJ = C / 2 / 16;                   // 16-column output fragments per wave (half of C)
  extern __shared__ __attribute__((aligned(16))) uint8_t ml_smem[];
  uint8_t* XS = ml_smem;
  uint8_t* HS = XS + XP * ML_PANEL;
  uint8_t* RING = HS + 2 * ML_PANEL;
  const uint32_t ring_base = lds_addr(RING);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int m0 = blockIdx.x * 128;
  const int rl = tid >> 3;            // row inside a 64-row DMA instruction
  const int slot = tid & 7;           // physical 16-byte slot of the lane

  // ---- x tile: 2 * XP instructions per thread
  {
    const uint32_t xs_base = lds_addr(XS);
#pragma unroll
    for (int i = 0; i < 2 * XP; ++i) {
      const int pn = i >> 1, row = (i & 1) * 64 + rl;
      const int chunk = slot ^ (row & 7);
      const uint16_t* src = (m0 + row < a.rows) ? a.x + (size_t)(m0 + row) * C + pn * 64 + chunk * 8 : mlp_zero_page + chunk * 8;
      glds16_at(src, xs_base + pn * ML_PANEL + ((i & 1) * 64 + wave * 8) * ML_ROWB);
    }
  }
  const int nchunks = a.H / ML_HC;
  const int nsteps = nchunks * STEPS;
  // slice of step s -> ring stage s & 1
  auto issue = [&](int s) {
    const int hc = s / STEPS, k = s - hc * STEPS;
    const uint32_t stage = ring_base + (uint32_t)(s & 1) * STAGE;
    if (k < XP) {  // fc1: rows = hidden hc*128 + r, columns k*64 ..
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = i * 64 + rl;
        const uint16_t* src = a.w1 + (size_t)(hc * ML_HC + row) * C + k * 64 + (slot ^ (row & 7)) * 8;
        glds16_at(src, stage + (i * 64 + wave * 8) * ML_ROWB);
      }
    } else {       // fc2: rows = output channel, columns = hidden hc*128 + (k - XP)*64 ..
#pragma unroll
      for (int i = 0; i < C / 64; ++i) {
        const int row = i * 64 + rl;
        const uint16_t* src = a.w2 + (size_t)row * a.H + hc * ML_HC + (k - XP) * 64 + (slot ^ (row & 7)) * 8;
        glds16_at(src, stage + (i * 64 + wave * 8) * ML_ROWB);
      }
    }
  };

  f32x4_t acc1[4][2];      // fc1: 64 hidden x 32 tokens per wave
  f32x4_t acc2[OJ][2];     // fc2: C/2 outputs x 32 tokens per wave
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) acc1[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < OJ; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) acc2[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto frag = [&](const uint8_t* base, int row, int ks) -> bf16x8_t {
    const int c = ks * 4 + fg;
    return *reinterpret_cast<const bf16x8_t*>(base + row * ML_ROWB + ((c ^ (row & 7)) << 4));
  };

  issue(0);
  for (int s = 0; s < nsteps; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wm_barrier();
    if (s + 1 < nsteps) issue(s + 1);
    else wm_barrier();  // (a phase between the retiring wait and the reads when nothing is issued: see conv3x3_patch)
    const int hc = s / STEPS, k = s - hc * STEPS;
    const uint8_t* stage = RING + (s & 1) * STAGE;
    if (k < XP) {
      const uint8_t* xp = XS + k * ML_PANEL;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t xf[2], wf[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) xf[i] = frag(xp, wm * 32 + i * 16 + fr, ks);
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = frag(stage, wn * 64 + j * 16 + fr, ks);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc1[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc1[j][i], 0, 0, 0);
      }
      if (k == XP - 1) {
        // hidden chunk complete: pre = bf16(bf16(acc) + b1) (the two-launch path stages bf16 accumulators, then adds
        // the bias), h = bf16(gelu(pre)) -> HS; the barrier of the next step orders it before fc2's reads
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int hl = wn * 64 + j * 16 + fg * 4;  // hidden unit inside the chunk
          const float4 bb = *reinterpret_cast<const float4*>(a.b1 + hc * ML_HC + hl);
          const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int row = wm * 32 + i * 16 + fr;
            float h[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float pre = bf2f(f2bf(bf2f(f2bf(acc1[j][i][e])) + bv[e]));
              h[e] = wm_gelu(pre);
            }
            const int cl = (hl & 63) >> 3;  // 16-byte chunk inside the 64-wide panel
            *reinterpret_cast<uint2*>(HS + (hl >> 6) * ML_PANEL + row * ML_ROWB + ((cl ^ (row & 7)) << 4) + (fg & 1) * 8) =
                make_uint2(pack_bf2(h[0], h[1]), pack_bf2(h[2], h[3]));
            acc1[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          }
        }
      }
    } else {
      const uint8_t* hp = HS + (k - XP) * ML_PANEL;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t hf[2], wf[OJ];
#pragma unroll
        for (int i = 0; i < 2; ++i) hf[i] = frag(hp, wm * 32 + i * 16 + fr, ks);
#pragma unroll
        for (int j = 0; j < OJ; ++j) wf[j] = frag(stage, wn * (C / 2) + j * 16 + fr, ks);
#pragma unroll
        for (int j = 0; j < OJ; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc2[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], hf[i], acc2[j][i], 0, 0, 0);
      }
    }
  }
  __syncthreads();

  // ---- output: bf16 accumulators staged in LDS ([row][C] + 16 B pad), then bias (+ residual) on coalesced 16-byte rows
  constexpr int CS = C * 2 + 16;
  static_assert(128 * CS <= (XP + 2) * ML_PANEL, "output staging fits XS + HS");
#pragma unroll
  for (int j = 0; j < OJ; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wm * 32 + i * 16 + fr;
      const int ch = wn * (C / 2) + j * 16 + fg * 4;
      *reinterpret_cast<uint2*>(ml_smem + row * CS + ch * 2) =
          make_uint2(pack_bf2(acc2[j][i][0], acc2[j][i][1]), pack_bf2(acc2[j][i][2], acc2[j][i][3]));
    }
  __syncthreads();
  constexpr int CPR = C / 8;
  for (int p = tid; p < 128 * CPR; p += ML_THREADS) {
    const int row = p / CPR, ch = p - row * CPR;
    if (m0 + row >= a.rows) continue;
    const uint4 v = *reinterpret_cast<const uint4*>(ml_smem + row * CS + ch * 16);
    const float4 b0 = *reinterpret_cast<const float4*>(a.b2 + ch * 8), b1v = *reinterpret_cast<const float4*>(a.b2 + ch * 8 + 4);
    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1v.x, b1v.y, b1v.z, b1v.w};
    const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      o[q] = pack_bf2(bf2f((uint16_t)(vv[q] & 0xffff)) + bb[2 * q], bf2f((uint16_t)(vv[q] >> 16)) + bb[2 * q + 1]);
    const size_t off = (size_t)(m0 + row) * C + ch * 8;
    if (a.res != nullptr) {
      const uint4 r4 = *reinterpret_cast<const uint4*>(a.res + off);
      const uint32_t rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
      for (int q = 0; q < 4; ++q)
        o[q] = pack_bf2(bf2f((uint16_t)(o[q] & 0xffff)) + bf2f((uint16_t)(rr[q] & 0xffff)),
                        bf2f((uint16_t)(o[q] >> 16)) + bf2f((uint16_t)(rr[q] >> 16)));
    }
    *reinterpret_cast<uint4*>(a.y + off) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

}  // namespace

extern "C" int wm_mlp_fused_fwd_ok(int rows, int C, int H) { return rows > 0 && C == 192 && H > 0 && H % 128 == 0 ? 1 : 0; }

extern "C" int wm_mlp_fused_fwd(const void* x, const void* w1_krsc, const float* b1, const void* w2_krsc, const float* b2,
                                const void* residual, void* y, int rows, int C, int H, void* stream) {
  WM_REQUIRE(x && w1_krsc && b1 && w2_krsc && b2 && y, WM_EINVAL);
  WM_REQUIRE(wm_mlp_fused_fwd_ok(rows, C, H), WM_EUNSUPPORTED);
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  WM_REQUIRE(al(x) && al(w1_krsc) && al(b1) && al(w2_krsc) && al(b2) && al(y) && (residual == nullptr || al(residual)), WM_EALIGN);
  MlpArgs a{static_cast<const uint16_t*>(x), static_cast<const uint16_t*>(w1_krsc), b1, static_cast<const uint16_t*>(w2_krsc),
            b2, static_cast<const uint16_t*>(residual), static_cast<uint16_t*>(y), rows, H};
  constexpr int lds = (192 / 64 + 2) * ML_PANEL + 2 * 192 * ML_ROWB;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_fwd<192>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  mlp_fused_fwd<192><<<wm_cdiv(rows, 128), ML_THREADS, lds, static_cast<hipStream_t>(stream)>>>(a);
  WM_LAUNCH_CHECK();
  return WM_OK;
}
