// Transformer MLP as ONE kernel: y = fc2(gelu(fc1(x))) (+ residual), the hidden activation never leaves the chip.
//
// Replaces, for forward passes that need no saved activations (the DINO teacher, kNN / embedding inference,
// validation), the pair of GEMM launches behind dino's Mlp / torchvision's MLPBlock
// (reference call sites scripts/WM811k_benchmark.py:548-551 student + teacher backbones, :578-588 the DINO step).
// Unfused, a d = 192 layer moves rows x (192 + 768 + 768 + 768 + 192) x 2 bytes through HBM for 2 x rows x 192 x 768 x 2
// FLOP: 154 FLOP per byte against a ridge of 312 -- its HBM roofline is below half of the MFMA peak.  Fused, only x and
// y move (and the two weight matrices, 590 KB, stay in L2): 1 536 FLOP per byte.
//
// Block = 128 token rows, 512 threads (8 waves: 4 row groups of 32 x 2 column halves).  One block per CU:
//   x rows   REGISTERS: a lane's 16-byte pieces of its two token rows x six k-steps (48 VGPRs), loaded once from global
//            memory as MFMA B-operand fragments (round 3; they used to occupy 48 KB of LDS)
//   HS       [2 panels][128 rows][128 B]   one 128-wide chunk of the hidden activation, bf16    (32 KB)
//   ring     5 stages x 24 KB              weight slices by global_load_lds: fc1 [128 hidden][64 c] or fc2 [C out][64 hidden]
// Per hidden chunk: C/64 k-steps of fc1 (A = weight fragment, B = x fragment: a lane owns 4 consecutive hidden units of
// one token), bias + GELU on the accumulators -> HS, then 2 k-steps of fc2 with HS as the token operand.  FOUR weight
// slices are in flight ahead of the one being multiplied, retired by counted waits (2 DMA instructions per thread for an
// fc1 slice, 3 for an fc2 slice).  All LDS rows are 128 B with the 16-byte chunk index XOR-swizzled by (row & 7)
// (conflict-free ds_read_b128 fragments).
// What bounds it (tools/probes/mlp_probe.py): NOT the weight stream -- going from the two-stage ring of round 2 to this
// one moved a launch over 25 216 rows from 49.6 to 47.9 us, and three blocks alone on the chip still take 41 us.  A block
// evaluates 128 x 768 exact GELUs (exp, rcp and a degree-5 polynomial: ~30 VALU slots each) with two waves per SIMD:
// ~19 us of VALU time per block, barrier-synchronised with the MFMA phases of the same waves, plus 30 barriers.  The
// two-launch path pays the same VALU work spread over all 256 CUs (59.8 us for both launches at these rows); at 39 424
// rows (308 blocks, two rounds of one block per CU) the fused kernel loses (91.9 vs 82.7 us), so only the teacher /
// inference passes (<= 256 row tiles) use it.
// The arithmetic mirrors the two-launch path exactly (k order, MFMA shapes, bf16 rounding of the pre-activation and of
// the staged outputs before bias / residual), so the result is BIT-IDENTICAL to wm_linear_bias_gelu_fwd followed by
// wm_conv2d_fwd_bias_res (tests/test_gpu_vit.py).
#include "common.h"
#include "ln_regs.h"

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
  const float* ln_gamma;  // optional LayerNorm of the token rows first ([C]); the residual stays the un-normalised input
  const float* ln_beta;
  float ln_eps;
};


__device__ __forceinline__ void ml_wait(int n) {  // s_waitcnt vmcnt(n): the immediate is an instruction field
  switch (n) {
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

template <int C>
__global__ __launch_bounds__(ML_THREADS) void mlp_fused_fwd(const MlpArgs a) {
  static_assert(C % 64 == 0 && C <= 192, "x fragments in registers; the fc2 slice must fit a ring stage");
  constexpr int XP = C / 64;                       // fc1 k-steps (64 wide) per chunk
  constexpr int KS = C / 32;                       // MFMA k-steps over C
  constexpr int STEPS = XP + 2;                    // slices per hidden chunk
  constexpr int STAGE = C * ML_ROWB;               // fc2 slice [C rows][64 k] (>= the 16 KB fc1 slice)
  constexpr int NST = 5;                           // ring stages: four slices in flight ahead of the current one
  constexpr int OJ = C / 2 / 16;                   // 16-column output fragments per wave (half of C)
  extern __shared__ __attribute__((aligned(16))) uint8_t ml_smem[];
  uint8_t* HS = ml_smem;
  uint8_t* RING = HS + 2 * ML_PANEL;
  const uint32_t ring_base = lds_addr(RING);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int m0 = blockIdx.x * 128;
  const int rl = tid >> 3;            // row inside a 64-row DMA instruction
  const int slot = tid & 7;           // physical 16-byte slot of the lane

  const int nchunks = a.H / ML_HC;
  const int nsteps = nchunks * STEPS;
  // slice of step s -> ring stage s % NST; DMA instructions per thread: 2 (fc1) or C / 64 (fc2)
  auto n_instr = [&](int s) { return (s % STEPS) < XP ? 2 : C / 64; };
  auto issue = [&](int s) {
    const int hc = s / STEPS, k = s - hc * STEPS;
    const uint32_t stage = ring_base + (uint32_t)(s % NST) * STAGE;
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
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nsteps) issue(s);

  // ---- the block's token rows as MFMA B-operand fragments (rows past the end repeat the last row; never stored)
  bf16x8_t xr[2][KS];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row = m0 + wm * 32 + i * 16 + fr;
    row = row < a.rows ? row : a.rows - 1;
    const uint16_t* xp = a.x + (size_t)row * C + fg * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xr[i][ks] = *reinterpret_cast<const bf16x8_t*>(xp + ks * 32);
  }

  if (a.ln_gamma != nullptr) wm_ln_fragments<KS>(xr, a.ln_gamma, a.ln_beta, a.ln_eps, fg);

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

  for (int hc = 0; hc < nchunks; ++hc) {
#pragma unroll
    for (int k = 0; k < STEPS; ++k) {
      const int s = hc * STEPS + k;
      // slice s has landed; the slices s + 1 .. s + 3 (issued before it was needed) may stay in flight
      {
        int later = 0;
#pragma unroll
        for (int d = 1; d < NST - 1; ++d)
          if (s + d < nsteps) later += n_instr(s + d);
        ml_wait(s == 0 ? 0 : later);   // (step 0: the x rows are needed right away as well)
      }
      wm_barrier();   // everyone's pieces of slice s; the stage of slice s - 1 has been read by all
      if (s + NST - 1 < nsteps) issue(s + NST - 1);
      const uint8_t* stage = RING + (s % NST) * STAGE;
      if (k < XP) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          bf16x8_t wf[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) wf[j] = frag(stage, wn * 64 + j * 16 + fr, ks);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
              acc1[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xr[i][k * 2 + ks], acc1[j][i], 0, 0, 0);
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
  }
  __syncthreads();

  // ---- output: bf16 accumulators staged in LDS ([row][C] + 16 B pad), then bias (+ residual) on coalesced 16-byte rows
  constexpr int CS = C * 2 + 16;
  static_assert(128 * CS <= 2 * ML_PANEL + NST * STAGE, "output staging fits HS + ring");
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

// (rows <= 256 row tiles: one block per CU, so beyond one round of the chip the two-launch path is faster)
extern "C" int wm_mlp_fused_fwd_ok(int rows, int C, int H) {
  return rows > 0 && rows <= 256 * 128 && C == 192 && H > 0 && H % 128 == 0 ? 1 : 0;
}

static int mlp_launch(const void* x, const void* w1_krsc, const float* b1, const void* w2_krsc, const float* b2,
                      const void* residual, void* y, int rows, int C, int H, const float* ln_gamma, const float* ln_beta,
                      float ln_eps, void* stream);

extern "C" int wm_mlp_fused_fwd(const void* x, const void* w1_krsc, const float* b1, const void* w2_krsc, const float* b2,
                                const void* residual, void* y, int rows, int C, int H, void* stream) {
  return mlp_launch(x, w1_krsc, b1, w2_krsc, b2, residual, y, rows, C, H, nullptr, nullptr, 0.f, stream);
}

// y = fc2(gelu(fc1(LayerNorm(x)) + b1)) + b2 (+ residual): the second half of a pre-norm transformer block in one launch
// (residual = x gives x + mlp(norm2(x))); the normalised rows exist only as register fragments.
extern "C" int wm_ln_mlp_fused_fwd(const void* x, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                   const void* w1_krsc, const float* b1, const void* w2_krsc, const float* b2,
                                   const void* residual, void* y, int rows, int C, int H, void* stream) {
  WM_REQUIRE(ln_gamma && ln_beta, WM_EINVAL);
  WM_REQUIRE((reinterpret_cast<uintptr_t>(ln_gamma) & 15) == 0 && (reinterpret_cast<uintptr_t>(ln_beta) & 15) == 0, WM_EALIGN);
  return mlp_launch(x, w1_krsc, b1, w2_krsc, b2, residual, y, rows, C, H, ln_gamma, ln_beta, ln_eps, stream);
}

static int mlp_launch(const void* x, const void* w1_krsc, const float* b1, const void* w2_krsc, const float* b2,
                      const void* residual, void* y, int rows, int C, int H, const float* ln_gamma, const float* ln_beta,
                      float ln_eps, void* stream) {
  WM_REQUIRE(x && w1_krsc && b1 && w2_krsc && b2 && y, WM_EINVAL);
  WM_REQUIRE(wm_mlp_fused_fwd_ok(rows, C, H), WM_EUNSUPPORTED);
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  WM_REQUIRE(al(x) && al(w1_krsc) && al(b1) && al(w2_krsc) && al(b2) && al(y) && (residual == nullptr || al(residual)), WM_EALIGN);
  MlpArgs a{static_cast<const uint16_t*>(x), static_cast<const uint16_t*>(w1_krsc), b1, static_cast<const uint16_t*>(w2_krsc),
            b2, static_cast<const uint16_t*>(residual), static_cast<uint16_t*>(y), rows, H, ln_gamma, ln_beta, ln_eps};
  constexpr int lds = 2 * ML_PANEL + 5 * 192 * ML_ROWB;
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
