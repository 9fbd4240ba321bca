// Linear layers with a SHORT reduction (in_features = 192: ViT-Tiny's qkv / proj / fc1 forward, proj / fc2 input
// gradients): y = act(x W^T + bias) (+ residual), x [rows][192], W [N][192].
//
// Why not conv_igemm: with three 64-wide k-steps a 128 x 64 output tile is all prologue and epilogue -- every block
// waits three times in a row for a DMA round trip (its x tile comes from HBM / Infinity Cache: ~2 us each under load),
// and the x tile is fetched once per column tile (9 x for qkv, 12 x for fc1).  Measured (tools/bench_linear.py, 25 216
// rows): qkv 25.3 us = 221 TFLOP/s, fc1 + GELU 38.6 us = 193 TFLOP/s; the launch's HBM floor is 8 / 15 us.
//
// Here a block owns 128 token rows for ALL of its output columns:
//   * the x rows live in REGISTERS as MFMA fragments for the whole block (a lane's 16-byte pieces of two token rows x
//     six k-steps: 48 VGPRs), loaded once, straight from global memory -- one memory round trip per block;
//   * weight tiles [64 output columns][192] (24 KB, three swizzled [64][128 B] panels) stream through a three-stage
//     ring by global_load_lds, two tiles ahead of the one being multiplied; they are L2 hits (the matrix is 72 - 295 KB);
//   * per tile: 24 MFMAs per wave (8 waves: 4 row groups x 2 column halves), accumulators -> bf16 tile staged in the
//     weight stage just consumed -> coalesced 16-byte rows with bias / GELU, exactly the arithmetic (k order, MFMA
//     shape, bf16 roundings) of conv_igemm's EPI epilogue: results are bit-identical to that path;
//   * NO ordinary global load inside the tile loop (the bias slice sits in LDS): hipcc cannot see the DMA
//     instructions, so its wait for any VGPR load would drain the whole queue, DMA included -- the first build of this
//     kernel (bias and residual fetched per tile) was no faster than conv_igemm for exactly that reason.  Waits are
//     counted by hand: stores of the two previous tiles and the next tile's DMA stay in flight.
// 76 KB of LDS and <= 128 VGPRs: two blocks (16 waves) per CU.  Launches with a residual or a saved pre-activation
// operand (proj forward, fc2 input gradient) stay on conv_igemm: their tile would have to come through LDS too, which
// leaves room for one block per CU only.
#include "common.h"
#include "panel.h"
#include "ln_regs.h"
#include <stdlib.h>

namespace {

constexpr int PN_THREADS = 512;
constexpr int PN_ROWB = 128;                  // bytes per LDS row of a weight panel (64 bf16)
constexpr int PN_PANEL = 64 * PN_ROWB;        // [64 output columns][64 k]
constexpr int PN_CS = 64 * 2 + 16;            // staged output row: 64 bf16 + 16 B pad
constexpr int PN_STAGES = 3;
constexpr int PN_MAX_N = 1024;                // bias slice of a block kept in LDS

// s_waitcnt vmcnt(n) for the counts the loop needs (the immediate is an instruction field)
__device__ __forceinline__ void pn_wait(int n) {
  switch (n) {
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// LN: LayerNorm of the token rows on the register fragments first (its own instantiation: the normalisation's
// temporaries would otherwise raise the plain kernel's register count from 94 to 252 and halve its blocks per CU)
template <int C, bool LN>
__global__ __launch_bounds__(PN_THREADS, 2) void linear_panel(const WmPanelArgs a) {
  static_assert(C % 64 == 0 && C <= 192, "x fragments must fit the register budget of two blocks per CU");
  constexpr int XP = C / 64;                  // 64-wide k panels = DMA instructions per thread and tile
  constexpr int KS = C / 32;                  // MFMA k-steps
  constexpr int STAGE = XP * PN_PANEL;
  static_assert(128 * PN_CS <= STAGE, "the staged output tile reuses the weight stage it was computed from");
  extern __shared__ __attribute__((aligned(16))) uint8_t pn_smem[];
  uint8_t* RING = pn_smem;
  float* BIAS = reinterpret_cast<float*>(pn_smem + PN_STAGES * STAGE);
  const uint32_t ring_base = lds_addr(RING);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int m0 = blockIdx.x * 128;
  const int t0 = blockIdx.y * a.tiles_per_block;
  int nt = a.N / 64 - t0;
  nt = nt < a.tiles_per_block ? nt : a.tiles_per_block;
  // Counted waits need every thread to have issued the same number of stores per tile: full blocks only.
  const bool full = m0 + 128 <= a.rows && !(a.debug & 1);
  const int nst = a.act == 1 ? 4 : 2;         // 16-byte stores per thread and tile

  const int rl = tid >> 3, slot = tid & 7;    // weight row inside a panel / physical 16-byte slot
  auto issue = [&](int t) {
    const uint32_t stage = ring_base + (uint32_t)(t % PN_STAGES) * STAGE;
    const uint16_t* src = a.w + (size_t)((t0 + t) * 64 + rl) * C + (slot ^ (rl & 7)) * 8;
#pragma unroll
    for (int p = 0; p < XP; ++p) glds16_at(src + p * 64, stage + p * PN_PANEL + wave * 8 * PN_ROWB);
  };
  if (nt > 0) issue(0);
  if (nt > 1) issue(1);

  // ---- the block's token rows as MFMA B-operand fragments (rows past the end repeat the last row; never stored)
  bf16x8_t xf[2][KS];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row = m0 + wm * 32 + i * 16 + fr;
    row = row < a.rows ? row : a.rows - 1;
    if (a.debug & 8) row = fr;
    const uint16_t* xr = a.x + (size_t)row * C + fg * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[i][ks] = *reinterpret_cast<const bf16x8_t*>(xr + ks * 32);
  }
  if constexpr (LN) wm_ln_fragments<KS>(xf, a.ln_gamma, a.ln_beta, a.ln_eps, fg);
  // the block's bias slice -> LDS (the loop below must not consume ordinary loads: with DMA instructions in flight
  // every such use drains the whole queue)
  if (a.bias != nullptr) {
    for (int i = tid; i < nt * 64; i += PN_THREADS) BIAS[i] = a.bias[t0 * 64 + i];
  }

  const int erow = tid >> 3, ech = tid & 7;   // epilogue: rows erow, erow + 64; 16-byte chunk ech of the 64 columns
  for (int t = 0; t < nt; ++t) {
    // This wave's pieces of tile t have landed.  Issued after them and allowed to stay in flight: the stores of tiles
    // t - 2 and t - 1 and the DMA of tile t + 1 (VMEM operations retire in issue order).
    {
      int later = 0;
      if (t >= 2) later += nst;
      if (t >= 1) later += nst;
      if (t >= 1 && t + 1 < nt) later += XP;
      pn_wait(full && t >= 1 ? later : 0);    // (t = 0: the x rows are needed right away as well)
    }
    wm_barrier();   // everyone's pieces of tile t; the staged tile t - 1 has been read by all
    if (t + 2 < nt) issue(t + 2);             // into the stage of tile t - 1
    uint8_t* stage = RING + (t % PN_STAGES) * STAGE;
    f32x4_t acc[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (!(a.debug & 2))
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8_t wf[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = wn * 32 + j * 16 + fr;
        const int c = (ks & 1) * 4 + fg;
        wf[j] = *reinterpret_cast<const bf16x8_t*>(stage + (ks >> 1) * PN_PANEL + row * PN_ROWB + ((c ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i][ks], acc[j][i], 0, 0, 0);
    }
    wm_barrier();   // every wave is done with the weights of this stage: it becomes the staging tile
    if (!(a.debug & 4))
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wm * 32 + i * 16 + fr;
        const int col = wn * 32 + j * 16 + fg * 4;
        *reinterpret_cast<uint2*>(stage + row * PN_CS + col * 2) =
            make_uint2(pack_bf2(acc[j][i][0], acc[j][i][1]), pack_bf2(acc[j][i][2], acc[j][i][3]));
      }
    wm_barrier();
    float bb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bb[e] = 0.f;
    if (a.bias != nullptr) {
      const float4 b0 = *reinterpret_cast<const float4*>(BIAS + t * 64 + ech * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(BIAS + t * 64 + ech * 8 + 4);
      bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
    }
    const int n0 = (t0 + t) * 64;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int rloc = erow + it * 64;
      if (m0 + rloc >= a.rows || (a.debug & 1)) continue;
      const uint4 v4 = *reinterpret_cast<const uint4*>(stage + rloc * PN_CS + ech * 16);
      uint32_t vv[4] = {v4.x, v4.y, v4.z, v4.w};
      const size_t off = (size_t)(m0 + rloc) * a.N + n0 + ech * 8;
      if (a.bias != nullptr) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          vv[q] = pack_bf2(bf2f((uint16_t)(vv[q] & 0xffff)) + bb[2 * q], bf2f((uint16_t)(vv[q] >> 16)) + bb[2 * q + 1]);
      }
      if (a.act == 1) {          // biased pre-activation kept for the backward pass; gelu(pre) is the output
        *reinterpret_cast<uint4*>(a.pre_out + off) = make_uint4(vv[0], vv[1], vv[2], vv[3]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          vv[q] = pack_bf2(wm_gelu(bf2f((uint16_t)(vv[q] & 0xffff))), wm_gelu(bf2f((uint16_t)(vv[q] >> 16))));
      }
      *reinterpret_cast<uint4*>(a.y + off) = make_uint4(vv[0], vv[1], vv[2], vv[3]);
    }
  }
}

bool panel_disabled() {  // WM_LINEAR_PANEL=0: the conv_igemm path (read per call: the tests compare the two)
  const char* e = getenv("WM_LINEAR_PANEL");
  return e != nullptr && atoi(e) == 0;
}

}  // namespace

bool wm_panel_ok(long long rows, int C, int N, bool has_aux) {
  // (a residual or pre-activation operand would have to come through LDS as well: those launches stay on conv_igemm)
  // N >= 384: with three column tiles (proj) the launch is too short for the resident rows to pay (9.4 vs 8.4 us)
  return !panel_disabled() && !has_aux && C == 192 && N >= 384 && N % 64 == 0 && rows > 0 && rows < (1ll << 31) - 128;
}

int wm_panel_launch(WmPanelArgs a, hipStream_t st) {
  if (!wm_panel_ok(a.rows, 192, a.N, a.res != nullptr || a.pre_in != nullptr) || a.act == 2) return WM_EUNSUPPORTED;
  constexpr int lds = PN_STAGES * (192 / 64) * PN_PANEL + PN_MAX_N * 4;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_panel<192, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_panel<192, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  // column tiles per block: as many blocks as fit twice into the chip's 512 block slots (two per CU); measured
  // (tools/probes/panel_probe.py, 39 424 rows): fc1 + GELU 65 us with all 12 tiles in one block (308 blocks), 49 us with
  // 4 tiles per block (924 blocks); conv_igemm 57 us.  Never more than PN_MAX_N columns per block.
  const int rowtiles = (a.rows + 127) / 128, tiles = a.N / 64;
  int split = 0;
  for (int s = 1; s <= tiles; ++s) {
    if (tiles % s != 0 || tiles / s * 64 > PN_MAX_N) continue;
    if (split == 0 || (long long)rowtiles * s <= 1024) split = s;
  }
  a.tiles_per_block = tiles / split;
  {
    const char* e = getenv("WM_PANEL_DEBUG");
    a.debug = e ? atoi(e) : 0;
    const char* f = getenv("WM_PANEL_SPLIT");
    if (f && atoi(f) > 0 && tiles % atoi(f) == 0) { split = atoi(f); a.tiles_per_block = tiles / split; }
  }
  if (a.ln_gamma != nullptr) linear_panel<192, true><<<dim3(rowtiles, split), PN_THREADS, lds, st>>>(a);
  else linear_panel<192, false><<<dim3(rowtiles, split), PN_THREADS, lds, st>>>(a);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

// y = LayerNorm(x; gamma, beta, eps) W^T + bias in ONE launch, for forward passes that keep nothing for a backward pass
// (DINO teacher, inference): the normalised rows exist only as register fragments.  x [rows][192] bf16, w_krsc [N][192]
// bf16, bias [N] f32 or NULL, y [rows][N] bf16.
extern "C" int wm_ln_linear_fwd_ok(int rows, int C, int N) { return wm_panel_ok(rows, C, N, false) ? 1 : 0; }

extern "C" int wm_ln_linear_fwd(const void* x, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* w_krsc,
                                const float* bias, void* y, int rows, int C, int N, void* stream) {
  WM_REQUIRE(x && ln_gamma && ln_beta && w_krsc && y, WM_EINVAL);
  WM_REQUIRE(wm_panel_ok(rows, C, N, false), WM_EUNSUPPORTED);
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  WM_REQUIRE(al(x) && al(ln_gamma) && al(ln_beta) && al(w_krsc) && al(y) && (bias == nullptr || al(bias)), WM_EALIGN);
  WmPanelArgs pa{static_cast<const uint16_t*>(x), static_cast<const uint16_t*>(w_krsc), bias, nullptr, nullptr, nullptr,
                 static_cast<uint16_t*>(y), rows, N, 0, 0, ln_gamma, ln_beta, ln_eps, 0};
  return wm_panel_launch(pa, static_cast<hipStream_t>(stream));
}
