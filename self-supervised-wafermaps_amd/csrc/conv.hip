// Convolution as implicit GEMM on MFMA (bf16 in, f32 accumulate), NHWC activations.
//
// Replaces the cuDNN convolutions behind timm.create_model("resnet18") in the reference
// (scripts/WM811k_benchmark.py:231, forward :236-240, backward via Lightning) and the nn.Linear
// layers of lightly's SimCLRProjectionHead (:233), which are 1x1 convolutions on a 1x1 image.
//
//   forward : Y[m][k]      = sum_{r,s,c} X[n, p*st-pad+r, q*st-pad+s, c] * Wk[k][r][s][c]
//   dgrad   : dX[m'][c]    = sum_{r,s,k} dY[n, (h+pad-r)/st, (w+pad-s)/st, k] * Wc[c][r][s][k]
//   wgrad   : dW[k][r][s][c] += sum_m dY[m][k] * X[n, p*st-pad+r, q*st-pad+s, c]
//
// forward and dgrad share one kernel: GEMM rows are destination pixels (tile 128), GEMM columns
// destination channels (tile 64/128), the reduction runs over (tap, source channel) in 64-element
// tiles, each a 128-byte contiguous run of the source tensor (NHWC) or zeros.  Tiles are staged
// global -> registers -> LDS (double buffered: the next tile's loads are in flight under the MFMAs),
// LDS rows are 128 B with the 16-byte chunk index XOR-swizzled by (row & 7): fragment reads with
// ds_read_b128 are bank-conflict free.  MFMA is v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A
// operand, so an accumulator lane owns 4 consecutive output channels of one pixel and the epilogue
// packs them into 8-byte LDS writes, then the tile leaves as coalesced 16-byte row stores.
//
// The stem (7x7 stride 2 on 3 channels) is run as a 4x4 stride-1 convolution over the 2x2
// space-to-depth image with 16 channels (12 used): a reduction tile is then one kernel row = four
// taps x 16 channels = 128 contiguous bytes, with per-tap bounds (CPT = 2 chunks per tap).
//
// wgrad reduces over pixels, so both operands need the pixel index along k: they are staged in
// their natural [pixel][channel] layout and read with ds_read_b64_tr_b16 (hardware transpose);
// the k-slot <-> pixel assignment is permuted identically for both operands so that the two
// 16-lane groups of each 32-lane half read 8 consecutive rows (conflict free with rows padded by
// 32 B).  Partial sums over pixel ranges (split-K) are combined with f32 atomics into an
// [K][R][S][C] f32 buffer the caller zeroes.
//
// Roofline: MFMA (dense bf16), 2*M*K*R*S*C FLOP per launch.
#include "common.h"
#include "panel.h"
#include <stdlib.h>

namespace {

// Compile-time ablation for profiling builds (-DWM_CONV_ABLATE=1: no tile fetches after the first,
// =2: no MFMA phase).  A run-time switch costs ~30 %: it makes hipcc shuttle the accumulators
// between AGPRs and VGPRs around the branch on every k-step.
#ifndef WM_CONV_ABLATE
#define WM_CONV_ABLATE 0
#endif

constexpr int CV_THREADS = 256;
constexpr int CV_ROW = 128;  // bytes per LDS row of an operand tile (64 bf16)

struct ConvArgs {
  const uint16_t* src;  // [N][SH][SW][SC]
  const uint16_t* wt;   // [DC][R][S][SC]
  uint16_t* dst;        // [N][DH][DW][DC]
  int N, SH, SW, SC, DH, DW, DC, R, S, stride, pad, M, nkt;
  // optional fused BatchNorm statistics (forward only): per-channel (sum, sum of squares) of the
  // bf16-rounded outputs, added into bucket (tile % stat_nb) of the tile's row group
  float* stat;          // [G][stat_nb][2][DC] f32: tile t of group g STORES its sums into slot (g, t) -- no atomics; or NULL
  int stat_nb, stat_rpg;  // stat_nb = slots (tiles) per group
  // optional BatchNorm-BACKWARD epilogue (dgrad only, template flag BNB): this convolution's input was
  // relu(BN(bn_y) (+ shortcut)), so the gradient this kernel produces is the one entering that ReLU.  The epilogue
  // applies the ReLU mask (bn_x > 0 when the convolution's forward input bn_x is given, else recomputed from bn_y as
  // bf16(bn_y * gamma * invstd + beta - mean * gamma * invstd) > 0), stores the MASKED gradient, and accumulates the
  // BatchNorm backward sums (sum g, sum g * bn_y) per channel into `stat`: the separate reduction pass over
  // (bn_y, gradient, mask) and the mask / dz handling of the BatchNorm backward apply pass disappear.
  const uint16_t* bn_y;
  const uint16_t* bn_x;
  const uint8_t* bn_mask;  // alternative to bn_x: [pixels][DC / 8] bytes, bit e = (bn_x channel 8 chunk + e > 0), written by
                           // the BatchNorm forward (wm_bn_train_fwd*): 1/16 of the bytes of bn_x
  const float* bn_mean;    // [G][DC]
  const float* bn_invstd;  // [G][DC]
  const float* bn_gamma;   // [DC]
  const float* bn_beta;    // [DC]
  // optional tensor added to the result in the epilogue (dgrad: the gradient that reached the same
  // input through a second path, e.g. the identity shortcut of a residual block), same shape as dst
  const uint16_t* res;
  // optional per-output-channel bias added in the epilogue (forward only: Linear layers)
  const float* bias;
  // optional activation in the EPI epilogue (ViT MLP): act 1 = forward GELU: the biased pre-activation goes to
  // pre_out (saved for the backward pass) and gelu(pre) to dst; act 2 = dgrad: result *= gelu'(pre_in) (the
  // gradient through the activation that follows the layer whose input gradient this is)
  const uint16_t* pre_in;
  uint16_t* pre_out;
  int act;
  // multiply-high forms of the divisions by DH * DW and by DW in the row decode (common.h)
  WmDiv d_dhw, d_dw, d_h2w2, d_w2;  // (the last two: DH/2 * DW/2 and DW/2, the parity-class order of MODE 2)
  int xcd;  // 1: XCD-aware tile order (WM_XCD_SWIZZLE, default on)
};

__device__ __forceinline__ float cv_gelu(float v) { return wm_gelu(v); }
__device__ __forceinline__ float cv_gelu_grad(float v) { return wm_gelu_grad(v); }

// 128 zero bytes: the global_load_lds source of padded / out-of-range taps
__device__ __attribute__((aligned(256))) uint16_t conv_zero_page[128];


// Per-channel sums over the block of two per-thread running sums: a thread holds s1[8], s2[8] for the 8 channels of chunk
// column tid % (BNC / 8), accumulated over its rows of the tile.  Lanes of a wave that share the column are summed on the
// DPP network / lane swaps (row_ror:8, then the 16- and 32-lane swaps), the four waves through 2 x 4 x BNC floats of
// LDS (`red`, outside the staged tile), and threads 0 .. 2 BNC - 1 store the block's sums into the tile's slot -- plain
// stores, fixed order: bit-reproducible.  (First build: a separate pass of 32 two-byte LDS reads per thread and a
// [2][32][BNC] LDS reduction -- 50-80 us per layer1 launch.)
template <int CPR>
__device__ __forceinline__ float strided_lane_sum(float v) {
  static_assert(CPR == 8 || CPR == 16, "chunk columns per row");
  if constexpr (CPR == 8) v += wm_dpp<0x128>(v);  // row_ror:8: lane i + lane i ^ 8
  v = wm_xor16_sum(v);
  v = wm_xor32_sum(v);
  return v;
}
template <int BNC>
__device__ __forceinline__ void block_colsums_store(float (&s1)[8], float (&s2)[8], float* red, float* slot, int DC,
                                                    int c_base, int tid) {
  constexpr int CPR = BNC / 8;
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    s1[e] = strided_lane_sum<CPR>(s1[e]);
    s2[e] = strided_lane_sum<CPR>(s2[e]);
  }
  if (lane < CPR) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[wave * BNC + lane * 8 + e] = s1[e];
      red[(4 + wave) * BNC + lane * 8 + e] = s2[e];
    }
  }
  __syncthreads();
  if (tid < 2 * BNC) {
    const int which = tid / BNC, cc = tid % BNC;
    const float* r = red + (size_t)which * 4 * BNC + cc;
    slot[(size_t)which * DC + c_base + cc] = ((r[0] + r[BNC]) + r[2 * BNC]) + r[3 * BNC];
  }
}

// Store a FULL staged bf16 tile [128 rows][CS bytes] (BNC channels, row -> pixel by pix_of) and leave its per-channel
// (sum, sum of squares) in the tile's statistics slot: the sums ride on the store loop's own LDS reads.
template <int BNC, typename PixOf>
__device__ __forceinline__ void store_tile_with_stats(const uint8_t* smem, int CS, uint16_t* dst, int DC, int n0, float* red,
                                                      float* slot, int tid, PixOf pix_of) {
  constexpr int CPR = BNC / 8;
  constexpr int NIT = 128 * CPR / CV_THREADS;
  const int chl = tid % CPR;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = (tid + it * CV_THREADS) / CPR;
    const uint4 v = *reinterpret_cast<const uint4*>(smem + row * CS + chl * 16);
    const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float lo = __builtin_bit_cast(float, vv[q] << 16), hi = __builtin_bit_cast(float, vv[q] & 0xffff0000u);
      s1[2 * q] += lo;
      s2[2 * q] = fmaf(lo, lo, s2[2 * q]);
      s1[2 * q + 1] += hi;
      s2[2 * q + 1] = fmaf(hi, hi, s2[2 * q + 1]);
    }
    *reinterpret_cast<uint4*>(dst + pix_of(row) * DC + n0 + chl * 8) = v;
  }
  block_colsums_store<BNC>(s1, s2, red, slot, DC, n0, tid);
}

// BatchNorm-backward epilogue of a dgrad tile (ConvArgs: bn_*), in two parts.
// bnb_prefetch (at kernel START): the epilogue's global operands -- the BatchNorm input tile, the residual-gradient
// tile, the mask source tile, 16 bytes per (thread, pass) each -- are requested with inline-asm loads before the MFMA
// loop, so their HBM latency passes under the loop instead of at the tail of every block (first build, with the loads
// in the epilogue: the dgrad launches took 0.7 ms per step longer, as much as the removed reduction passes).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
template <int BNC>
struct BnbRegs {
  static constexpr int NIT = 128 * (BNC / 8) / CV_THREADS;
  u32x4_t yv[NIT], rv[NIT], xv[NIT];
  uint32_t mk[NIT];   // ReLU mask byte of the pass (bn_mask form)
  size_t pixs[NIT];
};
// Plain (compiler-tracked) loads: the k-loop's asm statements clobber "memory", so hipcc cannot sink these below the
// loop, and because it tracks them it waits before any copy of their registers.  (Untracked inline-asm loads are NOT
// safe here: under register pressure hipcc splits the live range -- copies the destination registers before the data
// has landed -- and the late-landing load then overwrites whatever lives in the old registers.)
__device__ __forceinline__ void bnb_load16(u32x4_t& dst, const void* p) {
  // read-once operand tiles: non-temporal, so that they do not displace the dY rows and weights the DMA loop re-reads
  // through L2
  dst = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
}
template <int BNC, typename PixOf>
__device__ __forceinline__ void bnb_prefetch(const ConvArgs& a, BnbRegs<BNC>& R, int n0, int tid, PixOf pix_of) {
  constexpr int CPR = BNC / 8;
  const int c0 = n0 + (tid % CPR) * 8;
#pragma unroll
  for (int it = 0; it < BnbRegs<BNC>::NIT; ++it) {
    const int row = (tid + it * CV_THREADS) / CPR;
    const size_t pix = pix_of(row);  // (tiles are full: the host admits only shapes whose 128-row tiles tile a group)
    R.pixs[it] = pix;
    bnb_load16(R.yv[it], a.bn_y + pix * a.DC + c0);
    if (a.res != nullptr) bnb_load16(R.rv[it], a.res + pix * a.DC + c0);
    if (a.bn_x != nullptr) bnb_load16(R.xv[it], a.bn_x + pix * a.DC + c0);
    R.mk[it] = a.bn_mask != nullptr ? a.bn_mask[pix * (a.DC >> 3) + (c0 >> 3)] : 0u;
  }
}

// bnb_epilogue: staged bf16 tile [128 rows][CS bytes] of BNC channels -> (+ residual) -> ReLU mask -> store, and the
// tile's (sum g, sum g * (bn_y - mean)) into its statistics slot.  `red`: 2 x 4 x BNC floats of LDS outside the tile.
template <int BNC>
__device__ __forceinline__ void bnb_epilogue(const ConvArgs& a, const BnbRegs<BNC>& R, const uint8_t* smem, int CS, float* red,
                                             int g, int n0, int tile, int tid) {
  constexpr int CPR = BNC / 8;
  constexpr int NIT = BnbRegs<BNC>::NIT;
  const int chl = tid % CPR;
  const int c0 = n0 + chl * 8;
  // ReLU mask recomputed from the BatchNorm input (no shortcut): bn_y * scale + shift > 0 with the forward's scale and
  // shift.  (The forward rounds to bf16 before its ReLU; that rounding changes the sign test only for |value| < 2^-133.)
  const bool remask = a.bn_x == nullptr && a.bn_mask == nullptr;
  const bool bitmask = a.bn_mask != nullptr;
  float sc[8], sh[8], mu[8];
  const float* pm = a.bn_mean + (size_t)g * a.DC + c0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = sh[e] = 0.f;
    mu[e] = pm[e];
  }
  if (remask) {
    const float* pi = a.bn_invstd + (size_t)g * a.DC + c0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {  // bn_fwd_finalize: scale = gamma * invstd, shift = beta - mean * scale
      sc[e] = a.bn_gamma[c0 + e] * pi[e];
      sh[e] = a.bn_beta[c0 + e] - mu[e] * sc[e];
    }
  }
  // per element: unpack, (+ residual), mask, two running sums (sum g and sum g * (y - mean): the finalize kernel scales
  // the second by invstd to sum g * xhat; centring per element instead of correcting sum g * y by mean * sum g keeps
  // the f32 partials free of cancellation when |mean| >> std), pack: ~11 VALU instructions.  The first build
  // (xhat per element, bf16 round trips for the residual sum and the mask) spent as long here as in the MFMA loop.
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = (tid + it * CV_THREADS) / CPR;
    const uint4 v4 = *reinterpret_cast<const uint4*>(smem + row * CS + chl * 16);
    const uint32_t vv[4] = {v4.x, v4.y, v4.z, v4.w};
    const uint32_t rr[4] = {R.rv[it][0], R.rv[it][1], R.rv[it][2], R.rv[it][3]};
    const uint32_t yy[4] = {R.yv[it][0], R.yv[it][1], R.yv[it][2], R.yv[it][3]};
    const uint32_t xx[4] = {R.xv[it][0], R.xv[it][1], R.xv[it][2], R.xv[it][3]};
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float gv[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = q * 2 + h;
        float v = __builtin_bit_cast(float, h ? (vv[q] & 0xffff0000u) : (vv[q] << 16));
        if (a.res != nullptr) v += __builtin_bit_cast(float, h ? (rr[q] & 0xffff0000u) : (rr[q] << 16));
        const float y = __builtin_bit_cast(float, h ? (yy[q] & 0xffff0000u) : (yy[q] << 16));
        // (x is a bf16 pattern: > 0 <=> the 16-bit pattern is a positive non-zero number)
        const bool keep = remask ? fmaf(y, sc[e], sh[e]) > 0.f
                          : bitmask ? ((R.mk[it] >> e) & 1u) != 0u
                                    : __builtin_bit_cast(float, h ? (xx[q] & 0xffff0000u) : (xx[q] << 16)) > 0.f;
        v = keep ? v : 0.f;
        s1[e] += v;
        s2[e] = fmaf(v, y - mu[e], s2[e]);
        gv[h] = v;
      }
      o[q] = pack_bf2(gv[0], gv[1]);
    }
    *reinterpret_cast<uint4*>(a.dst + R.pixs[it] * a.DC + c0) = make_uint4(o[0], o[1], o[2], o[3]);
  }
  block_colsums_store<BNC>(s1, s2, red, a.stat + ((size_t)(g * a.stat_nb + tile) * 2) * a.DC, a.DC, n0, tid);
}

// Both operand tiles go HBM -> LDS by global_load_lds (16 B per lane, 1 KiB per wave instruction,
// no VGPR staging, no ds_write).  The DMA destination is lane-linear, so a lane fetches the LOGICAL
// chunk that belongs at its physical slot: chunk = slot ^ (row & 7) (the same involution is applied
// on the fragment reads).  Two stages; the tile for k-step kt+1 is in flight under the MFMAs of kt:
//   s_waitcnt vmcnt(0) ; s_barrier ; issue(kt+1) ; compute(kt)
// MODE 0: forward.  MODE 1: dgrad (any stride; stride-2 taps that do not hit a source pixel are
// fetched as zeros).  MODE 3: dgrad, stride 1 only (the common case, without the stride-2 address path).  MODE 2: dgrad of a stride-2 convolution with the destination pixels ordered by
// parity class (h&1, w&1): all 128 rows of a tile then share the set of taps that hit a source
// pixel (r = r0 + 2 jr, s = s0 + 2 js), so only those k-tiles are executed — 9/4 instead of 9 per
// pixel for 3x3, and three of four classes of the 1x1 downsample just store zeros.
// EPI: epilogue with bias / prefetched residual (Linear layers).  It is a separate instantiation
// because carrying its registers in the convolution kernels cost them ~8 %.
template <int BM, int BN, int CPT, int MODE, bool EPI = false, bool BNB = false>
__global__ __launch_bounds__(CV_THREADS) void conv_igemm(const ConvArgs a) {
  constexpr bool DGRAD = MODE != 0;
  static_assert(!BNB || (DGRAD && !EPI && BM == 128), "BatchNorm-backward epilogue: dgrad tiles of 128 rows");
  extern __shared__ __attribute__((aligned(16))) uint8_t cv_smem[];
  // BM x BN tile, 4 waves: 2 x 2 waves of 64 px x BN/2 ch for BM = 128; 4 x 1 waves of 64 px x BN ch
  // for BM = 256 (measured no faster than 128 x 64 on the 64-channel layers: those are bound by the
  // 9-fold re-read of the input through L2, not by MFMA issue; kept as a template option)
  constexpr int WN = BM == 128 ? 2 : 1;
  constexpr int A_BYTES = BM * CV_ROW;
  constexpr int B_BYTES = BN * CV_ROW;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int NR = BM / 32;          // pixel rows fetched per thread
  constexpr int NB = BN / 32;          // weight rows fetched per thread
  constexpr int NJ = BN / WN / 16;     // 16-channel fragments per wave

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
  // tile of this workgroup: column tiles fastest (they share the whole pixel tile), pixel tiles next (neighbours share
  // their halo rows), each XCD a contiguous range of that order (common.h: wm_xcd_swizzle)
  // (MODE 2: its tiles are ordered by parity class and the classes cost different amounts -- three of the four classes
  // of a 1x1 / stride-2 gradient only store zeros -- so a plain contiguous range per XCD gives one XCD all the
  // expensive tiles: measured 18.6 -> 40 us on layer4's downsample, 128 -> 164 us on layer2.0's 3x3)
  int bm = blockIdx.x, bn = blockIdx.y;
  if (a.xcd) {
    const uint32_t lin = wm_xcd_swizzle(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
    bm = (int)(lin / gridDim.y);
    bn = (int)(lin - (uint32_t)bm * gridDim.y);
    if constexpr (MODE == 2) {
      // tiles are stored class by class (gridDim.x / 4 tiles each): walk the four classes INTERLEAVED, so that every
      // XCD's contiguous range holds the same mix of cheap and expensive tiles and neighbours inside a class stay close
      const int tpc = (int)(gridDim.x >> 2);
      if (a.xcd == 3 && (gridDim.x & 3) == 0) bm = (bm & 3) * tpc + (bm >> 2);
      else { bm = blockIdx.x; bn = blockIdx.y; }
    }
  }
  const int m0 = bm * BM, n0 = bn * BN;
  const int rowl = tid >> 3;                   // rows rowl + 32 i; (rowl + 32 i) & 7 == rowl & 7
  const int chunk = (tid & 7) ^ (rowl & 7);    // logical 16-byte chunk this lane fetches

  int bh[NR], bw[NR], nb[NR];
  bool mv[NR];
  const uint16_t* p0[NR];
  // MODE 2: parity class of this tile and its valid-tap grid
  int r0 = 0, s0 = 0, nr = 0, ns = 0, nkt = a.nkt;
  if constexpr (MODE == 2) {
    const int cls = a.M >> 2;
    const int pc = m0 / cls;
    const int ph = pc >> 1, pw = pc & 1;
    r0 = (ph + a.pad) & 1;
    s0 = (pw + a.pad) & 1;
    nr = r0 < a.R ? (a.R - r0 + 1) >> 1 : 0;
    ns = s0 < a.S ? (a.S - s0 + 1) >> 1 : 0;
    nkt = nr * ns * (a.SC >> 6);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int m = m0 + rowl + 32 * i - pc * cls;  // index inside the class
      mv[i] = true;                                   // class size % 128 == 0 (host-checked)
      uint32_t urem2, uw2;
      const int n = (int)wm_divmod((uint32_t)m, a.d_h2w2, urem2);
      const int h2 = (int)wm_divmod(urem2, a.d_w2, uw2), w2 = (int)uw2;
      // source pixel of tap (r, s): (h2 + (ph + pad - r)/2, w2 + (pw + pad - s)/2)
      bh[i] = h2 + ((ph + a.pad - r0) >> 1);
      bw[i] = w2 + ((pw + a.pad - s0) >> 1);
      nb[i] = n * a.SH * a.SW;
      p0[i] = a.src + ((long long)(nb[i] + bh[i] * a.SW + bw[i]) * a.SC + chunk * 8);
    }
  } else {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int m = m0 + rowl + 32 * i;
      mv[i] = m < a.M;
      const int mm = mv[i] ? m : 0;
      uint32_t urem, udw;
      const int n = (int)wm_divmod((uint32_t)mm, a.d_dhw, urem);
      const int dh = (int)wm_divmod(urem, a.d_dw, udw), dw = (int)udw;
      if constexpr (!DGRAD) {
        bh[i] = dh * a.stride - a.pad;
        bw[i] = dw * a.stride - a.pad;
      } else {
        bh[i] = dh + a.pad;
        bw[i] = dw + a.pad;
      }
      nb[i] = n * a.SH * a.SW;
      // address of (tap 0, channel chunk) for this row; may lie outside the tensor (never read then)
      p0[i] = a.src + ((long long)(nb[i] + bh[i] * a.SW + bw[i]) * a.SC + chunk * 8);
    }
  }
  const size_t wrow = (size_t)a.R * a.S * a.SC;
  const uint16_t* pb[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) pb[i] = a.wt + (size_t)(n0 + rowl + 32 * i) * wrow + chunk * 8;

  // running decode of the k-tile index (issue() is called for kt = 0, 1, 2, ... in order): channel
  // offset and tap coordinates advance by carries instead of two integer divisions per k-step
  int it_c0 = 0, it_r = 0, it_s = 0;
  const uint32_t smem_base = lds_addr(cv_smem);
  auto issue = [&](int kt, uint32_t stage) {
    int r, s;
    long long koff;  // uniform element offset of this k-tile relative to p0
    const int c0 = it_c0;
    if constexpr (MODE == 2) {
      const int jr = it_r, js = it_s;
      it_c0 += 64;
      if (it_c0 == a.SC) {
        it_c0 = 0;
        if (++it_s == ns) {
          it_s = 0;
          ++it_r;
        }
      }
      r = jr;  // steps of -1 source pixel per valid tap
      s = js;
      koff = (long long)(-jr * a.SW - js) * a.SC + c0;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int sh = bh[i] - jr, sw = bw[i] - js;
        const bool ok = (unsigned)sh < (unsigned)a.SH && (unsigned)sw < (unsigned)a.SW;
        const uint16_t* src = ok ? p0[i] + koff : conv_zero_page + chunk * 8;
        glds16_at(src, stage + (wave * 8 + 32 * i) * CV_ROW);
      }
      // weights: tap (r0 + 2 jr, s0 + 2 js)
      const size_t wk = ((size_t)((r0 + 2 * jr) * a.S + (s0 + 2 * js)) * a.SC) + c0;
#pragma unroll
      for (int i = 0; i < NB; ++i)
        glds16_at(pb[i] + wk, stage + A_BYTES + (wave * 8 + 32 * i) * CV_ROW);
      return;
    } else if constexpr (CPT == 8) {
      r = it_r;
      s = it_s;
      it_c0 += 64;
      if (it_c0 == a.SC) {
        it_c0 = 0;
        if (++it_s == a.S) {
          it_s = 0;
          ++it_r;
        }
      }
      koff = DGRAD ? ((long long)(-r * a.SW - s) * a.SC + c0) : ((long long)(r * a.SW + s) * a.SC + c0);
    } else {  // 16-channel source: one k-tile = kernel row kt; the lane's chunk picks the tap
      r = kt;
      s = chunk >> 1;
      koff = (long long)r * a.SW * a.SC;
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      bool ok = mv[i];
      const uint16_t* src;
      if constexpr (!DGRAD) {
        const int sh = bh[i] + r, sw = bw[i] + s;
        ok = ok && (unsigned)sh < (unsigned)a.SH && (unsigned)sw < (unsigned)a.SW;
        src = p0[i] + koff;
      } else {
        const int th = bh[i] - r, tw = bw[i] - s;
        ok = ok && th >= 0 && tw >= 0;
        if (MODE == 1 && a.stride == 2) {  // MODE 3: stride 1 known at compile time
          ok = ok && (((th | tw) & 1) == 0);
          const int sh = th >> 1, sw = tw >> 1;
          ok = ok && sh < a.SH && sw < a.SW;
          src = a.src + ((long long)(nb[i] + sh * a.SW + sw) * a.SC + c0 + chunk * 8);
        } else {
          ok = ok && th < a.SH && tw < a.SW;
          src = p0[i] + koff;
        }
      }
      if (!ok) src = conv_zero_page + chunk * 8;
      glds16_at(src, stage + (wave * 8 + 32 * i) * CV_ROW);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
      glds16_at(pb[i] + (size_t)kt * 64, stage + A_BYTES + (wave * 8 + 32 * i) * CV_ROW);
  };

  f32x4_t acc[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  auto compute = [&](const uint8_t* buf) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ks * 4 + fg;
      bf16x8_t xf[4], wf[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + fr;
        xf[i] = *reinterpret_cast<const bf16x8_t*>(buf + row * CV_ROW + ((c ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int row = wn * (BN / WN) + j * 16 + fr;
        wf[j] = *reinterpret_cast<const bf16x8_t*>(buf + A_BYTES + row * CV_ROW + ((c ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[j][i], 0, 0, 0);
    }
  };

  BnbRegs<BNB ? BN : 64> bnb;  // (unused and eliminated unless BNB)
  if constexpr (BNB) {
    bnb_prefetch<BN>(a, bnb, n0, tid, [&](int row) -> size_t {
      if constexpr (MODE == 2) {  // class-ordered row -> pixel (n, 2 h2 + ph, 2 w2 + pw)
        const int cls = a.M >> 2;
        const int pc = m0 / cls;
        const int m = m0 + row - pc * cls;
        uint32_t urem2, uw2;
        const int n = (int)wm_divmod((uint32_t)m, a.d_h2w2, urem2);
        const int h2 = (int)wm_divmod(urem2, a.d_w2, uw2), w2 = (int)uw2;
        return ((size_t)n * a.DH + 2 * h2 + (pc >> 1)) * a.DW + 2 * w2 + (pc & 1);
      } else {
        return (size_t)(m0 + row);
      }
    });
  }
  if (nkt > 0) issue(0, smem_base);
  for (int kt = 0; kt < nkt; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile kt have landed
    wm_barrier();                     // ... everyone's; and compute(kt-1) is over
    if (kt + 1 < nkt && !(WM_CONV_ABLATE & 1)) issue(kt + 1, smem_base + ((kt + 1) & 1) * STAGE);
    else wm_barrier();  // nothing to issue: keep a phase between the wait that retired this tile and its fragment reads
                        // (LDS-DMA data becomes readable a little after vmcnt retires it: see conv3x3_patch)
    if (!(WM_CONV_ABLATE & 2)) compute(cv_smem + (kt & 1) * STAGE);
  }
  __syncthreads();

  // ---- epilogue: accumulators -> bf16 tile in LDS ([pixel][channel], rows padded by 16 B) -> HBM
  constexpr int CS = BN * 2 + 16;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pix = wm * 64 + i * 16 + fr;
      const int ch = wn * (BN / WN) + j * 16 + fg * 4;
      const uint2 v = make_uint2(pack_bf2(acc[j][i][0], acc[j][i][1]), pack_bf2(acc[j][i][2], acc[j][i][3]));
      *reinterpret_cast<uint2*>(cv_smem + pix * CS + ch * 2) = v;
    }
  __syncthreads();
  if constexpr (MODE == 0 && !EPI) {
    if (a.stat != nullptr) {
      // rows_per_group % BM == 0: the tile is full and lies inside one statistics group; its column sums ride on the
      // store loop
      const int g = m0 / a.stat_rpg;
      float* slot = a.stat + ((size_t)(g * a.stat_nb + (m0 - g * a.stat_rpg) / BM) * 2) * a.DC;
      store_tile_with_stats<BN>(cv_smem, CS, a.dst, a.DC, n0, reinterpret_cast<float*>(cv_smem + BM * CS), slot, tid,
                                [&](int row) -> size_t { return (size_t)(m0 + row); });
      return;
    }
  }
  if constexpr (BNB) {
    int g, tile;  // statistics group of this tile and its slot inside the group
    if constexpr (MODE == 2) {
      const int cls = a.M >> 2;
      const int pc = m0 / cls, in_cls = m0 - pc * cls;
      g = in_cls / a.stat_rpg;  // stat_rpg: rows of one statistics group inside a parity class
      tile = pc * (a.stat_rpg / BM) + (in_cls - g * a.stat_rpg) / BM;
    } else {
      g = m0 / a.stat_rpg;
      tile = (m0 - g * a.stat_rpg) / BM;
    }
    bnb_epilogue<BN>(a, bnb, cv_smem, CS, reinterpret_cast<float*>(cv_smem + BM * CS), g, n0, tile, tid);
    return;
  }
  if constexpr (EPI) {
    constexpr int CPR = BN / 8;
    constexpr int NIT = BM * CPR / CV_THREADS;  // 16-byte chunks per thread
    size_t pixs[NIT];
    uint4 rv[NIT];
  #pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int p = tid + it * CV_THREADS;
      const int row = p / CPR, ch = p - row * CPR;
      size_t pix = (size_t)(m0 + row);
      if constexpr (MODE == 2) {  // class-ordered row -> pixel (n, 2 h2 + ph, 2 w2 + pw)
        const int cls = a.M >> 2;
        const int pc = m0 / cls;
        const int m = m0 + row - pc * cls;
            uint32_t urem2, uw2;
        const int n = (int)wm_divmod((uint32_t)m, a.d_h2w2, urem2);
        const int h2 = (int)wm_divmod(urem2, a.d_w2, uw2), w2 = (int)uw2;
        pix = ((size_t)n * a.DH + 2 * h2 + (pc >> 1)) * a.DW + 2 * w2 + (pc & 1);
      }
      pixs[it] = pix;
      // all residual chunks of the thread are requested before the first store (they may alias dst as
      // far as the compiler knows, so it would not hoist them itself)
      rv[it] = make_uint4(0, 0, 0, 0);
      if (a.res != nullptr && m0 + row < a.M) rv[it] = *reinterpret_cast<const uint4*>(a.res + pix * a.DC + n0 + ch * 8);
      // (act 2 never comes with a residual: the same registers carry the pre-activation chunk)
      if (a.act == 2 && m0 + row < a.M) rv[it] = *reinterpret_cast<const uint4*>(a.pre_in + pix * a.DC + n0 + ch * 8);
    }
    // the lane's eight bias values: its chunk column (tid % CPR) is the same in every pass (CV_THREADS % CPR == 0),
    // and inside the loop the loads could not move above the previous pass's stores (they may alias as far as hipcc
    // knows): eight dependent L2 round trips per tile
    static_assert(CV_THREADS % CPR == 0, "chunk column constant per lane");
    float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
    if constexpr (MODE == 0) {
      if (a.bias != nullptr) {
        b0 = *reinterpret_cast<const float4*>(a.bias + n0 + (tid % CPR) * 8);
        b1 = *reinterpret_cast<const float4*>(a.bias + n0 + (tid % CPR) * 8 + 4);
      }
    }
  #pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int p = tid + it * CV_THREADS;
      const int row = p / CPR, ch = p - row * CPR;
      const size_t pix = pixs[it];
      if (m0 + row < a.M) {
        uint4 v = *reinterpret_cast<const uint4*>(cv_smem + row * CS + ch * 16);
        if constexpr (MODE == 0) {
          if (a.bias != nullptr) {
            v = make_uint4(pack_bf2(bf2f((uint16_t)(v.x & 0xffff)) + b0.x, bf2f((uint16_t)(v.x >> 16)) + b0.y),
                           pack_bf2(bf2f((uint16_t)(v.y & 0xffff)) + b0.z, bf2f((uint16_t)(v.y >> 16)) + b0.w),
                           pack_bf2(bf2f((uint16_t)(v.z & 0xffff)) + b1.x, bf2f((uint16_t)(v.z >> 16)) + b1.y),
                           pack_bf2(bf2f((uint16_t)(v.w & 0xffff)) + b1.z, bf2f((uint16_t)(v.w >> 16)) + b1.w));
          }
        }
        if (a.act == 1) {  // v = biased pre-activation (bf16): keep it for the backward pass, emit gelu(v)
          *reinterpret_cast<uint4*>(a.pre_out + pix * a.DC + n0 + ch * 8) = v;
          const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
          uint32_t o[4];
  #pragma unroll
          for (int e = 0; e < 4; ++e)
            o[e] = pack_bf2(cv_gelu(bf2f((uint16_t)(vv[e] & 0xffff))), cv_gelu(bf2f((uint16_t)(vv[e] >> 16))));
          v = make_uint4(o[0], o[1], o[2], o[3]);
        } else if (a.act == 2) {  // v = gradient w.r.t. gelu(pre): times gelu'(pre)
          const uint4 r4 = rv[it];
          const uint32_t vv[4] = {v.x, v.y, v.z, v.w}, rr[4] = {r4.x, r4.y, r4.z, r4.w};
          uint32_t o[4];
  #pragma unroll
          for (int e = 0; e < 4; ++e)
            o[e] = pack_bf2(bf2f((uint16_t)(vv[e] & 0xffff)) * cv_gelu_grad(bf2f((uint16_t)(rr[e] & 0xffff))),
                            bf2f((uint16_t)(vv[e] >> 16)) * cv_gelu_grad(bf2f((uint16_t)(rr[e] >> 16))));
          v = make_uint4(o[0], o[1], o[2], o[3]);
        } else if (a.res != nullptr) {
          const uint4 r4 = rv[it];
          const uint32_t vv[4] = {v.x, v.y, v.z, v.w}, rr[4] = {r4.x, r4.y, r4.z, r4.w};
          uint32_t o[4];
  #pragma unroll
          for (int e = 0; e < 4; ++e)
            o[e] = pack_bf2(bf2f((uint16_t)(vv[e] & 0xffff)) + bf2f((uint16_t)(rr[e] & 0xffff)),
                            bf2f((uint16_t)(vv[e] >> 16)) + bf2f((uint16_t)(rr[e] >> 16)));
          v = make_uint4(o[0], o[1], o[2], o[3]);
        }
        *reinterpret_cast<uint4*>(a.dst + pix * a.DC + n0 + ch * 8) = v;
      }
    }

  } else {
    constexpr int CPR = BN / 8;
    for (int p = tid; p < BM * CPR; p += CV_THREADS) {
      const int row = p / CPR, ch = p - row * CPR;
      size_t pix = (size_t)(m0 + row);
      if constexpr (MODE == 2) {  // class-ordered row -> pixel (n, 2 h2 + ph, 2 w2 + pw)
        const int cls = a.M >> 2;
        const int pc = m0 / cls;
        const int m = m0 + row - pc * cls;
            uint32_t urem2, uw2;
        const int n = (int)wm_divmod((uint32_t)m, a.d_h2w2, urem2);
        const int h2 = (int)wm_divmod(urem2, a.d_w2, uw2), w2 = (int)uw2;
        pix = ((size_t)n * a.DH + 2 * h2 + (pc >> 1)) * a.DW + 2 * w2 + (pc & 1);
      }
      if (m0 + row < a.M) {
        uint4 v = *reinterpret_cast<const uint4*>(cv_smem + row * CS + ch * 16);
        if (a.res != nullptr) {
          const uint4 r4 = *reinterpret_cast<const uint4*>(a.res + pix * a.DC + n0 + ch * 8);
          const uint32_t vv[4] = {v.x, v.y, v.z, v.w}, rr[4] = {r4.x, r4.y, r4.z, r4.w};
          uint32_t o[4];
  #pragma unroll
          for (int e = 0; e < 4; ++e)
            o[e] = pack_bf2(bf2f((uint16_t)(vv[e] & 0xffff)) + bf2f((uint16_t)(rr[e] & 0xffff)),
                            bf2f((uint16_t)(vv[e] >> 16)) + bf2f((uint16_t)(rr[e] >> 16)));
          v = make_uint4(o[0], o[1], o[2], o[3]);
        }
        *reinterpret_cast<uint4*>(a.dst + pix * a.DC + n0 + ch * 8) = v;
      }
    }

  }
}

// ------------------------------------------------------------- 3x3 / stride 1 on 64 channels (layer1)
// conv_igemm fetches the pixel operand once per TAP: nine 16-KB tiles per 128 output pixels, each
// through L2.  On the 64-channel layers (9 k-tiles, 64 output channels per block) that re-read, not
// MFMA issue, bounds the kernel (~545 TFLOP/s where the 128/256-channel layers reach 850-990).  Here a
// block owns two 8x8 output tiles and DMAs their 10x10-pixel source patches into LDS ONCE (28 KB
// instead of 144 KB); the nine taps then only shift the fragment-read address inside the patch, and
// the loop fetches nothing but the 8-KB weight slices of the coming taps (3-stage ring, two in flight).
// (A persistent variant -- one block per CU, all nine taps' weights resident, patches double buffered, no
// barrier inside a step -- was measured SLOWER, 219 vs 177 us: with one wave per SIMD nothing overlaps a
// block's fragment reads, epilogue and stores; three independent blocks per CU do.  A 4-stage weight ring
// (61 KB of LDS, two blocks per CU) was slower as well: 194 vs 168 us.)
//   patch layout: [tile][row 0..9][col 0..10][64 ch] = 128-byte pixels, row pitch 11 pixels (col 10 is
//   padding); 16-byte slot = chunk ^ (col & 7).  A fragment read touches 2 rows x 8 cols: equal cols
//   of the two rows differ in the parity of the pixel index (pitch 11 is odd), i.e. in the half of the
//   256-byte bank line, and the 8 cols of a row take 8 distinct slots: conflict free.
//   The per-lane addresses are 6 loop-invariant registers (3 column shifts x 2 k-steps) + immediates.
// forward (MODE 0): source pixel of tap (r, s) = (h - 1 + r, w - 1 + s); dgrad (MODE 1, weights
// [c][r][s][k]): (h + 1 - r, w + 1 - s).  Waves, accumulators and the epilogue (bf16 tile through LDS,
// fused BatchNorm sums, optional residual) are those of conv_igemm<128, 64>.
constexpr int PT_PITCH = 11;
constexpr int PT_PIX = 10 * PT_PITCH;        // pixel slots per patch
constexpr int PT_SLOTS = 224;                // 2 patches = 220 slots, rounded up to 28 DMA instructions
constexpr int PT_PATCH_BYTES = PT_SLOTS * CV_ROW;
constexpr int PT_W_BYTES = 64 * CV_ROW;      // one tap's weights: 64 rows x 64 source channels
constexpr int PT_WSTAGES = 3;                 // weight ring: two taps in flight ahead of the one being multiplied

constexpr int PT_LDS = PT_PATCH_BYTES + PT_WSTAGES * PT_W_BYTES;

template <int MODE, bool BNB = false>
__global__ __launch_bounds__(CV_THREADS, BNB ? 3 : 1) void conv3x3_patch(const ConvArgs a) {  // BNB: keep three blocks per CU
  constexpr bool DGRAD = MODE != 0;
  static_assert(!BNB || DGRAD, "BatchNorm-backward epilogue: dgrad only");
  extern __shared__ __attribute__((aligned(16))) uint8_t cv_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;  // 8x8 tile of the pair, 32-channel half
  const int tw_n = a.DW >> 3, tiles_img = (a.DH >> 3) * tw_n;
  const uint32_t smem_base = lds_addr(cv_smem);

  // ---- patches: 28 instructions x 8 pixel slots; wave w issues instructions w, w + 4, ...
  // (the two tile origins are decoded ONCE from the block index, as wave-uniform values: integer divisions by the
  // run-time image sizes per fetched piece and per stored chunk were a third of this kernel's instructions)
  // (tile pairs in XCD-contiguous order: neighbouring tiles share their halo columns / rows through one L2)
  // (measured neutral to slightly negative here, 157 -> 164 us forward: the 10 x 10 patches of neighbouring tile pairs
  // overlap by two columns only; kept as an experiment switch, WM_XCD_SWIZZLE=2)
  const int bx = a.xcd == 2 ? (int)wm_xcd_swizzle(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  int tn[2], th0[2], tw0[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int T = bx * 2 + t;
    tn[t] = T / tiles_img;
    const int tr = T - tn[t] * tiles_img;
    const int th = tr / tw_n;
    th0[t] = th * 8;
    tw0[t] = (tr - th * tw_n) * 8;
  }
  {
    const int sl = lane & 7;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int j = wave + 4 * i;
      const int gp = 8 * j + (lane >> 3);
      const int t = gp >= PT_PIX ? 1 : 0;
      const int pp = gp - t * PT_PIX;
      const int py = pp / PT_PITCH, px = pp - py * PT_PITCH;
      const int h = (t ? th0[1] : th0[0]) - 1 + py, w = (t ? tw0[1] : tw0[0]) - 1 + px;
      const bool ok = gp < 2 * PT_PIX && px < 10 && (unsigned)h < (unsigned)a.SH && (unsigned)w < (unsigned)a.SW;
      const int chunk = sl ^ (px & 7);
      const uint16_t* src = ok ? a.src + ((size_t)((t ? tn[1] : tn[0]) * a.SH + h) * a.SW + w) * 64 + chunk * 8
                               : conv_zero_page + chunk * 8;
      glds16_at(src, smem_base + j * 1024);
    }
  }
  BnbRegs<64> bnb;  // (unused and eliminated unless BNB)
  if constexpr (BNB) {
    const size_t porg[2] = {((size_t)tn[0] * a.DH + th0[0]) * a.DW + tw0[0], ((size_t)tn[1] * a.DH + th0[1]) * a.DW + tw0[1]};
    bnb_prefetch<64>(a, bnb, 0, tid, [&](int row) -> size_t {
      return porg[row >> 6] + (size_t)((row >> 3) & 7) * a.DW + (row & 7);
    });
  }
  // ---- weights of one tap: 8 instructions, two per wave (rows rowl, rowl + 32)
  const int rowl = tid >> 3;
  const int wchunk = (tid & 7) ^ (rowl & 7);
  const size_t wrow = (size_t)9 * 64;
  const uint16_t* pb0 = a.wt + (size_t)rowl * wrow + wchunk * 8;
  const uint16_t* pb1 = pb0 + 32 * wrow;
  auto issue_w = [&](int tap, uint32_t stage) {
    glds16_at(pb0 + tap * 64, stage + (wave * 8) * CV_ROW);
    glds16_at(pb1 + tap * 64, stage + (wave * 8 + 32) * CV_ROW);
  };
  issue_w(0, smem_base + PT_PATCH_BYTES);
  issue_w(1, smem_base + PT_PATCH_BYTES + PT_W_BYTES);
  if (PT_WSTAGES > 3) issue_w(2, smem_base + PT_PATCH_BYTES + 2 * PT_W_BYTES);

  f32x4_t acc[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  const int dy = fr >> 3, dx = fr & 7;
  // pixel-fragment addresses: lane base + (column shift, k-step) variant + immediate row offset
  uint32_t xoff[3][2];
#pragma unroll
  for (int sh = 0; sh < 3; ++sh)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      xoff[sh][ks] = (uint32_t)((wm * PT_PIX + dy * PT_PITCH + dx + sh) * CV_ROW + (((ks * 4 + fg) ^ ((dx + sh) & 7)) << 4));
  uint32_t woff[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int row = wn * 32 + j * 16 + fr;
      woff[j][ks] = (uint32_t)(PT_PATCH_BYTES + row * CV_ROW + (((ks * 4 + fg) ^ (row & 7)) << 4));
    }

  // Ring discipline (see wm_barrier in common.h for the race this kernel exposed): RAW -- a wave waits for ITS pieces of
  // tap `tap` with a counted vmcnt (the two instructions of tap + 1 may stay in flight; vector-memory operations
  // retire in issue order: tools/probes/ldsdma_order_probe.hip) and the barrier makes that everyone's; WAR -- the
  // barrier's lgkmcnt(0) retires every wave's fragment reads of the previous tap before any wave restages that tap's
  // ring slot.
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    constexpr int AHEAD = PT_WSTAGES - 1;  // taps in flight ahead (incl. the one waited for)
    if (tap + AHEAD - 1 < 9) {
      if (AHEAD == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else if (tap + 1 < 9 && AHEAD == 3) {
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    wm_barrier();                                     // ... everyone's; and the previous tap's reads are over
    if (tap + AHEAD < 9) issue_w(tap + AHEAD, smem_base + PT_PATCH_BYTES + ((tap + AHEAD) % PT_WSTAGES) * PT_W_BYTES);
    const int r = tap / 3, sx = tap % 3;
    const int prow = DGRAD ? 2 - r : r;      // patch row shift
    const int pcol = DGRAD ? 2 - sx : sx;    // patch column shift
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t xf[4], wf[2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        xf[i] = *reinterpret_cast<const bf16x8_t*>(cv_smem + xoff[pcol][ks] + (2 * i + prow) * PT_PITCH * CV_ROW);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        wf[j] = *reinterpret_cast<const bf16x8_t*>(cv_smem + woff[j][ks] + (tap % PT_WSTAGES) * PT_W_BYTES);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[j][i], 0, 0, 0);
    }
  }
  __syncthreads();

  // ---- epilogue: accumulators -> bf16 tile in LDS ([pixel][channel], rows padded by 16 B) -> HBM
  // tile row = wm*64 + ty*8 + tx of 8x8 tile wm
  constexpr int CS = 64 * 2 + 16;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pix = wm * 64 + i * 16 + fr;  // = wm*64 + (2i + dy)*8 + dx
      const int ch = wn * 32 + j * 16 + fg * 4;
      const uint2 v = make_uint2(pack_bf2(acc[j][i][0], acc[j][i][1]), pack_bf2(acc[j][i][2], acc[j][i][3]));
      *reinterpret_cast<uint2*>(cv_smem + pix * CS + ch * 2) = v;
    }
  __syncthreads();
  const size_t org[2] = {((size_t)tn[0] * a.DH + th0[0]) * a.DW + tw0[0], ((size_t)tn[1] * a.DH + th0[1]) * a.DW + tw0[1]};
  if constexpr (BNB) {
    const int g = (int)(((long long)tn[0] * a.DH * a.DW) / a.stat_rpg);
    bnb_epilogue<64>(a, bnb, cv_smem, CS, reinterpret_cast<float*>(cv_smem + 128 * CS), g, 0,
                     bx - g * (a.stat_rpg >> 7), tid);
    return;
  }
  if constexpr (MODE == 0) {
    if (a.stat != nullptr) {
      // both 8x8 tiles lie in one statistics group (host-checked: an even number of tiles per group)
      const int g = (int)(((long long)tn[0] * a.DH * a.DW) / a.stat_rpg);
      float* slot = a.stat + ((size_t)(g * a.stat_nb + (bx - g * (a.stat_rpg >> 7))) * 2) * 64;
      store_tile_with_stats<64>(cv_smem, CS, a.dst, 64, 0, reinterpret_cast<float*>(cv_smem + 128 * CS), slot, tid,
                                [&](int row) -> size_t { return org[row >> 6] + (size_t)((row >> 3) & 7) * a.DW + (row & 7); });
      return;
    }
  }
#pragma unroll
  for (int q = 0; q < 128 * 8 / CV_THREADS; ++q) {
    const int p = tid + q * CV_THREADS;
    const int row = p >> 3, ch = p & 7;
    const size_t pix = org[row >> 6] + (size_t)((row >> 3) & 7) * a.DW + (row & 7);
    uint4 v = *reinterpret_cast<const uint4*>(cv_smem + row * CS + ch * 16);
    if (a.res != nullptr) {
      const uint4 r4 = *reinterpret_cast<const uint4*>(a.res + pix * 64 + ch * 8);
      const uint32_t vv[4] = {v.x, v.y, v.z, v.w}, rr[4] = {r4.x, r4.y, r4.z, r4.w};
      uint32_t o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[e] = pack_bf2(bf2f((uint16_t)(vv[e] & 0xffff)) + bf2f((uint16_t)(rr[e] & 0xffff)),
                        bf2f((uint16_t)(vv[e] >> 16)) + bf2f((uint16_t)(rr[e] >> 16)));
      v = make_uint4(o[0], o[1], o[2], o[3]);
    }
    *reinterpret_cast<uint4*>(a.dst + pix * 64 + ch * 8) = v;
  }
}

// ------------------------------------------------------------- the stem as a patch-resident kernel
// The stem (4x4 window over the 2x2 space-to-depth image, 16 channels, 64 outputs) through conv_igemm fetches, per
// 128 output pixels, four 16-KB pixel tiles (every source pixel once per kernel row and four times along a row) and
// the 32 KB of weights again: 96 KB of L2 -> LDS traffic and one cold start per 16 KB of output, 50 176 blocks of
// ~7 us each (525 us where the output alone costs ~200).  Here a block is PERSISTENT (four per CU): it keeps its
// share of the weights in REGISTERS (a wave's 32 outputs x 256 reduction elements = 16 fragments, 64 VGPRs) and
// walks over pairs of 8x8 output tiles; per pair it DMAs the two 11x11-pixel source patches (7.7 KB; the next pair's
// are in flight under the MFMAs and the epilogue of this one), and the sixteen taps only shift the fragment-read
// address inside the patch.
//   patch: [tile][row 0..10][col 0..11] x 32-byte pixels (col 11 is padding: a row pitch of 12 pixels = 384 B
//   puts the two pixel rows of a fragment read in different halves of the 256-byte bank line; the 8 pixels x 2
//   channel halves of a row are 256 contiguous bytes): conflict free without a swizzle.
// The MFMA sequence (kernel row, then the two 32-element halves of it) and the fragments are those of
// conv_igemm<128, 64, 2, 0>, so the results are bit-identical; the epilogue (bf16 tile through LDS, fused
// BatchNorm sums) is conv3x3_patch's.
constexpr int ST_PITCH = 12;
constexpr int ST_PIX = 11 * ST_PITCH;              // pixel slots per patch
constexpr int ST_PATCH_INSTR = 9;                  // 2 x 132 x 32 B = 8448 B -> nine 1-KB DMA instructions
constexpr int ST_PATCH_BYTES = ST_PATCH_INSTR * 1024;
constexpr int ST_CS = 64 * 2 + 16;                 // staged output row (64 channels bf16 + 16 B)
constexpr int ST_LDS = 2 * ST_PATCH_BYTES + 128 * ST_CS + 2 * 4 * 64 * 4;  // + scratch of the statistics flush

__global__ __launch_bounds__(CV_THREADS, 3) void conv_stem_patch(const ConvArgs a, int npairs, const WmDiv d_timg,
                                                                 const WmDiv d_twn) {
  extern __shared__ __attribute__((aligned(16))) uint8_t cv_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;  // 8x8 tile of the pair, 32-output half
  const uint32_t smem_base = lds_addr(cv_smem);
  uint8_t* stage = cv_smem + 2 * ST_PATCH_BYTES;
  const int fr = lane & 15, fg = lane >> 4;

  // ---- this wave's weights: outputs wn*32 + j*16 + fr, reduction chunk r*8 + ks*4 + fg (8 elements each)
  bf16x8_t wreg[2][4][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        wreg[j][r][ks] = *reinterpret_cast<const bf16x8_t*>(a.wt + (size_t)(wn * 32 + j * 16 + fr) * 256 + (r * 8 + ks * 4 + fg) * 8);

  // ---- patches of one tile pair: 9 instructions x 64 sixteen-byte pieces (piece = (pixel slot, channel half)).
  // The lane's slot decode (tile, row, column, half) does not depend on the pair: it is done once, here; per pair
  // only the two tile origins are decoded, from the wave-uniform pair index (the integer divisions by run-time
  // image sizes, repeated per piece and per stored chunk, were most of this kernel's instruction count).
  int pc_t[3], pc_py[3], pc_px[3], pc_half[3];
  bool pc_on[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int piece = 64 * (wave + 4 * i) + lane;
    const int gp = piece >> 1;
    pc_half[i] = piece & 1;
    pc_t[i] = gp >= ST_PIX ? 1 : 0;
    const int pp = gp - pc_t[i] * ST_PIX;
    pc_py[i] = pp / ST_PITCH;
    pc_px[i] = pp - pc_py[i] * ST_PITCH;
    pc_on[i] = gp < 2 * ST_PIX && pc_px[i] < 11;
  }
  auto tile_origin = [&](int T, int& n, int& h0, int& w0) {  // T is wave-uniform
    T = __builtin_amdgcn_readfirstlane(T);
    uint32_t tr, tw;
    n = (int)wm_divmod((uint32_t)T, d_timg, tr);
    h0 = (int)wm_divmod(tr, d_twn, tw) * 8;
    w0 = (int)tw * 8;
  };
  auto issue_patch = [&](int pair, uint32_t buf) {
    int n[2], h0[2], w0[2];
    tile_origin(pair * 2, n[0], h0[0], w0[0]);
    tile_origin(pair * 2 + 1, n[1], h0[1], w0[1]);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int j = wave + 4 * i;
      if (j < ST_PATCH_INSTR) {
        const int t = pc_t[i];
        const int h = (t ? h0[1] : h0[0]) - a.pad + pc_py[i], w = (t ? w0[1] : w0[0]) - a.pad + pc_px[i];
        const bool ok = pc_on[i] && (unsigned)h < (unsigned)a.SH && (unsigned)w < (unsigned)a.SW;
        const uint16_t* src = ok ? a.src + ((size_t)((t ? n[1] : n[0]) * a.SH + h) * a.SW + w) * 16 + pc_half[i] * 8
                                 : conv_zero_page + pc_half[i] * 8;
        glds16_at(src, buf + j * 1024);
      }
    }
  };

  const int dy = fr >> 3, dx = fr & 7;
  // pixel fragment: pixel (2 i + dy + r, dx + 2 ks + (fg >> 1)) of the patch, channel half fg & 1
  const uint32_t xbase = (uint32_t)((wm * ST_PIX + dy * ST_PITCH + dx + (fg >> 1)) * 32 + (fg & 1) * 16);

  // statistics: every thread owns the LDS slots (row slice, channel) of the two running sums over this block's pairs
  // (static schedule: pair = blockIdx.x + it * gridDim.x, so the order is fixed); ONE flush after the last pair: a launch
  // covers a single statistics group (the host launches once per group).  Flushing inside the loop when the group
  // changes spilled registers (168 VGPRs is the three-blocks-per-CU cap; the weights alone hold 64).
  float* st_red = reinterpret_cast<float*>(cv_smem + 2 * ST_PATCH_BYTES + 128 * ST_CS);  // [2][4][64]
  st_red[tid] = 0.f;
  st_red[256 + tid] = 0.f;

  int pair = blockIdx.x;
  if (pair < npairs) issue_patch(pair, smem_base);
  for (int it = 0; pair < npairs; pair += gridDim.x, ++it) {
    // the patches of this pair have landed (this wave's pieces; the barrier: everyone's), the previous pair's
    // staged tile has been read by every wave.  Vector-memory operations retire in issue order, so the counted
    // wait leaves the previous pair's 4 output stores per lane in flight: they were issued
    // AFTER these patch fetches (waiting for them too cost 0.5 us per pair and block).
    if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // (a statistics flush's atomics precede the stores: covered)
    wm_barrier();
    if (pair + (int)gridDim.x < npairs) issue_patch(pair + gridDim.x, smem_base + ((it + 1) & 1) * ST_PATCH_BYTES);
    const uint8_t* pb = cv_smem + (it & 1) * ST_PATCH_BYTES;

    f32x4_t acc[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t xf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          xf[i] = *reinterpret_cast<const bf16x8_t*>(pb + xbase + ((2 * i + r) * ST_PITCH + 2 * ks) * 32);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[j][r][ks], xf[i], acc[j][i], 0, 0, 0);
      }
    }

    // ---- epilogue: accumulators -> bf16 tile in LDS ([pixel][channel], rows padded by 16 B) -> HBM
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pix = wm * 64 + i * 16 + fr;  // = wm*64 + (2i + dy)*8 + dx
        const int ch = wn * 32 + j * 16 + fg * 4;
        const uint2 v = make_uint2(pack_bf2(acc[j][i][0], acc[j][i][1]), pack_bf2(acc[j][i][2], acc[j][i][3]));
        *reinterpret_cast<uint2*>(stage + pix * ST_CS + ch * 2) = v;
      }
    __syncthreads();
    const int T0 = pair * 2;
    if (a.stat != nullptr) {
      const int c = tid & 63, part = tid >> 6;
      float sm = 0.f, sq = 0.f;
      for (int rr = part * 32; rr < (part + 1) * 32; ++rr) {
        const float v = bf2f(*reinterpret_cast<const uint16_t*>(stage + rr * ST_CS + c * 2));
        sm += v;
        sq = fmaf(v, v, sq);
      }
      st_red[tid] += sm;        // slot (part, c) of the first sum: tid = part * 64 + c
      st_red[256 + tid] += sq;
    }
    {
      int n[2], h0[2], w0[2];
      tile_origin(T0, n[0], h0[0], w0[0]);
      tile_origin(T0 + 1, n[1], h0[1], w0[1]);
      const size_t org[2] = {((size_t)n[0] * a.DH + h0[0]) * a.DW + w0[0], ((size_t)n[1] * a.DH + h0[1]) * a.DW + w0[1]};
#pragma unroll
      for (int q = 0; q < 128 * 8 / CV_THREADS; ++q) {
        const int p = tid + q * CV_THREADS;
        const int row = p >> 3, ch = p & 7;
        const size_t pix = org[row >> 6] + (size_t)((row >> 3) & 7) * a.DW + (row & 7);
        *reinterpret_cast<uint4*>(a.dst + pix * 64 + ch * 8) = *reinterpret_cast<const uint4*>(stage + row * ST_CS + ch * 16);
      }
    }
  }
  if (a.stat != nullptr) {
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      const float t = st_red[(which * 4) * 64 + c] + st_red[(which * 4 + 1) * 64 + c] + st_red[(which * 4 + 2) * 64 + c] +
                      st_red[(which * 4 + 3) * 64 + c];
      a.stat[((size_t)blockIdx.x * 2 + which) * 64 + c] = t;  // slot = this block (group 0 of this launch)
    }
  }
}

// ------------------------------------------------------------------------------------ wgrad
struct WgradArgs {
  const uint16_t* dy;  // [M][K]
  const uint16_t* x;   // [N][H][W][C]
  float* dw;           // [nsplit][K][R][S][C] f32: split z STORES its partial sums into slab z (no atomics: the
                       // slabs are summed in a fixed order by wm_wgrad_fold / wm_wgrad_finalize -- bit-reproducible)
  int N, H, W, C, K, R, S, P, Q, stride, pad, M, chunks_per_split, total_chunks;
  // optional bias gradient dbias[K] += sum over pixels of dy: the blocks of the first column group
  // multiply their dY fragments with an all-ones B operand (one extra MFMA per fragment and k-step),
  // so a Linear layer's bias gradient costs no extra pass over dY
  float* dbias;        // [nsplit][K] slabs, like dw
  int xcd;             // 1: XCD-aware block order
};

constexpr int WG_PIX = 64;  // pixels per staged chunk (two MFMA k-steps)

// Tiles keep their natural [pixel][channel] layout (rows of 256 B for 128 channels, 128 B for 64)
// and are read with ds_read_b64_tr_b16.  A half-wave reads 8 consecutive rows x one 32-byte column
// block, so the 32-byte block index is XOR-swizzled by the row: block ^ (row & 7) in 256-byte rows,
// block ^ ((row >> 1) & 3) in 128-byte rows (two rows per 64-bank line) — conflict free, and
// compatible with the lane-linear global_load_lds destination (swizzle applied to the source).
template <int ROWB>
__device__ __forceinline__ int wg_swz(int row, int blk) {
  return ROWB == 256 ? (blk ^ (row & 7)) : (blk ^ ((row >> 1) & 3));
}

// A block owns BMO output channels x NT consecutive 64-column tiles of the (r,s,c) axis: the dY
// tile is fetched once per 64-pixel chunk and multiplied against NT gathered X tiles, so the MFMA
// work per barrier is NT x that of a single tile (layer1 and the stem have only 64 channels).
template <int BMO, int CPT, int NT, bool BIAS>
__global__ __launch_bounds__(CV_THREADS) void conv_wgrad(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t wg_smem[];
  constexpr int RA = BMO * 2;  // dY tile row bytes
  constexpr int RB = 128;      // X tile row bytes
  constexpr int A_BYTES = WG_PIX * RA;
  constexpr int X_BYTES = WG_PIX * RB;
  constexpr int STAGE = A_BYTES + NT * X_BYTES;
  constexpr int NA = A_BYTES / 4096;  // dY DMA instructions per wave (1 KiB each, 4 waves)
  constexpr int RPI_A = 1024 / RA;    // rows per dY instruction (4 or 8)
  // wave tiling.  SQ (BMO = 128, NT = 2): 2 x 2 waves, each 64 channels x ONE 64-column tile
  // (4 x 4 fragments: 16 transposed reads per 16 MFMAs).  Otherwise: BMO = 128 -> 4 waves x 32
  // channels x all NT tiles; BMO = 64 -> 2 x 2 waves of 32 channels x 32 columns of every tile.
  constexpr bool SQ = BMO == 128 && NT == 2;
  constexpr int MJ = SQ ? 4 : 2;               // 16-channel fragments per wave along K
  constexpr int NJ = BMO == 128 ? 4 : 2;       // 16-column fragments per wave per tile
  constexpr int WT = SQ ? 1 : NT;              // tiles a wave multiplies

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // logical block (column-tile group, output-channel tile, row split): the blocks of one row split read the same dY
  // and X rows -- keep them on one XCD (contiguous logical ranges per XCD: common.h wm_xcd_swizzle)
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (a.xcd) {
    const uint32_t gxy = gridDim.x * gridDim.y;
    const uint32_t lin = wm_xcd_swizzle(blockIdx.x + gridDim.x * blockIdx.y + gxy * blockIdx.z, gxy * gridDim.z);
    bz = (int)(lin / gxy);
    const uint32_t rem = lin - (uint32_t)bz * gxy;
    by = (int)(rem / gridDim.x);
    bx = (int)(rem - (uint32_t)by * gridDim.x);
  }
  const int ct0 = bx * NT;  // first 64-column tile
  const int k0 = by * BMO;
  const int cout_w = SQ ? (wave >> 1) * 64 : (BMO == 128 ? wave * 32 : (wave >> 1) * 32);
  const int col_w = BMO == 128 ? 0 : (wave & 1) * 32;
  const int tile_w = SQ ? (wave & 1) : 0;  // first tile this wave multiplies

  // ---- DMA lane geometry
  const int a_ro = RA == 256 ? (lane >> 4) : (lane >> 3);  // row inside a dY instruction
  const int a_pc = RA == 256 ? (lane & 15) : (lane & 7);   // physical 16-byte slot
  const int x_ro = lane >> 3, x_pc = lane & 7;             // X: 8 rows x 8 slots per instruction

  const int chunk_begin = bz * a.chunks_per_split;
  int chunk_end = chunk_begin + a.chunks_per_split;
  if (chunk_end > a.total_chunks) chunk_end = a.total_chunks;
  const int iters = chunk_end - chunk_begin;  // >= 1: the host derives the split count from chunks_per_split

  // running (n, p, q) of this lane's two X rows (shared by the NT tiles).  Every chunk advances a row
  // by 64 pixels: (dq, dpp, dn) is that step in mixed radix (Q, P), applied with two carries.
  int pn[2], pp[2], pq[2], xrow[2], xlc[2];
  const int pqn = a.P * a.Q;
  const int step_q = WG_PIX % a.Q, step_rows = WG_PIX / a.Q;
  const int step_p = step_rows % a.P, step_n = step_rows / a.P;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    xrow[i] = (i * 4 + wave) * 8 + x_ro;
    xlc[i] = (wg_swz<RB>(xrow[i], x_pc >> 1) << 1) | (x_pc & 1);  // logical 16-byte chunk
    const int m = chunk_begin * WG_PIX + xrow[i];
    pn[i] = m / pqn;
    const int rem = m - pn[i] * pqn;
    pp[i] = rem / a.Q;
    pq[i] = rem - pp[i] * a.Q;
  }
  // lane-constant part of the dY addresses (element offsets fit 32 bits: host-checked)
  uint32_t dy_off[NA];
  int dy_row[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = (i * 4 + wave) * RPI_A + a_ro;
    const int lc = (wg_swz<RA>(row, a_pc >> 1) << 1) | (a_pc & 1);
    dy_row[i] = row;
    dy_off[i] = (uint32_t)row * (uint32_t)a.K + (uint32_t)(k0 + lc * 8);
  }

  const uint32_t smem_base = lds_addr(wg_smem);
  auto issue = [&](int it, uint32_t stage) {
    const int pix0 = (chunk_begin + it) * WG_PIX;
    const uint32_t dy_base = (uint32_t)pix0 * (uint32_t)a.K;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const uint16_t* src = a.dy + (dy_base + dy_off[i]);
      if (pix0 + dy_row[i] >= a.M) src = conv_zero_page + (dy_off[i] & 63);  // any 16-byte aligned zeros
      glds16_at(src, stage + ((i * 4 + wave) * RPI_A) * RA);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool inb = pix0 + xrow[i] < a.M;
      const int hb = pp[i] * a.stride - a.pad, wb = pq[i] * a.stride - a.pad;
      const int lin = (pn[i] * a.H + hb) * a.W + wb;  // linear input pixel of tap (0, 0); may be "negative"
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int ct = ct0 + t;
        int r, s, coff;
        if constexpr (CPT == 8) {
          const int cpk = a.C >> 6;
          const int tap = ct / cpk;
          r = tap / a.S;
          s = tap - r * a.S;
          coff = (ct - tap * cpk) * 64 + xlc[i] * 8;
        } else {
          r = ct;
          s = xlc[i] >> 1;
          coff = (xlc[i] & 1) * 8;
        }
        const int sh = hb + r, sw = wb + s;
        const bool ok = inb & ((unsigned)sh < (unsigned)a.H) & ((unsigned)sw < (unsigned)a.W);
        const uint32_t off = (uint32_t)(lin + r * a.W + s) * (uint32_t)a.C + (uint32_t)coff;
        const uint16_t* src = a.x + off;
        if (!ok) src = conv_zero_page + xlc[i] * 8;
        glds16_at(src, stage + A_BYTES + t * X_BYTES + ((i * 4 + wave) * 8) * RB);
      }
      // advance this row by one chunk (64 pixels)
      pq[i] += step_q;
      int carry = pq[i] >= a.Q;
      pq[i] -= carry ? a.Q : 0;
      pp[i] += step_p + carry;
      carry = pp[i] >= a.P;
      pp[i] -= carry ? a.P : 0;
      pn[i] += step_n + carry;
    }
  };

  f32x4_t acc[MJ][WT * NJ];
#pragma unroll
  for (int i = 0; i < MJ; ++i)
#pragma unroll
    for (int j = 0; j < WT * NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // bias gradient: only column group 0, and only one of the waves that share a channel range
  // (BIAS is a template flag: carrying the extra accumulators in the convolution instantiations
  // cost them ~12 %)
  const bool bias_wave = BIAS && bx == 0 && (SQ ? (wave & 1) == 0 : (BMO == 128 || (wave & 1) == 0));
  f32x4_t bacc[MJ];
#pragma unroll
  for (int i = 0; i < MJ; ++i) bacc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const s16x8_t ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);

  // transposed-read geometry: 16-lane group g, lane (q, p) inside it; MFMA k-slot (g, e) holds
  // pixel 4g + e (e < 4) or 16 + 4g + (e - 4) of the 32-pixel k-step — same map for both operands.
  const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  auto tr_read = [&](const uint8_t* p) -> s16x4_t {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
  };
  auto compute = [&](const uint8_t* buf) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r0 = ks * 32 + 4 * tg + tq, r1 = r0 + 16;
      bf16x8_t af[MJ];
#pragma unroll
      for (int i = 0; i < MJ; ++i) {
        const int blk = (cout_w + i * 16) >> 4;
        const s16x4_t lo = tr_read(buf + r0 * RA + wg_swz<RA>(r0, blk) * 32 + 8 * tp);
        const s16x4_t hi = tr_read(buf + r1 * RA + wg_swz<RA>(r1, blk) * 32 + 8 * tp);
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        af[i] = __builtin_bit_cast(bf16x8_t, v);
      }
      if (BIAS && bias_wave) {
#pragma unroll
        for (int i = 0; i < MJ; ++i) bacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, bacc[i], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < WT; ++t) {
        const uint8_t* xb = buf + A_BYTES + (tile_w + t) * X_BYTES;
        bf16x8_t bfr[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int blk = (col_w + j * 16) >> 4;
          const s16x4_t lo = tr_read(xb + r0 * RB + wg_swz<RB>(r0, blk) * 32 + 8 * tp);
          const s16x4_t hi = tr_read(xb + r1 * RB + wg_swz<RB>(r1, blk) * 32 + 8 * tp);
          const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          bfr[j] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int i = 0; i < MJ; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][t * NJ + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][t * NJ + j], 0, 0, 0);
      }
    }
  };

  issue(0, smem_base);
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wm_barrier();
    if (it + 1 < iters && !(WM_CONV_ABLATE & 1)) issue(it + 1, smem_base + ((it + 1) & 1) * STAGE);
    else wm_barrier();  // (as in conv_igemm: a phase between the retiring wait and the reads when nothing is issued)
    if (!(WM_CONV_ABLATE & 2)) compute(wg_smem + (it & 1) * STAGE);
  }

  const size_t rsc = (size_t)a.R * a.S * a.C;
  const int fr = lane & 15, fg = lane >> 4;
  float* slab = a.dw + (size_t)bz * (size_t)a.K * rsc;
#pragma unroll
  for (int i = 0; i < MJ; ++i)
#pragma unroll
    for (int t = 0; t < WT; ++t)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kk = k0 + cout_w + i * 16 + fg * 4 + e;
          const int col = (ct0 + tile_w + t) * 64 + col_w + j * 16 + fr;
          __builtin_nontemporal_store(acc[i][t * NJ + j][e], &slab[(size_t)kk * rsc + col]);  // (read again only by the fold at the end of the pass)
        }
  if (BIAS && bias_wave && fr == 0) {  // every column of bacc holds the same sums: lane column 0 reports
#pragma unroll
    for (int i = 0; i < MJ; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) a.dbias[(size_t)bz * a.K + k0 + cout_w + i * 16 + fg * 4 + e] = bacc[i][e];
  }
}


// ------------------------------------------------------------- weight gradient of the 64 -> 64 channel 3x3 layers
// conv_wgrad<64, 8, 3> on layer1 is bound by its tile fetches (ablation builds: fetches alone 157 of 170 us, at the
// chip's L2 -> LDS rate): three blocks per 64-pixel chunk each fetch the dY tile and gather three 8-KB X tiles, one per
// tap -- 96 KB per chunk, the input re-read nine times through L2.  Here a chunk is an 8 x 8 output tile, its 10 x 10
// input patch is fetched ONCE (12.8 KB) and the nine taps shift the transposed-read address inside the patch; one block
// multiplies all nine taps (the dY fragments are read once per k-step for 36 MFMAs): 20.8 KB per chunk, 72 MFMAs per
// barrier instead of 24.  Layout: dY tile [ty * 8 + tx][64 k] in 128-byte rows, 32-byte blocks XOR-swizzled as in
// conv_wgrad; patch [py * 10 + px] in 160-BYTE pixel slots (128 B of channels + 32 B never read): a half-wave's
// transposed read takes one 32-byte block of eight consecutive pixels, 160 k + 32 b bytes -> banks 40 k + 8 b mod 64,
// eight disjoint groups of eight: conflict free WITHOUT a swizzle, so a tap is a constant byte offset (an immediate of
// the ds_read) from one per-lane base address.  (First build: 128-byte slots with the XOR swizzle keyed on the patch
// pixel -- 72 distinct read addresses per lane, recomputed every chunk for want of registers: 250 us against 175.)
// MFMA k-slot <-> pixel map that of conv_wgrad.  Results: per-split partial sums in slabs as before (the pixel order
// inside a split differs from conv_wgrad's, so the sums agree to rounding, not bit for bit).
constexpr int WP_PITCH = 10;
constexpr int WP_XSLOT = 160;   // bytes per patch pixel
constexpr int WP_X_INSTR = 16;  // 100 pixels x 10 sixteen-byte slots = 1 000 lanes -> 16 x 64
constexpr int WP_A_BYTES = 64 * 128;
constexpr int WP_STAGE = WP_A_BYTES + WP_X_INSTR * 1024;
constexpr int WP_LDS = 2 * WP_STAGE;

__global__ __launch_bounds__(CV_THREADS, 2) void conv_wgrad_patch64(const WgradArgs a, const WmDiv d_tpi, const WmDiv d_tw) {
  extern __shared__ __attribute__((aligned(16))) uint8_t wg_smem[];
  constexpr int RA = 128, RB = 128, MJ = 2, NJ = 2, NT = 9;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bz = blockIdx.x;
  if (a.xcd) bz = (int)wm_xcd_swizzle(blockIdx.x, gridDim.x);  // neighbouring tiles (shared halo rows) on one XCD
  const int cout_w = (wave >> 1) * 32, col_w = (wave & 1) * 32;  // 2 x 2 waves: 32 output channels x 32 columns of every tap

  const int chunk_begin = bz * a.chunks_per_split;
  int chunk_end = chunk_begin + a.chunks_per_split;
  if (chunk_end > a.total_chunks) chunk_end = a.total_chunks;
  const int iters = chunk_end - chunk_begin;

  // ---- DMA lane geometry.  dY: instruction i covers tile row ty = i * 4 + wave (8 pixels x 128 B)
  const int a_ro = lane >> 3, a_pc = lane & 7;
  uint32_t a_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ty = i * 4 + wave, row = ty * 8 + a_ro;
    const int lc = (wg_swz<RA>(row, a_pc >> 1) << 1) | (a_pc & 1);
    a_off[i] = (uint32_t)((ty * a.Q + a_ro) * a.K + lc * 8);  // relative to the tile's first pixel
  }
  // patch: instruction j = wave + 4 q fills the sixteen-byte slots 64 j .. 64 j + 63; slot = pixel * 10 + piece,
  // pieces 8 and 9 of a pixel (and the slots past pixel 99) are padding
  int x_off[4], x_py[4], x_px[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int slot = (wave + 4 * q) * 64 + lane;
    const int pidx = slot / 10, piece = slot - pidx * 10;
    const int py = pidx / WP_PITCH, px = pidx - py * WP_PITCH;
    x_py[q] = (pidx < WP_PITCH * WP_PITCH && piece < 8) ? py : -100;  // (padding: never in range -> zeros)
    x_px[q] = px;
    x_off[q] = ((py - 1) * a.W + (px - 1)) * a.C + (piece & 7) * 8;
  }

  const uint32_t smem_base = lds_addr(wg_smem);
  auto issue = [&](int it, uint32_t stage) {
    const uint32_t c = (uint32_t)(chunk_begin + it);
    uint32_t t, tw;
    const int n = (int)wm_divmod(c, d_tpi, t);
    const int th = (int)wm_divmod(t, d_tw, tw);
    const int h0 = th * 8, w0 = (int)tw * 8;
    const uint32_t org = (uint32_t)((n * a.H + h0) * a.W + w0);  // (P = H, Q = W; element offsets fit 32 bits: host-checked)
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16_at(a.dy + (org * (uint32_t)a.K + a_off[i]), stage + ((i * 4 + wave) * 8) * RA);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = wave + 4 * q;
      const int sh = h0 - 1 + x_py[q], sw = w0 - 1 + x_px[q];
      const bool ok = (unsigned)sh < (unsigned)a.H && (unsigned)sw < (unsigned)a.W;
      const uint16_t* src = ok ? a.x + (long long)org * a.C + x_off[q] : conv_zero_page;
      glds16_at(src, stage + WP_A_BYTES + j * 1024);
    }
  };

  f32x4_t acc[MJ][NT * NJ];
#pragma unroll
  for (int i = 0; i < MJ; ++i)
#pragma unroll
    for (int j = 0; j < NT * NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  auto tr_read = [&](const uint8_t* p) -> s16x4_t {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
  };
  // pixel 4 tg + tq of the tile = (tg >> 1, 4 (tg & 1) + tq); column block col_w / 16; 8 bytes per tp
  const int x_lane = ((tg >> 1) * WP_PITCH + 4 * (tg & 1) + tq) * WP_XSLOT + (col_w >> 4) * 32 + 8 * tp;
  auto compute = [&](const uint8_t* buf) {
    const uint8_t* xb = buf + WP_A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r0 = ks * 32 + 4 * tg + tq, r1 = r0 + 16;  // pixels of the tile: (r >> 3, r & 7)
      bf16x8_t af[MJ];
#pragma unroll
      for (int i = 0; i < MJ; ++i) {
        const int blk = (cout_w + i * 16) >> 4;
        const s16x4_t lo = tr_read(buf + r0 * RA + wg_swz<RA>(r0, blk) * 32 + 8 * tp);
        const s16x4_t hi = tr_read(buf + r1 * RA + wg_swz<RA>(r1, blk) * 32 + 8 * tp);
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        af[i] = __builtin_bit_cast(bf16x8_t, v);
      }
      // this lane's patch address for tap (0, 0), k-step 0, column block 0; everything else is a constant offset
      const uint8_t* xl = xb + x_lane + ks * (4 * WP_PITCH * WP_XSLOT);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int toff = ((t / 3) * WP_PITCH + (t % 3)) * WP_XSLOT;
        bf16x8_t bfr[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const s16x4_t lo = tr_read(xl + toff + j * 32);
          const s16x4_t hi = tr_read(xl + toff + j * 32 + 2 * WP_PITCH * WP_XSLOT);
          const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          bfr[j] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int i = 0; i < MJ; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][t * NJ + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][t * NJ + j], 0, 0, 0);
      }
    }
  };

  issue(0, smem_base);
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wm_barrier();
    if (it + 1 < iters) issue(it + 1, smem_base + ((it + 1) & 1) * WP_STAGE);
    else wm_barrier();  // (as in conv_igemm: a phase between the retiring wait and the reads when nothing is issued)
    compute(wg_smem + (it & 1) * WP_STAGE);
  }

  const size_t rsc = (size_t)9 * a.C;
  const int fr = lane & 15, fg = lane >> 4;
  float* slab = a.dw + (size_t)bz * (size_t)a.K * rsc;
#pragma unroll
  for (int i = 0; i < MJ; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kk = cout_w + i * 16 + fg * 4 + e;
          const int col = t * 64 + col_w + j * 16 + fr;
          __builtin_nontemporal_store(acc[i][t * NJ + j][e], &slab[(size_t)kk * rsc + col]);  // (read again only by the fold at the end of the pass)
        }
}

template <typename K>
int set_lds(K kernel, int bytes) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  return e == hipSuccess ? WM_OK : (int)e;
}

template <int BM, int BN, int CPT, int MODE, bool EPI = false, bool BNB = false>
int launch_igemm(const ConvArgs& a, hipStream_t st) {
  // two operand stages; the epilogue reuses them: staged tile BM x (BN * 2 + 16) B + the forward-statistics scratch
  // (2 x 256 floats) or the 16 KB of cross-thread sums of the BatchNorm-backward epilogue -- always smaller
  constexpr int lds = 2 * (BM * CV_ROW + BN * CV_ROW);
  static_assert(BM * (BN * 2 + 16) + 2 * CV_THREADS * 4 <= lds, "epilogue LDS");
  static bool attr = false;
  if (!attr) {
    const int rc = set_lds(&conv_igemm<BM, BN, CPT, MODE, EPI, BNB>, lds);
    if (rc != WM_OK) return rc;
    attr = true;
  }
  dim3 grid(wm_cdiv(a.M, BM), a.DC / BN);
  conv_igemm<BM, BN, CPT, MODE, EPI, BNB><<<grid, CV_THREADS, lds, st>>>(a);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

// 3x3 / stride 1 / pad 1, 64 -> 64 channels, image sides multiples of 8, an even number of 8x8 tiles
// (and of tiles per statistics group): the patch-resident kernel.  WM_CONV_PATCH=0 keeps conv_igemm.
inline bool conv_patch_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("WM_CONV_PATCH");
    v = e ? atoi(e) != 0 : 1;
  }
  return v != 0;
}

inline bool conv_patch_ok(const ConvArgs& a) {
  if (!conv_patch_enabled()) return false;
  if (a.SC != 64 || a.DC != 64 || a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1) return false;
  if (a.SH != a.DH || a.SW != a.DW || (a.DH & 7) || (a.DW & 7) || a.bias != nullptr) return false;
  const long long tiles = (long long)a.N * (a.DH >> 3) * (a.DW >> 3);
  if (tiles & 1) return false;
  if (a.stat != nullptr) {
    if (a.stat_rpg % 128 != 0) return false;           // tiles per group even
    if (a.stat_rpg % (a.DH * a.DW) != 0) return false;  // groups are whole images
  }
  return true;
}

// The space-to-depth stem in its patch-resident form: 4x4 window, stride 1, 16 source channels, 64 outputs, output
// the size of the source, sides multiples of 8, an even number of 8x8 tiles (per statistics group too).
inline bool stem_patch_ok(const ConvArgs& a) {
  if (!conv_patch_enabled()) return false;
  if (a.SC != 16 || a.DC != 64 || a.R != 4 || a.S != 4 || a.stride != 1 || a.pad < 0 || a.pad > 3) return false;
  if (a.SH != a.DH || a.SW != a.DW || (a.DH & 7) || (a.DW & 7) || a.bias != nullptr || a.res != nullptr) return false;
  const long long tiles = (long long)a.N * (a.DH >> 3) * (a.DW >> 3);
  if (tiles & 1) return false;
  if (a.stat != nullptr) {
    if (a.stat_rpg % 128 != 0) return false;
    if (a.stat_rpg % (a.DH * a.DW) != 0) return false;
  }
  return true;
}

template <int MODE, bool BNB = false>
int launch_patch(const ConvArgs& a, hipStream_t st) {
  static_assert(128 * (64 * 2 + 16) + 2 * CV_THREADS * 4 <= PT_LDS, "epilogue LDS");
  static bool attr = false;
  if (!attr) {
    const int rc = set_lds(&conv3x3_patch<MODE, BNB>, PT_LDS);
    if (rc != WM_OK) return rc;
    attr = true;
  }
  const int blocks = (int)((long long)a.N * (a.DH >> 3) * (a.DW >> 3) / 2);
  static int lds_pad = -1;  // WM_PATCH_LDS_PAD: extra dynamic LDS per block (experiment: 12288 -> two blocks per CU)
  if (lds_pad < 0) {
    const char* e = getenv("WM_PATCH_LDS_PAD");
    lds_pad = e ? atoi(e) : 0;
    if (lds_pad > 0 && set_lds(&conv3x3_patch<MODE, BNB>, PT_LDS + lds_pad) != WM_OK) lds_pad = 0;
  }
  conv3x3_patch<MODE, BNB><<<blocks, CV_THREADS, PT_LDS + lds_pad, st>>>(a);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

// split-K plan of a weight-gradient launch: enough blocks to fill the chip (2 resident per CU), no more -- every split
// writes (and the fold reads) one K x R x S x C slab
inline void wgrad_plan(const WgradArgs& a, int BMO, int NT, int& nsplit, int& chunks_per_split, int& total_chunks) {
  const int colgroups = a.R * a.S * a.C / 64 / NT;
  const int ktiles = a.K / BMO;
  total_chunks = wm_cdiv((long long)a.N * a.P * a.Q, WG_PIX);
  static int target = 0;
  if (target == 0) {
    const char* e = getenv("WM_WGRAD_BLOCKS");
    target = e ? atoi(e) : 512;
  }
  // Linear layers (1 x 1 on a 1 x 1 image: the transformer GEMMs): the output tile is small and every split ends in
  // a K x C f32 slab (round 2, with atomics: half the blocks, 3.06 vs 3.57 ms of wgrad per DINO ViT-Tiny step).
  // Re-measured with slabs (round 3, ms per step at 256 / 512 blocks): DINO ViT-Tiny 9.79 / 9.53, DINO ViT-S 19.68 / 18.62
  // (39 424 token rows: the slab traffic is small beside the operands), MAE ViT-S/16 4.64 / 4.74 (3 200 and 12 608 rows:
  // there every extra split is mostly slab bytes).  So: the full target from 16 384 rows on, half of it below.
  static int target_lin = -1;
  if (target_lin < 0) {
    const char* e = getenv("WM_WGRAD_BLOCKS_LINEAR");
    target_lin = e ? atoi(e) : 0;
  }
  int tgt = target;
  if (a.R * a.S == 1 && a.H * a.W == 1 && target == 512) tgt = target_lin > 0 ? target_lin : (total_chunks >= 256 ? 512 : 256);
  if (NT == 9) {  // conv_wgrad_patch64: one block owns the whole K x R x S x C gradient, every split costs a full slab
    // (kernel alone 112 us with 512 blocks, 130 with 256 -- but the fold then reads 150 instead of 300 MB of slabs
    // for the four layer1 convolutions: SimCLR step 11.32 ms with 256 or 384 blocks, 11.38 with 512, 11.51 without this
    // kernel, one box)
    const char* e = getenv("WM_WGRAD_PATCH_BLOCKS");  // (read per call)
    tgt = e ? atoi(e) : 256;
  }
  nsplit = tgt / (colgroups * ktiles);
  if (nsplit < 1) nsplit = 1;
  if (nsplit > total_chunks) nsplit = total_chunks;
  chunks_per_split = wm_cdiv(total_chunks, nsplit);
  nsplit = wm_cdiv(total_chunks, chunks_per_split);
}

template <int BMO, int CPT, int NT, bool BIAS>
int launch_wgrad_impl(WgradArgs a, hipStream_t st) {
  constexpr int lds = 2 * (WG_PIX * BMO * 2 + NT * WG_PIX * 128);
  static bool attr = false;
  if (!attr) {
    const int rc = set_lds(&conv_wgrad<BMO, CPT, NT, BIAS>, lds);
    if (rc != WM_OK) return rc;
    attr = true;
  }
  int nsplit;
  wgrad_plan(a, BMO, NT, nsplit, a.chunks_per_split, a.total_chunks);
  dim3 grid(a.R * a.S * a.C / 64 / NT, a.K / BMO, nsplit);
  conv_wgrad<BMO, CPT, NT, BIAS><<<grid, CV_THREADS, lds, st>>>(a);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

template <int BMO, int CPT, int NT>
int launch_wgrad(const WgradArgs& a, hipStream_t st) {
  if (a.dbias == nullptr) return launch_wgrad_impl<BMO, CPT, NT, false>(a, st);
  if constexpr (CPT == 8) return launch_wgrad_impl<BMO, CPT, NT, true>(a, st);
  return WM_EUNSUPPORTED;  // no bias gradient on the space-to-depth stem form
}

// the patch-resident form (conv_wgrad_patch64): WM_WGRAD_PATCH=0 keeps conv_wgrad<64, 8, 3>
inline bool wgrad_patch_ok(const WgradArgs& a) {
  const char* e = getenv("WM_WGRAD_PATCH");  // (read per call: A/B switch)
  if (e != nullptr && atoi(e) == 0) return false;
  return a.C == 64 && a.K == 64 && a.R == 3 && a.S == 3 && a.stride == 1 && a.pad == 1 && a.P == a.H && a.Q == a.W &&
         (a.H & 7) == 0 && (a.W & 7) == 0 && a.dbias == nullptr;
}
inline int launch_wgrad_patch(WgradArgs a, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    const int rc = set_lds(&conv_wgrad_patch64, WP_LDS);
    if (rc != WM_OK) return rc;
    attr = true;
  }
  int nsplit;
  wgrad_plan(a, 64, 9, nsplit, a.chunks_per_split, a.total_chunks);  // one block per split: all nine taps, all 64 channels
  const int tiles_w = a.W >> 3, tpi = (a.H >> 3) * tiles_w;
  conv_wgrad_patch64<<<nsplit, CV_THREADS, WP_LDS, st>>>(a, wm_div_make((uint32_t)tpi), wm_div_make((uint32_t)tiles_w));
  WM_LAUNCH_CHECK();
  return WM_OK;
}

// tile configuration of a weight-gradient shape (one place: the launch and wm_conv2d_wgrad_splits must agree)
inline void wgrad_config(int C, int K, int R, int S, int& bmo, int& cpt, int& nt) {
  const int coltiles = R * S * C / 64;
  if (C == 16) {  // stem: the 4 kernel rows together
    cpt = 2;
    if (K % 128 == 0) { bmo = 128; nt = coltiles % 2 == 0 ? 2 : 1; }
    else { bmo = 64; nt = coltiles % 4 == 0 ? 4 : 1; }
    return;
  }
  cpt = 8;
  if (K % 128 == 0) {
    bmo = 128;
    static int force_nt = -1;  // WM_WGRAD_NT: experiment switch (1, 2 or 3 column tiles per block)
    if (force_nt < 0) {
      const char* e = getenv("WM_WGRAD_NT");
      force_nt = e ? atoi(e) : 0;
    }
    if (force_nt == 3 && coltiles % 3 == 0) { nt = 3; return; }
    if (force_nt == 2 && coltiles % 2 == 0) { nt = 2; return; }
    if (force_nt == 1) { nt = 1; return; }
    // column tiles per block, measured per ResNet-18 shape at batch 512 (profiles/r01_conv_layers_v3.txt):
    // three taps per block where the dY tile is the larger share of the traffic (C <= 128 with K = 128)
    // and for the 512-channel layers, two for 256 channels, one for the stride-2 128 -> 256 layer
    nt = (coltiles % 2 == 0 && C > 128) ? 2 : 1;
    // 3x3: three taps per block everywhere.  (Round 1 measured one tap per block best for the stride-2 128 -> 256 layer
    // and two for 256 channels -- with f32 atomics and hardware block order.  With slabs and the blocks of a row split on
    // one XCD, tools/bench_conv.py at batch 512: layer3.0 105.5 -> 73.0 us, layer3 3x3 145.7 -> 132.1 us.)
    if (R * S == 9) nt = 3;
    if (nt == 3 && coltiles % 3 != 0) nt = 1;
    if (nt == 2 && coltiles % 2 != 0) nt = 1;
    return;
  }
  bmo = 64;
  nt = coltiles % 3 == 0 ? 3 : 1;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline int xcd_order() {  // WM_XCD_SWIZZLE=0: hardware block order (A/B switch; read per call)
  const char* e = getenv("WM_XCD_SWIZZLE");
  return e != nullptr ? atoi(e) : 1;
}

}  // namespace

// Geometry checks shared by the three entry points.  "x" is always the forward input
// [N][H][W][C], "y" the forward output [N][P][Q][K].
static int conv_check(int N, int H, int W, int C, int K, int R, int S, int P, int Q, int stride, int pad) {
  WM_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && R > 0 && S > 0 && P > 0 && Q > 0, WM_EINVAL);
  WM_REQUIRE(stride == 1 || stride == 2, WM_EUNSUPPORTED);
  WM_REQUIRE(pad >= 0 && pad <= R, WM_EUNSUPPORTED);
  WM_REQUIRE(K % 64 == 0, WM_EUNSUPPORTED);
  WM_REQUIRE(C % 64 == 0 || (C == 16 && S == 4), WM_EUNSUPPORTED);
  // every output pixel's window must start no later than the input's far edge
  WM_REQUIRE((long long)(P - 1) * stride - pad < H && (long long)(Q - 1) * stride - pad < W, WM_EINVAL);
  WM_REQUIRE((long long)N * P * Q < (1ll << 31) && (long long)N * H * W * C < (1ll << 40), WM_EUNSUPPORTED);
  WM_REQUIRE((long long)N * H * W < (1ll << 31), WM_EUNSUPPORTED);
  return WM_OK;
}

static int conv_fwd_impl(const void* x, const void* w_krsc, void* y, int N, int H, int W, int C, int K, int R,
                         int S, int P, int Q, int stride, int pad, float* stat, int stat_nb, int stat_rpg,
                         void* stream, const float* bias = nullptr, const void* residual = nullptr,
                         void* pre_out = nullptr);

extern "C" int wm_conv2d_fwd(const void* x, const void* w_krsc, void* y, int N, int H, int W, int C,
                             int K, int R, int S, int P, int Q, int stride, int pad, void* stream) {
  return conv_fwd_impl(x, w_krsc, y, N, H, W, C, K, R, S, P, Q, stride, pad, nullptr, 0, 0, stream);
}

extern "C" int wm_conv2d_fwd_bias_res(const void* x, const void* w_krsc, const float* bias, const void* residual,
                                      void* y, int N, int H, int W, int C, int K, int R, int S, int P, int Q,
                                      int stride, int pad, void* stream) {
  WM_REQUIRE((reinterpret_cast<uintptr_t>(bias) & 15) == 0 && (reinterpret_cast<uintptr_t>(residual) & 15) == 0, WM_EALIGN);
  return conv_fwd_impl(x, w_krsc, y, N, H, W, C, K, R, S, P, Q, stride, pad, nullptr, 0, 0, stream, bias, residual);
}

// Statistics slots per group a forward-with-statistics launch of this geometry writes: one per 128-row tile, except
// the persistent stem kernel, whose blocks accumulate over their tiles and write one slot each.
static int stem_patch_slots();
extern "C" int wm_conv2d_fwd_stats_tiles(int N, int H, int W, int C, int K, int R, int S, int P, int Q, int stride,
                                         int pad, int rows_per_group) {
  if (conv_check(N, H, W, C, K, R, S, P, Q, stride, pad) != WM_OK || rows_per_group <= 0) return WM_EINVAL;
  const long long M = (long long)N * P * Q;
  if (rows_per_group % 128 != 0 || M % rows_per_group != 0) return WM_EUNSUPPORTED;
  if (C == 16) {
    ConvArgs a{};
    a.N = N; a.SH = H; a.SW = W; a.SC = C; a.DH = P; a.DW = Q; a.DC = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad;
    a.M = (int)M; a.stat = reinterpret_cast<float*>(1); a.stat_rpg = rows_per_group; a.bias = nullptr; a.res = nullptr;
    if (stem_patch_ok(a)) {
      const int pairs_g = rows_per_group / 128;
      const int slots = stem_patch_slots();
      return pairs_g < slots ? pairs_g : slots;
    }
  }
  return rows_per_group / 128;
}

extern "C" int wm_conv2d_fwd_stats(const void* x, const void* w_krsc, void* y, int N, int H, int W, int C,
                                   int K, int R, int S, int P, int Q, int stride, int pad, float* stat_part,
                                   int stat_tiles, int rows_per_group, void* stream) {
  WM_REQUIRE(stat_part && rows_per_group > 0, WM_EINVAL);
  WM_REQUIRE(rows_per_group % 128 == 0 && ((long long)N * P * Q) % rows_per_group == 0, WM_EUNSUPPORTED);
  WM_REQUIRE(stat_tiles == wm_conv2d_fwd_stats_tiles(N, H, W, C, K, R, S, P, Q, stride, pad, rows_per_group), WM_EINVAL);
  return conv_fwd_impl(x, w_krsc, y, N, H, W, C, K, R, S, P, Q, stride, pad, stat_part, stat_tiles, rows_per_group, stream);
}

// resident blocks of the persistent stem kernel: 38 KB of LDS, 168 registers per lane -> three per CU
static int stem_patch_slots() {
  static int slots = 0;
  if (slots == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    const char* e = getenv("WM_STEM_BLOCKS_PER_CU");
    slots = cus * (e ? atoi(e) : 3);
  }
  return slots;
}

static int conv_fwd_impl(const void* x, const void* w_krsc, void* y, int N, int H, int W, int C, int K, int R,
                         int S, int P, int Q, int stride, int pad, float* stat, int stat_nb, int stat_rpg,
                         void* stream, const float* bias, const void* residual, void* pre_out) {
  WM_REQUIRE(pre_out == nullptr || (residual == nullptr && R == 1 && S == 1), WM_EUNSUPPORTED);
  WM_REQUIRE(x && w_krsc && y, WM_EINVAL);
  const int rc = conv_check(N, H, W, C, K, R, S, P, Q, stride, pad);
  if (rc != WM_OK) return rc;
  WM_REQUIRE(aligned16(x) && aligned16(w_krsc) && aligned16(y), WM_EALIGN);
  ConvArgs a;
  a.xcd = xcd_order();
  a.src = static_cast<const uint16_t*>(x);
  a.wt = static_cast<const uint16_t*>(w_krsc);
  a.dst = static_cast<uint16_t*>(y);
  a.N = N; a.SH = H; a.SW = W; a.SC = C; a.DH = P; a.DW = Q; a.DC = K;
  a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.M = N * P * Q;
  a.d_dhw = wm_div_make((uint32_t)(P * Q)); a.d_dw = wm_div_make((uint32_t)Q);
  a.d_h2w2 = wm_div_make(1); a.d_w2 = wm_div_make(1);
  a.stat = stat; a.stat_nb = stat_nb; a.stat_rpg = stat_rpg;
  a.res = static_cast<const uint16_t*>(residual);
  a.bias = bias;
  a.pre_in = nullptr;
  a.pre_out = static_cast<uint16_t*>(pre_out);
  a.act = pre_out != nullptr ? 1 : 0;
  a.bn_y = a.bn_x = nullptr;
  a.bn_mask = nullptr;
  a.bn_mean = a.bn_invstd = a.bn_gamma = a.bn_beta = nullptr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (C == 16) {
    WM_REQUIRE(bias == nullptr && residual == nullptr, WM_EUNSUPPORTED);
    a.nkt = R;
    if (stem_patch_ok(a)) {
      const int npairs = (int)((long long)a.N * (a.DH >> 3) * (a.DW >> 3) / 2);
      static bool attr = false;
      if (!attr) {
        const int rc2 = set_lds(&conv_stem_patch, ST_LDS);
        if (rc2 != WM_OK) return rc2;
        attr = true;
      }
      const int slots = stem_patch_slots();
      // one launch per statistics group (whole images, an even number of tiles: stem_patch_ok): a block then flushes
      // its running sums once, after its last tile pair
      const int groups = a.stat != nullptr ? a.M / a.stat_rpg : 1;
      const int n_g = a.N / groups;
      const int pairs_g = npairs / groups;
      for (int g = 0; g < groups; ++g) {
        ConvArgs ag = a;
        ag.N = n_g;
        ag.M = n_g * a.DH * a.DW;
        ag.src = a.src + (size_t)g * n_g * a.SH * a.SW * 16;
        ag.dst = a.dst + (size_t)g * n_g * a.DH * a.DW * 64;
        if (a.stat != nullptr) ag.stat = a.stat + (size_t)g * a.stat_nb * 2 * 64;  // (stat_nb = blocks of a launch)
        conv_stem_patch<<<pairs_g < slots ? pairs_g : slots, CV_THREADS, ST_LDS, st>>>(
            ag, pairs_g, wm_div_make((uint32_t)((a.DH >> 3) * (a.DW >> 3))), wm_div_make((uint32_t)(a.DW >> 3)));
        WM_LAUNCH_CHECK();
      }
      return WM_OK;
    }
    // (256-pixel tiles for the 64-channel stem are 7 % faster alone but 10 % slower inside the training
    // step, where the epilogue also accumulates the BatchNorm statistics: 564 vs 509 us)
    return K % 128 == 0 ? launch_igemm<128, 128, 2, 0>(a, st) : launch_igemm<128, 64, 2, 0>(a, st);
  }
  a.nkt = R * S * (C / 64);
  if (R == 1 && S == 1 && stride == 1 && pad == 0 && stat == nullptr && wm_panel_ok(a.M, C, K, residual != nullptr)) {
    // Linear with a 192-wide input (ViT-Tiny): token rows resident in registers, weight tiles streamed (panel.hip)
    WmPanelArgs pa{a.src, a.wt, bias, a.res, nullptr, a.pre_out, a.dst, a.M, K, a.act, 0, nullptr, nullptr, 0.f, 0};
    return wm_panel_launch(pa, st);
  }
  if (residual == nullptr && conv_patch_ok(a)) return launch_patch<0>(a, st);
  if (bias != nullptr || residual != nullptr || pre_out != nullptr)
    return K % 128 == 0 ? launch_igemm<128, 128, 8, 0, true>(a, st) : launch_igemm<128, 64, 8, 0, true>(a, st);
  return K % 128 == 0 ? launch_igemm<128, 128, 8, 0>(a, st) : launch_igemm<128, 64, 8, 0>(a, st);
}

struct BnbArgs {  // BatchNorm-backward epilogue (ConvArgs: bn_*)
  const void* bn_y;
  const void* bn_x;
  const void* bn_mask;
  const float *mean, *invstd, *gamma, *beta;
  int G;
  float* stat;
  int stat_nb;
};

static int conv_dgrad_impl(const void* dy, const void* w_crsk, void* dx, const void* residual, int N, int H,
                           int W, int C, int K, int R, int S, int P, int Q, int stride, int pad, void* stream,
                           const void* pre_in = nullptr, const BnbArgs* bnb = nullptr);

// Linear + bias + GELU in one launch (ViT MLP fc1): pre = x W^T + bias -> pre_out (bf16, saved for the backward
// pass), gelu(pre) -> y.
extern "C" int wm_linear_bias_gelu_fwd(const void* x, const void* w_krsc, const float* bias, void* pre_out, void* y,
                                       int rows, int C, int K, void* stream) {
  WM_REQUIRE(pre_out && bias, WM_EINVAL);
  WM_REQUIRE((reinterpret_cast<uintptr_t>(bias) & 15) == 0 && (reinterpret_cast<uintptr_t>(pre_out) & 15) == 0, WM_EALIGN);
  return conv_fwd_impl(x, w_krsc, y, rows, 1, 1, C, K, 1, 1, 1, 1, 1, 0, nullptr, 0, 0, stream, bias, nullptr, pre_out);
}

// Input gradient of a Linear whose INPUT was gelu(pre): dx = (dy W) * gelu'(pre)  (ViT MLP fc2 backward): the
// activation's backward pass rides in the dgrad epilogue.  dy [rows][K], w_crsk [C][K], pre / dx [rows][C].
extern "C" int wm_linear_dgrad_gelu(const void* dy, const void* w_crsk, const void* pre, void* dx, int rows, int C,
                                    int K, void* stream) {
  WM_REQUIRE(pre, WM_EINVAL);
  return conv_dgrad_impl(dy, w_crsk, dx, nullptr, rows, 1, 1, C, K, 1, 1, 1, 1, 1, 0, stream, pre);
}

extern "C" int wm_conv2d_dgrad(const void* dy, const void* w_crsk, void* dx, int N, int H, int W,
                               int C, int K, int R, int S, int P, int Q, int stride, int pad,
                               void* stream) {
  return conv_dgrad_impl(dy, w_crsk, dx, nullptr, N, H, W, C, K, R, S, P, Q, stride, pad, stream);
}

extern "C" int wm_conv2d_dgrad_add(const void* dy, const void* w_crsk, const void* residual, void* dx, int N,
                                   int H, int W, int C, int K, int R, int S, int P, int Q, int stride, int pad,
                                   void* stream) {
  WM_REQUIRE(residual, WM_EINVAL);
  WM_REQUIRE((reinterpret_cast<uintptr_t>(residual) & 15) == 0, WM_EALIGN);
  return conv_dgrad_impl(dy, w_crsk, dx, residual, N, H, W, C, K, R, S, P, Q, stride, pad, stream);
}

// Can the BatchNorm-backward epilogue serve this dgrad?  Every 128-row tile must lie inside one statistics group
// (stride 2: inside one group of one parity class) and the kernel must be one of the BNB instantiations.
extern "C" int wm_conv2d_dgrad_bnstat_ok(int N, int H, int W, int C, int K, int R, int S, int P, int Q, int stride,
                                         int pad, int G) {
  if (conv_check(N, H, W, C, K, R, S, P, Q, stride, pad) != WM_OK || C % 64 != 0 || G <= 0 || N % G != 0) return 0;
  const long long rows = (long long)N * H * W;
  if (stride == 1) return (rows / G) % 128 == 0 ? 1 : 0;
  if (H % 2 || W % 2) return 0;
  const long long cls = (long long)N * (H / 2) * (W / 2);
  return (cls % 128 == 0 && (cls / G) % 128 == 0) ? 1 : 0;
}

extern "C" int wm_conv2d_dgrad_bnstat(const void* dy, const void* w_crsk, const void* residual, void* dx, int N, int H,
                                      int W, int C, int K, int R, int S, int P, int Q, int stride, int pad,
                                      const void* bn_y, const void* relu_x, const void* relu_mask, const float* gamma,
                                      const float* beta, const float* save_mean, const float* save_invstd, int G,
                                      float* stat_part, int stat_tiles, void* stream) {
  WM_REQUIRE(bn_y && save_mean && save_invstd && stat_part, WM_EINVAL);
  WM_REQUIRE(G > 0 && stat_tiles == (int)((long long)N * H * W / G / 128), WM_EINVAL);  // one slot per 128-row tile
  WM_REQUIRE(relu_x || relu_mask || (gamma && beta), WM_EINVAL);
  WM_REQUIRE(!(relu_x && relu_mask), WM_EINVAL);
  WM_REQUIRE(wm_conv2d_dgrad_bnstat_ok(N, H, W, C, K, R, S, P, Q, stride, pad, G), WM_EUNSUPPORTED);
  WM_REQUIRE(aligned16(bn_y) && aligned16(save_mean) && aligned16(save_invstd) && (relu_x == nullptr || aligned16(relu_x)) &&
                 (residual == nullptr || aligned16(residual)),
             WM_EALIGN);
  BnbArgs b{bn_y, relu_x, relu_mask, save_mean, save_invstd, gamma, beta, G, stat_part, stat_tiles};
  return conv_dgrad_impl(dy, w_crsk, dx, residual, N, H, W, C, K, R, S, P, Q, stride, pad, stream, nullptr, &b);
}

static int conv_dgrad_impl(const void* dy, const void* w_crsk, void* dx, const void* residual, int N, int H,
                           int W, int C, int K, int R, int S, int P, int Q, int stride, int pad, void* stream,
                           const void* pre_in, const BnbArgs* bnb) {
  WM_REQUIRE(dy && w_crsk && dx, WM_EINVAL);
  WM_REQUIRE(pre_in == nullptr || (residual == nullptr && R == 1 && S == 1 && stride == 1 && aligned16(pre_in)), WM_EUNSUPPORTED);
  const int rc = conv_check(N, H, W, C, K, R, S, P, Q, stride, pad);
  if (rc != WM_OK) return rc;
  WM_REQUIRE(C % 64 == 0, WM_EUNSUPPORTED);  // the stem needs no input gradient
  WM_REQUIRE(aligned16(dy) && aligned16(w_crsk) && aligned16(dx), WM_EALIGN);
  ConvArgs a;
  a.xcd = xcd_order();
  a.src = static_cast<const uint16_t*>(dy);
  a.wt = static_cast<const uint16_t*>(w_crsk);
  a.dst = static_cast<uint16_t*>(dx);
  a.N = N; a.SH = P; a.SW = Q; a.SC = K; a.DH = H; a.DW = W; a.DC = C;
  a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.M = N * H * W;
  a.d_dhw = wm_div_make((uint32_t)(H * W)); a.d_dw = wm_div_make((uint32_t)W);
  a.d_h2w2 = wm_div_make((uint32_t)((H >> 1) * (W >> 1) > 0 ? (H >> 1) * (W >> 1) : 1));
  a.d_w2 = wm_div_make((uint32_t)((W >> 1) > 0 ? (W >> 1) : 1));
  a.stat = nullptr; a.stat_nb = 0; a.stat_rpg = 1;
  a.res = static_cast<const uint16_t*>(residual);
  a.nkt = R * S * (K / 64);
  hipStream_t st = static_cast<hipStream_t>(stream);
  a.bias = nullptr;
  a.pre_in = static_cast<const uint16_t*>(pre_in);
  a.pre_out = nullptr;
  a.act = pre_in != nullptr ? 2 : 0;
  a.bn_y = a.bn_x = nullptr;
  a.bn_mask = nullptr;
  a.bn_mean = a.bn_invstd = a.bn_gamma = a.bn_beta = nullptr;
  if (bnb != nullptr) {
    WM_REQUIRE(pre_in == nullptr, WM_EUNSUPPORTED);
    a.bn_y = static_cast<const uint16_t*>(bnb->bn_y);
    a.bn_x = static_cast<const uint16_t*>(bnb->bn_x);
    a.bn_mask = static_cast<const uint8_t*>(bnb->bn_mask);
    a.bn_mean = bnb->mean; a.bn_invstd = bnb->invstd; a.bn_gamma = bnb->gamma; a.bn_beta = bnb->beta;
    a.stat = bnb->stat; a.stat_nb = bnb->stat_nb;
    if (stride == 1) {
      a.stat_rpg = (int)((long long)N * H * W / bnb->G);
      if (conv_patch_ok(a)) return launch_patch<1, true>(a, st);
      return C % 128 == 0 ? launch_igemm<128, 128, 8, 3, false, true>(a, st) : launch_igemm<128, 64, 8, 3, false, true>(a, st);
    }
    a.stat_rpg = (int)((long long)N * (H / 2) * (W / 2) / bnb->G);
    return C % 128 == 0 ? launch_igemm<128, 128, 8, 2, false, true>(a, st) : launch_igemm<128, 64, 8, 2, false, true>(a, st);
  }
  if (R == 1 && S == 1 && stride == 1 && pad == 0 && bnb == nullptr &&
      wm_panel_ok(a.M, K, C, residual != nullptr || pre_in != nullptr)) {
    // input gradient of a Linear with 192 OUTPUT features: dx = dy [rows][192] . w_crsk^T, w_crsk [C][192]
    WmPanelArgs pa{a.src, a.wt, nullptr, a.res, a.pre_in, nullptr, a.dst, a.M, C, a.act, 0, nullptr, nullptr, 0.f, 0};
    return wm_panel_launch(pa, st);
  }
  if (pre_in != nullptr)
    return C % 128 == 0 ? launch_igemm<128, 128, 8, 3, true>(a, st) : launch_igemm<128, 64, 8, 3, true>(a, st);
  if (conv_patch_ok(a)) return launch_patch<1>(a, st);
  // stride 2 with even image sides and class size % 128 == 0: parity-class ordering (no wasted taps)
  const long long cls = (long long)N * (H / 2) * (W / 2);
  if (stride == 2 && H % 2 == 0 && W % 2 == 0) {
    if (C % 128 == 0 && cls % 128 == 0) return launch_igemm<128, 128, 8, 2>(a, st);
    if (C % 128 != 0 && cls % 128 == 0) return launch_igemm<128, 64, 8, 2>(a, st);
  }
  // stride 1: MODE 3 = MODE 1 without the stride-2 address path in the k-loop's issue phase (hipcc
  // if-converts the run-time branch: its 64-bit address arithmetic was executed on every k-step)
  if (stride == 1) return C % 128 == 0 ? launch_igemm<128, 128, 8, 3>(a, st) : launch_igemm<128, 64, 8, 3>(a, st);
  return C % 128 == 0 ? launch_igemm<128, 128, 8, 1>(a, st) : launch_igemm<128, 64, 8, 1>(a, st);
}

extern "C" int wm_conv2d_wgrad(const void* dy, const void* x, float* dw_krsc, int N, int H, int W,
                               int C, int K, int R, int S, int P, int Q, int stride, int pad,
                               void* stream) {
  return wm_conv2d_wgrad_bias(dy, x, dw_krsc, nullptr, N, H, W, C, K, R, S, P, Q, stride, pad, stream);
}

extern "C" int wm_conv2d_wgrad_bias(const void* dy, const void* x, float* dw_krsc, float* dbias, int N, int H,
                                    int W, int C, int K, int R, int S, int P, int Q, int stride, int pad,
                                    void* stream) {
  WM_REQUIRE(dy && x && dw_krsc, WM_EINVAL);
  const int rc = conv_check(N, H, W, C, K, R, S, P, Q, stride, pad);
  if (rc != WM_OK) return rc;
  WM_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(dw_krsc), WM_EALIGN);
  // the kernel forms element offsets in 32 bits
  WM_REQUIRE((long long)N * H * W * C < (1ll << 32) - (1 << 20) && (long long)N * P * Q * K < (1ll << 32) - (1 << 20),
             WM_EUNSUPPORTED);
  WgradArgs a;
  a.xcd = xcd_order();
  a.dy = static_cast<const uint16_t*>(dy);
  a.x = static_cast<const uint16_t*>(x);
  a.dw = dw_krsc;
  a.dbias = dbias;
  a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.P = P; a.Q = Q;
  a.stride = stride; a.pad = pad; a.M = N * P * Q;
  a.chunks_per_split = 0; a.total_chunks = 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dbias != nullptr) {
    // wm_conv2d_wgrad_splits answers for the launch WITHOUT a bias gradient; a shape whose plan would differ with one
    // (the patch-resident 64 -> 64 3x3 form has no bias path) is refused instead of leaving slabs of the caller's
    // buffer unwritten.  Only Linear layers (1x1) ask for a bias gradient.
    WgradArgs q = a;
    q.dbias = nullptr;
    WM_REQUIRE(!wgrad_patch_ok(q), WM_EUNSUPPORTED);
  }
  if (wgrad_patch_ok(a)) return launch_wgrad_patch(a, st);
  int bmo, cpt, nt;
  wgrad_config(C, K, R, S, bmo, cpt, nt);
  if (cpt == 2) {
    if (bmo == 128) return nt == 2 ? launch_wgrad<128, 2, 2>(a, st) : launch_wgrad<128, 2, 1>(a, st);
    return nt == 4 ? launch_wgrad<64, 2, 4>(a, st) : launch_wgrad<64, 2, 1>(a, st);
  }
  if (bmo == 128) {
    if (nt == 3) return launch_wgrad<128, 8, 3>(a, st);
    if (nt == 2) return launch_wgrad<128, 8, 2>(a, st);
    return launch_wgrad<128, 8, 1>(a, st);
  }
  return nt == 3 ? launch_wgrad<64, 8, 3>(a, st) : launch_wgrad<64, 8, 1>(a, st);
}

extern "C" int wm_conv2d_wgrad_splits(int N, int H, int W, int C, int K, int R, int S, int P, int Q, int stride, int pad) {
  const int rc = conv_check(N, H, W, C, K, R, S, P, Q, stride, pad);
  if (rc != WM_OK) return rc;
  WgradArgs a{};
  a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.P = P; a.Q = Q;
  a.stride = stride; a.pad = pad; a.dbias = nullptr;
  int bmo, cpt, nt, nsplit, cps, tc;
  wgrad_config(C, K, R, S, bmo, cpt, nt);
  if (wgrad_patch_ok(a)) { bmo = 64; nt = 9; }  // (a bias gradient is only asked of Linear layers: never this shape)
  wgrad_plan(a, bmo, nt, nsplit, cps, tc);
  return nsplit;
}
