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

namespace {

constexpr int CV_THREADS = 256;
constexpr int CV_BM = 128;
constexpr int CV_ROW = 128;  // bytes per LDS row of an operand tile (64 bf16)

struct ConvArgs {
  const uint16_t* src;  // [N][SH][SW][SC]
  const uint16_t* wt;   // [DC][R][S][SC]
  uint16_t* dst;        // [N][DH][DW][DC]
  int N, SH, SW, SC, DH, DW, DC, R, S, stride, pad, M, nkt;
};

__device__ __forceinline__ uint4 ldg128(const uint16_t* p) { return *reinterpret_cast<const uint4*>(p); }

template <int BN, int CPT, bool DGRAD>
__global__ __launch_bounds__(CV_THREADS) void conv_igemm(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t cv_smem[];
  constexpr int A_BYTES = CV_BM * CV_ROW;
  constexpr int B_BYTES = BN * CV_ROW;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int NB = BN / 32;  // weight rows staged per thread
  constexpr int NJ = BN / 32;  // 16-channel fragments per wave (a wave owns BN/2 channels)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * CV_BM, n0 = blockIdx.y * BN;
  const int chunk = tid & 7, rowl = tid >> 3;
  const int swz = (chunk ^ (rowl & 7)) << 4;  // (rowl + 32 i) & 7 == rowl & 7

  int bh[4], bw[4], nb[4];
  bool mv[4];
  const int dhw = a.DH * a.DW;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + rowl + 32 * i;
    mv[i] = m < a.M;
    const int mm = mv[i] ? m : 0;
    const int n = mm / dhw, rem = mm - n * dhw;
    const int dh = rem / a.DW, dw = rem - dh * a.DW;
    if constexpr (!DGRAD) {
      bh[i] = dh * a.stride - a.pad;
      bw[i] = dw * a.stride - a.pad;
    } else {
      bh[i] = dh + a.pad;
      bw[i] = dw + a.pad;
    }
    nb[i] = n * a.SH * a.SW;
  }
  const size_t wrow = (size_t)a.R * a.S * a.SC;

  uint4 ra[4], rb[NB];
  auto gload = [&](int kt) {
    int r, s, coff;
    if constexpr (CPT == 8) {
      const int cpk = a.SC >> 6;
      const int tap = kt / cpk;
      coff = (kt - tap * cpk) * 64 + chunk * 8;
      r = tap / a.S;
      s = tap - r * a.S;
    } else {  // 16-channel source: one k-tile = kernel row kt, taps s = 0..3
      r = kt;
      s = chunk >> 1;
      coff = (chunk & 1) * 8;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int sh, sw;
      bool ok = mv[i];
      if constexpr (!DGRAD) {
        sh = bh[i] + r;
        sw = bw[i] + s;
      } else {
        const int th = bh[i] - r, tw = bw[i] - s;
        ok = ok && th >= 0 && tw >= 0;
        if (a.stride == 2) {
          ok = ok && (((th | tw) & 1) == 0);
          sh = th >> 1;
          sw = tw >> 1;
        } else {
          sh = th;
          sw = tw;
        }
      }
      ok = ok && (unsigned)sh < (unsigned)a.SH && (unsigned)sw < (unsigned)a.SW;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) v = ldg128(a.src + ((size_t)(nb[i] + sh * a.SW + sw) * a.SC + coff));
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
      rb[i] = ldg128(a.wt + (size_t)(n0 + rowl + 32 * i) * wrow + (size_t)kt * 64 + chunk * 8);
  };
  auto sstore = [&](uint8_t* buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<uint4*>(buf + (rowl + 32 * i) * CV_ROW + swz) = ra[i];
#pragma unroll
    for (int i = 0; i < NB; ++i)
      *reinterpret_cast<uint4*>(buf + A_BYTES + (rowl + 32 * i) * CV_ROW + swz) = rb[i];
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
        const int row = wn * (BN / 2) + j * 16 + fr;
        wf[j] = *reinterpret_cast<const bf16x8_t*>(buf + A_BYTES + row * CV_ROW + ((c ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[j][i], 0, 0, 0);
    }
  };

  gload(0);
  sstore(cv_smem);
  __syncthreads();
  for (int kt = 0; kt < a.nkt; ++kt) {
    uint8_t* cur = cv_smem + (kt & 1) * STAGE;
    if (kt + 1 < a.nkt) gload(kt + 1);
    compute(cur);
    if (kt + 1 < a.nkt) sstore(cv_smem + ((kt + 1) & 1) * STAGE);
    __syncthreads();
  }

  // ---- epilogue: accumulators -> bf16 tile in LDS ([pixel][channel], rows padded by 16 B) -> HBM
  constexpr int CS = BN * 2 + 16;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pix = wm * 64 + i * 16 + fr;
      const int ch = wn * (BN / 2) + j * 16 + fg * 4;
      const uint2 v = make_uint2(pack_bf2(acc[j][i][0], acc[j][i][1]), pack_bf2(acc[j][i][2], acc[j][i][3]));
      *reinterpret_cast<uint2*>(cv_smem + pix * CS + ch * 2) = v;
    }
  __syncthreads();
  constexpr int CPR = BN / 8;
  for (int p = tid; p < CV_BM * CPR; p += CV_THREADS) {
    const int row = p / CPR, ch = p - row * CPR;
    if (m0 + row < a.M)
      *reinterpret_cast<uint4*>(a.dst + (size_t)(m0 + row) * a.DC + n0 + ch * 8) =
          *reinterpret_cast<const uint4*>(cv_smem + row * CS + ch * 16);
  }
}

// ------------------------------------------------------------------------------------ wgrad
struct WgradArgs {
  const uint16_t* dy;  // [M][K]
  const uint16_t* x;   // [N][H][W][C]
  float* dw;           // [K][R][S][C] f32, accumulated with atomics
  int N, H, W, C, K, R, S, P, Q, stride, pad, M, chunks_per_split, total_chunks;
};

constexpr int WG_PIX = 64;  // pixels per staged chunk (two MFMA k-steps)

template <int BMO, int CPT>
__global__ __launch_bounds__(CV_THREADS) void conv_wgrad(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t wg_smem[];
  constexpr int SA = BMO * 2 + 32;  // dY tile row stride (bytes)
  constexpr int SB = 64 * 2 + 32;   // X tile row stride
  constexpr int A_BYTES = WG_PIX * SA;
  constexpr int STAGE = A_BYTES + WG_PIX * SB;
  constexpr int NA = BMO / 32;       // dY pieces per thread (64 rows x BMO/8 chunks / 256)
  constexpr int ACH = BMO / 8;       // 16-byte chunks per dY row
  constexpr int MJ = 2;              // 16-channel fragments per wave along K
  constexpr int NJ = BMO == 128 ? 4 : 2;  // 16-column fragments per wave

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = blockIdx.x;  // 64-column tile of the (r,s,c) axis
  const int k0 = blockIdx.y * BMO;
  const int cout_w = BMO == 128 ? wave * 32 : (wave >> 1) * 32;
  const int col_w = BMO == 128 ? 0 : (wave & 1) * 32;

  // column tile -> tap / channel offset for this thread's X chunk
  const int xchunk = tid & 7, xrow = tid >> 3;  // rows xrow, xrow + 32
  int r, s, coff;
  if constexpr (CPT == 8) {
    const int cpk = a.C >> 6;
    const int tap = ct / cpk;
    coff = (ct - tap * cpk) * 64 + xchunk * 8;
    r = tap / a.S;
    s = tap - r * a.S;
  } else {
    r = ct;
    s = xchunk >> 1;
    coff = (xchunk & 1) * 8;
  }

  const int chunk_begin = blockIdx.z * a.chunks_per_split;
  int chunk_end = chunk_begin + a.chunks_per_split;
  if (chunk_end > a.total_chunks) chunk_end = a.total_chunks;
  const int iters = chunk_end - chunk_begin;
  if (iters <= 0) return;

  // running (n, p, q) of this thread's two X rows
  int pn[2], pp[2], pq[2];
  const int pqn = a.P * a.Q;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = chunk_begin * WG_PIX + xrow + 32 * i;
    pn[i] = m / pqn;
    const int rem = m - pn[i] * pqn;
    pp[i] = rem / a.Q;
    pq[i] = rem - pp[i] * a.Q;
  }

  uint4 ra[NA], rx[2];
  auto gload = [&](int it) {
    const int pix0 = (chunk_begin + it) * WG_PIX;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int p = tid + CV_THREADS * i;
      const int row = p / ACH, ch = p - row * ACH;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (pix0 + row < a.M) v = ldg128(a.dy + (size_t)(pix0 + row) * a.K + k0 + ch * 8);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int sh = pp[i] * a.stride - a.pad + r, sw = pq[i] * a.stride - a.pad + s;
      const bool ok = (pix0 + xrow + 32 * i < a.M) && (unsigned)sh < (unsigned)a.H && (unsigned)sw < (unsigned)a.W;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) v = ldg128(a.x + ((size_t)((pn[i] * a.H + sh) * a.W + sw) * a.C + coff));
      rx[i] = v;
      // advance this row by one chunk (64 pixels)
      pq[i] += WG_PIX;
      while (pq[i] >= a.Q) {
        pq[i] -= a.Q;
        if (++pp[i] == a.P) {
          pp[i] = 0;
          ++pn[i];
        }
      }
    }
  };
  auto sstore = [&](uint8_t* buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int p = tid + CV_THREADS * i;
      const int row = p / ACH, ch = p - row * ACH;
      *reinterpret_cast<uint4*>(buf + row * SA + ch * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *reinterpret_cast<uint4*>(buf + A_BYTES + (xrow + 32 * i) * SB + xchunk * 16) = rx[i];
  };

  f32x4_t acc[MJ][NJ];
#pragma unroll
  for (int i = 0; i < MJ; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // transposed-read geometry: 16-lane group g, lane (q, p) inside it; MFMA k-slot (g, e) holds
  // pixel 4g + e (e < 4) or 16 + 4g + (e - 4) of the 32-pixel k-step — same map for both operands.
  const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  auto tr_read = [&](const uint8_t* base, int stride, int row, int colbyte) -> s16x4_t {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4_t*)(base + row * stride + colbyte));
  };
  auto compute = [&](const uint8_t* buf) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r0 = ks * 32 + 4 * tg + tq, r1 = r0 + 16;
      bf16x8_t af[MJ], bfr[NJ];
#pragma unroll
      for (int i = 0; i < MJ; ++i) {
        const int cb = (cout_w + i * 16 + 4 * tp) * 2;
        const s16x4_t lo = tr_read(buf, SA, r0, cb), hi = tr_read(buf, SA, r1, cb);
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        af[i] = __builtin_bit_cast(bf16x8_t, v);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int cb = (col_w + j * 16 + 4 * tp) * 2;
        const s16x4_t lo = tr_read(buf + A_BYTES, SB, r0, cb), hi = tr_read(buf + A_BYTES, SB, r1, cb);
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        bfr[j] = __builtin_bit_cast(bf16x8_t, v);
      }
#pragma unroll
      for (int i = 0; i < MJ; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };

  gload(0);
  sstore(wg_smem);
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    uint8_t* cur = wg_smem + (it & 1) * STAGE;
    if (it + 1 < iters) gload(it + 1);
    compute(cur);
    if (it + 1 < iters) sstore(wg_smem + ((it + 1) & 1) * STAGE);
    __syncthreads();
  }

  const size_t rsc = (size_t)a.R * a.S * a.C;
  const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
  for (int i = 0; i < MJ; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kk = k0 + cout_w + i * 16 + fg * 4 + e;
        const int col = ct * 64 + col_w + j * 16 + fr;
        atomicAdd(a.dw + (size_t)kk * rsc + col, acc[i][j][e]);
      }
}

template <typename K>
int set_lds(K kernel, int bytes) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  return e == hipSuccess ? WM_OK : (int)e;
}

template <int BN, int CPT, bool DGRAD>
int launch_igemm(const ConvArgs& a, hipStream_t st) {
  constexpr int lds = 2 * (CV_BM * CV_ROW + BN * CV_ROW);
  static bool attr = false;
  if (!attr) {
    const int rc = set_lds(&conv_igemm<BN, CPT, DGRAD>, lds);
    if (rc != WM_OK) return rc;
    attr = true;
  }
  dim3 grid(wm_cdiv(a.M, CV_BM), a.DC / BN);
  conv_igemm<BN, CPT, DGRAD><<<grid, CV_THREADS, lds, st>>>(a);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

template <int BMO, int CPT>
int launch_wgrad(WgradArgs a, hipStream_t st) {
  constexpr int lds = 2 * (WG_PIX * (BMO * 2 + 32) + WG_PIX * (64 * 2 + 32));
  static bool attr = false;
  if (!attr) {
    const int rc = set_lds(&conv_wgrad<BMO, CPT>, lds);
    if (rc != WM_OK) return rc;
    attr = true;
  }
  const int coltiles = a.R * a.S * a.C / 64;
  const int ktiles = a.K / BMO;
  a.total_chunks = wm_cdiv(a.M, WG_PIX);
  int nsplit = 2048 / (coltiles * ktiles);
  if (nsplit < 1) nsplit = 1;
  if (nsplit > a.total_chunks) nsplit = a.total_chunks;
  a.chunks_per_split = wm_cdiv(a.total_chunks, nsplit);
  nsplit = wm_cdiv(a.total_chunks, a.chunks_per_split);
  dim3 grid(coltiles, ktiles, nsplit);
  conv_wgrad<BMO, CPT><<<grid, CV_THREADS, lds, st>>>(a);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

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

extern "C" int wm_conv2d_fwd(const void* x, const void* w_krsc, void* y, int N, int H, int W, int C,
                             int K, int R, int S, int P, int Q, int stride, int pad, void* stream) {
  WM_REQUIRE(x && w_krsc && y, WM_EINVAL);
  const int rc = conv_check(N, H, W, C, K, R, S, P, Q, stride, pad);
  if (rc != WM_OK) return rc;
  WM_REQUIRE(aligned16(x) && aligned16(w_krsc) && aligned16(y), WM_EALIGN);
  ConvArgs a;
  a.src = static_cast<const uint16_t*>(x);
  a.wt = static_cast<const uint16_t*>(w_krsc);
  a.dst = static_cast<uint16_t*>(y);
  a.N = N; a.SH = H; a.SW = W; a.SC = C; a.DH = P; a.DW = Q; a.DC = K;
  a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.M = N * P * Q;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (C == 16) {
    a.nkt = R;
    return K % 128 == 0 ? launch_igemm<128, 2, false>(a, st) : launch_igemm<64, 2, false>(a, st);
  }
  a.nkt = R * S * (C / 64);
  return K % 128 == 0 ? launch_igemm<128, 8, false>(a, st) : launch_igemm<64, 8, false>(a, st);
}

extern "C" int wm_conv2d_dgrad(const void* dy, const void* w_crsk, void* dx, int N, int H, int W,
                               int C, int K, int R, int S, int P, int Q, int stride, int pad,
                               void* stream) {
  WM_REQUIRE(dy && w_crsk && dx, WM_EINVAL);
  const int rc = conv_check(N, H, W, C, K, R, S, P, Q, stride, pad);
  if (rc != WM_OK) return rc;
  WM_REQUIRE(C % 64 == 0, WM_EUNSUPPORTED);  // the stem needs no input gradient
  WM_REQUIRE(aligned16(dy) && aligned16(w_crsk) && aligned16(dx), WM_EALIGN);
  ConvArgs a;
  a.src = static_cast<const uint16_t*>(dy);
  a.wt = static_cast<const uint16_t*>(w_crsk);
  a.dst = static_cast<uint16_t*>(dx);
  a.N = N; a.SH = P; a.SW = Q; a.SC = K; a.DH = H; a.DW = W; a.DC = C;
  a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.M = N * H * W;
  a.nkt = R * S * (K / 64);
  hipStream_t st = static_cast<hipStream_t>(stream);
  return C % 128 == 0 ? launch_igemm<128, 8, true>(a, st) : launch_igemm<64, 8, true>(a, st);
}

extern "C" int wm_conv2d_wgrad(const void* dy, const void* x, float* dw_krsc, int N, int H, int W,
                               int C, int K, int R, int S, int P, int Q, int stride, int pad,
                               void* stream) {
  WM_REQUIRE(dy && x && dw_krsc, WM_EINVAL);
  const int rc = conv_check(N, H, W, C, K, R, S, P, Q, stride, pad);
  if (rc != WM_OK) return rc;
  WM_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(dw_krsc), WM_EALIGN);
  WgradArgs a;
  a.dy = static_cast<const uint16_t*>(dy);
  a.x = static_cast<const uint16_t*>(x);
  a.dw = dw_krsc;
  a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.P = P; a.Q = Q;
  a.stride = stride; a.pad = pad; a.M = N * P * Q;
  a.chunks_per_split = 0; a.total_chunks = 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (C == 16) return K % 128 == 0 ? launch_wgrad<128, 2>(a, st) : launch_wgrad<64, 2>(a, st);
  return K % 128 == 0 ? launch_wgrad<128, 8>(a, st) : launch_wgrad<64, 8>(a, st);
}
