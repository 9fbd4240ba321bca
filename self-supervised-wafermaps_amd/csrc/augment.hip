// Fused multi-view wafer-map augmentation: ragged uint8 store -> normalised image tensors.
//
// One launch replaces the reference's per-sample CPU pipeline (DataLoader workers running
// torch + PIL + OpenCV + numpy), i.e. get_base_transforms / get_inference_transforms
// (src/ssl_wafermap/transforms/augmentations.py:253-357) and the RandomResizedCrop of
// MultiCropViewTransform (src/ssl_wafermap/transforms/wafer_multicrop_transform.py:66-85):
//
//   stage 1 (in LDS, at wafer resolution):
//       DieNoise      augmentations.py:27-36    x[flip] = 383 - x[flip]  (128 <-> 255)
//       DPWTransform  augmentations.py:182-227  die (r,c) -> (int((r+.5)/H*newH), int((c+.5)/W*newW)),
//                                               128s written first, 255s win collisions
//       MedianFilter  augmentations.py:103-107  cv2.medianBlur(x, 3): 3x3 median, replicated border
//   stage 2 (per output pixel, inverse index maps):
//       Resize([S,S], NEAREST)            PIL nearest: xin = int(xo), xo = 0.5*a + k*a accumulated in
//                                         double by repeated addition (ImagingScaleAffine)
//       RandomRotate -> transpose(ROTATE_90): out[y][x] = in[x][S-1-y]
//       RandomVerticalFlip, RandomHorizontalFlip
//       [RandomResizedCrop box (i,j,h,w) -> crop, then the same nearest resize to out_size]
//       Grayscale(3) + ToTensor (/255) + Normalize((x-mean)/std): a 256-entry LUT, R == G == B
//
// Roofline: HBM write-bound (algorithmic bytes per view = H*W + 3*O*O*elsize); a wafer is ~1.3 kB
// so everything upstream of the store lives in LDS.  grid = (views, row blocks): each block
// re-derives the (tiny) stage-1 image and emits a band of output rows with 16-byte stores.
#include "common.h"

namespace {

constexpr int AUG_THREADS = 256;
constexpr int AUG_ROW_BLOCKS = 4;
constexpr int AUG_MAX_SIDE = 256;  // largest wafer side / img_size / out_size supported
constexpr int AUG_LDS_ELEMS = 8192; // elements per LDS image (wafers up to ~90 x 90 run from LDS)

// Counter RNG shared with the oracle (oracle/augment.py: rand01): lowbias32 finaliser.
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float rand01(uint32_t seed, uint32_t idx) {
  const uint32_t x = lowbias32(idx ^ lowbias32(seed ^ 0x9E3779B9U));
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ uint8_t die_only(uint8_t v) { return (v == 128 || v == 255) ? v : 0; }

// PIL nearest-resize source index for every destination index (sequential double accumulation).
__device__ void nearest_map(int n_in, int n_out, int base, short* dst) {
  const double a = (double)n_in / (double)n_out;
  double xo = a * 0.5;
  for (int k = 0; k < n_out; ++k) {
    int xi = (int)xo;
    if (xi > n_in - 1) xi = n_in - 1;
    dst[k] = (short)(base + xi);
    xo += a;
  }
}

__global__ __launch_bounds__(AUG_THREADS) void augment_kernel(
    const uint8_t* __restrict__ wafers, const long long* __restrict__ offsets,
    const int* __restrict__ heights, const int* __restrict__ widths,
    const WmViewParams* __restrict__ params, int S, int O, int fmt, int normalize, float mean,
    float stdv, int lds_elems, void* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t aug_smem[];
  __shared__ float lut[256];
  __shared__ short ymap[AUG_MAX_SIDE], xmap[AUG_MAX_SIDE], ycmap[AUG_MAX_SIDE], xcmap[AUG_MAX_SIDE];
  __shared__ short rmap[AUG_MAX_SIDE], cmap[AUG_MAX_SIDE];
  // stage 2, composed: source index of output pixel (oy, ox) = fx[ox] + gy[oy] (crop, flips, rotation and both nearest
  // maps folded into two tables of O entries: 8 consecutive fx entries are one 16-byte LDS read)
  __shared__ __attribute__((aligned(16))) int fx[AUG_MAX_SIDE];
  __shared__ int gy[AUG_MAX_SIDE];
  __shared__ int rfirst[AUG_MAX_SIDE], rlast[AUG_MAX_SIDE], cfirst[AUG_MAX_SIDE], clast[AUG_MAX_SIDE];

  const int tid = threadIdx.x;
  const WmViewParams P = params[blockIdx.x];
  const int H = heights[P.sample], W = widths[P.sample];
  const uint8_t* src = wafers + offsets[P.sample];
  uint8_t* raw_lds = aug_smem;              // [H][W]
  uint8_t* img1 = aug_smem + lds_elems;     // [H1][W1]

  if (H > AUG_MAX_SIDE || W > AUG_MAX_SIDE || H < 1 || W < 1) return;  // host-validated; never overrun the tables
  // The two LDS images are sized for an 8 K-element wafer (90 x 90: 18 KB per block, ~6 blocks per CU), not for
  // the largest wafer of the store (212 x 204 -> 86 KB, ONE block per CU: the launch then ran 8 rounds of
  // single-block CUs).  The rare larger wafer takes the same code with `raw` pointing at the store itself and the
  // stage-1 value of a pixel computed on demand in stage 2 (block-uniform branch).
  const bool small = H * W <= lds_elems;
  const uint8_t* raw = small ? raw_lds : src;
  if (small)
    for (int i = tid; i < H * W; i += AUG_THREADS) raw_lds[i] = src[i];
  {
    // ToTensor: uint8 -> float32 / 255 ; Normalize: (x - mean) / std, all float32 IEEE ops
    float v = __fdiv_rn((float)tid, 255.0f);
    if (normalize) v = __fdiv_rn(__fsub_rn(v, mean), stdv);
    lut[tid] = v;
  }
  int H1 = H, W1 = W;
  if (P.op == WM_AUG_DPW) {
    H1 = P.dpw_h;
    W1 = P.dpw_w;
    for (int i = tid; i < AUG_MAX_SIDE; i += AUG_THREADS) {
      rfirst[i] = 1 << 30; rlast[i] = -1; cfirst[i] = 1 << 30; clast[i] = -1;
    }
  }
  __syncthreads();

  // index tables (four lanes in four different waves run the sequential maps concurrently)
  if (tid == 0) nearest_map(H1, S, 0, ymap);
  if (tid == 64) nearest_map(W1, S, 0, xmap);
  if (P.crop) {
    if (tid == 128) nearest_map(P.crop_h, O, P.crop_i, ycmap);
    if (tid == 192) nearest_map(P.crop_w, O, P.crop_j, xcmap);
  }

  // ---- stage 1: value of pixel i of the stage-1 image, from `raw`
  auto noise_at = [&](int i) -> uint8_t {
    const uint8_t v = raw[i];
    const bool die = (v == 128) || (v == 255);
    const bool flip = die && (rand01(P.noise_seed, (uint32_t)i) < P.noise_p);
    return flip ? (uint8_t)(383 - v) : v;
  };
  auto median_at = [&](int i) -> uint8_t {
    const int r = i / W, c = i - r * W;
    // values need not be in {0,128,255}: 9-element median by partial selection sort
    uint8_t n[9];
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
      for (int dc = -1; dc <= 1; ++dc) {
        int rr = r + dr, cc = c + dc;
        rr = rr < 0 ? 0 : (rr > H - 1 ? H - 1 : rr);
        cc = cc < 0 ? 0 : (cc > W - 1 ? W - 1 : cc);
        n[(dr + 1) * 3 + (dc + 1)] = raw[rr * W + cc];
      }
#pragma unroll
    for (int a = 0; a < 5; ++a) {
#pragma unroll
      for (int b = a + 1; b < 9; ++b) {
        const uint8_t lo = n[a] < n[b] ? n[a] : n[b];
        const uint8_t hi = n[a] < n[b] ? n[b] : n[a];
        n[a] = lo;
        n[b] = hi;
      }
    }
    return n[4];
  };
  auto dpw_at = [&](int i) -> uint8_t {
    const int R = i / W1, C = i - R * W1;
    uint8_t best = 0;  // 255 beats 128 beats empty: the max over the source rectangle
    for (int r = rfirst[R]; r <= rlast[R]; ++r)
      for (int c = cfirst[C]; c <= clast[C]; ++c)
        if (rmap[r] == R && cmap[c] == C) {
          const uint8_t v = die_only(raw[r * W + c]);
          best = v > best ? v : best;
        }
    return best;
  };
  if (P.op == WM_AUG_DPW) {
    // reference arithmetic, float32 throughout: ((idx + 0.5) / shape) * new_shape, truncated
    if (tid < H) rmap[tid] = (short)(int)(__fmul_rn(__fdiv_rn((float)tid + 0.5f, (float)H), (float)H1));
    if (tid < W) cmap[tid] = (short)(int)(__fmul_rn(__fdiv_rn((float)tid + 0.5f, (float)W), (float)W1));
    __syncthreads();
    if (tid < H) {
      const int R = rmap[tid];
      if (R >= 0 && R < H1) { atomicMin(&rfirst[R], tid); atomicMax(&rlast[R], tid); }
    }
    if (tid < W) {
      const int C = cmap[tid];
      if (C >= 0 && C < W1) { atomicMin(&cfirst[C], tid); atomicMax(&clast[C], tid); }
    }
    __syncthreads();
  }
  const int op = P.op;
  if (small) {
    if (op == WM_AUG_DIENOISE) {
      for (int i = tid; i < H * W; i += AUG_THREADS) img1[i] = noise_at(i);
    } else if (op == WM_AUG_MEDIAN3) {
      for (int i = tid; i < H * W; i += AUG_THREADS) img1[i] = median_at(i);
    } else if (op == WM_AUG_DPW) {
      for (int i = tid; i < H1 * W1; i += AUG_THREADS) img1[i] = dpw_at(i);
    } else {
      img1 = raw_lds;
    }
  }
  __syncthreads();
  // composed index tables (ymap / xmap / ycmap / xcmap are complete: barriers above)
  for (int o = tid; o < O; o += AUG_THREADS) {
    int x = P.crop ? xcmap[o] : o;
    int y = P.crop ? ycmap[o] : o;
    if (P.hflip) x = S - 1 - x;
    if (P.vflip) y = S - 1 - y;
    if (P.rot90) {  // I1[y][x] = I0[x][S-1-y]: the row index comes from ox, the column index from oy
      fx[o] = ymap[x] * W1;
      gy[o] = xmap[S - 1 - y];
    } else {
      fx[o] = xmap[x];
      gy[o] = ymap[y] * W1;
    }
  }
  __syncthreads();
  // stage-1 pixel for stage 2: the LDS image, or (wafer larger than the LDS images) computed on demand
  auto pix1 = [&](int i) -> uint8_t {
    if (small) return img1[i];
    if (op == WM_AUG_DIENOISE) return noise_at(i);
    if (op == WM_AUG_MEDIAN3) return median_at(i);
    if (op == WM_AUG_DPW) return dpw_at(i);
    return raw[i];
  };

  // ---- stage 2: this block's band of output rows
  const int rows_per_block = (O + gridDim.y - 1) / gridDim.y;
  const int oy0 = blockIdx.y * rows_per_block;
  int oy1 = oy0 + rows_per_block;
  if (oy1 > O) oy1 = O;
  const int groups = O >> 3;  // 8 pixels per work item
  const size_t slot = (size_t)P.out_slot;
  auto gather8 = [&](int oy, int g, uint8_t* px) {
    const int base = gy[oy];
    const int4 f0 = *reinterpret_cast<const int4*>(fx + g * 8), f1 = *reinterpret_cast<const int4*>(fx + g * 8 + 4);
    const int f[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) px[e] = pix1(base + f[e]);
  };
  if (fmt == WM_IMG_S2D_BF16) {
    // 2x2 space-to-depth: s2d pixel (r2, c2) = 32 bytes = channels dh*6 + dw*3 + c of image pixels (2 r2 + dh, 2 c2 + dw)
    // (the three channels of a wafer image are equal), 12..15 zero.  A work item is ONE 16-byte half of an s2d pixel, so
    // the 64 lanes of a store instruction write 1 KiB of consecutive memory (8 full 128-byte lines).  [An item owning
    // four whole s2d pixels stored the same bytes as 16 B per lane at a 128-byte stride: every instruction touched 64
    // lines partially, 2.4x a plain fill of the tensor instead of ~1.3x.]
    //   half 0: a a a b | b b c c     half 1: c d d d | 0 0 0 0     (a, b = row 2 r2; c, d = row 2 r2 + 1)
    const int O2 = O >> 1;
    const int halves = O;                              // 2 halves x O/2 s2d pixels per s2d row
    const int items2 = ((oy1 - oy0) >> 1) * halves;    // bands start and end on even rows (O % 8 == 0, <= 4 bands)
    uint4* orow = reinterpret_cast<uint4*>(static_cast<uint16_t*>(out) + (slot * O2 + (size_t)(oy0 >> 1)) * O2 * 16);
    for (int it = tid; it < items2; it += AUG_THREADS) {
      const int rr = it / halves, hh = it - rr * halves;
      const int c2 = hh >> 1, half = hh & 1;
      const int g0 = gy[oy0 + 2 * rr], g1 = gy[oy0 + 2 * rr + 1];
      const int f0 = fx[2 * c2], f1 = fx[2 * c2 + 1];
      const uint32_t c = f2bf(lut[pix1(g1 + f0)]);
      uint4 v;
      if (half == 0) {
        const uint32_t a = f2bf(lut[pix1(g0 + f0)]), b = f2bf(lut[pix1(g0 + f1)]);
        v = make_uint4(a | (a << 16), a | (b << 16), b | (b << 16), c | (c << 16));
      } else {
        const uint32_t d = f2bf(lut[pix1(g1 + f1)]);
        v = make_uint4(c | (d << 16), d | (d << 16), 0u, 0u);
      }
      orow[(size_t)rr * halves + hh] = v;
    }
    return;
  }
  const int items = (oy1 - oy0) * groups;
  for (int it = tid; it < items; it += AUG_THREADS) {
    const int oy = oy0 + it / groups, g = it % groups;
    uint8_t px[8];
    gather8(oy, g, px);
    if (fmt == WM_IMG_NCHW_F32) {
      float* o = static_cast<float*>(out) + slot * 3 * O * O + (size_t)oy * O + g * 8;
      const float4 a = make_float4(lut[px[0]], lut[px[1]], lut[px[2]], lut[px[3]]);
      const float4 b = make_float4(lut[px[4]], lut[px[5]], lut[px[6]], lut[px[7]]);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        *reinterpret_cast<float4*>(o + (size_t)c * O * O) = a;
        *reinterpret_cast<float4*>(o + (size_t)c * O * O + 4) = b;
      }
    } else if (fmt == WM_IMG_NHWC_BF16) {
      uint16_t* o = static_cast<uint16_t*>(out) + (slot * O * O + (size_t)oy * O + g * 8) * 3;
      uint16_t hv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) hv[e] = f2bf(lut[px[e]]);
      // 24 bf16 = 3 x 16 bytes: p0 p0 p0 p1 p1 p1 p2 p2 | p2 p3 p3 p3 p4 p4 p4 p5 | p5 p5 p6 ...
      uint32_t w[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        const int e0 = (2 * q) / 3, e1 = (2 * q + 1) / 3;
        w[q] = (uint32_t)hv[e0] | ((uint32_t)hv[e1] << 16);
      }
      uint4* o4 = reinterpret_cast<uint4*>(o);
      o4[0] = make_uint4(w[0], w[1], w[2], w[3]);
      o4[1] = make_uint4(w[4], w[5], w[6], w[7]);
      o4[2] = make_uint4(w[8], w[9], w[10], w[11]);
    } else {
      uint8_t* o = static_cast<uint8_t*>(out) + slot * O * O + (size_t)oy * O + g * 8;
      uint32_t lo = px[0] | (px[1] << 8) | (px[2] << 16) | ((uint32_t)px[3] << 24);
      uint32_t hi = px[4] | (px[5] << 8) | (px[6] << 16) | ((uint32_t)px[7] << 24);
      *reinterpret_cast<uint2*>(o) = make_uint2(lo, hi);
    }
  }
}

// Validates the per-view parameters against the store on the device side would cost a sync, so the
// host wrapper (transforms/gpu.py) validates before upload; the kernel clamps nothing silently
// except the nearest-map upper edge.
}  // namespace

extern "C" int wm_augment_views(const uint8_t* wafers, const int64_t* offsets,
                                const int32_t* heights, const int32_t* widths, int n_wafers,
                                int max_wafer_elems, const WmViewParams* params, int n_views,
                                int img_size, int out_size, int out_format, int normalize,
                                float mean, float std, void* out, void* stream) {
  WM_REQUIRE(wafers && offsets && heights && widths && params && out, WM_EINVAL);
  WM_REQUIRE(n_wafers > 0 && n_views > 0 && img_size > 0 && out_size > 0, WM_EINVAL);
  WM_REQUIRE(img_size <= AUG_MAX_SIDE && out_size <= AUG_MAX_SIDE && out_size % 8 == 0,
             WM_EUNSUPPORTED);
  WM_REQUIRE(out_format == WM_IMG_NCHW_F32 || out_format == WM_IMG_NHWC_BF16 ||
                 out_format == WM_IMG_HW_U8 || out_format == WM_IMG_S2D_BF16,
             WM_EUNSUPPORTED);
  WM_REQUIRE(std != 0.f || !normalize, WM_EINVAL);
  WM_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, WM_EALIGN);
  WM_REQUIRE(max_wafer_elems > 0 && max_wafer_elems <= AUG_MAX_SIDE * AUG_MAX_SIDE, WM_EUNSUPPORTED);
  // two LDS images of min(largest wafer, 8 K elements): larger wafers take the on-demand path inside the kernel
  int lds_elems = (max_wafer_elems + 15) & ~15;
  if (lds_elems > AUG_LDS_ELEMS) lds_elems = AUG_LDS_ELEMS;
  // bands per view: every band repeats the view's prologue (index maps, stage 1), so two when the launch has enough
  // views to fill the chip either way (measured at 512 views: 47.5 us with 2, 54.0 with 4, 63.7 with 1)
  const int rb = n_views >= 384 ? 2 : AUG_ROW_BLOCKS;
  dim3 grid(n_views, rb);
  augment_kernel<<<grid, AUG_THREADS, 2 * lds_elems, static_cast<hipStream_t>(stream)>>>(
      wafers, reinterpret_cast<const long long*>(offsets), heights, widths, params, img_size,
      out_size, out_format, normalize, mean, std, lds_elems, out);
  WM_LAUNCH_CHECK();
  return WM_OK;
}
