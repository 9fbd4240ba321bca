// Max-pool 3x3/2 (ResNet stem) and global average pool, NHWC bf16.
//
// Replaces nn.MaxPool2d(3, 2, 1) and the SelectAdaptivePool2d(avg) of timm's resnet18 in the
// reference (scripts/WM811k_benchmark.py:231).  The forward records the winning window position
// per output element (first maximum in (kh, kw) scan order, as PyTorch's max_pool2d does — ReLU
// outputs tie at 0 all the time), and the backward is a gather over the <= 4 windows covering an
// input pixel, so no atomics are needed.  Roofline: HBM.
#include "common.h"

namespace {

constexpr int PL_THREADS = 256;

__device__ __forceinline__ void up8(const uint4 v, float (&f)[8]) {
  f[0] = bf2f((uint16_t)(v.x & 0xffff)); f[1] = bf2f((uint16_t)(v.x >> 16));
  f[2] = bf2f((uint16_t)(v.y & 0xffff)); f[3] = bf2f((uint16_t)(v.y >> 16));
  f[4] = bf2f((uint16_t)(v.z & 0xffff)); f[5] = bf2f((uint16_t)(v.z >> 16));
  f[6] = bf2f((uint16_t)(v.w & 0xffff)); f[7] = bf2f((uint16_t)(v.w >> 16));
}

__global__ __launch_bounds__(PL_THREADS) void maxpool_fwd(const uint16_t* __restrict__ x, int N, int H,
                                                          int W, int C, int P, int Q,
                                                          uint16_t* __restrict__ y,
                                                          uint8_t* __restrict__ idx) {
  const int cpr = C >> 3;
  const long long total = (long long)N * P * Q * cpr;
  for (long long t = (long long)blockIdx.x * PL_THREADS + threadIdx.x; t < total;
       t += (long long)gridDim.x * PL_THREADS) {
    const int c0 = (int)(t % cpr) * 8;
    long long pix = t / cpr;
    const int q = (int)(pix % Q);
    pix /= Q;
    const int p = (int)(pix % P), n = (int)(pix / P);
    float best[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      best[e] = -INFINITY;
      bi[e] = 0;
    }
    bool first = true;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int h = p * 2 - 1 + kh, w = q * 2 - 1 + kw;
        if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
          float f[8];
          up8(*reinterpret_cast<const uint4*>(x + ((size_t)(n * H + h) * W + w) * C + c0), f);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (first || f[e] > best[e]) {
              best[e] = f[e];
              bi[e] = kh * 3 + kw;
            }
          first = false;
        }
      }
    const size_t o = ((size_t)(n * P + p) * Q + q) * C + c0;
    *reinterpret_cast<uint4*>(y + o) = make_uint4(pack_bf2(best[0], best[1]), pack_bf2(best[2], best[3]),
                                                  pack_bf2(best[4], best[5]), pack_bf2(best[6], best[7]));
    uint2 pk;
    pk.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    pk.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
    *reinterpret_cast<uint2*>(idx + o) = pk;
  }
}

// Stem: relu(y*scale + shift) is pooled on the fly, so the 112x112x64 activation (the largest tensor
// of the network) is never written or re-read.  scale/shift are [G][C]; a row group = N/G images.
__global__ __launch_bounds__(PL_THREADS) void bn_relu_maxpool_fwd(const uint16_t* __restrict__ x,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, int N,
                                                                  int H, int W, int C, int P, int Q,
                                                                  int imgs_per_group,
                                                                  uint16_t* __restrict__ y,
                                                                  uint8_t* __restrict__ idx,
                                                                  uint16_t* __restrict__ xsel, const WmDiv d_cpr,
                                                                  const WmDiv d_q, const WmDiv d_p, const WmDiv d_ipg) {
  // 32-bit index arithmetic (the host checks N*P*Q*C/8 < 2^31): the kernel is VALU-bound and 64-bit
  // divisions by run-time values cost ~150 instructions per item
  const uint32_t cpr = (uint32_t)C >> 3;
  const uint32_t total = (uint32_t)N * P * Q * cpr;
  for (uint32_t t = blockIdx.x * PL_THREADS + threadIdx.x; t < total; t += gridDim.x * PL_THREADS) {
    uint32_t rc, rq, rp;
    uint32_t pix = wm_divmod(t, d_cpr, rc);
    const int c0 = (int)rc * 8;
    pix = wm_divmod(pix, d_q, rq);
    const int n = (int)wm_divmod(pix, d_p, rp);
    const int q = (int)rq, p = (int)rp;
    const int g = (int)wm_div((uint32_t)n, d_ipg);
    float sc[8], sh[8], bx[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = scale[(size_t)g * C + c0 + e];
      sh[e] = shift[(size_t)g * C + c0 + e];
    }
    float best[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      best[e] = -INFINITY;
      bi[e] = 0;
    }
    bool first = true;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int h = p * 2 - 1 + kh, w = q * 2 - 1 + kw;
        if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
          float f[8];
          up8(*reinterpret_cast<const uint4*>(x + ((size_t)(n * H + h) * W + w) * C + c0), f);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            // exactly what bn_apply would have stored: bf16(relu(fma(y, scale, shift)))
            const float v = bf2f(f2bf(fmaxf(fmaf(f[e], sc[e], sh[e]), 0.f)));
            if (first || v > best[e]) {
              best[e] = v;
              bi[e] = kh * 3 + kw;
              bx[e] = f[e];
            }
          }
          first = false;
        }
      }
    const size_t o = ((size_t)(n * P + p) * Q + q) * C + c0;
    *reinterpret_cast<uint4*>(y + o) = make_uint4(pack_bf2(best[0], best[1]), pack_bf2(best[2], best[3]),
                                                  pack_bf2(best[4], best[5]), pack_bf2(best[6], best[7]));
    uint2 pk;
    pk.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    pk.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
    *reinterpret_cast<uint2*>(idx + o) = pk;
    // the selected INPUT values: with them the backward's per-channel sums run over the pooled tensor
    // only (sum_pixels dz f(x) = sum_windows dy_pooled f(x at argmax), dz being a scatter of dy_pooled)
    if (xsel != nullptr)
      *reinterpret_cast<uint4*>(xsel + o) = make_uint4(pack_bf2(bx[0], bx[1]), pack_bf2(bx[2], bx[3]),
                                                       pack_bf2(bx[4], bx[5]), pack_bf2(bx[6], bx[7]));
  }
}

__global__ __launch_bounds__(PL_THREADS) void maxpool_bwd(const uint16_t* __restrict__ dy,
                                                          const uint8_t* __restrict__ idx, int N, int H,
                                                          int W, int C, int P, int Q,
                                                          uint16_t* __restrict__ dx) {
  const int cpr = C >> 3;
  const long long total = (long long)N * H * W * cpr;
  for (long long t = (long long)blockIdx.x * PL_THREADS + threadIdx.x; t < total;
       t += (long long)gridDim.x * PL_THREADS) {
    const int c0 = (int)(t % cpr) * 8;
    long long pix = t / cpr;
    const int w = (int)(pix % W);
    pix /= W;
    const int h = (int)(pix % H), n = (int)(pix / H);
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = 0.f;
    // windows p with 2p-1 <= h <= 2p+1
    const int p_lo = h >> 1, p_hi = (h + 1) >> 1, q_lo = w >> 1, q_hi = (w + 1) >> 1;
    for (int p = p_lo; p <= p_hi; ++p) {
      if (p >= P) continue;
      const int kh = h - (2 * p - 1);
      for (int q = q_lo; q <= q_hi; ++q) {
        if (q >= Q) continue;
        const int code = kh * 3 + (w - (2 * q - 1));
        const size_t o = ((size_t)(n * P + p) * Q + q) * C + c0;
        const uint2 pk = *reinterpret_cast<const uint2*>(idx + o);
        float d[8];
        up8(*reinterpret_cast<const uint4*>(dy + o), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int ie = (int)(((e < 4 ? pk.x : pk.y) >> (8 * (e & 3))) & 0xff);
          g[e] += ie == code ? d[e] : 0.f;
        }
      }
    }
    *reinterpret_cast<uint4*>(dx + ((size_t)(n * H + h) * W + w) * C + c0) =
        make_uint4(pack_bf2(g[0], g[1]), pack_bf2(g[2], g[3]), pack_bf2(g[4], g[5]), pack_bf2(g[6], g[7]));
  }
}

// x [N][HW][C] -> y [N][C] (mean over HW).  One thread per (n, 8 channels).
__global__ __launch_bounds__(PL_THREADS) void gap_fwd(const uint16_t* __restrict__ x, int N, int HW, int C,
                                                      uint16_t* __restrict__ y) {
  const int cpr = C >> 3;
  const int t = blockIdx.x * PL_THREADS + threadIdx.x;
  if (t >= N * cpr) return;
  const int n = t / cpr, c0 = (t - n * cpr) * 8;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  for (int i = 0; i < HW; ++i) {
    float f[8];
    up8(*reinterpret_cast<const uint4*>(x + ((size_t)n * HW + i) * C + c0), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] += f[e];
  }
  const float inv = 1.0f / (float)HW;
  *reinterpret_cast<uint4*>(y + (size_t)n * C + c0) =
      make_uint4(pack_bf2(s[0] * inv, s[1] * inv), pack_bf2(s[2] * inv, s[3] * inv),
                 pack_bf2(s[4] * inv, s[5] * inv), pack_bf2(s[6] * inv, s[7] * inv));
}

__global__ __launch_bounds__(PL_THREADS) void gap_bwd(const uint16_t* __restrict__ dy, int N, int HW, int C,
                                                      uint16_t* __restrict__ dx) {
  const int cpr = C >> 3;
  const long long total = (long long)N * HW * cpr;
  const float inv = 1.0f / (float)HW;
  for (long long t = (long long)blockIdx.x * PL_THREADS + threadIdx.x; t < total;
       t += (long long)gridDim.x * PL_THREADS) {
    const int c0 = (int)(t % cpr) * 8;
    const long long pix = t / cpr;
    const int n = (int)(pix / HW);
    float d[8];
    up8(*reinterpret_cast<const uint4*>(dy + (size_t)n * C + c0), d);
    *reinterpret_cast<uint4*>(dx + (size_t)pix * C + c0) =
        make_uint4(pack_bf2(d[0] * inv, d[1] * inv), pack_bf2(d[2] * inv, d[3] * inv),
                   pack_bf2(d[4] * inv, d[5] * inv), pack_bf2(d[6] * inv, d[7] * inv));
  }
}

inline int grid_for(long long items) {
  long long b = (items + PL_THREADS - 1) / PL_THREADS;
  if (b > 4096) b = 4096;
  return b < 1 ? 1 : (int)b;
}

}  // namespace

extern "C" int wm_maxpool3x3s2_fwd(const void* x, int N, int H, int W, int C, void* y, void* idx, void* stream) {
  WM_REQUIRE(x && y && idx, WM_EINVAL);
  WM_REQUIRE(N > 0 && H > 1 && W > 1 && C > 0, WM_EINVAL);
  WM_REQUIRE(C % 8 == 0, WM_EUNSUPPORTED);
  const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
  maxpool_fwd<<<grid_for((long long)N * P * Q * (C >> 3)), PL_THREADS, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const uint16_t*>(x), N, H, W, C, P, Q, static_cast<uint16_t*>(y), static_cast<uint8_t*>(idx));
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_bn_relu_maxpool3x3s2_fwd(const void* x, const float* scale, const float* shift, int N, int H,
                                           int W, int C, int G, void* y, void* idx, void* xsel, void* stream) {
  WM_REQUIRE(x && scale && shift && y && idx, WM_EINVAL);
  WM_REQUIRE(N > 0 && H > 1 && W > 1 && C > 0 && G > 0, WM_EINVAL);
  WM_REQUIRE(C % 8 == 0 && N % G == 0, WM_EUNSUPPORTED);
  WM_REQUIRE((long long)N * H * W * (C / 8) < (1ll << 31), WM_EUNSUPPORTED);  // 32-bit item indices
  const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
  bn_relu_maxpool_fwd<<<grid_for((long long)N * P * Q * (C >> 3)), PL_THREADS, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const uint16_t*>(x), scale, shift, N, H, W, C, P, Q, N / G, static_cast<uint16_t*>(y),
      static_cast<uint8_t*>(idx), static_cast<uint16_t*>(xsel), wm_div_make((uint32_t)(C >> 3)), wm_div_make((uint32_t)Q),
      wm_div_make((uint32_t)P), wm_div_make((uint32_t)(N / G)));
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_maxpool3x3s2_bwd(const void* dy, const void* idx, int N, int H, int W, int C, void* dx,
                                   void* stream) {
  WM_REQUIRE(dy && idx && dx, WM_EINVAL);
  WM_REQUIRE(N > 0 && H > 1 && W > 1 && C > 0, WM_EINVAL);
  WM_REQUIRE(C % 8 == 0, WM_EUNSUPPORTED);
  const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
  maxpool_bwd<<<grid_for((long long)N * H * W * (C >> 3)), PL_THREADS, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const uint16_t*>(dy), static_cast<const uint8_t*>(idx), N, H, W, C, P, Q,
      static_cast<uint16_t*>(dx));
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_gap_fwd(const void* x, int N, int HW, int C, void* y, void* stream) {
  WM_REQUIRE(x && y, WM_EINVAL);
  WM_REQUIRE(N > 0 && HW > 0 && C > 0, WM_EINVAL);
  WM_REQUIRE(C % 8 == 0, WM_EUNSUPPORTED);
  gap_fwd<<<wm_cdiv((long long)N * (C >> 3), PL_THREADS), PL_THREADS, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const uint16_t*>(x), N, HW, C, static_cast<uint16_t*>(y));
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_gap_bwd(const void* dy, int N, int HW, int C, void* dx, void* stream) {
  WM_REQUIRE(dy && dx, WM_EINVAL);
  WM_REQUIRE(N > 0 && HW > 0 && C > 0, WM_EINVAL);
  WM_REQUIRE(C % 8 == 0, WM_EUNSUPPORTED);
  gap_bwd<<<grid_for((long long)N * HW * (C >> 3)), PL_THREADS, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const uint16_t*>(dy), N, HW, C, static_cast<uint16_t*>(dx));
  WM_LAUNCH_CHECK();
  return WM_OK;
}
