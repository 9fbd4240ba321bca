// Weight / image layout transforms and the fused SGD step.
//
//  * f32 OIHW master weights (timm / torch state_dict layout, kept so that reference checkpoints
//    load: SURVEY §5) -> bf16 [K][R][S][C] (forward, wgrad order) and [C][R][S][K] (dgrad);
//  * wgrad's f32 [K][R][S][C] accumulator -> OIHW gradient;
//  * the 7x7/2 stem as a 4x4/1 convolution over the 2x2 space-to-depth image: channel
//    (dh*2+dw)*3 + c of pixel (h2, w2) is x[c][2*h2+dh][2*w2+dw]; kernel row/col r' = r + 1 = 2a+dh;
//  * SGD with momentum and weight decay over flat f32 arenas, as torch.optim.SGD computes it for
//    the reference's SimCLR (scripts/WM811k_benchmark.py:250-255).
#include "common.h"

namespace {

constexpr int LT_THREADS = 256;

__global__ void weights_prepare(const float* __restrict__ w, int K, int C, int R, int S,
                                uint16_t* __restrict__ krsc, uint16_t* __restrict__ crsk) {
  const long long total = (long long)K * C * R * S;
  for (long long t = (long long)blockIdx.x * LT_THREADS + threadIdx.x; t < total;
       t += (long long)gridDim.x * LT_THREADS) {
    // t enumerates the KRSC order (coalesced writes of the forward layout)
    const int c = (int)(t % C);
    long long u = t / C;
    const int s = (int)(u % S);
    u /= S;
    const int r = (int)(u % R), k = (int)(u / R);
    const uint16_t v = f2bf(w[(((size_t)k * C + c) * R + r) * S + s]);
    if (krsc) krsc[t] = v;
    if (crsk) crsk[(((size_t)c * R + r) * S + s) * K + k] = v;
  }
}

__global__ void wgrad_finalize(const float* __restrict__ ws, int nsplit, int K, int C, int R, int S,
                               float* __restrict__ grad, int accumulate) {
  const long long total = (long long)K * C * R * S;
  for (long long t = (long long)blockIdx.x * LT_THREADS + threadIdx.x; t < total;
       t += (long long)gridDim.x * LT_THREADS) {
    // t enumerates OIHW
    const int s = (int)(t % S);
    long long u = t / S;
    const int r = (int)(u % R);
    u /= R;
    const int c = (int)(u % C), k = (int)(u / C);
    const size_t wi = (((size_t)k * R + r) * S + s) * C + c;
    float v = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) v += ws[(size_t)sp * total + wi];  // slabs in order: bit-reproducible
    grad[t] = accumulate ? grad[t] + v : v;
  }
}

// stem: w [K][3][7][7] -> [K][4][4][16] bf16
__global__ void stem_weights_prepare(const float* __restrict__ w, int K, uint16_t* __restrict__ out) {
  const int total = K * 256;
  for (int t = blockIdx.x * LT_THREADS + threadIdx.x; t < total; t += gridDim.x * LT_THREADS) {
    const int ch = t & 15, b = (t >> 4) & 3, a = (t >> 6) & 3, k = t >> 8;
    float v = 0.f;
    if (ch < 12) {
      const int c = ch % 3, dw = (ch / 3) & 1, dh = ch / 6;
      const int r = 2 * a + dh - 1, s = 2 * b + dw - 1;
      if (r >= 0 && s >= 0) v = w[((k * 3 + c) * 7 + r) * 7 + s];
    }
    out[t] = f2bf(v);
  }
}

// One block per output channel k: 1024 threads = 4 slab ranges x the 256 space-to-depth positions of k.  Each range is
// summed in slab order (eight loads in flight), the four partial sums are combined in range order: the result does
// not depend on scheduling.  (The stem's weight gradient has one column group, so its split count is ~512.)
__global__ __launch_bounds__(1024) void stem_wgrad_finalize(const float* __restrict__ ws, int nsplit, int K,
                                                            float* __restrict__ grad, int accumulate) {
  __shared__ float red[4][256];
  const int k = blockIdx.x, wi = threadIdx.x & 255, q = threadIdx.x >> 8;
  const int per = (nsplit + 3) >> 2;
  const int s0 = q * per, s1 = min(nsplit, s0 + per);
  const size_t slab = (size_t)K * 256;
  const float* src = ws + (size_t)k * 256 + wi;
  float v = 0.f;
  int sp = s0;
  for (; sp + 8 <= s1; sp += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = src[(size_t)(sp + u) * slab];
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  for (; sp < s1; ++sp) v += src[(size_t)sp * slab];
  red[q][wi] = v;
  __syncthreads();
  if (q == 0) {
    const float tot = ((red[0][wi] + red[1][wi]) + red[2][wi]) + red[3][wi];
    const int ch = wi & 15, b = (wi >> 4) & 3, a = wi >> 6;
    if (ch < 12) {
      const int c = ch % 3, dw = (ch / 3) & 1, dh = ch / 6;
      const int r = 2 * a + dh - 1, s = 2 * b + dw - 1;
      if (r >= 0 && s >= 0) {
        float* g = grad + ((k * 3 + c) * 7 + r) * 7 + s;
        *g = accumulate ? *g + tot : tot;
      }
    }
  }
}

// image -> space-to-depth [N][H/2][W/2][16] bf16.  FMT 0: f32 NCHW [N][3][H][W]; 1: bf16 NHWC.
template <int FMT>
__global__ void image_to_s2d(const void* __restrict__ img, int N, int H, int W, uint16_t* __restrict__ out) {
  const int H2 = H >> 1, W2 = W >> 1;
  const long long total = (long long)N * H2 * W2;
  for (long long t = (long long)blockIdx.x * LT_THREADS + threadIdx.x; t < total;
       t += (long long)gridDim.x * LT_THREADS) {
    const int w2 = (int)(t % W2);
    long long u = t / W2;
    const int h2 = (int)(u % H2), n = (int)(u / H2);
    uint16_t v[16];
#pragma unroll
    for (int ch = 0; ch < 16; ++ch) {
      uint16_t o = 0;
      if (ch < 12) {
        const int c = ch % 3, dw = (ch / 3) & 1, dh = ch / 6;
        const int h = 2 * h2 + dh, w = 2 * w2 + dw;
        if (FMT == 0)
          o = f2bf(static_cast<const float*>(img)[(((size_t)n * 3 + c) * H + h) * W + w]);
        else
          o = static_cast<const uint16_t*>(img)[(((size_t)n * H + h) * W + w) * 3 + c];
      }
      v[ch] = o;
    }
    uint4* o4 = reinterpret_cast<uint4*>(out + (size_t)t * 16);
    o4[0] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
    o4[1] = make_uint4(v[8] | (v[9] << 16), v[10] | (v[11] << 16), v[12] | (v[13] << 16), v[14] | (v[15] << 16));
  }
}

__global__ void cast_f32_bf16(const float* __restrict__ x, long long n, uint16_t* __restrict__ y) {
  for (long long t = (long long)blockIdx.x * LT_THREADS + threadIdx.x; t < n; t += (long long)gridDim.x * LT_THREADS)
    y[t] = f2bf(x[t]);
}

// hyper[0] = lr, [1] = momentum, [2] = weight_decay, [3] = grad scale (1/world for averaged grads)
__global__ void sgd_step(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mom,
                         long long n, const float* __restrict__ hyper) {
  const float lr = hyper[0], mu = hyper[1], wd = hyper[2], gs = hyper[3];
  for (long long t = (long long)blockIdx.x * LT_THREADS + threadIdx.x; t < n; t += (long long)gridDim.x * LT_THREADS) {
    const float pv = p[t];
    float gv = g[t] * gs;
    gv = fmaf(wd, pv, gv);          // d_p = grad + weight_decay * p
    const float b = fmaf(mu, mom[t], gv);  // buf = momentum * buf + d_p  (buf starts at 0 == clone on step 1)
    mom[t] = b;
    p[t] = pv - lr * b;             // p -= lr * buf
  }
}

// ---- batched forms: every convolution / Linear parameter of a model in ONE launch, 32 x 32 (k, c) tiles through LDS.
// The per-parameter kernels above walk one layout linearly and scatter into the other (an OIHW element every R*S*4
// bytes, a [C][R][S][K] element every R*S*K*2 bytes): ~0.5 TB/s, and 24 (ResNet-18) to 100 (ViT) launches of
// 4-8 us per step.  Here a block owns tile (k0..k0+31, c0..c0+31, all R*S taps) of one parameter, found by binary
// search of its block index in the descriptors' tile prefix; both sides move in runs of >= 64 bytes.
constexpr int LB_T = 32;        // tile side
constexpr int LB_MAX_RS = 9;    // 3 x 3 (1 x 1 Linear / downsample: RS = 1)

__device__ __forceinline__ int lb_find(const WmLayoutDesc* __restrict__ d, int n, int tile) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {  // last descriptor whose tile0 <= tile
    const int mid = (lo + hi + 1) >> 1;
    if (d[mid].tile0 <= tile) lo = mid; else hi = mid - 1;
  }
  return lo;
}

template <int RS>
__device__ __forceinline__ void refresh_tile(const WmLayoutDesc& d, int k0, int c0, uint16_t (*tile)[LB_T][LB_T + 2]) {
  const int K = d.K, C = d.C;
  const int nk = min(LB_T, K - k0), nc = min(LB_T, C - c0);
  // load: for each k row the (c, rs) block is nc * RS contiguous floats of the OIHW tensor; 8 rows per pass
  // (thread = (row in pass, position in row): no division by a run-time value anywhere)
  const int rowl = threadIdx.x >> 5, pos = threadIdx.x & 31;
  for (int k = rowl; k < nk; k += LT_THREADS / 32) {
    const float* src = d.w + ((size_t)(k0 + k) * C + c0) * RS;
    for (int j = pos; j < nc * RS; j += 32) {
      const int c = j / RS, rs = j - c * RS;  // RS is a compile-time constant
      tile[rs][k][c] = f2bf(src[j]);
    }
  }
  __syncthreads();
  if (d.krsc != nullptr) {  // [K][RS][C]: runs of nc channels
    for (int u = rowl; u < nk * RS; u += LT_THREADS / 32) {
      const int k = u / RS, rs = u - k * RS;
      if (pos < nc) d.krsc[((size_t)(k0 + k) * RS + rs) * C + c0 + pos] = tile[rs][k][pos];
    }
  }
  if (d.crsk != nullptr) {  // [C][RS][K]: runs of nk output channels
    for (int u = rowl; u < nc * RS; u += LT_THREADS / 32) {
      const int c = u / RS, rs = u - c * RS;
      if (pos < nk) d.crsk[((size_t)(c0 + c) * RS + rs) * K + k0 + pos] = tile[rs][pos][c];
    }
  }
}

__global__ __launch_bounds__(LT_THREADS) void layouts_refresh_batched(const WmLayoutDesc* __restrict__ descs, int n_desc) {
  __shared__ uint16_t tile[LB_MAX_RS][LB_T][LB_T + 2];  // [rs][k][c], +2: the transposed read walks k at fixed c
  const int di = lb_find(descs, n_desc, blockIdx.x);
  const WmLayoutDesc d = descs[di];
  const int t = blockIdx.x - d.tile0;
  const int tk = t / d.tiles_c, tc = t - tk * d.tiles_c;
  if (d.RS == 9) refresh_tile<9>(d, tk * LB_T, tc * LB_T, tile);
  else refresh_tile<1>(d, tk * LB_T, tc * LB_T, tile);
}

// Weight-gradient fold: sum of a parameter's nsplit slabs [K][RS][C], in a FIXED order (bit-reproducible), added to the
// OIHW gradient; a block owns one (k, c) tile of one tap, a thread four channels (C % 4 == 0, host-checked).
//   nsplit <= FB_SPLIT_MIN: tile 8 k x 128 c x ALL taps, one thread per (k, four channels): all slabs in slab order;
//   nsplit  > FB_SPLIT_MIN: tile 4 k x 64 c, FOUR threads per (k, four channels), each summing a quarter of the slab
//     range in slab order; the quarters are combined in quarter order through LDS.  (The 64-channel layers run ~170
//     splits: one thread walking all of them was a chain of 20 - 40 dependent round trips, and those few blocks -- not
//     the bytes -- set the launch time: 265 us for 590 MB.)
// The OIHW side is a strided 4-byte read-modify-write (gradients are a few per cent of the slab bytes).
constexpr int FB_SPLIT_MIN = 32;
__device__ __forceinline__ float4 fold_range(const float* src, size_t slab, int s0, int s1) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  int sp = s0;
  for (; sp + 8 <= s1; sp += 8) {
    float4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const float4*>(src + (size_t)(sp + u) * slab);
#pragma unroll
    for (int u = 0; u < 8; ++u) {  // slab order
      v.x += t[u].x; v.y += t[u].y; v.z += t[u].z; v.w += t[u].w;
    }
  }
  for (; sp < s1; ++sp) {
    const float4 t = *reinterpret_cast<const float4*>(src + (size_t)sp * slab);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  return v;
}

template <int RS>
__device__ __forceinline__ void fold_tile(const WmLayoutDesc& d, const float* __restrict__ slabs, int tk, int tc, int rs,
                                          float4 (*part)[64]) {
  const int K = d.K, C = d.C, ns = d.nsplit;
  const size_t slab = (size_t)K * RS * C;
  const int tid = (int)threadIdx.x;
  if (ns <= FB_SPLIT_MIN) {
    // all RS taps of (k, four channels) in one thread: its 4 * RS gradient values are CONTIGUOUS in OIHW
    // (offset j * RS + rs), i.e. RS aligned float4 read-modify-writes instead of 4 * RS scattered words
    const int k = tk * 8 + (tid >> 5), c = tc * 128 + (tid & 31) * 4;
    if (k >= K || c >= C) return;
    float o[4 * RS];
#pragma unroll
    for (int t = 0; t < RS; ++t) {
      const float4 v = fold_range(slabs + ((size_t)k * RS + t) * C + c, slab, 0, ns);
      o[t] = v.x; o[RS + t] = v.y; o[2 * RS + t] = v.z; o[3 * RS + t] = v.w;
    }
    float4* g = reinterpret_cast<float4*>(d.grad + ((size_t)k * C + c) * RS);  // 16-byte aligned: c % 4 == 0
#pragma unroll
    for (int i = 0; i < RS; ++i) {
      float4 cur = g[i];
      cur.x += o[4 * i]; cur.y += o[4 * i + 1]; cur.z += o[4 * i + 2]; cur.w += o[4 * i + 3];
      g[i] = cur;
    }
    return;
  }
  const int q = tid >> 6, e = tid & 63;                       // quarter of the slab range, element of the 4 x 64 tile
  const int k = tk * 4 + (e >> 4), c = tc * 64 + (e & 15) * 4;
  const bool on = k < K && c < C;
  const int per = (ns + 3) >> 2;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (on) v = fold_range(slabs + ((size_t)k * RS + rs) * C + c, slab, q * per, min(ns, (q + 1) * per));
  part[q][e] = v;
  __syncthreads();
  if (q == 0 && on) {
    const float4 a = part[0][e], b = part[1][e], cc = part[2][e], dd = part[3][e];
    float* g = d.grad + ((size_t)k * C + c) * RS + rs;
    g[0] += ((a.x + b.x) + cc.x) + dd.x;
    g[RS] += ((a.y + b.y) + cc.y) + dd.y;
    g[2 * RS] += ((a.z + b.z) + cc.z) + dd.z;
    g[3 * RS] += ((a.w + b.w) + cc.w) + dd.w;
  }
}

// weight-gradient slabs -> += OIHW gradients (and bias slabs -> += bias gradients), all parameters of a backward pass.
// tiles_c = ceil(C / 128), tiles = ceil(K / 8) * tiles_c (nsplit <= 32); else tiles_c = ceil(C / 64), tiles =
// ceil(K / 4) * tiles_c * RS.
__global__ __launch_bounds__(LT_THREADS) void wgrad_fold_batched(const WmLayoutDesc* __restrict__ descs, int n_desc) {
  __shared__ float4 part[4][64];
  const int di = lb_find(descs, n_desc, blockIdx.x);
  const WmLayoutDesc d = descs[di];
  int t = blockIdx.x - d.tile0;
  int rs = 0;
  if (d.nsplit > FB_SPLIT_MIN) {  // one block per tap only in the split mode
    rs = t % d.RS;
    t /= d.RS;
  }
  const int tk = t / d.tiles_c, tc = t - tk * d.tiles_c;
  if (d.RS == 9) fold_tile<9>(d, d.ws, tk, tc, rs, part);
  else fold_tile<1>(d, d.ws, tk, tc, rs, part);
  if (d.crsk != nullptr) {
    // a SECOND slab set of the same shape (the field is unused by the fold otherwise): the parameter's weight gradient from
    // the second of two parallel branches of the pass (nn.ViewBranches), folded behind the first by the same threads --
    // grad = (grad + set 0) + set 1, to the bit what a second fold launch would compute
    __syncthreads();
    const float* second = reinterpret_cast<const float*>(d.crsk);
    if (d.RS == 9) fold_tile<9>(d, second, tk, tc, rs, part);
    else fold_tile<1>(d, second, tk, tc, rs, part);
  }
  const int kt = d.nsplit <= FB_SPLIT_MIN ? 8 : 4;
  if (d.w != nullptr && tc == 0 && rs == 0) {  // bias slabs [nsplit][K] -> bias gradient
    const int k = tk * kt + (int)threadIdx.x;
    if ((int)threadIdx.x < kt && k < d.K) {
      float v = 0.f;
      for (int sp = 0; sp < d.nsplit; ++sp) v += d.w[(size_t)sp * d.K + k];
      reinterpret_cast<float*>(d.krsc)[k] += v;
    }
  }
}

inline int grid_for(long long items) {
  long long b = (items + LT_THREADS - 1) / LT_THREADS;
  if (b > 2048) b = 2048;
  return b < 1 ? 1 : (int)b;
}

}  // namespace

extern "C" int wm_weights_prepare(const float* w_oihw, int K, int C, int R, int S, void* w_krsc, void* w_crsk,
                                  void* stream) {
  WM_REQUIRE(w_oihw && (w_krsc || w_crsk), WM_EINVAL);
  WM_REQUIRE(K > 0 && C > 0 && R > 0 && S > 0, WM_EINVAL);
  weights_prepare<<<grid_for((long long)K * C * R * S), LT_THREADS, 0, static_cast<hipStream_t>(stream)>>>(
      w_oihw, K, C, R, S, static_cast<uint16_t*>(w_krsc), static_cast<uint16_t*>(w_crsk));
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_layouts_refresh(const WmLayoutDesc* descs_dev, int n_desc, int total_tiles, void* stream) {
  WM_REQUIRE(descs_dev && n_desc > 0 && total_tiles > 0, WM_EINVAL);
  layouts_refresh_batched<<<total_tiles, LT_THREADS, 0, static_cast<hipStream_t>(stream)>>>(descs_dev, n_desc);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_wgrad_fold(const WmLayoutDesc* descs_dev, int n_desc, int total_tiles, void* stream) {
  WM_REQUIRE(descs_dev && n_desc > 0 && total_tiles > 0, WM_EINVAL);
  wgrad_fold_batched<<<total_tiles, LT_THREADS, 0, static_cast<hipStream_t>(stream)>>>(descs_dev, n_desc);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_wgrad_finalize(const float* dw_slabs, int nsplit, int K, int C, int R, int S, float* grad_oihw,
                                 int accumulate, void* stream) {
  WM_REQUIRE(dw_slabs && grad_oihw, WM_EINVAL);
  WM_REQUIRE(nsplit > 0 && K > 0 && C > 0 && R > 0 && S > 0, WM_EINVAL);
  wgrad_finalize<<<grid_for((long long)K * C * R * S), LT_THREADS, 0, static_cast<hipStream_t>(stream)>>>(
      dw_slabs, nsplit, K, C, R, S, grad_oihw, accumulate);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_stem_weights_prepare(const float* w_oihw, int K, void* w_s2d, void* stream) {
  WM_REQUIRE(w_oihw && w_s2d && K > 0, WM_EINVAL);
  stem_weights_prepare<<<grid_for((long long)K * 256), LT_THREADS, 0, static_cast<hipStream_t>(stream)>>>(
      w_oihw, K, static_cast<uint16_t*>(w_s2d));
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_stem_wgrad_finalize(const float* dw_s2d_slabs, int nsplit, int K, float* grad_oihw, int accumulate,
                                      void* stream) {
  WM_REQUIRE(dw_s2d_slabs && grad_oihw && K > 0 && nsplit > 0, WM_EINVAL);
  stem_wgrad_finalize<<<K, 1024, 0, static_cast<hipStream_t>(stream)>>>(dw_s2d_slabs, nsplit, K, grad_oihw, accumulate);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_image_to_s2d(const void* img, int fmt, int N, int H, int W, void* out, void* stream) {
  WM_REQUIRE(img && out, WM_EINVAL);
  WM_REQUIRE(N > 0 && H > 0 && W > 0, WM_EINVAL);
  WM_REQUIRE(H % 2 == 0 && W % 2 == 0, WM_EUNSUPPORTED);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long items = (long long)N * (H / 2) * (W / 2);
  if (fmt == WM_IMG_NCHW_F32)
    image_to_s2d<0><<<grid_for(items), LT_THREADS, 0, st>>>(img, N, H, W, static_cast<uint16_t*>(out));
  else if (fmt == WM_IMG_NHWC_BF16)
    image_to_s2d<1><<<grid_for(items), LT_THREADS, 0, st>>>(img, N, H, W, static_cast<uint16_t*>(out));
  else
    return WM_EUNSUPPORTED;
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_cast_f32_bf16(const float* x, long long n, void* y, void* stream) {
  WM_REQUIRE(x && y && n > 0, WM_EINVAL);
  cast_f32_bf16<<<grid_for(n), LT_THREADS, 0, static_cast<hipStream_t>(stream)>>>(x, n, static_cast<uint16_t*>(y));
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_sgd_step(float* params, const float* grads, float* momentum_buf, long long n,
                           const float* hyper, void* stream) {
  WM_REQUIRE(params && grads && momentum_buf && hyper && n > 0, WM_EINVAL);
  sgd_step<<<grid_for(n), LT_THREADS, 0, static_cast<hipStream_t>(stream)>>>(params, grads, momentum_buf, n, hyper);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

// ------------------------------------------------------------------------------------ LARS
// timm.optim.lars.Lars as the reference's BarlowTwins / VICReg use it (scripts/WM811k_benchmark.py:383-392,
// 418-427): per PARAMETER trust ratio  r = trust_coeff |w| / (|g| + wd |w| + eps)  (1 when |w| or |g| is 0),
// g <- (g + wd w) r, then SGD with momentum.  Parameters live in one flat arena; seg[] holds the element
// offsets of the parameters (seg[n_seg] = end), a block handles one chunk of one parameter.
namespace {
__global__ void segment_sqnorms(const float* __restrict__ p, const float* __restrict__ g, const long long* __restrict__ seg,
                                const float* __restrict__ hyper, float* __restrict__ out /* [n_seg][2], zeroed */) {
  __shared__ float red[2][4];
  const float gscale = hyper[5];
  const int s = blockIdx.y;
  const long long b = seg[s], e = seg[s + 1];
  float a = 0.f, c = 0.f;
  for (long long i = b + (long long)blockIdx.x * LT_THREADS + threadIdx.x; i < e; i += (long long)gridDim.x * LT_THREADS) {
    const float pv = p[i], gv = g[i] * gscale;
    a = fmaf(pv, pv, a);
    c = fmaf(gv, gv, c);
  }
  a = wave_sum(a);
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a;
    red[1][threadIdx.x >> 6] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(out + 2 * s, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(out + 2 * s + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

// hyper: {lr, momentum, weight_decay, trust_coeff, eps, grad_scale}
__global__ void lars_step(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mom,
                          const long long* __restrict__ seg, const float* __restrict__ norms,
                          const float* __restrict__ hyper) {
  const float lr = hyper[0], mu = hyper[1], wd = hyper[2], tc = hyper[3], eps = hyper[4], gs = hyper[5];
  const int s = blockIdx.y;
  const long long b = seg[s], e = seg[s + 1];
  float ratio = 1.f;
  if (wd != 0.f) {
    const float wn = sqrtf(norms[2 * s]), gn = sqrtf(norms[2 * s + 1]);
    if (wn > 0.f && gn > 0.f) ratio = tc * wn / (gn + wn * wd + eps);
  }
  for (long long i = b + (long long)blockIdx.x * LT_THREADS + threadIdx.x; i < e; i += (long long)gridDim.x * LT_THREADS) {
    const float pv = p[i];
    float gv = g[i] * gs;
    if (wd != 0.f) gv = fmaf(wd, pv, gv) * ratio;
    const float bu = fmaf(mu, mom[i], gv);
    mom[i] = bu;
    p[i] = pv - lr * bu;
  }
}
}  // namespace

extern "C" int wm_lars_step(float* params, const float* grads, float* momentum_buf, const long long* seg_offsets,
                            int n_seg, const float* hyper, float* norms_ws, void* stream) {
  WM_REQUIRE(params && grads && momentum_buf && seg_offsets && hyper && norms_ws && n_seg > 0 && n_seg <= 65535, WM_EINVAL);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e = wm_zero_async(norms_ws, (size_t)2 * n_seg * sizeof(float), st);
  if (e != hipSuccess) return (int)e;
  segment_sqnorms<<<dim3(32, n_seg), LT_THREADS, 0, st>>>(params, grads, seg_offsets, hyper, norms_ws);
  WM_LAUNCH_CHECK();
  lars_step<<<dim3(32, n_seg), LT_THREADS, 0, st>>>(params, grads, momentum_buf, seg_offsets, norms_ws, hyper);
  WM_LAUNCH_CHECK();
  return WM_OK;
}
