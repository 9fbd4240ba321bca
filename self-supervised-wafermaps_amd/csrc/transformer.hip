// Row-wise / elementwise kernels of the vision-transformer path (DINO ViT-S/16, MAE ViT-B/32).
//
// Replaces the torch.nn.LayerNorm / GELU / bias adds inside facebookresearch/dino's
// VisionTransformer blocks and torchvision's vit_b_32 encoder blocks as the reference runs them
// (scripts/WM811k_benchmark.py:548-550,566-588 DINOViT; :881-957 MAE), lightly's
// get_at_index / set_at_index / patchify helpers (:911-947), torch.optim.AdamW (:591-598),
// lightly.models.utils.update_momentum (:579-581) and torch.nn.MSELoss (:900).
//
// Activations are bf16 [rows][C] with rows = tokens; every kernel moves 16 bytes per lane.
// Roofline: HBM (one read + one write of the activation per pass; reductions finish with f32 atomics).
#include "common.h"

namespace {

constexpr int TF_THREADS = 256;

__device__ __forceinline__ void unpack8(const uint4 v, float (&f)[8]) {
  f[0] = bf2f((uint16_t)(v.x & 0xffff)); f[1] = bf2f((uint16_t)(v.x >> 16));
  f[2] = bf2f((uint16_t)(v.y & 0xffff)); f[3] = bf2f((uint16_t)(v.y >> 16));
  f[4] = bf2f((uint16_t)(v.z & 0xffff)); f[5] = bf2f((uint16_t)(v.z >> 16));
  f[6] = bf2f((uint16_t)(v.w & 0xffff)); f[7] = bf2f((uint16_t)(v.w >> 16));
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
}
__device__ __forceinline__ void load8f(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// ------------------------------------------------------------------------------------ LayerNorm
// One wave per row; a lane owns chunks lane, lane + 64, ... of 8 channels (C <= 2048 -> <= 4 chunks),
// the row stays in registers between the mean, the variance and the output pass.
constexpr int LN_MAXV = 4;

// LPR = lanes per row: 64 (one row per wave, up to LN_MAXV chunks per lane) or 32 (rows of <= 32 chunks, i.e.
// C <= 256 -- ViT-Tiny's 192: two rows per wave, one chunk per lane; with one row per wave 40 of the 64 lanes idled).
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
  return group_sum<LPR>(v);  // DPP / lane-swap all-reduce (common.h)
}

// NV = chunks per lane (C <= 512 NV: 1 for the ViT-Tiny / -Small / MAE-decoder widths, 2 for ViT-B, 4 up to 2048);
// U = independent rows (row pairs for LPR = 32) a wave loads before it reduces any of them: one 16-byte load per lane
// and row leaves ~12 KB in flight per CU, a quarter of what HBM latency needs (measured 1.8-2.5 TB/s); with U rows in
// flight the kernels are bandwidth- instead of latency-bound.
template <int LPR, int NV, int U>
__global__ __launch_bounds__(TF_THREADS) void ln_fwd(const uint16_t* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, long long rows,
                                                     int C, uint16_t* __restrict__ y, float* __restrict__ mean,
                                                     float* __restrict__ rstd) {
  constexpr int RPW = 64 / LPR;                 // rows per wave and trip-slot
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, rsel = lane / LPR;
  const int nch = C >> 3;
  const float inv_c = 1.f / (float)C;
  float g[NV][8], b[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int ch = sub + LPR * i;
#pragma unroll
    for (int e = 0; e < 8; ++e) g[i][e] = b[i][e] = 0.f;
    if (ch < nch) {
      load8f(gamma + ch * 8, g[i]);
      load8f(beta + ch * 8, b[i]);
    }
  }
  for (long long base = ((long long)blockIdx.x * 4 + wave) * (RPW * U); base < rows;
       base += (long long)gridDim.x * 4 * (RPW * U)) {
    float v[U][NV][8];
    float s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long row = base + u * RPW + rsel;
      s[u] = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int ch = sub + LPR * i;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[u][i][e] = 0.f;
        if (row < rows && ch < nch) unpack8(*reinterpret_cast<const uint4*>(x + row * C + ch * 8), v[u][i]);
      }
    }
    float mu[U], r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) s[u] += v[u][i][e];
      mu[u] = row_sum<LPR>(s[u]) * inv_c;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (sub + LPR * i < nch) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = v[u][i][e] - mu[u];
            q = fmaf(d, d, q);
          }
        }
      }
      r[u] = rsqrtf(row_sum<LPR>(q) * inv_c + eps);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long row = base + u * RPW + rsel;
      if (row >= rows) continue;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int ch = sub + LPR * i;
        if (ch < nch) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = fmaf((v[u][i][e] - mu[u]) * r[u], g[i][e], b[i][e]);
          *reinterpret_cast<uint4*>(y + row * C + ch * 8) = pack8(o);
        }
      }
      if (sub == 0) {
        mean[row] = mu[u];
        rstd[row] = r[u];
      }
    }
  }
}

template <int LPR, int NV, int U>
__global__ __launch_bounds__(TF_THREADS) void ln_bwd(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, long long rows, int C,
                                                     const uint16_t* __restrict__ dres, uint16_t* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     float* __restrict__ part) {
  // part (optional, round 3): [2][gridDim.x][C] f32 -- the block STORES its per-channel sums into its own slot instead
  // of adding them to dgamma / dbeta with f32 atomics; the batched fold adds the slots in order (bit-reproducible)
  // dres (optional): the gradient that reaches x along the residual connection around the normalised branch
  // (x -> LN -> f -> + x); added here so autograd needs no separate accumulation pass over the activation
  __shared__ float red[2 * 2048];
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, rsel = lane / LPR;
  const int nch = C >> 3;
  const float inv_c = 1.f / (float)C;
  float dg[NV][8], db[NV][8], gm[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int ch = sub + LPR * i;
#pragma unroll
    for (int e = 0; e < 8; ++e) dg[i][e] = db[i][e] = gm[i][e] = 0.f;
    if (ch < nch) load8f(gamma + ch * 8, gm[i]);
  }
  for (long long base = ((long long)blockIdx.x * 4 + wave) * (RPW * U); base < rows;
       base += (long long)gridDim.x * 4 * (RPW * U)) {
    uint4 vx[U][NV], vd[U][NV], vr[U][NV];
    float mu[U], r[U];
    // every load of the trip first
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long row = base + u * RPW + rsel;
      const bool rok = row < rows;
      mu[u] = rok ? mean[row] : 0.f;
      r[u] = rok ? rstd[row] : 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int ch = sub + LPR * i;
        vx[u][i] = vd[u][i] = vr[u][i] = make_uint4(0, 0, 0, 0);
        if (rok && ch < nch) {
          vx[u][i] = *reinterpret_cast<const uint4*>(x + row * C + ch * 8);
          vd[u][i] = *reinterpret_cast<const uint4*>(dy + row * C + ch * 8);
          if (dres != nullptr) vr[u][i] = *reinterpret_cast<const uint4*>(dres + row * C + ch * 8);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long row = base + u * RPW + rsel;
      const bool rok = row < rows;
      float xh[NV][8], g[NV][8];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        float fx[8], fd[8];
        unpack8(vx[u][i], fx);
        unpack8(vd[u][i], fd);
        const bool on = rok && sub + LPR * i < nch;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[i][e] = on ? (fx[e] - mu[u]) * r[u] : 0.f;
          g[i][e] = fd[e] * gm[i][e];   // (zero where off: fd is zero there)
          s1 += g[i][e];
          s2 = fmaf(g[i][e], xh[i][e], s2);
          dg[i][e] = fmaf(fd[e], xh[i][e], dg[i][e]);
          db[i][e] += fd[e];
        }
      }
      const float c1 = row_sum<LPR>(s1) * inv_c, c2 = row_sum<LPR>(s2) * inv_c;
      if (rok) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int ch = sub + LPR * i;
          if (ch < nch) {
            float o[8], sk[8];
            unpack8(vr[u][i], sk);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = r[u] * (g[i][e] - c1 - xh[i][e] * c2);
            if (dres != nullptr) {  // the un-fused chain rounds the LayerNorm gradient to bf16 before the add
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = bf2f(f2bf(o[e])) + sk[e];
            }
            *reinterpret_cast<uint4*>(dx + row * C + ch * 8) = pack8(o);
          }
        }
      }
    }
  }
  // per-channel sums: with two rows per wave the halves of a wave hold the same channels: fold them first
  if (RPW == 2) {
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        dg[i][e] = wm_xor32_sum(dg[i][e]);
        db[i][e] = wm_xor32_sum(db[i][e]);
      }
  }
  // per-channel sums: the four waves fold into LDS one after the other, then one atomic per channel
  for (int w = 0; w < 4; ++w) {
    if (wave == w && rsel == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int ch = sub + LPR * i;
        if (ch < nch) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int c = ch * 8 + e;
            red[c] = (w == 0 ? 0.f : red[c]) + dg[i][e];
            red[2048 + c] = (w == 0 ? 0.f : red[2048 + c]) + db[i][e];
          }
        }
      }
    }
    __syncthreads();
  }
  if (part != nullptr) {
    for (int c = threadIdx.x; c < C; c += TF_THREADS) {
      part[(size_t)blockIdx.x * C + c] = red[c];
      part[((size_t)gridDim.x + blockIdx.x) * C + c] = red[2048 + c];
    }
    return;
  }
  for (int c = threadIdx.x; c < C; c += TF_THREADS) {
    atomicAdd(dgamma + c, red[c]);
    atomicAdd(dbeta + c, red[2048 + c]);
  }
}

// ------------------------------------------------------------------------------------ bias / GELU
__device__ __forceinline__ float gelu_f(float v) { return wm_gelu(v); }
__device__ __forceinline__ float gelu_grad(float v) { return wm_gelu_grad(v); }

template <int ACT>
__global__ __launch_bounds__(TF_THREADS) void bias_act_fwd(const uint16_t* __restrict__ x, const float* __restrict__ bias,
                                                           const uint16_t* __restrict__ res, long long rows, int C,
                                                           uint16_t* __restrict__ y) {
  const int nch = C >> 3;
  const long long total = rows * nch;
  for (long long t = (long long)blockIdx.x * TF_THREADS + threadIdx.x; t < total;
       t += (long long)gridDim.x * TF_THREADS) {
    const int ch = (int)(t % nch);
    float f[8], b[8];
    unpack8(*reinterpret_cast<const uint4*>(x + t * 8), f);
    if (bias != nullptr) {
      load8f(bias + ch * 8, b);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += b[e];
    }
    if (ACT == WM_ACT_GELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = gelu_f(f[e]);
    } else if (ACT == WM_ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
    }
    if (res != nullptr) {
      float r[8];
      unpack8(*reinterpret_cast<const uint4*>(res + t * 8), r);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += r[e];
    }
    *reinterpret_cast<uint4*>(y + t * 8) = pack8(f);
  }
}

// Column slabs of CB chunks (8 channels each) x RL row lanes per block: a thread keeps its chunk and
// walks rows, so the per-channel sums stay in registers until the end of the block.
// MODE 0: colsum of src only.  MODE 1: dx = dy (identity act) is not written, colsum of dy.
// MODE 2: GELU: dx = dy * gelu'(x + bias), colsum of dx.  MODE 3: ReLU: dx = dy * (x + bias > 0).
template <int MODE>
__global__ __launch_bounds__(TF_THREADS) void colsum_kernel(const uint16_t* __restrict__ x, const float* __restrict__ bias,
                                                            const uint16_t* __restrict__ dy, long long rows, int C,
                                                            int CB, int rows_per_block, uint16_t* __restrict__ dx,
                                                            float* __restrict__ out, float* __restrict__ part) {
  // part (optional): [gridDim.x][C] f32 -- the block STORES its column sums into its own slot (no f32 atomics; the
  // caller adds the slots in order)
  __shared__ float red[TF_THREADS * 8];
  const int nch = C >> 3;
  const int RL = TF_THREADS / CB;
  const int cl = threadIdx.x % CB, rl = threadIdx.x / CB;
  const int ch = blockIdx.y * CB + cl;
  const bool on = rl < RL && ch < nch;
  float acc[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = b[e] = 0.f;
  if (on) {
    if (MODE >= 2 && bias != nullptr) load8f(bias + ch * 8, b);
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    // four rows per trip: the loads of a trip are independent and issued together
    long long r = r0 + rl;
    for (; r + 3 * RL < r1; r += 4 * RL) {
      uint4 vd[4], vx[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t off = (size_t)(r + u * RL) * C + ch * 8;
        vd[u] = *reinterpret_cast<const uint4*>(dy + off);
        if (MODE >= 2) vx[u] = *reinterpret_cast<const uint4*>(x + off);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float fd[8];
        unpack8(vd[u], fd);
        if (MODE >= 2) {
          float fx[8];
          unpack8(vx[u], fx);
#pragma unroll
          for (int e = 0; e < 8; ++e) fd[e] *= MODE == 2 ? gelu_grad(fx[e] + b[e]) : (fx[e] + b[e] > 0.f ? 1.f : 0.f);
          const uint4 pk = pack8(fd);
          *reinterpret_cast<uint4*>(dx + (size_t)(r + u * RL) * C + ch * 8) = pk;
          unpack8(pk, fd);  // the sums see the rounded gradient, like a separate reduction would
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += fd[e];
      }
    }
    for (; r < r1; r += RL) {
      const size_t off = (size_t)r * C + ch * 8;
      float fd[8];
      unpack8(*reinterpret_cast<const uint4*>(dy + off), fd);
      if (MODE >= 2) {
        float fx[8];
        unpack8(*reinterpret_cast<const uint4*>(x + off), fx);
#pragma unroll
        for (int e = 0; e < 8; ++e) fd[e] *= MODE == 2 ? gelu_grad(fx[e] + b[e]) : (fx[e] + b[e] > 0.f ? 1.f : 0.f);
        const uint4 pk = pack8(fd);
        *reinterpret_cast<uint4*>(dx + off) = pk;
        unpack8(pk, fd);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += fd[e];
    }
  }
  if (out == nullptr && part == nullptr) return;
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = on ? acc[e] : 0.f;
  __syncthreads();
  if (rl == 0 && ch < nch) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s = 0.f;
      for (int j = 0; j < RL; ++j) s += red[(j * CB + cl) * 8 + e];
      if (part != nullptr) part[(size_t)blockIdx.x * C + ch * 8 + e] = s;
      else atomicAdd(out + ch * 8 + e, s);
    }
  }
}

__global__ __launch_bounds__(TF_THREADS) void zero_f32(float* p, long long n) {
  for (long long i = (long long)blockIdx.x * TF_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * TF_THREADS)
    p[i] = 0.f;
}

inline void colsum_geometry(long long rows, int C, int& CB, int& slabs, int& rb, long long& rpb) {
  const int nch = C >> 3;
  CB = nch <= 64 ? nch : 64;
  slabs = wm_cdiv(nch, CB);
  // ~768 blocks in total (3 per CU), at least 64 rows each: every block ends with one f32 atomic per
  // channel on the SAME addresses, and a chain of N same-address atomics costs ~0.1 us x N
  rb = wm_cdiv(768, slabs);
  rpb = (rows + rb - 1) / rb;
  if (rpb < 64) rpb = 64;
  rb = (int)wm_cdiv(rows, rpb);
}

template <int MODE>
int launch_colsum(const void* x, const float* bias, const void* dy, long long rows, int C, void* dx, float* out,
                  hipStream_t st, float* part = nullptr) {
  int CB, slabs, rb;
  long long rpb;
  colsum_geometry(rows, C, CB, slabs, rb, rpb);
  dim3 grid(rb, slabs);
  colsum_kernel<MODE><<<grid, TF_THREADS, 0, st>>>(static_cast<const uint16_t*>(x), bias,
                                                    static_cast<const uint16_t*>(dy), rows, C, CB, (int)rpb,
                                                    static_cast<uint16_t*>(dx), out, part);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

// ------------------------------------------------------------------------------------ tokens
__global__ __launch_bounds__(TF_THREADS) void tokens_assemble(const uint16_t* __restrict__ patches,
                                                              const float* __restrict__ cls, const float* __restrict__ pos,
                                                              int N, int np, int D, uint16_t* __restrict__ tokens) {
  const int nch = D >> 3;
  const long long total = (long long)N * (np + 1) * nch;
  for (long long t = (long long)blockIdx.x * TF_THREADS + threadIdx.x; t < total;
       t += (long long)gridDim.x * TF_THREADS) {
    const int ch = (int)(t % nch);
    const long long tok = t / nch;
    const int s = (int)(tok % (np + 1));
    const long long n = tok / (np + 1);
    float f[8], p[8];
    load8f(pos + (size_t)s * D + ch * 8, p);
    if (s == 0) {
      load8f(cls + ch * 8, f);
    } else {
      unpack8(*reinterpret_cast<const uint4*>(patches + ((size_t)(n * np + s - 1) * D + ch * 8)), f);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += p[e];
    *reinterpret_cast<uint4*>(tokens + (size_t)tok * D + ch * 8) = pack8(f);
  }
}

// [N][S][S][3] -> [N * (S/p)^2][p][p][3]: a patch row is p pixels x 3 channels = 6p contiguous bytes
// (p = 16: 96 B = 6 chunks of 16 B; p = 32: 12 chunks), copied 16 bytes per lane.
__global__ __launch_bounds__(TF_THREADS) void patchify_kernel(const uint16_t* __restrict__ img, int N, int S, int p,
                                                              uint16_t* __restrict__ rows) {
  const int g = S / p;
  const int cpr = (p * 3) >> 3;  // 16-byte chunks per patch row
  const long long total = (long long)N * g * g * p * cpr;
  for (long long t = (long long)blockIdx.x * TF_THREADS + threadIdx.x; t < total;
       t += (long long)gridDim.x * TF_THREADS) {
    const int c = (int)(t % cpr);
    long long u = t / cpr;
    const int ph = (int)(u % p);
    u /= p;
    const int gx = (int)(u % g);
    u /= g;
    const int gy = (int)(u % g);
    const long long n = u / g;
    const size_t src = (((size_t)n * S + gy * p + ph) * S + gx * p) * 3 + c * 8;
    *reinterpret_cast<uint4*>(rows + (size_t)t * 8) = *reinterpret_cast<const uint4*>(img + src);
  }
}

template <bool SCATTER>
__global__ __launch_bounds__(TF_THREADS) void rows_by_index(const uint16_t* __restrict__ src,
                                                            const long long* __restrict__ idx, int B, int S, int K,
                                                            int C, uint16_t* __restrict__ dst) {
  const int nch = C >> 3;
  const long long total = (long long)B * K * nch;
  for (long long t = (long long)blockIdx.x * TF_THREADS + threadIdx.x; t < total;
       t += (long long)gridDim.x * TF_THREADS) {
    const int ch = (int)(t % nch);
    const long long bk = t / nch;
    const long long b = bk / K;
    const long long s = idx[bk];
    if (s < 0 || s >= S) continue;
    const size_t big = ((size_t)b * S + s) * C + ch * 8, small = (size_t)bk * C + ch * 8;
    if (SCATTER) *reinterpret_cast<uint4*>(dst + big) = *reinterpret_cast<const uint4*>(src + small);
    else *reinterpret_cast<uint4*>(dst + small) = *reinterpret_cast<const uint4*>(src + big);
  }
}

template <bool L1>
__global__ __launch_bounds__(TF_THREADS) void mse_kernel(const uint16_t* __restrict__ pred,
                                                         const uint16_t* __restrict__ target, long long n,
                                                         float* __restrict__ loss, uint16_t* __restrict__ dpred) {
  __shared__ float red[4];
  const float inv = 1.f / (float)n;
  float s = 0.f;
  const long long nch = n >> 3;
  for (long long t = (long long)blockIdx.x * TF_THREADS + threadIdx.x; t < nch; t += (long long)gridDim.x * TF_THREADS) {
    float a[8], b[8], d[8];
    unpack8(*reinterpret_cast<const uint4*>(pred + t * 8), a);
    unpack8(*reinterpret_cast<const uint4*>(target + t * 8), b);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float df = a[e] - b[e];
      if (L1) {
        s += fabsf(df);
        d[e] = df > 0.f ? inv : (df < 0.f ? -inv : 0.f);
      } else {
        s = fmaf(df, df, s);
        d[e] = 2.f * df * inv;
      }
    }
    *reinterpret_cast<uint4*>(dpred + t * 8) = pack8(d);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv);
}

// ------------------------------------------------------------------------------------ DINO loss
// One block per row of D logits (D <= 256 * 8 * DL_MAXV); the row stays in registers.
constexpr int DL_MAXV = 4;  // D <= 8192 (bf16 rows) / f32 prob rows are streamed

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(TF_THREADS) void dino_teacher_probs(const uint16_t* __restrict__ t,
                                                                 const float* __restrict__ center, float inv_temp,
                                                                 int D, float* __restrict__ probs) {
  __shared__ float red[4];
  const long long row = blockIdx.x;
  const int nch = D >> 3;
  float v[DL_MAXV][8];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < DL_MAXV; ++i) {
    const int ch = threadIdx.x + TF_THREADS * i;
    if (ch < nch) {
      float c[8];
      unpack8(*reinterpret_cast<const uint4*>(t + row * D + ch * 8), v[i]);
      load8f(center + ch * 8, c);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] = (v[i][e] - c[e]) * inv_temp;
        m = fmaxf(m, v[i][e]);
      }
    }
  }
  m = block_reduce(m, red, true);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < DL_MAXV; ++i) {
    if (threadIdx.x + TF_THREADS * i < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] = expf(v[i][e] - m);
        s += v[i][e];
      }
    }
  }
  s = 1.f / block_reduce(s, red, false);
#pragma unroll
  for (int i = 0; i < DL_MAXV; ++i) {
    const int ch = threadIdx.x + TF_THREADS * i;
    if (ch < nch) {
      float* o = probs + row * D + ch * 8;
      *reinterpret_cast<float4*>(o) = make_float4(v[i][0] * s, v[i][1] * s, v[i][2] * s, v[i][3] * s);
      *reinterpret_cast<float4*>(o + 4) = make_float4(v[i][4] * s, v[i][5] * s, v[i][6] * s, v[i][7] * s);
    }
  }
}

// block = (student view s, sample b).  lsm = x/T - lse;  for every teacher view t != s:
//   loss -= <p_t, lsm> * w;   dx += w/T * (softmax(x/T) - p_t),   w = 1 / (n_terms * B)
__global__ __launch_bounds__(TF_THREADS) void dino_loss_kernel(const uint16_t* __restrict__ student,
                                                               const float* __restrict__ probs, int Vs, int Vt, int B,
                                                               int D, float inv_temp, float w, int skip_same,
                                                               float* __restrict__ loss, uint16_t* __restrict__ dstudent) {
  __shared__ float red[4];
  const int s = blockIdx.x / B, b = blockIdx.x % B;
  const long long row = (long long)s * B + b;
  const int nch = D >> 3;
  float v[DL_MAXV][8];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < DL_MAXV; ++i) {
    const int ch = threadIdx.x + TF_THREADS * i;
    if (ch < nch) {
      unpack8(*reinterpret_cast<const uint4*>(student + row * D + ch * 8), v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] *= inv_temp;
        m = fmaxf(m, v[i][e]);
      }
    }
  }
  m = block_reduce(m, red, true);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < DL_MAXV; ++i) {
    if (threadIdx.x + TF_THREADS * i < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += expf(v[i][e] - m);
    }
  }
  sum = block_reduce(sum, red, false);
  const float lse = m + logf(sum);
  float dot = 0.f;
  int nt = 0;
  float g[DL_MAXV][8];
#pragma unroll
  for (int i = 0; i < DL_MAXV; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) g[i][e] = 0.f;
  for (int t = 0; t < Vt; ++t) {
    if (skip_same && t == s) continue;
    ++nt;
    const float* p = probs + ((size_t)t * B + b) * D;
#pragma unroll
    for (int i = 0; i < DL_MAXV; ++i) {
      const int ch = threadIdx.x + TF_THREADS * i;
      if (ch < nch) {
        float pv[8];
        load8f(p + ch * 8, pv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          dot = fmaf(pv[e], v[i][e] - lse, dot);
          g[i][e] -= pv[e];
        }
      }
    }
  }
  dot = block_reduce(dot, red, false);
  if (threadIdx.x == 0) atomicAdd(loss, -dot * w);
  const float k = w * inv_temp;
#pragma unroll
  for (int i = 0; i < DL_MAXV; ++i) {
    const int ch = threadIdx.x + TF_THREADS * i;
    if (ch < nch) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = k * ((float)nt * expf(v[i][e] - lse) + g[i][e]);
      *reinterpret_cast<uint4*>(dstudent + row * D + ch * 8) = pack8(o);
    }
  }
}

// Mean-entropy regulariser of MSN / PMSN on rows of logits x [N][K] (bf16): p = softmax(x / T), m = mean_rows p,
//   reg = sum_k m_k (log m_k - log prior_k)      (prior NULL: the me-max term sum m log m of MSN)
//   d reg / d x[i][k] = p_ik (u_k - sum_j p_ij u_j) / (N T),  u_k = log m_k - log prior_k + 1.
// Pass 1 accumulates m (one block per row, f32 atomics); pass 2 forms the gradient and the scalar.
__global__ __launch_bounds__(TF_THREADS) void memax_mean_kernel(const uint16_t* __restrict__ x, int K, float inv_temp,
                                                                float inv_n, float* __restrict__ m) {
  __shared__ float red[4];
  const long long row = blockIdx.x;
  const int nch = K >> 3;
  float v[DL_MAXV][8];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < DL_MAXV; ++i) {
    const int ch = threadIdx.x + TF_THREADS * i;
    if (ch < nch) {
      unpack8(*reinterpret_cast<const uint4*>(x + row * K + ch * 8), v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] *= inv_temp;
        mx = fmaxf(mx, v[i][e]);
      }
    }
  }
  mx = block_reduce(mx, red, true);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < DL_MAXV; ++i)
    if (threadIdx.x + TF_THREADS * i < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] = expf(v[i][e] - mx);
        s += v[i][e];
      }
    }
  s = inv_n / block_reduce(s, red, false);
#pragma unroll
  for (int i = 0; i < DL_MAXV; ++i) {
    const int ch = threadIdx.x + TF_THREADS * i;
    if (ch < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) atomicAdd(m + ch * 8 + e, v[i][e] * s);
    }
  }
}

__global__ __launch_bounds__(TF_THREADS) void memax_grad_kernel(const uint16_t* __restrict__ x, const float* __restrict__ m,
                                                                const float* __restrict__ log_prior, int K,
                                                                float inv_temp, float inv_n, float* __restrict__ loss,
                                                                float* __restrict__ dx) {
  __shared__ float red[4];
  const long long row = blockIdx.x;
  const int nch = K >> 3;
  float v[DL_MAXV][8], u[DL_MAXV][8];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < DL_MAXV; ++i) {
    const int ch = threadIdx.x + TF_THREADS * i;
    if (ch < nch) {
      unpack8(*reinterpret_cast<const uint4*>(x + row * K + ch * 8), v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] *= inv_temp;
        mx = fmaxf(mx, v[i][e]);
        const float mk = m[ch * 8 + e];
        u[i][e] = logf(fmaxf(mk, 1e-30f)) - (log_prior ? log_prior[ch * 8 + e] : 0.f) + 1.f;
      }
    }
  }
  mx = block_reduce(mx, red, true);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < DL_MAXV; ++i)
    if (threadIdx.x + TF_THREADS * i < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] = expf(v[i][e] - mx);
        s += v[i][e];
      }
    }
  s = 1.f / block_reduce(s, red, false);
  float pu = 0.f;
#pragma unroll
  for (int i = 0; i < DL_MAXV; ++i)
    if (threadIdx.x + TF_THREADS * i < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] *= s;
        pu = fmaf(v[i][e], u[i][e], pu);
      }
    }
  pu = block_reduce(pu, red, false);
  const float c = inv_n * inv_temp;
#pragma unroll
  for (int i = 0; i < DL_MAXV; ++i) {
    const int ch = threadIdx.x + TF_THREADS * i;
    if (ch < nch) {
      float* o = dx + row * K + ch * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = c * v[i][e] * (u[i][e] - pu);
    }
  }
  if (row == 0) {  // the scalar: sum_k m_k (log m_k - log prior_k)
    float r = 0.f;
    for (int k = threadIdx.x; k < K; k += TF_THREADS) {
      const float mk = m[k];
      r += mk * (logf(fmaxf(mk, 1e-30f)) - (log_prior ? log_prior[k] : 0.f));
    }
    r = block_reduce(r, red, false);
    if (threadIdx.x == 0) atomicAdd(loss, r);
  }
}

// center[d] = m * center[d] + (1 - m) * mean_rows teacher[r][d].  One block per 32 columns.
__global__ __launch_bounds__(TF_THREADS) void dino_center_kernel(const uint16_t* __restrict__ t, long long rows, int D,
                                                                 float momentum, float* __restrict__ center) {
  __shared__ float red[8][33];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), rl = threadIdx.x >> 5;
  float s = 0.f;
  if (c < D)
    for (long long r = rl; r < rows; r += 8) s += bf2f(t[r * D + c]);
  red[rl][threadIdx.x & 31] = s;
  __syncthreads();
  if (rl == 0 && c < D) {
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) tot += red[j][threadIdx.x & 31];
    center[c] = momentum * center[c] + (1.f - momentum) * tot / (float)rows;
  }
}

// ------------------------------------------------------------------------------------ optimiser
__global__ __launch_bounds__(TF_THREADS) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v, long long n,
                                                           const float* __restrict__ hyper) {
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], bc1 = hyper[5],
              bc2s = sqrtf(hyper[6]), gs = hyper[7];
  const bool l2 = hyper[8] != 0.f;  // torch.optim.Adam: weight decay is added to the gradient, not decoupled
  const float step = lr / bc1;
  for (long long i = (long long)blockIdx.x * TF_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * TF_THREADS) {
    float gi = g[i] * gs;
    float pi = p[i];
    if (l2) gi = fmaf(wd, pi, gi);
    else pi *= 1.f - lr * wd;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    pi -= step * mi / (sqrtf(vi) / bc2s + eps);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}

__global__ __launch_bounds__(TF_THREADS) void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, long long n,
                                                         float m) {
  for (long long i = (long long)blockIdx.x * TF_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * TF_THREADS)
    ema[i] = ema[i] * m + p[i] * (1.f - m);
}

// ------------------------------------------------------------------------------------ column stats
template <typename T>
__device__ __forceinline__ float ld1(const T* p, size_t i);
template <>
__device__ __forceinline__ float ld1<float>(const float* p, size_t i) { return p[i]; }
template <>
__device__ __forceinline__ float ld1<uint16_t>(const uint16_t* p, size_t i) { return bf2f(p[i]); }

// PASS 0: acc[c] += sum_r x[r][c] / rows.  PASS 1: acc[c] += sum_r (x[r][c] - mean[c])^2 / rows.
// Block = 32 columns x 8 row lanes over a slab of rows; one atomic per column and block.
template <typename T, int PASS>
__global__ __launch_bounds__(TF_THREADS) void colstats_kernel(const T* __restrict__ x, long long rows, int C,
                                                              int rows_per_block, const float* __restrict__ mean,
                                                              float* __restrict__ acc) {
  __shared__ float red[8][33];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.y * 32 + cl;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  float s = 0.f;
  if (c < C) {
    const float mu = PASS == 1 ? mean[c] : 0.f;
    for (long long r = r0 + rl; r < r1; r += 8) {
      const float v = ld1<T>(x, (size_t)r * C + c) - mu;
      s += PASS == 1 ? v * v : v;
    }
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) tot += red[j][cl];
    atomicAdd(acc + c, tot / (float)rows);
  }
}

template <typename T>
__global__ __launch_bounds__(TF_THREADS) void standardize_kernel(const T* __restrict__ x, long long rows, int C,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ inv_scale,
                                                                 float* __restrict__ out) {
  const long long total = rows * C;
  for (long long i = (long long)blockIdx.x * TF_THREADS + threadIdx.x; i < total; i += (long long)gridDim.x * TF_THREADS) {
    const int c = (int)(i % C);
    out[i] = (ld1<T>(x, (size_t)i) - mean[c]) * inv_scale[c];
  }
}

template <typename T>
int launch_colstats(const void* x, long long rows, int C, float* mean, float* var, hipStream_t st) {
  long long rpb = (rows + 255) / 256;
  if (rpb < 64) rpb = 64;
  dim3 grid(wm_cdiv(rows, rpb), wm_cdiv(C, 32));
  colstats_kernel<T, 0><<<grid, TF_THREADS, 0, st>>>(static_cast<const T*>(x), rows, C, (int)rpb, nullptr, mean);
  WM_LAUNCH_CHECK();
  colstats_kernel<T, 1><<<grid, TF_THREADS, 0, st>>>(static_cast<const T*>(x), rows, C, (int)rpb, mean, var);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

inline int ew_blocks(long long work) {
  long long b = (work + TF_THREADS - 1) / TF_THREADS;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int wm_layernorm_fwd(const void* x, const float* gamma, const float* beta, float eps, long long rows, int C,
                                void* y, float* mean, float* rstd, void* stream) {
  WM_REQUIRE(x && gamma && beta && y && mean && rstd, WM_EINVAL);
  WM_REQUIRE(rows > 0 && C > 0 && C % 8 == 0, WM_EINVAL);
  WM_REQUIRE(C <= 64 * 8 * LN_MAXV, WM_EUNSUPPORTED);
  WM_REQUIRE(al16(x) && al16(y) && al16(gamma) && al16(beta), WM_EALIGN);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const uint16_t* xp = static_cast<const uint16_t*>(x);
  uint16_t* yp = static_cast<uint16_t*>(y);
  auto grid = [&](int rows_per_block) {
    long long blocks = (rows + rows_per_block - 1) / rows_per_block;
    return (int)(blocks > 1536 ? 1536 : blocks);  // 6 resident blocks per CU (76-84 VGPRs): one round, rows strided
  };
  // rows per block and trip: 4 waves x (64 / LPR) x U
  if (C <= 256) ln_fwd<32, 1, 4><<<grid(32), TF_THREADS, 0, st>>>(xp, gamma, beta, eps, rows, C, yp, mean, rstd);
  else if (C <= 512) ln_fwd<64, 1, 4><<<grid(16), TF_THREADS, 0, st>>>(xp, gamma, beta, eps, rows, C, yp, mean, rstd);
  else if (C <= 1024) ln_fwd<64, 2, 2><<<grid(8), TF_THREADS, 0, st>>>(xp, gamma, beta, eps, rows, C, yp, mean, rstd);
  else ln_fwd<64, 4, 1><<<grid(4), TF_THREADS, 0, st>>>(xp, gamma, beta, eps, rows, C, yp, mean, rstd);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

static int ln_bwd_blocks(long long rows, int C) {
  const int rpb = C <= 256 ? 32 : C <= 512 ? 16 : C <= 1024 ? 8 : 4;
  const long long blocks = (rows + rpb - 1) / rpb;
  return (int)(blocks > 768 ? 768 : blocks);  // 3 resident blocks per CU: one round; short atomic chains
}

static int layernorm_bwd_impl(const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd,
                              long long rows, int C, const void* dres, void* dx, float* dgamma, float* dbeta,
                              void* stream, float* part = nullptr) {
  WM_REQUIRE(x && dy && gamma && mean && rstd && dx && ((dgamma && dbeta) || part), WM_EINVAL);
  WM_REQUIRE(rows > 0 && C > 0 && C % 8 == 0, WM_EINVAL);
  WM_REQUIRE(C <= 64 * 8 * LN_MAXV, WM_EUNSUPPORTED);
  WM_REQUIRE(al16(x) && al16(dy) && al16(dx) && al16(gamma) && al16(dres), WM_EALIGN);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const uint16_t* xp = static_cast<const uint16_t*>(x);
  const uint16_t* dp = static_cast<const uint16_t*>(dy);
  const uint16_t* rp = static_cast<const uint16_t*>(dres);
  uint16_t* op = static_cast<uint16_t*>(dx);
  const int nb = ln_bwd_blocks(rows, C);
  if (C <= 256) ln_bwd<32, 1, 4><<<nb, TF_THREADS, 0, st>>>(xp, dp, gamma, mean, rstd, rows, C, rp, op, dgamma, dbeta, part);
  else if (C <= 512) ln_bwd<64, 1, 4><<<nb, TF_THREADS, 0, st>>>(xp, dp, gamma, mean, rstd, rows, C, rp, op, dgamma, dbeta, part);
  else if (C <= 1024) ln_bwd<64, 2, 2><<<nb, TF_THREADS, 0, st>>>(xp, dp, gamma, mean, rstd, rows, C, rp, op, dgamma, dbeta, part);
  else ln_bwd<64, 4, 1><<<nb, TF_THREADS, 0, st>>>(xp, dp, gamma, mean, rstd, rows, C, rp, op, dgamma, dbeta, part);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_layernorm_bwd(const void* x, const void* dy, const float* gamma, const float* mean,
                                const float* rstd, long long rows, int C, void* dx, float* dgamma, float* dbeta,
                                void* stream) {
  return layernorm_bwd_impl(x, dy, gamma, mean, rstd, rows, C, nullptr, dx, dgamma, dbeta, stream);
}

extern "C" int wm_layernorm_bwd_add(const void* x, const void* dy, const float* gamma, const float* mean,
                                    const float* rstd, long long rows, int C, const void* dres, void* dx,
                                    float* dgamma, float* dbeta, void* stream) {
  WM_REQUIRE(dres, WM_EINVAL);
  return layernorm_bwd_impl(x, dy, gamma, mean, rstd, rows, C, dres, dx, dgamma, dbeta, stream);
}

// The same with the parameter-gradient sums as per-block slots: part [2][wm_layernorm_bwd_blocks(rows, C)][C] f32
// (dgamma slots, then dbeta slots), every slot overwritten; the caller adds the slots in order (wm_wgrad_fold /
// wm_wgrad_finalize with K = 1).  dres may be NULL.
extern "C" int wm_layernorm_bwd_blocks(long long rows, int C) {
  if (rows <= 0 || C <= 0 || C % 8) return 0;
  return ln_bwd_blocks(rows, C);
}
extern "C" int wm_layernorm_bwd_parts(const void* x, const void* dy, const float* gamma, const float* mean,
                                      const float* rstd, long long rows, int C, const void* dres, void* dx,
                                      float* part, void* stream) {
  WM_REQUIRE(part, WM_EINVAL);
  return layernorm_bwd_impl(x, dy, gamma, mean, rstd, rows, C, dres, dx, nullptr, nullptr, stream, part);
}

extern "C" int wm_bias_act_fwd(const void* x, const float* bias, const void* residual, int act, long long rows, int C,
                               void* y, void* stream) {
  WM_REQUIRE(x && y, WM_EINVAL);
  WM_REQUIRE(rows > 0 && C > 0 && C % 8 == 0, WM_EINVAL);
  WM_REQUIRE(act == WM_ACT_NONE || act == WM_ACT_GELU || act == WM_ACT_RELU, WM_EUNSUPPORTED);
  WM_REQUIRE(al16(x) && al16(y) && al16(bias) && al16(residual), WM_EALIGN);
  const int blocks = ew_blocks(rows * (C >> 3));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (act == WM_ACT_RELU)
    bias_act_fwd<WM_ACT_RELU><<<blocks, TF_THREADS, 0, st>>>(static_cast<const uint16_t*>(x), bias,
                                                             static_cast<const uint16_t*>(residual), rows, C,
                                                             static_cast<uint16_t*>(y));
  else if (act == WM_ACT_GELU)
    bias_act_fwd<WM_ACT_GELU><<<blocks, TF_THREADS, 0, st>>>(static_cast<const uint16_t*>(x), bias,
                                                             static_cast<const uint16_t*>(residual), rows, C,
                                                             static_cast<uint16_t*>(y));
  else
    bias_act_fwd<WM_ACT_NONE><<<blocks, TF_THREADS, 0, st>>>(static_cast<const uint16_t*>(x), bias,
                                                             static_cast<const uint16_t*>(residual), rows, C,
                                                             static_cast<uint16_t*>(y));
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_bias_act_bwd(const void* x, const float* bias, const void* dy, int act, long long rows, int C,
                               void* dx, float* dbias, void* stream) {
  WM_REQUIRE(dy, WM_EINVAL);
  WM_REQUIRE(rows > 0 && C > 0 && C % 8 == 0, WM_EINVAL);
  WM_REQUIRE(al16(x) && al16(dy) && al16(dx) && al16(bias), WM_EALIGN);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (act == WM_ACT_NONE) {
    if (dbias == nullptr) return WM_OK;
    return launch_colsum<1>(nullptr, nullptr, dy, rows, C, nullptr, dbias, st);
  }
  WM_REQUIRE(act == WM_ACT_GELU || act == WM_ACT_RELU, WM_EUNSUPPORTED);
  WM_REQUIRE(x && dx, WM_EINVAL);
  return act == WM_ACT_GELU ? launch_colsum<2>(x, bias, dy, rows, C, dx, dbias, st)
                            : launch_colsum<3>(x, bias, dy, rows, C, dx, dbias, st);
}

// Slot forms (bit-reproducible): part [wm_colsum_blocks(rows, C)][C] f32, every slot overwritten; add the slots in order
// (wm_wgrad_fold / wm_wgrad_finalize with K = 1, RS = 1).
extern "C" int wm_colsum_blocks(long long rows, int C) {
  if (rows <= 0 || C <= 0 || C % 8) return 0;
  int CB, slabs, rb;
  long long rpb;
  colsum_geometry(rows, C, CB, slabs, rb, rpb);
  return rb;
}

extern "C" int wm_bias_act_bwd_parts(const void* x, const float* bias, const void* dy, int act, long long rows, int C,
                                     void* dx, float* part, void* stream) {
  WM_REQUIRE(dy && part, WM_EINVAL);
  WM_REQUIRE(rows > 0 && C > 0 && C % 8 == 0, WM_EINVAL);
  WM_REQUIRE(al16(x) && al16(dy) && al16(dx) && al16(bias), WM_EALIGN);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (act == WM_ACT_NONE) return launch_colsum<1>(nullptr, nullptr, dy, rows, C, nullptr, nullptr, st, part);
  WM_REQUIRE(act == WM_ACT_GELU || act == WM_ACT_RELU, WM_EUNSUPPORTED);
  WM_REQUIRE(x && dx, WM_EINVAL);
  return act == WM_ACT_GELU ? launch_colsum<2>(x, bias, dy, rows, C, dx, nullptr, st, part)
                            : launch_colsum<3>(x, bias, dy, rows, C, dx, nullptr, st, part);
}

extern "C" int wm_colsum_bf16(const void* x, long long rows, int C, float* out, int accumulate, void* stream) {
  WM_REQUIRE(x && out, WM_EINVAL);
  WM_REQUIRE(rows > 0 && C > 0 && C % 8 == 0, WM_EINVAL);
  WM_REQUIRE(al16(x), WM_EALIGN);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!accumulate) {
    zero_f32<<<ew_blocks(C), TF_THREADS, 0, st>>>(out, C);
    WM_LAUNCH_CHECK();
  }
  return launch_colsum<1>(nullptr, nullptr, x, rows, C, nullptr, out, st);
}

extern "C" int wm_tokens_assemble(const void* patches, const float* cls, const float* pos, int N, int np, int D,
                                  void* tokens, void* stream) {
  WM_REQUIRE(patches && cls && pos && tokens, WM_EINVAL);
  WM_REQUIRE(N > 0 && np > 0 && D > 0 && D % 8 == 0, WM_EINVAL);
  WM_REQUIRE(al16(patches) && al16(cls) && al16(pos) && al16(tokens), WM_EALIGN);
  tokens_assemble<<<ew_blocks((long long)N * (np + 1) * (D >> 3)), TF_THREADS, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const uint16_t*>(patches), cls, pos, N, np, D, static_cast<uint16_t*>(tokens));
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_patchify(const void* images, int N, int S, int p, void* rows, void* stream) {
  WM_REQUIRE(images && rows, WM_EINVAL);
  WM_REQUIRE(N > 0 && S > 0 && p > 0 && S % p == 0 && (p * 3) % 8 == 0, WM_EINVAL);
  WM_REQUIRE(al16(images) && al16(rows), WM_EALIGN);
  const int g = S / p;
  patchify_kernel<<<ew_blocks((long long)N * g * g * p * ((p * 3) >> 3)), TF_THREADS, 0,
                    static_cast<hipStream_t>(stream)>>>(static_cast<const uint16_t*>(images), N, S, p,
                                                        static_cast<uint16_t*>(rows));
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_gather_rows(const void* x, const long long* idx, int B, int S, int K, int C, void* out,
                              void* stream) {
  WM_REQUIRE(x && idx && out, WM_EINVAL);
  WM_REQUIRE(B > 0 && S > 0 && K > 0 && C > 0 && C % 8 == 0, WM_EINVAL);
  WM_REQUIRE(al16(x) && al16(out), WM_EALIGN);
  rows_by_index<false><<<ew_blocks((long long)B * K * (C >> 3)), TF_THREADS, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const uint16_t*>(x), idx, B, S, K, C, static_cast<uint16_t*>(out));
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_scatter_rows(const void* src, const long long* idx, int B, int S, int K, int C, void* dst,
                               void* stream) {
  WM_REQUIRE(src && idx && dst, WM_EINVAL);
  WM_REQUIRE(B > 0 && S > 0 && K > 0 && C > 0 && C % 8 == 0, WM_EINVAL);
  WM_REQUIRE(al16(src) && al16(dst), WM_EALIGN);
  rows_by_index<true><<<ew_blocks((long long)B * K * (C >> 3)), TF_THREADS, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const uint16_t*>(src), idx, B, S, K, C, static_cast<uint16_t*>(dst));
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_mse_fwd_bwd(const void* pred, const void* target, long long n, float* loss, void* dpred,
                              void* stream) {
  WM_REQUIRE(pred && target && loss && dpred, WM_EINVAL);
  WM_REQUIRE(n > 0 && n % 8 == 0, WM_EINVAL);
  WM_REQUIRE(al16(pred) && al16(target) && al16(dpred), WM_EALIGN);
  mse_kernel<false><<<ew_blocks(n >> 3) > 1024 ? 1024 : ew_blocks(n >> 3), TF_THREADS, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const uint16_t*>(pred), static_cast<const uint16_t*>(target), n, loss,
      static_cast<uint16_t*>(dpred));
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_l1_fwd_bwd(const void* pred, const void* target, long long n, float* loss, void* dpred,
                             void* stream) {
  WM_REQUIRE(pred && target && loss && dpred, WM_EINVAL);
  WM_REQUIRE(n > 0 && n % 8 == 0, WM_EINVAL);
  WM_REQUIRE(al16(pred) && al16(target) && al16(dpred), WM_EALIGN);
  mse_kernel<true><<<ew_blocks(n >> 3) > 1024 ? 1024 : ew_blocks(n >> 3), TF_THREADS, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const uint16_t*>(pred), static_cast<const uint16_t*>(target), n, loss,
      static_cast<uint16_t*>(dpred));
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_dino_teacher_probs(const void* teacher, const float* center, float temp_t, long long rows, int D,
                                     float* probs, void* stream) {
  WM_REQUIRE(teacher && center && probs, WM_EINVAL);
  WM_REQUIRE(rows > 0 && rows < (1ll << 31) && D > 0 && D % 8 == 0 && temp_t > 0.f, WM_EINVAL);
  WM_REQUIRE(D <= TF_THREADS * 8 * DL_MAXV, WM_EUNSUPPORTED);
  WM_REQUIRE(al16(teacher) && al16(center) && al16(probs), WM_EALIGN);
  dino_teacher_probs<<<(int)rows, TF_THREADS, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const uint16_t*>(teacher), center, 1.f / temp_t, D, probs);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

static int dino_loss_impl(const void* student, const float* probs, int Vs, int Vt, int B, int D, float temp_s,
                          int skip_same, float* loss, void* dstudent, void* stream) {
  WM_REQUIRE(student && probs && loss && dstudent, WM_EINVAL);
  WM_REQUIRE(Vs > 0 && Vt > 0 && B > 0 && D > 0 && D % 8 == 0 && temp_s > 0.f, WM_EINVAL);
  WM_REQUIRE(D <= TF_THREADS * 8 * DL_MAXV, WM_EUNSUPPORTED);
  WM_REQUIRE(al16(student) && al16(probs) && al16(dstudent), WM_EALIGN);
  const int ndiag = skip_same ? (Vs < Vt ? Vs : Vt) : 0;
  const int n_terms = Vs * Vt - ndiag;
  WM_REQUIRE(n_terms > 0, WM_EINVAL);
  const float w = 1.f / ((float)n_terms * (float)B);
  dino_loss_kernel<<<Vs * B, TF_THREADS, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const uint16_t*>(student), probs, Vs, Vt, B, D, 1.f / temp_s, w, skip_same, loss,
      static_cast<uint16_t*>(dstudent));
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_dino_loss_fwd_bwd(const void* student, const float* probs, int Vs, int Vt, int B, int D,
                                    float temp_s, float* loss, void* dstudent, void* stream) {
  return dino_loss_impl(student, probs, Vs, Vt, B, D, temp_s, 1, loss, dstudent, stream);
}

extern "C" int wm_soft_cross_entropy_fwd_bwd(const void* student, const float* probs, int Vs, int B, int D, float temp_s,
                                             float* loss, void* dstudent, void* stream) {
  return dino_loss_impl(student, probs, Vs, 1, B, D, temp_s, 0, loss, dstudent, stream);
}

extern "C" int wm_dino_center_update(const void* teacher, long long rows, int D, float momentum, float* center,
                                     void* stream) {
  WM_REQUIRE(teacher && center, WM_EINVAL);
  WM_REQUIRE(rows > 0 && D > 0, WM_EINVAL);
  dino_center_kernel<<<wm_cdiv(D, 32), TF_THREADS, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const uint16_t*>(teacher), rows, D, momentum, center);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                             const float* hyper, void* stream) {
  WM_REQUIRE(params && grads && exp_avg && exp_avg_sq && hyper, WM_EINVAL);
  WM_REQUIRE(n > 0, WM_EINVAL);
  adamw_kernel<<<ew_blocks(n), TF_THREADS, 0, static_cast<hipStream_t>(stream)>>>(params, grads, exp_avg, exp_avg_sq, n,
                                                                                  hyper);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_ema_update(float* ema, const float* params, long long n, float m, void* stream) {
  WM_REQUIRE(ema && params, WM_EINVAL);
  WM_REQUIRE(n > 0, WM_EINVAL);
  ema_kernel<<<ew_blocks(n), TF_THREADS, 0, static_cast<hipStream_t>(stream)>>>(ema, params, n, m);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_colstats(const void* x, int dtype, long long rows, int C, float* mean, float* var, void* stream) {
  WM_REQUIRE(x && mean && var, WM_EINVAL);
  WM_REQUIRE(rows > 0 && C > 0, WM_EINVAL);
  WM_REQUIRE(dtype == WM_F32 || dtype == WM_BF16, WM_EUNSUPPORTED);
  hipStream_t st = static_cast<hipStream_t>(stream);
  return dtype == WM_F32 ? launch_colstats<float>(x, rows, C, mean, var, st)
                         : launch_colstats<uint16_t>(x, rows, C, mean, var, st);
}

extern "C" int wm_standardize(const void* x, int dtype, long long rows, int C, const float* mean,
                              const float* inv_scale, float* out, void* stream) {
  WM_REQUIRE(x && mean && inv_scale && out, WM_EINVAL);
  WM_REQUIRE(rows > 0 && C > 0, WM_EINVAL);
  WM_REQUIRE(dtype == WM_F32 || dtype == WM_BF16, WM_EUNSUPPORTED);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int blocks = ew_blocks(rows * C);
  if (dtype == WM_F32)
    standardize_kernel<float><<<blocks, TF_THREADS, 0, st>>>(static_cast<const float*>(x), rows, C, mean, inv_scale, out);
  else
    standardize_kernel<uint16_t><<<blocks, TF_THREADS, 0, st>>>(static_cast<const uint16_t*>(x), rows, C, mean,
                                                                inv_scale, out);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_mean_entropy_reg_fwd_bwd(const void* logits, const float* log_prior, int N, int K, float temperature,
                                           float* loss, float* dlogits, float* mean_ws, void* stream) {
  WM_REQUIRE(logits && loss && dlogits && mean_ws, WM_EINVAL);
  WM_REQUIRE(N > 0 && K > 0 && K % 8 == 0 && temperature > 0.f, WM_EINVAL);
  WM_REQUIRE(K <= TF_THREADS * 8 * DL_MAXV, WM_EUNSUPPORTED);
  WM_REQUIRE(al16(logits), WM_EALIGN);
  hipStream_t st = static_cast<hipStream_t>(stream);
  zero_f32<<<ew_blocks(K), TF_THREADS, 0, st>>>(mean_ws, K);
  WM_LAUNCH_CHECK();
  memax_mean_kernel<<<N, TF_THREADS, 0, st>>>(static_cast<const uint16_t*>(logits), K, 1.f / temperature, 1.f / (float)N,
                                             mean_ws);
  WM_LAUNCH_CHECK();
  memax_grad_kernel<<<N, TF_THREADS, 0, st>>>(static_cast<const uint16_t*>(logits), mean_ws, log_prior, K,
                                             1.f / temperature, 1.f / (float)N, loss, dlogits);
  WM_LAUNCH_CHECK();
  return WM_OK;
}


// ------------------------------------------------------------------------------------ small f32 matmul
// C [M][N] = op(A) B, op(A) = A [M][K] or A^T (A stored [K][M]); B [K][N]; all float32, row-major, any sizes.
// For the parameter-side products of the transformers (dino / lightly interpolate_pos_encoding as a fixed
// [g'^2][g^2] matrix times the [g^2][D] position table: 36 x 196 x 384, and its transpose product in the backward
// pass): a few MFLOP, not worth bf16 rounding or a library GEMM.  16 x 16 tiles through LDS, one thread per output.
namespace {
template <bool TA>
__global__ __launch_bounds__(256) void small_matmul_f32(const float* __restrict__ A, const float* __restrict__ B, int M,
                                                        int N, int K, float* __restrict__ C) {
  __shared__ float sa[16][17], sb[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int row = blockIdx.y * 16 + ty, col = blockIdx.x * 16 + tx;
  float acc = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    const int ka = k0 + tx, kb = k0 + ty;
    sa[ty][tx] = (row < M && ka < K) ? (TA ? A[(size_t)ka * M + row] : A[(size_t)row * K + ka]) : 0.f;
    sb[ty][tx] = (kb < K && col < N) ? B[(size_t)kb * N + col] : 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = fmaf(sa[ty][k], sb[k][tx], acc);
    __syncthreads();
  }
  if (row < M && col < N) C[(size_t)row * N + col] = acc;
}
}  // namespace

extern "C" int wm_matmul_f32(const float* a, const float* b, float* c, int M, int N, int K, int trans_a, void* stream) {
  WM_REQUIRE(a && b && c, WM_EINVAL);
  WM_REQUIRE(M > 0 && N > 0 && K > 0, WM_EINVAL);
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 grid(wm_cdiv(N, 16), wm_cdiv(M, 16));
  if (trans_a) small_matmul_f32<true><<<grid, 256, 0, st>>>(a, b, M, N, K, c);
  else small_matmul_f32<false><<<grid, 256, 0, st>>>(a, b, M, N, K, c);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

// ---- per-row ascending argsort of up to 256 float keys (lightly.models.utils.random_token_mask: the permutation of an
// image's tokens is the argsort of uniform noise; reference scripts/WM811k_benchmark.py:930).  One block per row, a
// bitonic network over (key, index) pairs in LDS; ties go to the lower index (a stable ascending order), rows are
// padded to 256 with +inf.
namespace {
__global__ __launch_bounds__(256) void argsort_rows_kernel(const float* __restrict__ keys, int S, long long* __restrict__ out) {
  __shared__ float sk[256];
  __shared__ int si[256];
  const int t = threadIdx.x;
  const size_t row = blockIdx.x;
  sk[t] = t < S ? keys[row * S + t] : INFINITY;
  si[t] = t;
  __syncthreads();
  for (int k = 2; k <= 256; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int p = t ^ j;
      if (p > t) {
        const float a = sk[t], b = sk[p];
        const int ia = si[t], ib = si[p];
        const bool a_after_b = a > b || (a == b && ia > ib);
        const bool up = (t & k) == 0;
        if (a_after_b == up) {
          sk[t] = b; sk[p] = a;
          si[t] = ib; si[p] = ia;
        }
      }
      __syncthreads();
    }
  }
  if (t < S) out[row * S + t] = si[t];
}

// y[r][c] = x[r][c] * g[c] (bf16 x / y, f32 g): the trainable gain of a weight-normalised Linear
// (DINOProjectionHead(norm_last_layer=False)).  Backward: dx = dy * g, dg[c] = sum_r dy[r][c] * x[r][c] (one thread per
// column walks the rows: a few hundred rows, coalesced across columns, deterministic).
__global__ __launch_bounds__(256) void colscale_fwd_kernel(const uint16_t* __restrict__ x, const float* __restrict__ g,
                                                           long long rows, int C, uint16_t* __restrict__ y) {
  const long long total = rows * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
    y[i] = f2bf(bf2f(x[i]) * g[i % C]);
}
__global__ __launch_bounds__(256) void colscale_bwd_kernel(const uint16_t* __restrict__ x, const float* __restrict__ g,
                                                           const uint16_t* __restrict__ dy, long long rows, int C,
                                                           uint16_t* __restrict__ dx, float* __restrict__ dg, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float gc = g[c];
  float acc = 0.f;
  for (long long r = 0; r < rows; ++r) {
    const float d = bf2f(dy[r * C + c]);
    acc = fmaf(d, bf2f(x[r * C + c]), acc);
    dx[r * C + c] = f2bf(d * gc);
  }
  dg[c] = accumulate ? dg[c] + acc : acc;
}
}  // namespace

extern "C" int wm_argsort_rows(const float* keys, int rows, int S, long long* out, void* stream) {
  WM_REQUIRE(keys && out && rows > 0 && S > 0, WM_EINVAL);
  WM_REQUIRE(S <= 256, WM_EUNSUPPORTED);
  argsort_rows_kernel<<<rows, 256, 0, static_cast<hipStream_t>(stream)>>>(keys, S, out);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_colscale_fwd(const void* x, const float* g, long long rows, int C, void* y, void* stream) {
  WM_REQUIRE(x && g && y && rows > 0 && C > 0, WM_EINVAL);
  long long b = (rows * C + 255) / 256;
  colscale_fwd_kernel<<<(int)(b < 4096 ? b : 4096), 256, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const uint16_t*>(x), g, rows, C, static_cast<uint16_t*>(y));
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_colscale_bwd(const void* x, const float* g, const void* dy, long long rows, int C, void* dx, float* dg,
                               int accumulate, void* stream) {
  WM_REQUIRE(x && g && dy && dx && dg && rows > 0 && C > 0, WM_EINVAL);
  colscale_bwd_kernel<<<wm_cdiv(C, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const uint16_t*>(x), g, static_cast<const uint16_t*>(dy), rows, C, static_cast<uint16_t*>(dx), dg, accumulate);
  WM_LAUNCH_CHECK();
  return WM_OK;
}
