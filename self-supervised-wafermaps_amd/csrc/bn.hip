// Batch normalisation (training + eval), fused with ReLU and the residual add, on NHWC bf16.
//
// Replaces the BatchNorm2d / ReLU / residual-add chain of timm's ResNet-18 BasicBlocks and the
// BatchNorm1d of lightly's SimCLRProjectionHead in the reference (scripts/WM811k_benchmark.py:
// 231-240).  A tensor is viewed as [rows][C]; the batch may be cut into G equal row groups with
// independent statistics: the reference runs forward(x0) and forward(x1) as two calls, so the two
// views are normalised separately — one launch over the concatenated [2B] batch with G = 2
// reproduces that exactly.
//
//   forward : partial sums (deterministic two-level reduction, no atomics) -> finalize
//             (mean/invstd in double, running-stat update, per-channel scale/shift)
//             -> apply: out = relu?(y*scale + shift (+ residual))
//   backward: dz = dout * (out > 0);  s1 = sum dz, s2 = sum dz*xhat  ->  finalize (dgamma, dbeta,
//             coefficients) -> apply: dy = gamma*invstd*(dz - s1/M - xhat*s2/M), optional dz copy
//             (the gradient of the residual branch).
//
// Roofline: HBM.  Every pass moves 16 bytes per lane; a row is covered by C/8 adjacent lanes.
#include "common.h"

namespace {

constexpr int BN_THREADS = 256;

__device__ __forceinline__ void unpack8(const uint4 v, float (&f)[8]) {
  f[0] = bf2f((uint16_t)(v.x & 0xffff)); f[1] = bf2f((uint16_t)(v.x >> 16));
  f[2] = bf2f((uint16_t)(v.y & 0xffff)); f[3] = bf2f((uint16_t)(v.y >> 16));
  f[4] = bf2f((uint16_t)(v.z & 0xffff)); f[5] = bf2f((uint16_t)(v.z >> 16));
  f[6] = bf2f((uint16_t)(v.w & 0xffff)); f[7] = bf2f((uint16_t)(v.w >> 16));
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
}

// Optional gradient source for the stem: instead of a materialised dout [rows][C], the gradient of
// max_pool3x3s2 is gathered on the fly from the POOLED gradient and the recorded window positions
// (what maxpool_bwd would have written): dout(n,h,w,c) = sum over the <= 4 windows covering (h,w)
// of [idx == position] * dyp.  Saves writing and twice re-reading the 112x112x64 tensor.
struct PoolSrc {
  const uint16_t* dy;   // [N][P][Q][C] pooled gradient, or NULL
  const uint8_t* idx;   // [N][P][Q][C] window positions
  int H, W, P, Q;
};

__device__ __forceinline__ void pool_grad8(const PoolSrc& ps, uint32_t row, int C, int c0, float (&g)[8]) {
  // rows < 2^31 (checked by the entry point): 32-bit divisions only
  const uint32_t hw = (uint32_t)(ps.H * ps.W);
  const int n = (int)(row / hw);
  const uint32_t rem = row - (uint32_t)n * hw;
  const int h = (int)(rem / (uint32_t)ps.W), w = (int)(rem - (uint32_t)h * (uint32_t)ps.W);
#pragma unroll
  for (int e = 0; e < 8; ++e) g[e] = 0.f;
  // the (at most) 2 x 2 covering windows, fully unrolled with clamped addresses so that all eight
  // loads are in flight together; an invalid window gets a code that never matches
  const int p_lo = h >> 1, q_lo = w >> 1;
  uint2 pk[4];
  uint4 dv[4];
  int code[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = p_lo + (j >> 1), q = q_lo + (j & 1);
    const bool ok = ((j >> 1) == 0 || (h & 1)) && ((j & 1) == 0 || (w & 1)) && p < ps.P && q < ps.Q;
    const int pc = p < ps.P ? p : ps.P - 1, qc = q < ps.Q ? q : ps.Q - 1;
    code[j] = ok ? (h - (2 * p - 1)) * 3 + (w - (2 * q - 1)) : 255;
    const size_t o = ((size_t)(n * ps.P + pc) * ps.Q + qc) * C + c0;
    pk[j] = *reinterpret_cast<const uint2*>(ps.idx + o);
    dv[j] = *reinterpret_cast<const uint4*>(ps.dy + o);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float d[8];
    unpack8(dv[j], d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ie = (int)(((e < 4 ? pk[j].x : pk[j].y) >> (8 * (e & 3))) & 0xff);
      g[e] += ie == code[j] ? d[e] : 0.f;
    }
  }
  // maxpool_bwd stores bf16: round like it does so that both paths see identical gradients
#pragma unroll
  for (int e = 0; e < 8; ++e) g[e] = bf2f(f2bf(g[e]));
}

// Two per-channel sums over a row range -> part[((g*nblk + blk)*2 + which)*C + c].
// MODE 0: (sum y, sum y^2).  MODE 1: (sum dz, sum dz*xhat) with dz = dout*(out>0 | no mask).
// MODE 1 with `out` NULL and gamma_m non-NULL: the ReLU mask is recomputed from y as
// bf16(y*scale + shift) > 0 with scale = gamma*invstd, shift = beta - mean*scale (the forward's own
// arithmetic), which saves reading `out` when the BN had no residual input.
template <int MODE>
__global__ __launch_bounds__(BN_THREADS) void bn_reduce(const uint16_t* __restrict__ y,
                                                        const uint16_t* __restrict__ dout,
                                                        const uint16_t* __restrict__ out,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ invstd,
                                                        const float* __restrict__ gamma_m,
                                                        const float* __restrict__ beta_m,
                                                        int rows_per_group, int C, int rows_per_block,
                                                        float* __restrict__ part, const PoolSrc ps) {
  extern __shared__ float red[];  // [2][rpp][C]
  const int tid = threadIdx.x;
  const int tpr = C >> 3;  // threads per row
  const int g = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  const int r_begin = blk * rows_per_block;
  int r_end = r_begin + rows_per_block;
  if (r_end > rows_per_group) r_end = rows_per_group;
  const size_t gbase = (size_t)g * rows_per_group;

  if (tpr <= BN_THREADS) {
    const int rpp = BN_THREADS / tpr;
    const int cidx = tid % tpr, rr = tid / tpr;
    float mu[8], is[8], msc[8], msh[8];
    const bool remask = MODE == 1 && out == nullptr && gamma_m != nullptr;
    if (MODE == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        mu[e] = mean[(size_t)g * C + cidx * 8 + e];
        is[e] = invstd[(size_t)g * C + cidx * 8 + e];
        msc[e] = msh[e] = 0.f;
        if (remask) {
          msc[e] = gamma_m[cidx * 8 + e] * is[e];
          msh[e] = beta_m[cidx * 8 + e] - mu[e] * msc[e];
        }
      }
    }
    if (rr < rpp) {
      int r = r_begin + rr;
      if (MODE == 1 && ps.dy == nullptr) {
        // four rows per trip: their 8-12 loads are independent and issued together (one row per trip
        // leaves a single 16-byte load per lane in flight and the pass latency-bound at ~3 TB/s)
        for (; r + 3 * rpp < r_end; r += 4 * rpp) {
          uint4 vy[4], vd[4], vo[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const size_t off = (gbase + r + u * rpp) * C + cidx * 8;
            vy[u] = *reinterpret_cast<const uint4*>(y + off);
            vd[u] = *reinterpret_cast<const uint4*>(dout + off);
            if (out) vo[u] = *reinterpret_cast<const uint4*>(out + off);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            float fy[8], fd[8];
            unpack8(vy[u], fy);
            unpack8(vd[u], fd);
            if (out) {
              float fo[8];
              unpack8(vo[u], fo);
#pragma unroll
              for (int e = 0; e < 8; ++e) fd[e] = fo[e] > 0.f ? fd[e] : 0.f;
            } else if (remask) {
#pragma unroll
              for (int e = 0; e < 8; ++e) fd[e] = bf2f(f2bf(fmaf(fy[e], msc[e], msh[e]))) > 0.f ? fd[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              s1[e] += fd[e];
              s2[e] = fmaf(fd[e], (fy[e] - mu[e]) * is[e], s2[e]);
            }
          }
        }
      }
      for (; r < r_end; r += rpp) {
        const size_t off = (gbase + r) * C + cidx * 8;
        float fy[8];
        unpack8(*reinterpret_cast<const uint4*>(y + off), fy);
        if (MODE == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s1[e] += fy[e];
            s2[e] = fmaf(fy[e], fy[e], s2[e]);
          }
        } else {
          float fd[8];
          if (ps.dy != nullptr) pool_grad8(ps, (uint32_t)(gbase + r), C, cidx * 8, fd);
          else unpack8(*reinterpret_cast<const uint4*>(dout + off), fd);
          if (out) {
            float fo[8];
            unpack8(*reinterpret_cast<const uint4*>(out + off), fo);
#pragma unroll
            for (int e = 0; e < 8; ++e) fd[e] = fo[e] > 0.f ? fd[e] : 0.f;
          } else if (remask) {
#pragma unroll
            for (int e = 0; e < 8; ++e) fd[e] = bf2f(f2bf(fmaf(fy[e], msc[e], msh[e]))) > 0.f ? fd[e] : 0.f;
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s1[e] += fd[e];
            s2[e] = fmaf(fd[e], (fy[e] - mu[e]) * is[e], s2[e]);
          }
        }
      }
    }
    float* r1 = red;
    float* r2 = red + rpp * C;
    if (rr < rpp) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        r1[rr * C + cidx * 8 + e] = s1[e];
        r2[rr * C + cidx * 8 + e] = s2[e];
      }
    }
    __syncthreads();
    for (int c = tid; c < C; c += BN_THREADS) {
      float a = 0.f, b = 0.f;
      for (int q = 0; q < rpp; ++q) {
        a += r1[q * C + c];
        b += r2[q * C + c];
      }
      part[((size_t)(g * nblk + blk) * 2 + 0) * C + c] = a;
      part[((size_t)(g * nblk + blk) * 2 + 1) * C + c] = b;
    }
  }
}

// Reduce the [nblk] partial sums of (group, channel) pairs: 32 lanes of a 1024-thread block stride over the partials
// of 32 adjacent channels (128-byte coalesced rows), then combine in LDS.  TWO groups (the two views of a training
// step) go through together: their loads are in flight at the same time and they share the barriers, which is most
// of what these few-microsecond launches cost (44 of them per SimCLR step).
__device__ __forceinline__ void finalize_sums2(float* __restrict__ part, int nblk, int g, int ng, int C, int c,
                                               int bl, int cl, double (*red)[2][32][33], double (&s)[2],
                                               double (&ss)[2], bool clear = false) {
  double a[2] = {0.0, 0.0}, b[2] = {0.0, 0.0};
  if (c < C) {
    for (int k = bl; k < nblk; k += 32) {
      float va[2] = {0.f, 0.f}, vb[2] = {0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u < ng) {
          float* p0 = part + ((size_t)((g + u) * nblk + k) * 2 + 0) * C + c;
          float* p1 = part + ((size_t)((g + u) * nblk + k) * 2 + 1) * C + c;
          va[u] = *p0;
          vb[u] = *p1;
          if (clear) {  // read-and-clear
            *p0 = 0.f;
            *p1 = 0.f;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        a[u] += (double)va[u];
        b[u] += (double)vb[u];
      }
    }
  }
  __syncthreads();  // previous pair's readers are done with `red`
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    red[0][u][bl][cl] = a[u];
    red[1][u][bl][cl] = b[u];
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    s[u] = 0.0;
    ss[u] = 0.0;
  }
  if (bl == 0) {
    for (int k = 0; k < 32; ++k) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        s[u] += red[0][u][k][cl];
        ss[u] += red[1][u][k][cl];
      }
    }
  }
}

// Forward finalize: grid = ceil(C/32) blocks of 1024 threads.
__global__ __launch_bounds__(1024) void bn_fwd_finalize(
    float* __restrict__ part, int nblk, int G, int C, int rows_per_group, int clear,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
    float* __restrict__ running_mean, float* __restrict__ running_var, long long* __restrict__ num_batches_tracked,
    float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift) {
  __shared__ double red[2][2][32][33];
  // torch's num_batches_tracked += 1 per forward call; the G groups are G calls (one lane of the launch)
  // (an integer atomic: two views of a step may run as parallel branches and count on the same module)
  if (num_batches_tracked != nullptr && blockIdx.x == 0 && threadIdx.x == 0)
    atomicAdd(reinterpret_cast<unsigned long long*>(num_batches_tracked), (unsigned long long)G);
  const int bl = threadIdx.x >> 5, cl = threadIdx.x & 31;
  const int c = blockIdx.x * 32 + cl;
  const bool owner = bl == 0 && c < C;
  float rm = 0.f, rv = 0.f;
  if (owner) {
    rm = running_mean ? running_mean[c] : 0.f;
    rv = running_var ? running_var[c] : 0.f;
  }
  for (int g0 = 0; g0 < G; g0 += 2) {
    const int ng = G - g0 < 2 ? G - g0 : 2;
    double s2[2], ss2[2];
    finalize_sums2(part, nblk, g0, ng, C, c, bl, cl, red, s2, ss2, clear != 0);
    if (owner) {
      for (int u = 0; u < ng; ++u) {
        const int g = g0 + u;
        const double s = s2[u], ss = ss2[u];
        const double m = s / rows_per_group;
        double var = ss / rows_per_group - m * m;
        if (var < 0.0) var = 0.0;
        const float fm = (float)m, is = (float)(1.0 / sqrt(var + (double)eps));
        mean[(size_t)g * C + c] = fm;
        invstd[(size_t)g * C + c] = is;
        const float sc = (gamma ? gamma[c] : 1.f) * is;
        scale[(size_t)g * C + c] = sc;
        shift[(size_t)g * C + c] = (beta ? beta[c] : 0.f) - fm * sc;
        // torch: running = (1-momentum)*running + momentum*stat, unbiased variance; the G groups
        // are the reference's G consecutive forward calls
        const double unb = rows_per_group > 1 ? var * rows_per_group / (rows_per_group - 1.0) : var;
        rm = (1.f - momentum) * rm + momentum * fm;
        rv = (1.f - momentum) * rv + momentum * (float)unb;
      }
    }
  }
  if (owner) {
    if (running_mean) running_mean[c] = rm;
    if (running_var) running_var[c] = rv;
  }
}

// Partial sums [G][nblk][2][C] -> their totals [G][2][C] (double accumulation in slot order, rounded once): what a
// rank contributes to the cross-rank all-reduce of synchronised BatchNorm.
__global__ __launch_bounds__(1024) void bn_sums_finalize(float* __restrict__ part, int nblk, int G, int C,
                                                          float* __restrict__ sums) {
  __shared__ double red[2][2][32][33];
  const int bl = threadIdx.x >> 5, cl = threadIdx.x & 31;
  const int c = blockIdx.x * 32 + cl;
  const bool owner = bl == 0 && c < C;
  for (int g0 = 0; g0 < G; g0 += 2) {
    const int ng = G - g0 < 2 ? G - g0 : 2;
    double a[2], b[2];
    finalize_sums2(part, nblk, g0, ng, C, c, bl, cl, red, a, b);
    if (owner) {
      for (int u = 0; u < ng; ++u) {
        sums[((size_t)(g0 + u) * 2 + 0) * C + c] = (float)a[u];
        sums[((size_t)(g0 + u) * 2 + 1) * C + c] = (float)b[u];
      }
    }
  }
}

// Eval-mode scale/shift from running statistics.
__global__ void bn_eval_params(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ running_mean,
                               const float* __restrict__ running_var, float eps, int C,
                               float* __restrict__ scale, float* __restrict__ shift) {
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float is = 1.0f / sqrtf(running_var[c] + eps);
    const float sc = (gamma ? gamma[c] : 1.f) * is;
    scale[c] = sc;
    shift[c] = (beta ? beta[c] : 0.f) - running_mean[c] * sc;
  }
}

// POW2: C/8 is a power of two dividing the block size, so a thread keeps its channel chunk for the whole
// grid-stride loop: no divisions, and the per-channel coefficients are loaded once per statistics
// group instead of once per 16-byte item.
template <bool POW2>
__global__ __launch_bounds__(BN_THREADS) void bn_apply(const uint16_t* __restrict__ y,
                                                       const uint16_t* __restrict__ residual,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift,
                                                       long long rows, int C, int rows_per_group,
                                                       int relu, int cshift, uint16_t* __restrict__ out,
                                                       uint8_t* __restrict__ relu_mask) {
  // relu_mask (optional): [rows][C / 8] bytes, bit e of byte (row, chunk) = output channel 8 chunk + e is > 0 -- what
  // the consuming convolution's dgrad epilogue needs of this tensor for the ReLU's backward (1/16 of its bytes)
  const int cpr = C >> 3;
  const long long total = rows * cpr;
  int cur_g = -1;
  float sc[8], sh[8];
  for (long long p = (long long)blockIdx.x * BN_THREADS + threadIdx.x; p < total;
       p += (long long)gridDim.x * BN_THREADS) {
    long long row;
    int c0, g;
    if (POW2) {
      row = p >> cshift;
      c0 = (int)(p & (cpr - 1)) * 8;
      g = 0;
      for (long long lim = rows_per_group; row >= lim; lim += rows_per_group) ++g;
    } else {
      row = p / cpr;
      c0 = (int)(p - row * cpr) * 8;
      g = (int)(row / rows_per_group);
    }
    if (!POW2 || g != cur_g) {
      const float4 sa = *reinterpret_cast<const float4*>(scale + (size_t)g * C + c0);
      const float4 sb = *reinterpret_cast<const float4*>(scale + (size_t)g * C + c0 + 4);
      const float4 ha = *reinterpret_cast<const float4*>(shift + (size_t)g * C + c0);
      const float4 hb = *reinterpret_cast<const float4*>(shift + (size_t)g * C + c0 + 4);
      sc[0] = sa.x; sc[1] = sa.y; sc[2] = sa.z; sc[3] = sa.w; sc[4] = sb.x; sc[5] = sb.y; sc[6] = sb.z; sc[7] = sb.w;
      sh[0] = ha.x; sh[1] = ha.y; sh[2] = ha.z; sh[3] = ha.w; sh[4] = hb.x; sh[5] = hb.y; sh[6] = hb.z; sh[7] = hb.w;
      cur_g = g;
    }
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(y + row * C + c0), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], sc[e], sh[e]);
    if (residual) {
      float r[8];
      unpack8(*reinterpret_cast<const uint4*>(residual + row * C + c0), r);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += r[e];
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
    }
    const uint4 pk = pack8(f);
    *reinterpret_cast<uint4*>(out + row * C + c0) = pk;
    if (relu_mask != nullptr) {
      float r8[8];
      unpack8(pk, r8);  // the stored (rounded) values decide, exactly as a test of the tensor itself would
      uint32_t m = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) m |= (r8[e] > 0.f ? 1u : 0u) << e;
      relu_mask[row * cpr + (c0 >> 3)] = (uint8_t)m;
    }
  }
}

// Backward finalize: dgamma/dbeta and the per-(group, channel) coefficients of the apply pass.
// coef[(g*7 + t)*C + c]: t = 0 mean, 1 invstd, 2 gamma*invstd, 3 s1/M, 4 s2/M, 5/6 scale/shift of the
// forward (for the recomputed ReLU mask).
__global__ __launch_bounds__(1024) void bn_bwd_finalize(
    float* __restrict__ part, int nblk, int G, int C, int rows_per_group,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
    const float* __restrict__ invstd, float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
    float* __restrict__ coef, int fx) {  // fx: the second sum is sum g * (y - mean) (dgrad epilogue), not sum g * xhat
  __shared__ double red[2][2][32][33];
  const int bl = threadIdx.x >> 5, cl = threadIdx.x & 31;
  const int c = blockIdx.x * 32 + cl;
  const bool owner = bl == 0 && c < C;
  double tg = 0.0, tb = 0.0;
  for (int g0 = 0; g0 < G; g0 += 2) {
    const int ng = G - g0 < 2 ? G - g0 : 2;
    double s1v[2], s2v[2];
    finalize_sums2(part, nblk, g0, ng, C, c, bl, cl, red, s1v, s2v);
    if (owner) {
      for (int u = 0; u < ng; ++u) {
        const int g = g0 + u;
        double s1 = s1v[u], s2 = s2v[u];
        if (fx) {  // the dgrad epilogue accumulated sum g * (y - mean): sum g * xhat = invstd * that
          s2 = (double)invstd[(size_t)g * C + c] * s2;
        }
        tb += s1;
        tg += s2;
        const float is = invstd[(size_t)g * C + c];
        const float mu = mean[(size_t)g * C + c];
        const float sc = (gamma ? gamma[c] : 1.f) * is;
        coef[((size_t)g * 7 + 0) * C + c] = mu;
        coef[((size_t)g * 7 + 1) * C + c] = is;
        coef[((size_t)g * 7 + 2) * C + c] = sc;
        coef[((size_t)g * 7 + 3) * C + c] = (float)(s1 / rows_per_group);
        coef[((size_t)g * 7 + 4) * C + c] = (float)(s2 / rows_per_group);
        coef[((size_t)g * 7 + 5) * C + c] = sc;
        coef[((size_t)g * 7 + 6) * C + c] = (beta ? beta[c] : 0.f) - mu * sc;
      }
    }
  }
  if (owner) {
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)tg;
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)tb;
  }
}

// FORM 0: gradient tensor in, nothing else (no ReLU mask, no dz copy: the form behind the dgrad epilogues, 21 launches of
// a SimCLR step); 1: + the optional ReLU mask (from `out` or recomputed) and dz output; 2: the gradient is gathered from a
// pooled gradient (PoolSrc).  Separate instantiations because the pass is bound by the bytes it keeps in flight: with
// everything compiled into one kernel it needed 142 registers (three waves per SIMD); form 0 runs five.
template <bool POW2, int FORM>
__global__ __launch_bounds__(BN_THREADS, FORM == 0 ? 4 : 1) void bn_bwd_apply(const uint16_t* __restrict__ y,
                                                           const uint16_t* __restrict__ dout,
                                                           const uint16_t* __restrict__ out,
                                                           const float* __restrict__ coef,
                                                           long long rows, int C, int rows_per_group,
                                                           int remask, int cshift, uint16_t* __restrict__ dy,
                                                           uint16_t* __restrict__ dz, const PoolSrc ps) {
  const int cpr = C >> 3;
  const long long total = rows * cpr;
  int cur_g = -1;
  constexpr int NK = FORM == 0 ? 5 : 7;
  float k[NK][8];  // mean, invstd, gamma*invstd, s1/M, s2/M, (forward scale, forward shift: the recomputed mask)
  for (long long p = (long long)blockIdx.x * BN_THREADS + threadIdx.x; p < total;
       p += (long long)gridDim.x * BN_THREADS) {
    long long row;
    int c0, g;
    if (POW2) {
      row = p >> cshift;
      c0 = (int)(p & (cpr - 1)) * 8;
      g = 0;
      for (long long lim = rows_per_group; row >= lim; lim += rows_per_group) ++g;
    } else {
      row = p / cpr;
      c0 = (int)(p - row * cpr) * 8;
      g = (int)(row / rows_per_group);
    }
    if (!POW2 || g != cur_g) {
      const float* cf = coef + (size_t)g * 7 * C + c0;
#pragma unroll
      for (int t = 0; t < NK; ++t) {
        const float4 a = *reinterpret_cast<const float4*>(cf + (size_t)t * C);
        const float4 b = *reinterpret_cast<const float4*>(cf + (size_t)t * C + 4);
        k[t][0] = a.x; k[t][1] = a.y; k[t][2] = a.z; k[t][3] = a.w;
        k[t][4] = b.x; k[t][5] = b.y; k[t][6] = b.z; k[t][7] = b.w;
      }
      cur_g = g;
    }
    const size_t off = row * C + c0;
    float fy[8], fd[8];
    unpack8(*reinterpret_cast<const uint4*>(y + off), fy);
    if constexpr (FORM == 2) pool_grad8(ps, (uint32_t)row, C, c0, fd);
    else unpack8(*reinterpret_cast<const uint4*>(dout + off), fd);
    if constexpr (FORM != 0) {
      if (out) {
        float fo[8];
        unpack8(*reinterpret_cast<const uint4*>(out + off), fo);
#pragma unroll
        for (int e = 0; e < 8; ++e) fd[e] = fo[e] > 0.f ? fd[e] : 0.f;
      }
      if (!out && remask) {
#pragma unroll
        for (int e = 0; e < 8; ++e) fd[e] = bf2f(f2bf(fmaf(fy[e], k[NK - 2][e], k[NK - 1][e]))) > 0.f ? fd[e] : 0.f;
      }
      if (dz) *reinterpret_cast<uint4*>(dz + off) = pack8(fd);
    }
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (fy[e] - k[0][e]) * k[1][e];
      r[e] = k[2][e] * (fd[e] - k[3][e] - xh * k[4][e]);
    }
    *reinterpret_cast<uint4*>(dy + off) = pack8(r);
  }
}

// Stem variant of the apply pass (max-pool 3x3 / stride 2 / pad 1 source, H and W even): one thread owns
// a 2 x 2 pixel quad x 8 channels.  The quad's pixels are covered by the four windows (a, b), (a, b+1),
// (a+1, b), (a+1, b+1) only, so the pooled gradient and the window positions are read 4 times per
// quad instead of 9 (1 + 2 + 2 + 4 per pixel), with fixed position codes per (pixel, window).
__global__ __launch_bounds__(BN_THREADS) void bn_pool_bwd_apply(const uint16_t* __restrict__ y,
                                                                const float* __restrict__ coef, int n_coef, int N, int C,
                                                                int imgs_per_group, int remask,
                                                                uint16_t* __restrict__ dy, const PoolSrc ps,
                                                                const WmDiv d_cpr, const WmDiv d_wb, const WmDiv d_ha,
                                                                const WmDiv d_ipg) {
  // 32-bit index arithmetic (host-checked range): the kernel is VALU-bound, see pool.hip
  const uint32_t cpr = (uint32_t)C >> 3;
  const uint32_t HA = (uint32_t)ps.H >> 1, WB = (uint32_t)ps.W >> 1;
  const uint32_t total = (uint32_t)N * HA * WB * cpr;
  // The coefficient rows of every group sit in LDS ([G][7][C] floats, a few KB) and are read where they are used: held in
  // registers across the loop (first build: 56 of them) the kernel needed 196 registers -- two waves per SIMD for a pass
  // that moves 2.3 GB.
  extern __shared__ __attribute__((aligned(16))) float pool_coef[];
  for (int i = threadIdx.x; i < n_coef; i += BN_THREADS) pool_coef[i] = coef[i];
  __syncthreads();
  for (uint32_t t = blockIdx.x * BN_THREADS + threadIdx.x; t < total; t += gridDim.x * BN_THREADS) {
    uint32_t rc, rb, ra;
    uint32_t u = wm_divmod(t, d_cpr, rc);
    const int c0 = (int)rc * 8;
    u = wm_divmod(u, d_wb, rb);
    const int n = (int)wm_divmod(u, d_ha, ra);
    const int b = (int)rb, a = (int)ra;
    const int g = (int)wm_div((uint32_t)n, d_ipg);
    const float* kk = pool_coef + (size_t)g * 7 * C + c0;
    auto krow = [&](int q, float (&v)[8]) {
      const float4 lo = *reinterpret_cast<const float4*>(kk + q * C);
      const float4 hi = *reinterpret_cast<const float4*>(kk + q * C + 4);
      v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    };
    // the four windows (clamped addresses; an out-of-range window contributes nothing)
    uint2 pk[4];
    uint4 dv[4];
    bool wok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = a + (j >> 1), q = b + (j & 1);
      wok[j] = p < ps.P && q < ps.Q;
      const int pc = p < ps.P ? p : ps.P - 1, qc = q < ps.Q ? q : ps.Q - 1;
      const size_t o = ((size_t)(n * ps.P + pc) * ps.Q + qc) * C + c0;
      pk[j] = *reinterpret_cast<const uint2*>(ps.idx + o);
      dv[j] = *reinterpret_cast<const uint4*>(ps.dy + o);
    }
    float dw[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j) unpack8(dv[j], dw[j]);
    // pixel (dh, dw) of the quad, window j = (jp, jq): position code (2 dh + 1 - 2 jp... ) = kh * 3 + kw with
    // kh = 2a + dh - (2 (a + jp) - 1) = dh + 1 - 2 jp, kw likewise; valid when 0 <= kh, kw <= 2
#pragma unroll
    for (int pix = 0; pix < 4; ++pix) {
      const int dh = pix >> 1, dwp = pix & 1;
      float fd[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) fd[e] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kh = dh + 1 - 2 * (j >> 1), kw = dwp + 1 - 2 * (j & 1);
        if (kh < 0 || kw < 0) continue;  // compile-time after unrolling
        const int code = kh * 3 + kw;
        if (wok[j]) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int ie = (int)(((e < 4 ? pk[j].x : pk[j].y) >> (8 * (e & 3))) & 0xff);
            fd[e] += ie == code ? dw[j][e] : 0.f;
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) fd[e] = bf2f(f2bf(fd[e]));  // as maxpool_bwd would have stored it
      const size_t off = ((size_t)(n * ps.H + 2 * a + dh) * ps.W + 2 * b + dwp) * C + c0;
      float fy[8];
      unpack8(*reinterpret_cast<const uint4*>(y + off), fy);
      if (remask) {
        float k5[8], k6[8];
        krow(5, k5);
        krow(6, k6);
#pragma unroll
        for (int e = 0; e < 8; ++e) fd[e] = bf2f(f2bf(fmaf(fy[e], k5[e], k6[e]))) > 0.f ? fd[e] : 0.f;
      }
      float r[8], k0[8], k1[8];
      krow(0, k0);
      krow(1, k1);
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = (fy[e] - k0[e]) * k1[e];  // xhat
      krow(4, k0);
      krow(3, k1);
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = fd[e] - k1[e] - r[e] * k0[e];
      krow(2, k0);
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = k0[e] * r[e];
      *reinterpret_cast<uint4*>(dy + off) = pack8(r);
    }
  }
}

// Tile slots [G][T][2][C] -> [G][T / R][2][C]: every output element is the sum of R consecutive slots, in slot order
// (fixed: bit-reproducible).  Used when a convolution wrote more slots per group than the finalize kernels' 32-lane
// strided walk handles quickly (64-channel layers at batch 512: 6272 tiles per view).
__global__ __launch_bounds__(BN_THREADS) void stats_prereduce(const float* __restrict__ in, int T, int R, int To, int GC2,
                                                              int C2, float* __restrict__ out) {
  // one thread per (group, output slot, statistic, channel): consecutive threads = consecutive channels
  const long long i = (long long)blockIdx.x * BN_THREADS + threadIdx.x;
  const long long total = (long long)GC2 / C2 * To * C2;  // G * To * (2 C)
  if (i >= total) return;
  const int c2 = (int)(i % C2);
  const long long u = i / C2;
  const int to = (int)(u % To), g = (int)(u / To);
  const float* src = in + ((size_t)g * T + (size_t)to * R) * C2 + c2;
  const int n = min(R, T - to * R);
  float v = 0.f;
  int k = 0;
  for (; k + 8 <= n; k += 8) {
    float t[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) t[q] = src[(size_t)(k + q) * C2];
#pragma unroll
    for (int q = 0; q < 8; ++q) v += t[q];
  }
  for (; k < n; ++k) v += src[(size_t)k * C2];
  out[i] = v;
}

// Statistics slots of a convolution epilogue -> the partial-sum array the finalize kernels read: the slots themselves
// (<= 512 per group), else their pre-reduction into <= 128 per group inside `scratch` (G * 128 * 2 * C floats).
inline const float* stat_partials(const float* stat_part, int T, int G, int C, float* scratch, int* nblk, hipStream_t st) {
  if (T <= 512) {
    *nblk = T;
    return stat_part;
  }
  int R = 1;
  while ((T + R - 1) / R > 128) R *= 2;
  const int To = (T + R - 1) / R;
  const long long total = (long long)G * To * 2 * C;
  stats_prereduce<<<wm_cdiv(total, BN_THREADS), BN_THREADS, 0, st>>>(stat_part, T, R, To, G * 2 * C, 2 * C, scratch);
  *nblk = To;
  return scratch;
}

inline bool chunk_pow2(int C, int* shift) {
  const int cpr = C >> 3;
  if (cpr <= 0 || (cpr & (cpr - 1)) || BN_THREADS % cpr) return false;
  int sft = 0;
  while ((1 << sft) < cpr) ++sft;
  *shift = sft;
  return true;
}

__global__ __launch_bounds__(BN_THREADS) void add_bf16_kernel(const uint16_t* __restrict__ a,
                                                              const uint16_t* __restrict__ b,
                                                              long long n8, uint16_t* __restrict__ o) {
  for (long long p = (long long)blockIdx.x * BN_THREADS + threadIdx.x; p < n8;
       p += (long long)gridDim.x * BN_THREADS) {
    float fa[8], fb[8];
    unpack8(reinterpret_cast<const uint4*>(a)[p], fa);
    unpack8(reinterpret_cast<const uint4*>(b)[p], fb);
#pragma unroll
    for (int e = 0; e < 8; ++e) fa[e] += fb[e];
    reinterpret_cast<uint4*>(o)[p] = pack8(fa);
  }
}

inline int reduce_blocks(int rows_per_group, int C) {
  const int tpr = C >> 3;
  const int rpp = BN_THREADS / tpr > 0 ? BN_THREADS / tpr : 1;
  int nblk = wm_cdiv(rows_per_group, rpp * 16);
  if (nblk > 256) nblk = 256;
  if (nblk < 1) nblk = 1;
  return nblk;
}

inline int stream_grid(long long items) {
  long long b = (items + BN_THREADS - 1) / BN_THREADS;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

inline void launch_bn_apply(const void* y, const void* residual, const float* scale, const float* shift, long long rows,
                            int C, int rpg, int relu, void* out, hipStream_t st, void* relu_mask = nullptr) {
  int csh = 0;
  if (chunk_pow2(C, &csh))
    bn_apply<true><<<stream_grid(rows * (C >> 3)), BN_THREADS, 0, st>>>(
        static_cast<const uint16_t*>(y), static_cast<const uint16_t*>(residual), scale, shift, rows, C, rpg, relu, csh,
        static_cast<uint16_t*>(out), static_cast<uint8_t*>(relu_mask));
  else
    bn_apply<false><<<stream_grid(rows * (C >> 3)), BN_THREADS, 0, st>>>(
        static_cast<const uint16_t*>(y), static_cast<const uint16_t*>(residual), scale, shift, rows, C, rpg, relu, csh,
        static_cast<uint16_t*>(out), static_cast<uint8_t*>(relu_mask));
}

// ---- wide, short matrices (projection heads: C > 2048, a few hundred rows): one thread per channel walks
// the rows of every statistics group (coalesced across channels); the whole BatchNorm is one launch.
constexpr int BN_WIDE_MAX_C = 16384;

__global__ __launch_bounds__(BN_THREADS) void bn_col_fwd(const uint16_t* __restrict__ y, const uint16_t* __restrict__ res,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ rmean, float* __restrict__ rvar,
                                                         long long* __restrict__ num_batches_tracked, int rpg,
                                                         int C, int G, float eps, float momentum, int relu, int training,
                                                         float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                         uint16_t* __restrict__ out) {
  const int c = blockIdx.x * BN_THREADS + threadIdx.x;
  if (training && num_batches_tracked != nullptr && c == 0)
    atomicAdd(reinterpret_cast<unsigned long long*>(num_batches_tracked), (unsigned long long)G);
  if (c >= C) return;
  const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  for (int g = 0; g < G; ++g) {
    const size_t base = (size_t)g * rpg;
    float mu, is;
    if (training) {
      float s = 0.f;
      for (int r = 0; r < rpg; ++r) s += bf2f(y[(base + r) * C + c]);
      mu = s / (float)rpg;
      float q = 0.f;
      for (int r = 0; r < rpg; ++r) {
        const float d = bf2f(y[(base + r) * C + c]) - mu;
        q = fmaf(d, d, q);
      }
      const float var = q / (float)rpg;
      is = rsqrtf(var + eps);
      save_mean[(size_t)g * C + c] = mu;
      save_invstd[(size_t)g * C + c] = is;
      if (rmean) {
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (rpg > 1 ? var * (float)rpg / (float)(rpg - 1) : var);
      }
    } else {
      mu = rmean[c];
      is = rsqrtf(rvar[c] + eps);
    }
    const float sc = ga * is, sh = be - mu * sc;
    for (int r = 0; r < rpg; ++r) {
      const size_t o = (base + r) * C + c;
      float v = fmaf(bf2f(y[o]), sc, sh);
      if (res) v += bf2f(res[o]);
      if (relu) v = fmaxf(v, 0.f);
      out[o] = f2bf(v);
    }
  }
}

__global__ __launch_bounds__(BN_THREADS) void bn_col_bwd(const uint16_t* __restrict__ y, const uint16_t* __restrict__ dout,
                                                         const uint16_t* __restrict__ out_relu, int remask,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         int rpg, int C, int G, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, int accumulate,
                                                         uint16_t* __restrict__ dy, uint16_t* __restrict__ dz) {
  const int c = blockIdx.x * BN_THREADS + threadIdx.x;
  if (c >= C) return;
  const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  float dg = 0.f, db = 0.f;
  for (int g = 0; g < G; ++g) {
    const size_t base = (size_t)g * rpg;
    const float mu = mean[(size_t)g * C + c], is = invstd[(size_t)g * C + c];
    const float sc = ga * is, sh = be - mu * sc;
    float s1 = 0.f, s2 = 0.f;
    for (int r = 0; r < rpg; ++r) {
      const size_t o = (base + r) * C + c;
      const float yv = bf2f(y[o]);
      float d = bf2f(dout[o]);
      if (out_relu) d = bf2f(out_relu[o]) > 0.f ? d : 0.f;
      else if (remask) d = bf2f(f2bf(fmaf(yv, sc, sh))) > 0.f ? d : 0.f;
      s1 += d;
      s2 = fmaf(d, (yv - mu) * is, s2);
    }
    dg += s2;
    db += s1;
    const float c1 = s1 / (float)rpg, c2 = s2 / (float)rpg;
    for (int r = 0; r < rpg; ++r) {
      const size_t o = (base + r) * C + c;
      const float yv = bf2f(y[o]);
      float d = bf2f(dout[o]);
      if (out_relu) d = bf2f(out_relu[o]) > 0.f ? d : 0.f;
      else if (remask) d = bf2f(f2bf(fmaf(yv, sc, sh))) > 0.f ? d : 0.f;
      if (dz) dz[o] = f2bf(d);
      dy[o] = f2bf(sc * (d - c1 - (yv - mu) * is * c2));
    }
  }
  if (dgamma) dgamma[c] = accumulate ? dgamma[c] + dg : dg;
  if (dbeta) dbeta[c] = accumulate ? dbeta[c] + db : db;
}

// Synchronised form of the wide case: this rank's per-(group, channel) totals [G][2][C] for the caller's all-reduce.
// BWD 0: (sum y, sum y^2); BWD 1: (sum d, sum d * xhat) with the ReLU mask applied to d as in bn_col_bwd.  One thread per
// channel (coalesced across channels), double accumulators: a few hundred rows, not a hot path.
template <int BWD>
__global__ __launch_bounds__(BN_THREADS) void bn_col_sums(const uint16_t* __restrict__ y, const uint16_t* __restrict__ dout,
                                                          const uint16_t* __restrict__ out_relu, int remask,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          int rpg, int C, int G, float* __restrict__ sums) {
  const int c = blockIdx.x * BN_THREADS + threadIdx.x;
  if (c >= C) return;
  for (int g = 0; g < G; ++g) {
    const size_t base = (size_t)g * rpg;
    double a = 0.0, b = 0.0;
    if constexpr (BWD == 0) {
      for (int r = 0; r < rpg; ++r) {
        const double v = (double)bf2f(y[(base + r) * C + c]);
        a += v;
        b += v * v;
      }
    } else {
      const float mu = mean[(size_t)g * C + c], is = invstd[(size_t)g * C + c];
      const float sc = (gamma ? gamma[c] : 1.f) * is, sh = (beta ? beta[c] : 0.f) - mu * sc;
      for (int r = 0; r < rpg; ++r) {
        const size_t o = (base + r) * C + c;
        const float yv = bf2f(y[o]);
        float d = bf2f(dout[o]);
        if (out_relu) d = bf2f(out_relu[o]) > 0.f ? d : 0.f;
        else if (remask) d = bf2f(f2bf(fmaf(yv, sc, sh))) > 0.f ? d : 0.f;
        a += (double)d;
        b += (double)d * (double)((yv - mu) * is);
      }
    }
    sums[((size_t)g * 2 + 0) * C + c] = (float)a;
    sums[((size_t)g * 2 + 1) * C + c] = (float)b;
  }
}

inline bool bn_wide(long long rows, int C, int G) {
  return C > 2048 && C <= BN_WIDE_MAX_C && rows / G <= 65536;
}

int bn_shape_check(long long rows, int C, int G) {
  WM_REQUIRE(rows > 0 && C > 0 && G > 0, WM_EINVAL);
  WM_REQUIRE(C % 8 == 0 && (C <= 2048 || bn_wide(rows, C, G)), WM_EUNSUPPORTED);
  WM_REQUIRE(rows % G == 0 && rows / G < (1ll << 31), WM_EUNSUPPORTED);
  return WM_OK;
}

}  // namespace

extern "C" size_t wm_bn_workspace_bytes(long long rows, int C, int G) {
  if (rows <= 0 || C <= 0 || G <= 0 || C % 8) return 0;
  const int nblk = reduce_blocks((int)(rows / G), C);
  // partial sums + the 7 backward coefficient planes
  return ((size_t)G * nblk * 2 * C + (size_t)G * 7 * C) * sizeof(float) + 256;
}

extern "C" int wm_bn_train_fwd(const void* y, const void* residual, const float* gamma,
                               const float* beta, float* running_mean, float* running_var,
                               long long* num_batches_tracked, long long rows, int C, int G, float eps, float momentum, int relu,
                               float* save_mean, float* save_invstd, void* out, void* relu_mask, void* workspace,
                               size_t workspace_bytes, void* stream) {
  WM_REQUIRE(y && out && save_mean && save_invstd && workspace, WM_EINVAL);
  const int rc = bn_shape_check(rows, C, G);
  if (rc != WM_OK) return rc;
  WM_REQUIRE(workspace_bytes >= wm_bn_workspace_bytes(rows, C, G), WM_EWORKSPACE);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rpg = (int)(rows / G);
  if (bn_wide(rows, C, G)) {
    WM_REQUIRE(relu_mask == nullptr, WM_EUNSUPPORTED);
    bn_col_fwd<<<wm_cdiv(C, BN_THREADS), BN_THREADS, 0, st>>>(
        static_cast<const uint16_t*>(y), static_cast<const uint16_t*>(residual), gamma, beta, running_mean, running_var,
        num_batches_tracked, rpg, C, G, eps, momentum, relu, 1, save_mean, save_invstd, static_cast<uint16_t*>(out));
    WM_LAUNCH_CHECK();
    return WM_OK;
  }
  const int nblk = reduce_blocks(rpg, C);
  float* part = static_cast<float*>(workspace);
  float* scale = part + (size_t)G * nblk * 2 * C;  // reuse the coefficient planes: scale, shift
  float* shift = scale + (size_t)G * C;
  const int tpr = C >> 3, rpp = BN_THREADS / tpr;
  const size_t lds = (size_t)2 * rpp * C * sizeof(float);
  bn_reduce<0><<<dim3(nblk, G), BN_THREADS, lds, st>>>(static_cast<const uint16_t*>(y), nullptr, nullptr, nullptr,
                                                       nullptr, nullptr, nullptr, rpg, C, wm_cdiv(rpg, nblk), part, PoolSrc{});
  WM_LAUNCH_CHECK();
  bn_fwd_finalize<<<wm_cdiv(C, 32), 1024, 0, st>>>(part, nblk, G, C, rpg, 0, gamma, beta, eps, momentum, running_mean,
                                                   running_var, num_batches_tracked, save_mean, save_invstd, scale, shift);
  WM_LAUNCH_CHECK();
  launch_bn_apply(y, residual, scale, shift, rows, C, rpg, relu, out, st, relu_mask);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

// Forward when the producing convolution already accumulated the statistics
// (wm_conv2d_fwd_stats): finalize (and clear) stat_part [G][stat_buckets][2][C], then apply.
extern "C" int wm_bn_train_fwd_from_stats(const void* y, const void* residual, const float* gamma,
                                          const float* beta, float* running_mean, float* running_var,
                                          long long* num_batches_tracked, long long rows, int C, int G, float eps,
                                          float momentum, int relu,
                                          float* save_mean, float* save_invstd, void* out, void* relu_mask,
                                          const float* stat_part, int stat_tiles, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  WM_REQUIRE(y && out && save_mean && save_invstd && workspace && stat_part, WM_EINVAL);
  WM_REQUIRE(stat_tiles > 0, WM_EINVAL);
  const int rc = bn_shape_check(rows, C, G);
  if (rc != WM_OK) return rc;
  WM_REQUIRE(workspace_bytes >= ((size_t)2 * G * C + (stat_tiles > 512 ? (size_t)G * 128 * 2 * C : 0)) * sizeof(float), WM_EWORKSPACE);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rpg = (int)(rows / G);
  float* scale = static_cast<float*>(workspace);
  float* shift = scale + (size_t)G * C;
  int nblk = 0;
  const float* part = stat_partials(stat_part, stat_tiles, G, C, shift + (size_t)G * C, &nblk, st);
  WM_LAUNCH_CHECK();
  bn_fwd_finalize<<<wm_cdiv(C, 32), 1024, 0, st>>>(const_cast<float*>(part), nblk, G, C, rpg, 0, gamma, beta, eps, momentum,
                                                   running_mean, running_var, num_batches_tracked, save_mean, save_invstd, scale, shift);
  WM_LAUNCH_CHECK();
  launch_bn_apply(y, residual, scale, shift, rows, C, rpg, relu, out, st, relu_mask);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

// Statistics only (no apply pass): mean/invstd/running stats + per-(group, channel) scale and shift
// [G][C] each, for a consumer that applies the normalisation itself (wm_bn_relu_maxpool3x3s2_fwd).
// stat_part non-NULL: statistics were fused into the producing convolution; else they are computed here.
extern "C" int wm_bn_train_stats(const void* y, const float* gamma, const float* beta, float* running_mean,
                                 float* running_var, long long* num_batches_tracked, long long rows, int C, int G, float eps, float momentum,
                                 float* save_mean, float* save_invstd, float* scale, float* shift, const float* stat_part,
                                 int stat_tiles, void* workspace, size_t workspace_bytes, void* stream) {
  WM_REQUIRE(y && save_mean && save_invstd && scale && shift && workspace, WM_EINVAL);
  const int rc = bn_shape_check(rows, C, G);
  if (rc != WM_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rpg = (int)(rows / G);
  if (stat_part) {
    WM_REQUIRE(stat_tiles > 0, WM_EINVAL);
    WM_REQUIRE(stat_tiles <= 512 || workspace_bytes >= (size_t)G * 128 * 2 * C * sizeof(float), WM_EWORKSPACE);
    int nblk = 0;
    const float* part = stat_partials(stat_part, stat_tiles, G, C, static_cast<float*>(workspace), &nblk, st);
    WM_LAUNCH_CHECK();
    bn_fwd_finalize<<<wm_cdiv(C, 32), 1024, 0, st>>>(const_cast<float*>(part), nblk, G, C, rpg, 0, gamma, beta, eps, momentum,
                                                     running_mean, running_var, num_batches_tracked, save_mean, save_invstd, scale, shift);
  } else {
    WM_REQUIRE(workspace_bytes >= wm_bn_workspace_bytes(rows, C, G), WM_EWORKSPACE);
    const int nblk = reduce_blocks(rpg, C);
    float* part = static_cast<float*>(workspace);
    const int tpr = C >> 3, rpp = BN_THREADS / tpr;
    const size_t lds = (size_t)2 * rpp * C * sizeof(float);
    bn_reduce<0><<<dim3(nblk, G), BN_THREADS, lds, st>>>(static_cast<const uint16_t*>(y), nullptr, nullptr, nullptr,
                                                         nullptr, nullptr, nullptr, rpg, C, wm_cdiv(rpg, nblk), part, PoolSrc{});
    WM_LAUNCH_CHECK();
    bn_fwd_finalize<<<wm_cdiv(C, 32), 1024, 0, st>>>(part, nblk, G, C, rpg, 0, gamma, beta, eps, momentum, running_mean,
                                                     running_var, num_batches_tracked, save_mean, save_invstd, scale, shift);
  }
  WM_LAUNCH_CHECK();
  return WM_OK;
}

// Eval-mode scale/shift [C] from the running statistics (for the fused stem in eval mode).
extern "C" int wm_bn_eval_scale_shift(const float* gamma, const float* beta, const float* running_mean,
                                      const float* running_var, int C, float eps, float* scale, float* shift,
                                      void* stream) {
  WM_REQUIRE(running_mean && running_var && scale && shift && C > 0, WM_EINVAL);
  bn_eval_params<<<1, C < 1024 ? ((C + 63) / 64) * 64 : 1024, 0, static_cast<hipStream_t>(stream)>>>(
      gamma, beta, running_mean, running_var, eps, C, scale, shift);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_bn_eval_fwd(const void* y, const void* residual, const float* gamma, const float* beta,
                              const float* running_mean, const float* running_var, long long rows, int C,
                              float eps, int relu, void* out, void* workspace, size_t workspace_bytes,
                              void* stream) {
  WM_REQUIRE(y && out && running_mean && running_var && workspace, WM_EINVAL);
  const int rc = bn_shape_check(rows, C, 1);
  if (rc != WM_OK) return rc;
  WM_REQUIRE(workspace_bytes >= (size_t)2 * C * sizeof(float), WM_EWORKSPACE);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (bn_wide(rows, C, 1)) {
    bn_col_fwd<<<wm_cdiv(C, BN_THREADS), BN_THREADS, 0, st>>>(
        static_cast<const uint16_t*>(y), static_cast<const uint16_t*>(residual), gamma, beta,
        const_cast<float*>(running_mean), const_cast<float*>(running_var), nullptr, (int)rows, C, 1, eps, 0.f, relu, 0, nullptr,
        nullptr, static_cast<uint16_t*>(out));
    WM_LAUNCH_CHECK();
    return WM_OK;
  }
  float* scale = static_cast<float*>(workspace);
  float* shift = scale + C;
  bn_eval_params<<<1, C < 1024 ? ((C + 63) / 64) * 64 : 1024, 0, st>>>(gamma, beta, running_mean, running_var, eps,
                                                                     C, scale, shift);
  WM_LAUNCH_CHECK();
  launch_bn_apply(y, residual, scale, shift, rows, C, (int)(rows < (1ll << 31) - 1 ? rows : (1ll << 31) - 1), relu, out,
                  st);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

static int bn_bwd_impl(const void* y, const void* dout, const void* out_relu, int relu_from_y, const float* gamma,
                       const float* beta, const float* save_mean, const float* save_invstd, long long rows, int C,
                       int G, float* dgamma, float* dbeta, int accumulate, void* dy, void* dz, void* workspace,
                       size_t workspace_bytes, void* stream, const PoolSrc ps, const void* ysel = nullptr);

extern "C" int wm_bn_train_bwd(const void* y, const void* dout, const void* out_relu, int relu_from_y,
                               const float* gamma, const float* beta, const float* save_mean,
                               const float* save_invstd, long long rows, int C, int G, float* dgamma,
                               float* dbeta, int accumulate, void* dy, void* dz, void* workspace,
                               size_t workspace_bytes, void* stream) {
  WM_REQUIRE(dout, WM_EINVAL);
  return bn_bwd_impl(y, dout, out_relu, relu_from_y, gamma, beta, save_mean, save_invstd, rows, C, G, dgamma, dbeta,
                     accumulate, dy, dz, workspace, workspace_bytes, stream, PoolSrc{});
}

// Backward when the producing dgrad (wm_conv2d_dgrad_bnstat) already took the gradient through the ReLU and
// accumulated (sum g, sum g * xhat): finalize (and clear) the buckets, then ONE pass over (y, g).
extern "C" int wm_bn_train_bwd_from_stats(const void* y, const void* g, const float* gamma, const float* beta,
                                          const float* save_mean, const float* save_invstd, long long rows, int C,
                                          int G, float* dgamma, float* dbeta, int accumulate, void* dy,
                                          const float* stat_part, int stat_tiles, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  WM_REQUIRE(y && g && save_mean && save_invstd && dy && stat_part && workspace && stat_tiles > 0, WM_EINVAL);
  const int rc = bn_shape_check(rows, C, G);
  if (rc != WM_OK) return rc;
  WM_REQUIRE(!bn_wide(rows, C, G), WM_EUNSUPPORTED);
  WM_REQUIRE(workspace_bytes >= ((size_t)7 * G * C + (stat_tiles > 512 ? (size_t)G * 128 * 2 * C : 0)) * sizeof(float), WM_EWORKSPACE);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rpg = (int)(rows / G);
  float* coef = static_cast<float*>(workspace);
  int nblk = 0;
  const float* part = stat_partials(stat_part, stat_tiles, G, C, coef + (size_t)7 * G * C, &nblk, st);
  WM_LAUNCH_CHECK();
  bn_bwd_finalize<<<wm_cdiv(C, 32), 1024, 0, st>>>(const_cast<float*>(part), nblk, G, C, rpg, gamma, beta,
                                                   save_mean, save_invstd, dgamma, dbeta, accumulate, coef, 1);
  WM_LAUNCH_CHECK();
  const int tpr = C >> 3;
  int csh = 0;
  if (chunk_pow2(C, &csh))
    bn_bwd_apply<true, 0><<<stream_grid(rows * tpr), BN_THREADS, 0, st>>>(
        static_cast<const uint16_t*>(y), static_cast<const uint16_t*>(g), nullptr, coef, rows, C, rpg, 0, csh,
        static_cast<uint16_t*>(dy), nullptr, PoolSrc{});
  else
    bn_bwd_apply<false, 0><<<stream_grid(rows * tpr), BN_THREADS, 0, st>>>(
        static_cast<const uint16_t*>(y), static_cast<const uint16_t*>(g), nullptr, coef, rows, C, rpg, 0, csh,
        static_cast<uint16_t*>(dy), nullptr, PoolSrc{});
  WM_LAUNCH_CHECK();
  return WM_OK;
}

// Backward of max_pool3x3s2(relu(BN(y))) (the fused stem): the gradient entering the BN is gathered
// from the pooled gradient + window positions instead of being materialised by wm_maxpool3x3s2_bwd.
extern "C" int wm_bn_relu_maxpool_bwd(const void* y, const void* ysel, const void* pooled_dy, const void* pool_idx,
                                      int N, int H, int W, int C, const float* gamma, const float* beta,
                                      const float* save_mean,
                                      const float* save_invstd, int G, float* dgamma, float* dbeta, int accumulate,
                                      void* dy, void* workspace, size_t workspace_bytes, void* stream) {
  WM_REQUIRE(pooled_dy && pool_idx && gamma && beta, WM_EINVAL);
  WM_REQUIRE(N > 0 && H > 1 && W > 1 && (long long)N * H * W < (1ll << 31), WM_EINVAL);
  PoolSrc ps;
  ps.dy = static_cast<const uint16_t*>(pooled_dy);
  ps.idx = static_cast<const uint8_t*>(pool_idx);
  ps.H = H; ps.W = W; ps.P = (H + 2 - 3) / 2 + 1; ps.Q = (W + 2 - 3) / 2 + 1;
  return bn_bwd_impl(y, nullptr, nullptr, 1, gamma, beta, save_mean, save_invstd, (long long)N * H * W, C, G, dgamma,
                     dbeta, accumulate, dy, nullptr, workspace, workspace_bytes, stream, ps, ysel);
}

static int bn_bwd_impl(const void* y, const void* dout, const void* out_relu, int relu_from_y, const float* gamma,
                       const float* beta, const float* save_mean, const float* save_invstd, long long rows, int C,
                       int G, float* dgamma, float* dbeta, int accumulate, void* dy, void* dz, void* workspace,
                       size_t workspace_bytes, void* stream, const PoolSrc ps, const void* ysel) {
  WM_REQUIRE(y && (dout || ps.dy) && save_mean && save_invstd && dy && workspace, WM_EINVAL);
  const int rc = bn_shape_check(rows, C, G);
  if (rc != WM_OK) return rc;
  WM_REQUIRE(workspace_bytes >= wm_bn_workspace_bytes(rows, C, G), WM_EWORKSPACE);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rpg = (int)(rows / G);
  if (bn_wide(rows, C, G)) {
    WM_REQUIRE(ps.dy == nullptr && dout, WM_EUNSUPPORTED);
    const bool rm = relu_from_y && !out_relu;
    bn_col_bwd<<<wm_cdiv(C, BN_THREADS), BN_THREADS, 0, st>>>(
        static_cast<const uint16_t*>(y), static_cast<const uint16_t*>(dout), static_cast<const uint16_t*>(out_relu),
        rm ? 1 : 0, gamma, beta, save_mean, save_invstd, rpg, C, G, dgamma, dbeta, accumulate, static_cast<uint16_t*>(dy),
        static_cast<uint16_t*>(dz));
    WM_LAUNCH_CHECK();
    return WM_OK;
  }
  const int nblk = reduce_blocks(rpg, C);
  float* part = static_cast<float*>(workspace);
  float* coef = part + (size_t)G * nblk * 2 * C;
  const int tpr = C >> 3, rpp = BN_THREADS / tpr;
  const size_t lds = (size_t)2 * rpp * C * sizeof(float);
  const bool remask = relu_from_y && !out_relu;
  WM_REQUIRE(!remask || (gamma && beta), WM_EINVAL);
  int nblk_used = nblk;
  if (ps.dy != nullptr && ysel != nullptr) {
    // pooled source with the selected inputs at hand: the sums run over the pooled tensor (a quarter of
    // the rows, no gather); 1/M in the finalize stays the full row count
    const long long prow = (long long)(rows / ((long long)ps.H * ps.W)) * ps.P * ps.Q;
    WM_REQUIRE(prow % G == 0, WM_EUNSUPPORTED);
    const int prpg = (int)(prow / G);
    nblk_used = reduce_blocks(prpg, C);
    if (nblk_used > nblk) nblk_used = nblk;  // the workspace was sized for nblk
    bn_reduce<1><<<dim3(nblk_used, G), BN_THREADS, lds, st>>>(
        static_cast<const uint16_t*>(ysel), ps.dy, nullptr, save_mean, save_invstd, gamma, beta, prpg, C,
        wm_cdiv(prpg, nblk_used), part, PoolSrc{});
  } else {
    bn_reduce<1><<<dim3(nblk, G), BN_THREADS, lds, st>>>(
        static_cast<const uint16_t*>(y), static_cast<const uint16_t*>(dout), static_cast<const uint16_t*>(out_relu),
        save_mean, save_invstd, remask ? gamma : nullptr, remask ? beta : nullptr, rpg, C, wm_cdiv(rpg, nblk), part, ps);
  }
  WM_LAUNCH_CHECK();
  bn_bwd_finalize<<<wm_cdiv(C, 32), 1024, 0, st>>>(part, nblk_used, G, C, rpg, gamma, beta, save_mean, save_invstd,
                                                   dgamma, dbeta, accumulate, coef, 0);
  WM_LAUNCH_CHECK();
  int csh = 0;
  if (ps.dy != nullptr && !dz && !out_relu && (ps.H & 1) == 0 && (ps.W & 1) == 0 && BN_THREADS % tpr == 0 &&
      rows * tpr < (1ll << 31) && (size_t)G * 7 * C * sizeof(float) <= 32768) {  // (32-bit item indices; coefficient rows in LDS)
    const int n_img = (int)(rows / ((long long)ps.H * ps.W));
    bn_pool_bwd_apply<<<stream_grid(rows / 4 * tpr), BN_THREADS, (size_t)G * 7 * C * sizeof(float), st>>>(
        static_cast<const uint16_t*>(y), coef, G * 7 * C, n_img, C, n_img / G, remask ? 1 : 0, static_cast<uint16_t*>(dy), ps,
        wm_div_make((uint32_t)(C >> 3)), wm_div_make((uint32_t)(ps.W >> 1)), wm_div_make((uint32_t)(ps.H >> 1)),
        wm_div_make((uint32_t)(n_img / G)));
    WM_LAUNCH_CHECK();
    return WM_OK;
  }
  const bool pow2 = chunk_pow2(C, &csh);
  auto launch = [&](auto kernel) {
    kernel<<<stream_grid(rows * tpr), BN_THREADS, 0, st>>>(
        static_cast<const uint16_t*>(y), static_cast<const uint16_t*>(dout), static_cast<const uint16_t*>(out_relu),
        coef, rows, C, rpg, remask ? 1 : 0, csh, static_cast<uint16_t*>(dy), static_cast<uint16_t*>(dz), ps);
  };
  if (ps.dy != nullptr) {
    if (pow2) launch(bn_bwd_apply<true, 2>);
    else launch(bn_bwd_apply<false, 2>);
  } else if (out_relu == nullptr && !remask && dz == nullptr) {
    if (pow2) launch(bn_bwd_apply<true, 0>);
    else launch(bn_bwd_apply<false, 0>);
  } else {
    if (pow2) launch(bn_bwd_apply<true, 1>);
    else launch(bn_bwd_apply<false, 1>);
  }
  WM_LAUNCH_CHECK();
  return WM_OK;
}

// ---- synchronised BatchNorm (torch.nn.SyncBatchNorm semantics: statistics over the batches of ALL ranks) -------------
// The library has no communicator: each pass is cut in two at the point where the caller all-reduces a [G][2][C]
// float vector (RCCL / gloo through torch.distributed).  `group_count` = rows per statistics group summed over ranks.
extern "C" int wm_bn_sync_fwd_sums(const void* y, long long rows, int C, int G, const float* stat_part, int stat_tiles,
                                   float* sums, void* workspace, size_t workspace_bytes, void* stream) {
  WM_REQUIRE(y && sums && workspace, WM_EINVAL);
  const int rc = bn_shape_check(rows, C, G);
  if (rc != WM_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rpg = (int)(rows / G);
  if (bn_wide(rows, C, G)) {   // projection-head widths (BYOL: BatchNorm1d(4096)): one thread per channel
    bn_col_sums<0><<<wm_cdiv(C, BN_THREADS), BN_THREADS, 0, st>>>(static_cast<const uint16_t*>(y), nullptr, nullptr, 0, nullptr,
                                                                 nullptr, nullptr, nullptr, rpg, C, G, sums);
    WM_LAUNCH_CHECK();
    return WM_OK;
  }
  int nblk = 0;
  const float* part;
  if (stat_part) {
    WM_REQUIRE(stat_tiles > 0, WM_EINVAL);
    WM_REQUIRE(stat_tiles <= 512 || workspace_bytes >= (size_t)G * 128 * 2 * C * sizeof(float), WM_EWORKSPACE);
    part = stat_partials(stat_part, stat_tiles, G, C, static_cast<float*>(workspace), &nblk, st);
  } else {
    WM_REQUIRE(workspace_bytes >= wm_bn_workspace_bytes(rows, C, G), WM_EWORKSPACE);
    nblk = reduce_blocks(rpg, C);
    const int tpr = C >> 3, rpp = BN_THREADS / tpr;
    const size_t lds = (size_t)2 * rpp * C * sizeof(float);
    bn_reduce<0><<<dim3(nblk, G), BN_THREADS, lds, st>>>(static_cast<const uint16_t*>(y), nullptr, nullptr, nullptr,
                                                         nullptr, nullptr, nullptr, rpg, C, wm_cdiv(rpg, nblk),
                                                         static_cast<float*>(workspace), PoolSrc{});
    part = static_cast<float*>(workspace);
  }
  WM_LAUNCH_CHECK();
  bn_sums_finalize<<<wm_cdiv(C, 32), 1024, 0, st>>>(const_cast<float*>(part), nblk, G, C, sums);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_bn_sync_fwd_apply(const void* y, const void* residual, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, long long* num_batches_tracked,
                                    long long rows, int C, int G, long long group_count, float eps, float momentum,
                                    int relu, float* save_mean, float* save_invstd, void* out, const float* sums,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  WM_REQUIRE(y && out && save_mean && save_invstd && sums && workspace, WM_EINVAL);
  const int rc = bn_shape_check(rows, C, G);
  if (rc != WM_OK) return rc;
  WM_REQUIRE(group_count >= rows / G && group_count < (1ll << 31), WM_EINVAL);   // (finalize + apply serve any width)
  WM_REQUIRE(workspace_bytes >= (size_t)2 * G * C * sizeof(float), WM_EWORKSPACE);
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* scale = static_cast<float*>(workspace);
  float* shift = scale + (size_t)G * C;
  bn_fwd_finalize<<<wm_cdiv(C, 32), 1024, 0, st>>>(const_cast<float*>(sums), 1, G, C, (int)group_count, 0, gamma, beta, eps,
                                                   momentum, running_mean, running_var, num_batches_tracked, save_mean,
                                                   save_invstd, scale, shift);
  WM_LAUNCH_CHECK();
  launch_bn_apply(y, residual, scale, shift, rows, C, (int)(rows / G), relu, out, st);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

// Local (sum g, sum g * xhat) per (group, channel) -> sums [G][2][C], and this rank's dgamma / dbeta (torch's
// SyncBatchNorm leaves their reduction to the gradient exchange).  Workspace as wm_bn_train_bwd.
extern "C" int wm_bn_sync_bwd_sums(const void* y, const void* dout, const void* out_relu, int relu_from_y,
                                   const float* gamma, const float* beta, const float* save_mean,
                                   const float* save_invstd, long long rows, int C, int G, float* dgamma, float* dbeta,
                                   int accumulate, float* sums, void* workspace, size_t workspace_bytes, void* stream) {
  WM_REQUIRE(y && dout && save_mean && save_invstd && sums && workspace, WM_EINVAL);
  const int rc = bn_shape_check(rows, C, G);
  if (rc != WM_OK) return rc;
  WM_REQUIRE(workspace_bytes >= wm_bn_workspace_bytes(rows, C, G), WM_EWORKSPACE);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rpg = (int)(rows / G);
  const int nblk = reduce_blocks(rpg, C);
  float* part = static_cast<float*>(workspace);
  float* coef = part + (size_t)G * nblk * 2 * C;
  const bool remask = relu_from_y && !out_relu;
  WM_REQUIRE(!remask || (gamma && beta), WM_EINVAL);
  if (bn_wide(rows, C, G)) {
    bn_col_sums<1><<<wm_cdiv(C, BN_THREADS), BN_THREADS, 0, st>>>(
        static_cast<const uint16_t*>(y), static_cast<const uint16_t*>(dout), static_cast<const uint16_t*>(out_relu),
        remask ? 1 : 0, gamma, beta, save_mean, save_invstd, rpg, C, G, sums);
    WM_LAUNCH_CHECK();
  } else {
    const int tpr = C >> 3, rpp = BN_THREADS / tpr;
    const size_t lds = (size_t)2 * rpp * C * sizeof(float);
    bn_reduce<1><<<dim3(nblk, G), BN_THREADS, lds, st>>>(
        static_cast<const uint16_t*>(y), static_cast<const uint16_t*>(dout), static_cast<const uint16_t*>(out_relu),
        save_mean, save_invstd, remask ? gamma : nullptr, remask ? beta : nullptr, rpg, C, wm_cdiv(rpg, nblk), part, PoolSrc{});
    WM_LAUNCH_CHECK();
    bn_sums_finalize<<<wm_cdiv(C, 32), 1024, 0, st>>>(part, nblk, G, C, sums);
    WM_LAUNCH_CHECK();
  }
  if (dgamma || dbeta) {  // the local parameter gradients from the local totals (coef is scratch here)
    bn_bwd_finalize<<<wm_cdiv(C, 32), 1024, 0, st>>>(sums, 1, G, C, rpg, gamma, beta, save_mean, save_invstd, dgamma,
                                                     dbeta, accumulate, coef, 0);
    WM_LAUNCH_CHECK();
  }
  return WM_OK;
}

// sums: the all-reduced totals.  dy = gamma * invstd * (dz - sum(dz)/M - xhat * sum(dz * xhat)/M), M = group_count.
extern "C" int wm_bn_sync_bwd_apply(const void* y, const void* dout, const void* out_relu, int relu_from_y,
                                    const float* gamma, const float* beta, const float* save_mean,
                                    const float* save_invstd, long long rows, int C, int G, long long group_count,
                                    const float* sums, void* dy, void* dz, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  WM_REQUIRE(y && dout && save_mean && save_invstd && sums && dy && workspace, WM_EINVAL);
  const int rc = bn_shape_check(rows, C, G);
  if (rc != WM_OK) return rc;
  WM_REQUIRE(group_count >= rows / G && group_count < (1ll << 31), WM_EINVAL);   // (finalize + apply serve any width)
  WM_REQUIRE(workspace_bytes >= (size_t)7 * G * C * sizeof(float), WM_EWORKSPACE);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rpg = (int)(rows / G);
  float* coef = static_cast<float*>(workspace);
  const bool remask = relu_from_y && !out_relu;
  WM_REQUIRE(!remask || (gamma && beta), WM_EINVAL);
  bn_bwd_finalize<<<wm_cdiv(C, 32), 1024, 0, st>>>(const_cast<float*>(sums), 1, G, C, (int)group_count, gamma, beta,
                                                   save_mean, save_invstd, nullptr, nullptr, 0, coef, 0);
  WM_LAUNCH_CHECK();
  const int tpr = C >> 3;
  int csh = 0;
  if (chunk_pow2(C, &csh))
    bn_bwd_apply<true, 1><<<stream_grid(rows * tpr), BN_THREADS, 0, st>>>(
        static_cast<const uint16_t*>(y), static_cast<const uint16_t*>(dout), static_cast<const uint16_t*>(out_relu),
        coef, rows, C, rpg, remask ? 1 : 0, csh, static_cast<uint16_t*>(dy), static_cast<uint16_t*>(dz), PoolSrc{});
  else
    bn_bwd_apply<false, 1><<<stream_grid(rows * tpr), BN_THREADS, 0, st>>>(
        static_cast<const uint16_t*>(y), static_cast<const uint16_t*>(dout), static_cast<const uint16_t*>(out_relu),
        coef, rows, C, rpg, remask ? 1 : 0, csh, static_cast<uint16_t*>(dy), static_cast<uint16_t*>(dz), PoolSrc{});
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_add_bf16(const void* a, const void* b, long long n, void* out, void* stream) {
  WM_REQUIRE(a && b && out && n > 0, WM_EINVAL);
  WM_REQUIRE(n % 8 == 0, WM_EUNSUPPORTED);
  add_bf16_kernel<<<stream_grid(n / 8), BN_THREADS, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const uint16_t*>(a), static_cast<const uint16_t*>(b), n / 8, static_cast<uint16_t*>(out));
  WM_LAUNCH_CHECK();
  return WM_OK;
}
