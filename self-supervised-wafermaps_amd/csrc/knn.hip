// kNN retrieval: blocked pairwise-dot on MFMA + running in-register top-k.
//
// Replaces lightly.utils.benchmarking.knn_predict as called by the reference at
// src/ssl_wafermap/models/knn.py:91-98 (torch.mm -> topk -> gather -> exp -> one-hot -> argsort).
// The reference materialises sim[B,N] in fp32 and re-reads it for topk; here a block streams a
// slice of the bank through LDS once, multiplies it against a resident query tile with MFMA and
// keeps each query's best K in registers, so HBM sees the bank once per query batch and nothing
// else.  Roofline: HBM (algorithmic bytes = n*d*elsize per query batch, SURVEY §8d formula (ii)).
//
// Geometry.  A row of either operand is cut into 256-byte slabs (128 bf16 / 64 f32).  LDS images
// keep 256-byte rows, the 16-byte chunk index XOR-swizzled with (row & 15) so that the
// ds_read_b128 fragment reads (lane (r,h) -> row r, chunk 2s+h) are bank-conflict free
// (cdna_hip_programming.md T2).  MFMA orientation is "swapped": A = 32 bank rows, B = 32 queries,
// so the accumulator puts the QUERY on the lane (col = lane&31) and 16 bank rows in registers —
// the top-k list of a query is lane-local and needs no cross-lane traffic in the hot loop.
//   bf16: v_mfma_f32_32x32x16_bf16, one 16-byte fragment = one MFMA (k = 8h+j)
//   f32 : v_mfma_f32_32x32x2_f32 (exact f32 fmaf chain), one 16-byte fragment = 4 MFMAs
//
// Selection is two-stage so that the streaming loop is branch-free: a lane folds the 16 values an
// accumulator tile gives it into a running MAXIMUM over a fixed group of 64 bank rows (4 chunks x 16
// rows) and offers that one (max, group id) pair to its sorted list once per group.  The k best
// elements always lie inside the k groups with the largest maxima (each group whose max reaches
// the k-th best value holds at least one of the k best), so after the cross-block merge a small
// rescoring kernel recomputes the <= K*64 candidate rows of every query exactly and orders them
// (value descending, bank index ascending).
#include "common.h"
#include <limits.h>

namespace {

constexpr int KNN_THREADS = 256;
constexpr int KNN_ROWS = 128;  // bank rows per LDS chunk: 4 waves x 32
constexpr int KNN_SLAB = 256;  // bytes per row per slab
constexpr int KNN_BUF = KNN_ROWS * KNN_SLAB;

__device__ __forceinline__ bool better(float av, int ai, float bv, int bi) {
  return av > bv || (av == bv && ai < bi);
}

// Insert (nv, ni) into a list sorted best-first; the worst entry falls off.
template <int K>
__device__ __forceinline__ void topk_insert(float (&v)[K], int (&ix)[K], float nv, int ni) {
#pragma unroll
  for (int j = K - 1; j >= 1; --j) {
    const bool cj = better(nv, ni, v[j], ix[j]);
    const bool cjm = better(nv, ni, v[j - 1], ix[j - 1]);
    const float tv = cjm ? v[j - 1] : nv;
    const int ti = cjm ? ix[j - 1] : ni;
    v[j] = cj ? tv : v[j];
    ix[j] = cj ? ti : ix[j];
  }
  if (better(nv, ni, v[0], ix[0])) {
    v[0] = nv;
    ix[0] = ni;
  }
}

__device__ __forceinline__ int acc_row(int reg, int half) {
  // C/D map of the 32x32 MFMA family: row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  return (reg & 3) + 8 * (reg >> 2) + 4 * half;
}

template <int DT, int QT, int K>
__global__ __launch_bounds__(KNN_THREADS) void knn_block_topk(
    const uint8_t* __restrict__ query, const uint8_t* __restrict__ bank, int nq, int n,
    int rowbytes, int chunks_per_slice, int nslices, float* __restrict__ part_sim,
    int* __restrict__ part_idx) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr int QB = QT * 32;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int nslab = rowbytes / KNN_SLAB;
  const int q0 = blockIdx.y * QB;
  const int slice = blockIdx.x;

  uint8_t* bankbuf = smem;                // 2 x KNN_BUF
  uint8_t* qbuf = smem + 2 * KNN_BUF;     // QB x rowbytes

  // ---- stage the query tile (swizzled), zero rows past nq
  {
    const int ppr = rowbytes >> 4;  // 16-byte pieces per row
    const int total = QB * ppr;
    for (int p = tid; p < total; p += KNN_THREADS) {
      const int row = p / ppr, c = p - row * ppr;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q0 + row < nq)
        v = *reinterpret_cast<const uint4*>(query + (size_t)(q0 + row) * rowbytes + (size_t)c * 16);
      const int slab = c >> 4, ch = c & 15;
      *reinterpret_cast<uint4*>(qbuf + (size_t)row * rowbytes + slab * KNN_SLAB +
                                ((ch ^ (row & 15)) << 4)) = v;
    }
  }

  float lv[QT][K];
  int li[QT][K];
#pragma unroll
  for (int t = 0; t < QT; ++t)
#pragma unroll
    for (int j = 0; j < K; ++j) {
      lv[t][j] = -INFINITY;
      li[t][j] = INT_MAX;
    }

  const int chunk_begin = slice * chunks_per_slice;
  const int total_chunks = (n + KNN_ROWS - 1) / KNN_ROWS;
  int chunk_end = chunk_begin + chunks_per_slice;
  if (chunk_end > total_chunks) chunk_end = total_chunks;
  const int iters = (chunk_end - chunk_begin) * nslab;

  uint4 stage[8];
  auto load_iter = [&](int it) {
    const int chunk = chunk_begin + it / nslab, slab = it % nslab;
    const int nb = chunk * KNN_ROWS;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = tid + KNN_THREADS * i;
      const int row = p >> 4, ch = p & 15;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (nb + row < n)
        v = *reinterpret_cast<const uint4*>(bank + (size_t)(nb + row) * rowbytes +
                                            (size_t)slab * KNN_SLAB + ch * 16);
      stage[i] = v;
    }
  };
  auto store_iter = [&](uint8_t* buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = tid + KNN_THREADS * i;
      const int row = p >> 4, ch = p & 15;
      *reinterpret_cast<uint4*>(buf + row * KNN_SLAB + ((ch ^ (row & 15)) << 4)) = stage[i];
    }
  };

  f32x16_t acc[QT];
  float gmax[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) gmax[t] = -INFINITY;

  if (iters > 0) {
    load_iter(0);
    store_iter(bankbuf);
  }
  __syncthreads();

  for (int it = 0; it < iters; ++it) {
    const int cur = it & 1;
    const int slab = it % nslab;
    if (it + 1 < iters) load_iter(it + 1);
    if (slab == 0) {
#pragma unroll
      for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    }
    const uint8_t* abuf = bankbuf + cur * KNN_BUF + (wave * 32 + r) * KNN_SLAB;
    const uint8_t* qrow = qbuf + (size_t)r * rowbytes + slab * KNN_SLAB;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int off = ((2 * s + h) ^ (r & 15)) << 4;
      const uint4 a = *reinterpret_cast<const uint4*>(abuf + off);
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        const uint4 b = *reinterpret_cast<const uint4*>(qrow + (size_t)t * 32 * rowbytes + off);
        if constexpr (DT == WM_BF16) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                           __builtin_bit_cast(bf16x8_t, b),
                                                           acc[t], 0, 0, 0);
        } else {
          const f32x4_t af = __builtin_bit_cast(f32x4_t, a);
          const f32x4_t bf = __builtin_bit_cast(f32x4_t, b);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc[t], 0, 0, 0);
        }
      }
    }
    if (slab == nslab - 1) {
      const int crel = it / nslab;  // chunk index inside this slice
      const int nb = (chunk_begin + crel) * KNN_ROWS + wave * 32;
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        float m = gmax[t];
#pragma unroll
        for (int e = 0; e < 16; ++e) m = fmaxf(m, (nb + acc_row(e, h) < n) ? acc[t][e] : -INFINITY);
        gmax[t] = m;
      }
      if ((crel & 3) == 3 || it == iters - 1) {  // group of 4 chunks complete (or slice ends)
        const int gid = ((chunk_begin + (crel & ~3)) << 3) | (wave << 1) | h;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          if (gmax[t] > lv[t][K - 1]) topk_insert<K>(lv[t], li[t], gmax[t], gid);
          gmax[t] = -INFINITY;
        }
      }
    }
    if (it + 1 < iters) store_iter(bankbuf + (cur ^ 1) * KNN_BUF);
    __syncthreads();
  }

  // ---- merge: (lane, lane+32) by shuffle, then the 4 waves through LDS (bank buffers are free)
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    float pv[K];
    int pi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      pv[j] = __shfl(lv[t][j], lane ^ 32, 64);
      pi[j] = __shfl(li[t][j], lane ^ 32, 64);
    }
    if (h == 0) {
#pragma unroll
      for (int j = 0; j < K; ++j)
        if (better(pv[j], pi[j], lv[t][K - 1], li[t][K - 1])) topk_insert<K>(lv[t], li[t], pv[j], pi[j]);
    }
  }
  float* msim = reinterpret_cast<float*>(smem);                  // [QB][4][K]
  int* midx = reinterpret_cast<int*>(smem + QB * 4 * K * 4);     // [QB][4][K]
  if (h == 0) {
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
      for (int j = 0; j < K; ++j) {
        msim[((t * 32 + r) * 4 + wave) * K + j] = lv[t][j];
        midx[((t * 32 + r) * 4 + wave) * K + j] = li[t][j];
      }
  }
  __syncthreads();
  if (tid < QB) {
    float fv[K];
    int fi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      fv[j] = msim[(tid * 4 + 0) * K + j];
      fi[j] = midx[(tid * 4 + 0) * K + j];
    }
    for (int w = 1; w < 4; ++w) {
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const float cv = msim[(tid * 4 + w) * K + j];
        const int ci = midx[(tid * 4 + w) * K + j];
        if (better(cv, ci, fv[K - 1], fi[K - 1])) topk_insert<K>(fv, fi, cv, ci);
      }
    }
    const size_t o = ((size_t)(q0 + tid) * nslices + slice) * K;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      part_sim[o + j] = fv[j];
      part_idx[o + j] = fi[j];
    }
  }
}

// One wave per query: merge `parts` sorted lists of `kin` candidates into the best `kout`.
// candidate (p, j) of query q lives at q*stride_q + p*stride_p + j.
template <int K>
__global__ __launch_bounds__(64) void knn_merge_lists(const float* __restrict__ in_sim,
                                                      const int* __restrict__ in_idx, int parts,
                                                      int kin, long long stride_q,
                                                      long long stride_p, int kout,
                                                      float* __restrict__ out_sim,
                                                      int* __restrict__ out_idx) {
  const int q = blockIdx.x;
  const int lane = threadIdx.x;
  float v[K];
  int ix[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    v[j] = -INFINITY;
    ix[j] = INT_MAX;
  }
  const int total = parts * kin;
  for (int c = lane; c < total; c += 64) {
    const int p = c / kin, j = c - p * kin;
    const size_t o = (size_t)q * stride_q + (size_t)p * stride_p + j;
    const float cv = in_sim[o];
    const int ci = in_idx[o];
    if (better(cv, ci, v[K - 1], ix[K - 1])) topk_insert<K>(v, ix, cv, ci);
  }
  for (int t = 0; t < kout; ++t) {
    float bv = v[0];
    int bi = ix[0];
    int bl = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      const int ol = __shfl_xor(bl, o, 64);
      if (better(ov, oi, bv, bi) || (ov == bv && oi == bi && ol < bl)) {
        bv = ov;
        bi = oi;
        bl = ol;
      }
    }
    if (lane == bl) {
#pragma unroll
      for (int j = 0; j < K - 1; ++j) {
        v[j] = v[j + 1];
        ix[j] = ix[j + 1];
      }
      v[K - 1] = -INFINITY;
      ix[K - 1] = INT_MAX;
    }
    if (lane == 0) {
      out_sim[(size_t)q * kout + t] = bv;
      out_idx[(size_t)q * kout + t] = bi;
    }
  }
}

// Exact scores of the candidate rows of one query: kg groups x 64 rows (group id -> rows as the
// streaming kernel enumerates them), then the best `kout` by (value desc, index asc).
template <int DT>
__global__ __launch_bounds__(256) void knn_rescore(const uint8_t* __restrict__ query,
                                                   const uint8_t* __restrict__ bank, int n, int d,
                                                   int rowbytes, const int* __restrict__ gids, int kg,
                                                   int chunks_per_slice, int total_chunks, int index_base,
                                                   int kout, float* __restrict__ out_sim,
                                                   int* __restrict__ out_idx) {
  extern __shared__ __attribute__((aligned(16))) uint8_t rs_smem[];
  float* qf = reinterpret_cast<float*>(rs_smem);            // [d]
  float* cv = qf + d;                                        // [kg*64]
  int* ci = reinterpret_cast<int*>(cv + kg * 64);            // [kg*64]
  const int q = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < d; c += 256) {
    if constexpr (DT == WM_BF16) qf[c] = bf2f(reinterpret_cast<const uint16_t*>(query + (size_t)q * rowbytes)[c]);
    else qf[c] = reinterpret_cast<const float*>(query + (size_t)q * rowbytes)[c];
  }
  __syncthreads();
  const int ncand = kg * 64;
  for (int c = tid; c < ncand; c += 256) {
    const int gid = gids[(size_t)q * kg + (c >> 6)];
    const int j = c & 63, cc = j >> 4, e = j & 15;
    float v = -INFINITY;
    int row = INT_MAX;
    if (gid != INT_MAX) {
      const int c0 = gid >> 3, w = (gid >> 1) & 3, hh = gid & 1;
      const int slice = c0 / chunks_per_slice;
      int cend = (slice + 1) * chunks_per_slice;
      if (cend > total_chunks) cend = total_chunks;
      const int chunk = c0 + cc;
      const int r = chunk * KNN_ROWS + w * 32 + acc_row(e, hh);
      if (chunk < cend && r < n) {
        row = r;
        float acc = 0.f;
        const uint8_t* br = bank + (size_t)r * rowbytes;
        for (int k = 0; k < d; k += 8) {
          if constexpr (DT == WM_BF16) {
            const uint4 u = *reinterpret_cast<const uint4*>(br + k * 2);
            const uint32_t ws[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              acc = fmaf(bf2f((uint16_t)(ws[x] & 0xffff)), qf[k + 2 * x], acc);
              acc = fmaf(bf2f((uint16_t)(ws[x] >> 16)), qf[k + 2 * x + 1], acc);
            }
          } else {
            const float4 a = *reinterpret_cast<const float4*>(br + k * 4);
            const float4 b = *reinterpret_cast<const float4*>(br + k * 4 + 16);
            acc = fmaf(a.x, qf[k], acc); acc = fmaf(a.y, qf[k + 1], acc);
            acc = fmaf(a.z, qf[k + 2], acc); acc = fmaf(a.w, qf[k + 3], acc);
            acc = fmaf(b.x, qf[k + 4], acc); acc = fmaf(b.y, qf[k + 5], acc);
            acc = fmaf(b.z, qf[k + 6], acc); acc = fmaf(b.w, qf[k + 7], acc);
          }
        }
        v = acc;
      }
    }
    cv[c] = v;
    ci[c] = row;
  }
  __syncthreads();
  if (tid < 64) {  // one wave: kout rounds of arg-best over the candidates
    for (int t = 0; t < kout; ++t) {
      float bv = -INFINITY;
      int bi = INT_MAX, bp = -1;
      for (int c = tid; c < ncand; c += 64) {
        const float v = cv[c];
        const int i = ci[c];
        if (i != INT_MAX && (bp < 0 || better(v, i, bv, bi))) {
          bv = v;
          bi = i;
          bp = c;
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        const int op = __shfl_xor(bp, o, 64);
        if (op >= 0 && (bp < 0 || better(ov, oi, bv, bi))) {
          bv = ov;
          bi = oi;
          bp = op;
        }
      }
      if (tid == 0) {
        out_sim[(size_t)q * kout + t] = bv;
        out_idx[(size_t)q * kout + t] = bp >= 0 ? bi + index_base : -1;
        if (bp >= 0) ci[bp] = INT_MAX;  // consumed
      }
      __builtin_amdgcn_s_waitcnt(0);  // tid 0's LDS write lands before the next round's reads
      __builtin_amdgcn_wave_barrier();
    }
  }
}

constexpr int VOTE_MAX_CLASSES = 64;

__global__ void knn_vote_kernel(const float* __restrict__ sim, const int* __restrict__ idx,
                                const long long* __restrict__ labels, int nq, int k, int nc,
                                float t, long long* __restrict__ pred,
                                float* __restrict__ scores_out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  float score[VOTE_MAX_CLASSES];
  for (int c = 0; c < nc; ++c) score[c] = 0.f;
  for (int j = 0; j < k; ++j) {
    // reference order of operations: (sim / t).exp(), then a sum over the k neighbours in order
    const float w = expf(sim[(size_t)q * k + j] / t);
    const int lab = (int)labels[idx[(size_t)q * k + j]];
    for (int c = 0; c < nc; ++c) score[c] += (c == lab) ? w : 0.f;
  }
  if (scores_out)
    for (int c = 0; c < nc; ++c) scores_out[(size_t)q * nc + c] = score[c];
  // selection sort: descending score, ties -> lower class id
  unsigned long long used = 0ull;
  for (int o = 0; o < nc; ++o) {
    int best = -1;
    for (int c = 0; c < nc; ++c) {
      if ((used >> c) & 1ull) continue;
      if (best < 0 || score[c] > score[best]) best = c;
    }
    pred[(size_t)q * nc + o] = best;
    used |= 1ull << best;
  }
}

inline int pick_qt(int rowbytes, int nq, int kt) {
  int qt = kt > 8 ? 2 : 4;  // K=16 lists at QT=4 would not fit the register file
  while (qt > 1 && qt * 32 * rowbytes > 65536) qt >>= 1;
  while (qt > 1 && (qt / 2) * 32 >= nq) qt >>= 1;  // do not carry empty query sub-tiles
  return qt;
}

struct KnnPlan {
  int qt, qtiles, nslices, chunks_per_slice, kt;
};

inline KnnPlan make_plan(int nq, int n, int rowbytes, int k) {
  KnnPlan p;
  p.kt = k <= 8 ? 8 : 16;
  p.qt = pick_qt(rowbytes, nq, p.kt);
  p.qtiles = wm_cdiv(nq, p.qt * 32);
  const int total_chunks = wm_cdiv(n, KNN_ROWS);
  int want = 512 / p.qtiles;
  if (want < 1) want = 1;
  p.nslices = total_chunks < want ? total_chunks : want;
  p.chunks_per_slice = wm_cdiv(total_chunks, p.nslices);
  p.nslices = wm_cdiv(total_chunks, p.chunks_per_slice);
  return p;
}

template <int DT, int QT, int K>
int launch_block(const KnnPlan& p, const void* query, const void* bank, int nq, int n,
                 int rowbytes, float* ps, int* pi, hipStream_t st) {
  const size_t lds = 2 * (size_t)KNN_BUF + (size_t)QT * 32 * rowbytes;
  static bool attr_set = false;  // idempotent; a race only repeats the call
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_block_topk<DT, QT, K>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KNN_BUF + 65536);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid(p.nslices, p.qtiles);
  knn_block_topk<DT, QT, K><<<grid, KNN_THREADS, lds, st>>>(
      static_cast<const uint8_t*>(query), static_cast<const uint8_t*>(bank), nq, n, rowbytes,
      p.chunks_per_slice, p.nslices, ps, pi);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

template <int DT, int K>
int dispatch_qt(const KnnPlan& p, const void* query, const void* bank, int nq, int n, int rowbytes,
                float* ps, int* pi, hipStream_t st) {
  if constexpr (K <= 8) {
    if (p.qt == 4) return launch_block<DT, 4, K>(p, query, bank, nq, n, rowbytes, ps, pi, st);
  }
  switch (p.qt) {
    case 2: return launch_block<DT, 2, K>(p, query, bank, nq, n, rowbytes, ps, pi, st);
    default: return launch_block<DT, 1, K>(p, query, bank, nq, n, rowbytes, ps, pi, st);
  }
}

}  // namespace

extern "C" size_t wm_knn_topk_workspace_bytes(int nq, int n, int d, int k) {
  if (nq <= 0 || n <= 0 || d <= 0 || k <= 0 || k > 16) return 0;
  // sized for the wider element type so one workspace serves both dtypes
  const KnnPlan pb = make_plan(nq, n, d * 2, k);
  const KnnPlan pf = make_plan(nq, n, d * 4, k);
  const size_t a = (size_t)pb.qtiles * pb.qt * 32 * pb.nslices * pb.kt * 8;
  const size_t b = (size_t)pf.qtiles * pf.qt * 32 * pf.nslices * pf.kt * 8;
  return (a > b ? a : b) + (size_t)nq * pb.kt * 8 + 256;
}

extern "C" int wm_knn_topk(const void* query, const void* bank, int nq, int n, int d, int dtype,
                           int k, int bank_index_base, float* out_sim, int32_t* out_idx,
                           void* workspace, size_t workspace_bytes, void* stream) {
  WM_REQUIRE(query && bank && out_sim && out_idx && workspace, WM_EINVAL);
  WM_REQUIRE(nq > 0 && n > 0 && d > 0 && k > 0, WM_EINVAL);
  WM_REQUIRE(dtype == WM_F32 || dtype == WM_BF16, WM_EUNSUPPORTED);
  WM_REQUIRE(k <= 16 && k <= n, WM_EUNSUPPORTED);
  const int rowbytes = d * (dtype == WM_BF16 ? 2 : 4);
  WM_REQUIRE(rowbytes % KNN_SLAB == 0 && rowbytes <= 2048, WM_EUNSUPPORTED);
  WM_REQUIRE((reinterpret_cast<uintptr_t>(query) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(bank) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
             WM_EALIGN);
  const KnnPlan p = make_plan(nq, n, rowbytes, k);
  const size_t cand = (size_t)p.qtiles * p.qt * 32 * p.nslices * p.kt;
  WM_REQUIRE(workspace_bytes >= cand * 8 + (size_t)nq * p.kt * 8, WM_EWORKSPACE);
  float* ps = static_cast<float*>(workspace);
  int* pi = reinterpret_cast<int*>(ps + cand);
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc;
  if (dtype == WM_BF16) {
    rc = p.kt == 8 ? dispatch_qt<WM_BF16, 8>(p, query, bank, nq, n, rowbytes, ps, pi, st)
                   : dispatch_qt<WM_BF16, 16>(p, query, bank, nq, n, rowbytes, ps, pi, st);
  } else {
    rc = p.kt == 8 ? dispatch_qt<WM_F32, 8>(p, query, bank, nq, n, rowbytes, ps, pi, st)
                   : dispatch_qt<WM_F32, 16>(p, query, bank, nq, n, rowbytes, ps, pi, st);
  }
  if (rc != WM_OK) return rc;
  // candidate groups of every query: [nq][kt] (value, group id), best first
  float* gsim = ps + 2 * cand;
  int* gidx = reinterpret_cast<int*>(gsim + (size_t)nq * p.kt);
  const long long sq = (long long)p.nslices * p.kt, sp = p.kt;
  if (p.kt == 8)
    knn_merge_lists<8><<<nq, 64, 0, st>>>(ps, pi, p.nslices, p.kt, sq, sp, p.kt, gsim, gidx);
  else
    knn_merge_lists<16><<<nq, 64, 0, st>>>(ps, pi, p.nslices, p.kt, sq, sp, p.kt, gsim, gidx);
  WM_LAUNCH_CHECK();
  const size_t lds = (size_t)d * 4 + (size_t)p.kt * 64 * 8;
  const int total_chunks = wm_cdiv(n, KNN_ROWS);
  if (dtype == WM_BF16)
    knn_rescore<WM_BF16><<<nq, 256, lds, st>>>(static_cast<const uint8_t*>(query), static_cast<const uint8_t*>(bank), n,
                                               d, rowbytes, gidx, p.kt, p.chunks_per_slice, total_chunks,
                                               bank_index_base, k, out_sim, out_idx);
  else
    knn_rescore<WM_F32><<<nq, 256, lds, st>>>(static_cast<const uint8_t*>(query), static_cast<const uint8_t*>(bank), n,
                                              d, rowbytes, gidx, p.kt, p.chunks_per_slice, total_chunks,
                                              bank_index_base, k, out_sim, out_idx);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_knn_merge(const float* in_sim, const int32_t* in_idx, int parts, int nq, int k,
                            float* out_sim, int32_t* out_idx, void* stream) {
  WM_REQUIRE(in_sim && in_idx && out_sim && out_idx, WM_EINVAL);
  WM_REQUIRE(parts > 0 && nq > 0 && k > 0, WM_EINVAL);
  WM_REQUIRE(k <= 16, WM_EUNSUPPORTED);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long sq = k, sp = (long long)nq * k;
  if (k <= 8)
    knn_merge_lists<8><<<nq, 64, 0, st>>>(in_sim, in_idx, parts, k, sq, sp, k, out_sim, out_idx);
  else
    knn_merge_lists<16><<<nq, 64, 0, st>>>(in_sim, in_idx, parts, k, sq, sp, k, out_sim, out_idx);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_knn_vote(const float* sim, const int32_t* idx, const int64_t* bank_labels, int nq,
                           int k, int num_classes, float temperature, int64_t* pred_labels,
                           float* scores, void* stream) {
  WM_REQUIRE(sim && idx && bank_labels && pred_labels, WM_EINVAL);
  WM_REQUIRE(nq > 0 && k > 0 && num_classes > 0 && temperature > 0.f, WM_EINVAL);
  WM_REQUIRE(num_classes <= VOTE_MAX_CLASSES, WM_EUNSUPPORTED);
  hipStream_t st = static_cast<hipStream_t>(stream);
  knn_vote_kernel<<<wm_cdiv(nq, 64), 64, 0, st>>>(sim, idx,
                                                  reinterpret_cast<const long long*>(bank_labels),
                                                  nq, k, num_classes, temperature,
                                                  reinterpret_cast<long long*>(pred_labels), scores);
  WM_LAUNCH_CHECK();
  return WM_OK;
}
