// General-shape inner-product top-k with an optional per-row bias: the retrieval path of the
// embedding-dump flow (reference notebooks 3.0-Embeddings-inference cell 7 and
// 2.0-Figures-nearest-neighbors cell 2: sklearn NearestNeighbors on StandardScaler'd features).
//
// wm_knn_topk (knn.hip) is the tuned streaming kernel but takes rows of at most 2 KB and no bias.
// Euclidean ranking needs one more column than the features have:
//     argmin ||q - x||^2 = argmax (q.x - ||x||^2 / 2)
// so this kernel scores  s = q.x + bias[row]  in float32 for any d <= 1024 (d % 4 == 0).
// One block = 8 queries x a slice of bank rows.  A wave owns whole rows: lane l holds chunks
// l, l + 64, ... of the row (float4 each, coalesced 1 KiB per instruction) and the same chunks of the
// 8 queries in registers; 8 wave reductions per row; lane q (0..7) keeps the sorted top-k of query q.
// Partial lists [slice * 4 + wave][nq][k] are combined by wm_knn_merge.
// Roofline: HBM (the bank is streamed once per 8 queries); algorithmic bytes n * d * 4 per 8 queries.
#include "common.h"

namespace {

constexpr int KG_THREADS = 256;
constexpr int KG_Q = 8;       // queries per block
constexpr int KG_MAXC = 4;    // float4 chunks per lane: d <= 64 * 4 * 4 = 1024
constexpr int KG_K = 16;

__global__ __launch_bounds__(KG_THREADS) void knn_general(const float* __restrict__ query, const float* __restrict__ bank,
                                                          const float* __restrict__ bias, int nq, int n, int d, int k,
                                                          int rows_per_slice, int index_base,
                                                          const float* __restrict__ after_sim,
                                                          const int* __restrict__ after_idx, float* __restrict__ ps,
                                                          int* __restrict__ pi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q0 = blockIdx.y * KG_Q;
  const int nch = d >> 2;
  float4 qv[KG_Q][KG_MAXC];
#pragma unroll
  for (int q = 0; q < KG_Q; ++q)
#pragma unroll
    for (int c = 0; c < KG_MAXC; ++c) {
      const int ch = lane + 64 * c;
      qv[q][c] = (q0 + q < nq && ch < nch) ? *reinterpret_cast<const float4*>(query + (size_t)(q0 + q) * d + ch * 4)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  float best[KG_K];
  int bidx[KG_K];
#pragma unroll
  for (int j = 0; j < KG_K; ++j) {
    best[j] = -INFINITY;
    bidx[j] = 0x7fffffff;
  }
  // optional cursor: only rows strictly AFTER (after_sim[q], after_idx[q]) in the list order (score descending,
  // index ascending) compete -- the next page of a top-k that is longer than KG_K
  float cur_v = INFINITY;
  int cur_i = -1;
  if (after_sim != nullptr && lane < KG_Q && q0 + lane < nq) {
    cur_v = after_sim[q0 + lane];
    cur_i = after_idx[q0 + lane] - index_base;
  }
  const int r0 = blockIdx.x * rows_per_slice;
  int r1 = r0 + rows_per_slice;
  if (r1 > n) r1 = n;
  for (int r = r0 + wave; r < r1; r += 4) {
    float4 xv[KG_MAXC];
#pragma unroll
    for (int c = 0; c < KG_MAXC; ++c) {
      const int ch = lane + 64 * c;
      xv[c] = ch < nch ? *reinterpret_cast<const float4*>(bank + (size_t)r * d + ch * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float b = bias ? bias[r] : 0.f;
    float mine = -INFINITY;  // lane q ends up with the score of query q
#pragma unroll
    for (int q = 0; q < KG_Q; ++q) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < KG_MAXC; ++c) {
        s = fmaf(qv[q][c].x, xv[c].x, s);
        s = fmaf(qv[q][c].y, xv[c].y, s);
        s = fmaf(qv[q][c].z, xv[c].z, s);
        s = fmaf(qv[q][c].w, xv[c].w, s);
      }
      s = wave_sum(s) + b;
      if (lane == q) mine = s;
    }
    // sorted insertion (descending; ties: lower index first) by the lane that owns the query
    const bool after_cursor = mine < cur_v || (mine == cur_v && r > cur_i);
    if (lane < KG_Q && after_cursor && (mine > best[KG_K - 1] || (mine == best[KG_K - 1] && r < bidx[KG_K - 1]))) {
      float v = mine;
      int vi = r;
#pragma unroll
      for (int j = 0; j < KG_K; ++j) {
        const bool better = v > best[j] || (v == best[j] && vi < bidx[j]);
        const float tv = best[j];
        const int ti = bidx[j];
        if (better) {
          best[j] = v;
          bidx[j] = vi;
          v = tv;
          vi = ti;
        }
      }
    }
  }
  if (lane < KG_Q && q0 + lane < nq) {
    const size_t part = (size_t)blockIdx.x * 4 + wave;
    float* os = ps + (part * nq + q0 + lane) * k;
    int* oi = pi + (part * nq + q0 + lane) * k;
    for (int j = 0; j < k; ++j) {
      os[j] = best[j];
      oi[j] = bidx[j] == 0x7fffffff ? 0x7fffffff : bidx[j] + index_base;
    }
  }
}

inline int kg_slices(int n) {
  int s = wm_cdiv(n, 4 * 64);  // at least 64 rows per wave
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  return s;
}

}  // namespace

extern "C" int wm_knn_topk_general_after(const float* query, const float* bank, const float* bias, int nq, int n, int d,
                                         int k, int bank_index_base, const float* after_sim, const int32_t* after_idx,
                                         float* out_sim, int32_t* out_idx, void* workspace, size_t workspace_bytes,
                                         void* stream);

extern "C" size_t wm_knn_topk_general_workspace_bytes(int nq, int n, int d, int k) {
  if (nq <= 0 || n <= 0 || d <= 0 || k <= 0 || k > KG_K) return 0;
  return (size_t)kg_slices(n) * 4 * nq * k * 8 + 256;
}

extern "C" int wm_knn_topk_general(const float* query, const float* bank, const float* bias, int nq, int n, int d,
                                   int k, int bank_index_base, float* out_sim, int32_t* out_idx, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  return wm_knn_topk_general_after(query, bank, bias, nq, n, d, k, bank_index_base, nullptr, nullptr, out_sim, out_idx,
                                   workspace, workspace_bytes, stream);
}

extern "C" int wm_knn_topk_general_after(const float* query, const float* bank, const float* bias, int nq, int n, int d,
                                         int k, int bank_index_base, const float* after_sim, const int32_t* after_idx,
                                         float* out_sim, int32_t* out_idx, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  WM_REQUIRE(query && bank && out_sim && out_idx && workspace, WM_EINVAL);
  WM_REQUIRE((after_sim == nullptr) == (after_idx == nullptr), WM_EINVAL);
  WM_REQUIRE(nq > 0 && n > 0 && d > 0 && k > 0 && k <= n, WM_EINVAL);
  WM_REQUIRE(k <= KG_K && d % 4 == 0 && d <= 64 * 4 * KG_MAXC, WM_EUNSUPPORTED);
  WM_REQUIRE((reinterpret_cast<uintptr_t>(query) & 15) == 0 && (reinterpret_cast<uintptr_t>(bank) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
             WM_EALIGN);
  const int slices = kg_slices(n);
  const size_t cand = (size_t)slices * 4 * nq * k;
  WM_REQUIRE(workspace_bytes >= cand * 8, WM_EWORKSPACE);
  float* ps = static_cast<float*>(workspace);
  int* pi = reinterpret_cast<int*>(ps + cand);
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 grid(slices, wm_cdiv(nq, KG_Q));
  knn_general<<<grid, KG_THREADS, 0, st>>>(query, bank, bias, nq, n, d, k, wm_cdiv(n, slices), bank_index_base, after_sim,
                                           after_idx, ps, pi);
  WM_LAUNCH_CHECK();
  return wm_knn_merge(ps, pi, slices * 4, nq, k, out_sim, out_idx, stream);
}
