/*
 * wafer_hip.h — C ABI of libwafer_hip.so, the MI355X (gfx950) hot path for the
 * self-supervised wafer-map pipeline of faris-k/self-supervised-wafermaps.
 *
 * The reference has no FFI of its own (pure Python on torch/lightly/timm/torchvision); its seam is
 * the set of Python callables listed in SURVEY.md §8(b).  Every entry point below names the
 * reference call site (file:line under the reference checkout) whose device work it replaces.
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes, no torch types; every pointer is DEVICE memory owned by the caller
 *     (the library never allocates, frees or retains device memory);
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no implicit sync;
 *   - return 0 on success, a negative WM_E* code for bad arguments, a positive hipError_t value
 *     when the HIP runtime reports a launch failure;
 *   - re-entrant, no mutable global state.
 *   - activation tensors are NHWC ("channels_last"); bf16 tensors are raw uint16 bit patterns.
 */
#ifndef WAFER_HIP_H
#define WAFER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WM_ABI_VERSION 1

/* error codes */
#define WM_OK 0
#define WM_EINVAL (-1)       /* null pointer / non-positive size */
#define WM_EUNSUPPORTED (-2) /* shape/dtype combination the kernels do not implement */
#define WM_EWORKSPACE (-3)   /* workspace too small */
#define WM_EALIGN (-4)       /* pointer or row pitch not aligned as required */

/* dtypes */
#define WM_F32 0
#define WM_BF16 1

int wm_version(void);
const char* wm_error_string(int code);

/* ---------------------------------------------------------------------------------------------
 * Two-view augmentation.
 * Replaces the per-sample CPU pipeline
 *   get_base_transforms          src/ssl_wafermap/transforms/augmentations.py:253-332
 *   get_inference_transforms     src/ssl_wafermap/transforms/augmentations.py:335-357
 *   DieNoise.__call__            augmentations.py:27-36
 *   MedianFilter.__call__        augmentations.py:103-107
 *   DPWTransform.dpw_transform   augmentations.py:182-227
 *   MultiCropViewTransform       src/ssl_wafermap/transforms/wafer_multicrop_transform.py:16-85
 *   WaferMapDataset.__getitem__  src/ssl_wafermap/data/dataset.py:29-34
 * with one launch over a ragged HBM-resident wafer store.  Every random decision is an input
 * (drawn on the host by the transform classes), so the CPU oracle sees identical choices.
 * ------------------------------------------------------------------------------------------- */

#define WM_AUG_NONE 0
#define WM_AUG_DIENOISE 1
#define WM_AUG_DPW 2
#define WM_AUG_MEDIAN3 3

#define WM_IMG_NCHW_F32 0  /* [n][3][S][S] float32 (the reference's tensor layout)            */
#define WM_IMG_NHWC_BF16 1 /* [n][S][S][3] bf16     (channels_last, feeds the conv kernels)   */
#define WM_IMG_HW_U8 2     /* [n][S][S]    uint8    (pre-ToTensor grey image, for parity checks) */

typedef struct WmViewParams {
  int32_t sample;      /* index into the wafer store                                         */
  int32_t out_slot;    /* image index inside `out`                                           */
  int32_t op;          /* WM_AUG_*                                                           */
  uint32_t noise_seed; /* DieNoise: key of the counter RNG (rand(r,c) = wm_rand01(seed, r*W+c)) */
  float noise_p;       /* DieNoise flip probability                                          */
  int32_t dpw_h;       /* DPW: int(H*scale), computed on the host in double like the reference */
  int32_t dpw_w;       /* DPW: int(W*scale)                                                  */
  int32_t rot90;       /* 1 = rotate 90 deg counter-clockwise after the resize               */
  int32_t vflip;       /* 1 = vertical flip   (after rot90)                                   */
  int32_t hflip;       /* 1 = horizontal flip (after vflip)                                   */
  int32_t crop;        /* 1 = RandomResizedCrop box (i,j,h,w) of the img_size image           */
  int32_t crop_i, crop_j, crop_h, crop_w;
  int32_t reserved;
} WmViewParams;

/* wafers: concatenated uint8 maps; wafer s is rows-major heights[s] x widths[s] at offsets[s].
 * params: n_views entries (device).  img_size: side of the square Resize target (224).
 * out_size: side of the emitted image (== img_size unless crop; % 8 == 0).  mean/std: Normalize
 * stats.  max_wafer_elems: largest H*W in the store (host-known; sizes the LDS images; a wafer
 * above it is skipped).  Largest supported wafer side, img_size and out_size: 256.
 * DieNoise RNG: rand(r,c) = (lowbias32(idx ^ lowbias32(seed ^ 0x9E3779B9)) >> 8) * 2^-24 with
 * idx = r*W + c; the flip test is rand < noise_p in float32 (oracle/augment.py: rand01). */
int wm_augment_views(const uint8_t* wafers, const int64_t* offsets, const int32_t* heights,
                     const int32_t* widths, int n_wafers, int max_wafer_elems,
                     const WmViewParams* params, int n_views, int img_size, int out_size,
                     int out_format, int normalize, float mean, float std, void* out,
                     void* stream);

/* ---------------------------------------------------------------------------------------------
 * kNN retrieval.
 * Replaces lightly.utils.benchmarking.knn_predict as called at
 *   src/ssl_wafermap/models/knn.py:91-98   (mm -> topk -> gather -> exp -> one-hot -> argsort)
 * and the bank build's F.normalize at src/ssl_wafermap/models/knn.py:76-80.
 * Blocked pairwise-dot on MFMA with a running in-register top-k; the [B,N] similarity matrix is
 * never written to HBM.
 * ------------------------------------------------------------------------------------------- */

/* query [nq][d], bank [n][d] row-major, both `dtype` (WM_F32 or WM_BF16); row bytes % 256 == 0.
 * out_sim [nq][k] float32 descending, out_idx [nq][k] int32 (ties: lower bank index first).
 * 1 <= k <= 16, k <= n.  bank_index_base is added to every emitted index (sharded banks). */
size_t wm_knn_topk_workspace_bytes(int nq, int n, int d, int k);
int wm_knn_topk(const void* query, const void* bank, int nq, int n, int d, int dtype, int k,
                int bank_index_base, float* out_sim, int32_t* out_idx, void* workspace,
                size_t workspace_bytes, void* stream);

/* Merge `parts` candidate lists per query ([parts][nq][k], e.g. all-gathered shard results)
 * into the global top-k.  in_* and out_* may not alias. */
int wm_knn_merge(const float* in_sim, const int32_t* in_idx, int parts, int nq, int k,
                 float* out_sim, int32_t* out_idx, void* stream);

/* Weighted vote: w = exp(sim/t); score[c] = sum_j w_j [labels[idx_j]==c]; classes sorted by score
 * descending (ties: lower class id first) into pred_labels [nq][num_classes] int64 — the tensor
 * knn_predict returns; scores [nq][num_classes] float32 optional (may be NULL). */
int wm_knn_vote(const float* sim, const int32_t* idx, const int64_t* bank_labels, int nq, int k,
                int num_classes, float temperature, int64_t* pred_labels, float* scores,
                void* stream);

/* Row-wise L2 normalisation y = x / max(||x||, eps)  (torch.nn.functional.normalize, dim=1),
 * x float32 or bf16 [rows][d] -> y `out_dtype`; inv_norm [rows] float32 optional. */
int wm_l2_normalize(const void* x, int in_dtype, int rows, int d, float eps, void* y,
                    int out_dtype, float* inv_norm, void* stream);
/* Backward of the above: dx = (dy - y * <dy,y>) * inv_norm ; all float32. */
int wm_l2_normalize_bwd(const float* dy, const float* y, const float* inv_norm, int rows, int d,
                        float* dx, void* stream);

/* ---------------------------------------------------------------------------------------------
 * NT-Xent.  Replaces lightly.loss.NTXentLoss()(z0, z1) at
 *   scripts/WM811k_benchmark.py:234,246
 * zn: L2-normalised local rows [2*b_local][d] float32 (z0 rows then z1 rows);
 * zall: all rows the loss contrasts against, [2][b_global][d] float32 (view-major; == zn when
 *   not gathered); rank_offset = rank * b_local (lightly gather_distributed label offset).
 * Global row id of (view v, sample m) is v*b_global + m; local row (v,i) is global
 * (v, rank_offset+i); its positive is (1-v, rank_offset+i); it is excluded from its own softmax.
 * Forward emits, per local row, lse_i = log sum_{j != i} exp(s_ij/T) and
 * loss_rows_i = lse_i - s_{i,pos(i)}/T  (the loss is their mean: CrossEntropyLoss(mean) over the
 * [2B, 2B-1] logits the reference builds, which are never materialised here).
 * Backward emits d(sum over ALL ranks' losses)/d zn for the local rows, i.e. lightly's
 * GatherLayer semantics (row part + column part):
 *   dzn_i = grad_scale/T * sum_{j != i} (p_ij + p_ji - 2*[j == pos(i)]) * zall_j,
 *   p_ij = exp(s_ij/T - lse_i);  lse_all [2][b_global] holds lse for every global row
 *   (all-gathered when distributed; == lse when not).  grad_scale = dL/dloss / (2*b_local).
 * d % 32 == 0, d <= 256.
 * ------------------------------------------------------------------------------------------- */
int wm_ntxent_fwd(const float* zn, const float* zall, int b_local, int b_global, int rank_offset,
                  int d, float temperature, float* lse, float* loss_rows, void* stream);
int wm_ntxent_bwd(const float* zn, const float* zall, const float* lse_all, int b_local,
                  int b_global, int rank_offset, int d, float temperature, float grad_scale,
                  float* dzn, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WAFER_HIP_H */
