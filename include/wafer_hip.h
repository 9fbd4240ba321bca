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

#define WM_ABI_VERSION 4

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
/* Small device-side pieces that keep framework launches out of the captured training step: clear a buffer (the flat
 * gradient arena of optimizer.zero_grad; a kernel, never a memset node: profiles/r02_nan_root_cause.md) and
 * out[0] = mean_i f(x_i), f = identity or sqrt(scale * x_i) (mean of the NT-Xent row losses, scripts/WM811k_benchmark.py:247;
 * lightly's std_of_l2_normalized monitor, :239), one block, fixed summation order. */
int wm_fill_zero(void* p, size_t bytes, void* stream);
int wm_mean_f32(const float* x, long long n, float scale, int sqrt_of, float* out, void* stream);
/* y = x * *scale_dev over n bf16 elements (n % 8 == 0, 16-byte aligned): the device-resident upstream gradient of a scalar
 * loss applied to the gradient tensor its forward pass saved (no host read of the scalar, no f32 round trip through HBM). */
int wm_scale_bf16(const void* x, long long n, const float* scale_dev, void* y, void* stream);

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
#define WM_IMG_S2D_BF16 3  /* [n][S/2][S/2][16] bf16: 2x2 space-to-depth of the NHWC image, channel = dh*6 + dw*3 + c,
                              12..15 zero -- the layout the stem convolution consumes (wm_image_to_s2d's output) */

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

/* query [nq][d], bank [n][d] row-major, both `dtype` (WM_F32 or WM_BF16); row bytes % 256 == 0; d <= 512 (k <= 8 when a row exceeds 1 KB).
 * out_sim [nq][k] float32 descending, out_idx [nq][k] int32 (ties: lower bank index first).
 * 1 <= k <= 16, k <= n.  bank_index_base is added to every emitted index (sharded banks). */
size_t wm_knn_topk_workspace_bytes(int nq, int n, int d, int k);
int wm_knn_topk(const void* query, const void* bank, int nq, int n, int d, int dtype, int k,
                int bank_index_base, float* out_sim, int32_t* out_idx, void* workspace,
                size_t workspace_bytes, void* stream);

/* General-shape variant for the retrieval flow: float32, any d <= 1024 (d % 4 == 0), k <= 16, score =
 * q.x + bias[row] (bias may be NULL).  With bias = -||x||^2 / 2 the ranking is the Euclidean one
 * (sklearn NearestNeighbors on the dumped embeddings, notebooks/2.0-Figures-nearest-neighbors cell 2). */
size_t wm_knn_topk_general_workspace_bytes(int nq, int n, int d, int k);
int wm_knn_topk_general(const float* query, const float* bank, const float* bias, int nq, int n, int d, int k,
                        int bank_index_base, float* out_sim, int32_t* out_idx, void* workspace,
                        size_t workspace_bytes, void* stream);
/* The next page: top-k among the rows strictly AFTER the cursor (after_sim[q], after_idx[q]) in the list order
 * (score descending, index ascending; after_idx in the output index space, i.e. including bank_index_base).
 * Calling it with the last entry of the previous page walks a top-K of any length in pages of <= 16
 * (lightly's knn_predict default knn_k = 200; the reference itself uses 5).  NULL cursors = the first page. */
int wm_knn_topk_general_after(const float* query, const float* bank, const float* bias, int nq, int n, int d, int k,
                              int bank_index_base, const float* after_sim, const int32_t* after_idx, float* out_sim,
                              int32_t* out_idx, void* workspace, size_t workspace_bytes, void* stream);

/* wm_knn_topk for nq queries in batches of `batch` rows, batch i (streaming + selection kernel) on
 * streams[i % n_streams] with slice i % n_streams of `workspaces` (n_streams slices of workspace_bytes_per_lane >=
 * wm_knn_topk_workspace_bytes(batch, ...) bytes, a multiple of 256), every launch queued by this one call: the
 * selection kernel of a batch overlaps the streaming kernels queued behind it on the other streams.  The caller
 * orders the streams against its own (fork before, join after).  (Embedding retrieval over a whole set:
 * notebooks/3.0-Embeddings-inference.ipynb cell 7; BASELINE configs[4].) */
int wm_knn_topk_many(const void* query, const void* bank, int nq, int n, int d, int dtype, int k, int bank_index_base,
                     float* out_sim, int32_t* out_idx, int batch, void* workspaces, size_t workspace_bytes_per_lane,
                     void* const* streams, int n_streams);

/* Merge `parts` candidate lists per query ([parts][nq][k], e.g. all-gathered shard results)
 * into the global top-k.  in_* and out_* may not alias. */
int wm_knn_merge(const float* in_sim, const int32_t* in_idx, int parts, int nq, int k,
                 float* out_sim, int32_t* out_idx, void* stream);

/* Weighted vote: w = exp(sim/t); score[c] = sum_j w_j [labels[idx_j]==c]; classes sorted by score
 * descending (ties: lower class id first) into pred_labels [nq][num_classes] int64 — the tensor
 * knn_predict returns; scores [nq][num_classes] float32 optional (may be NULL).  bank_labels holds
 * n_labels entries; a neighbour index outside [0, n_labels) (the padding of a shard list shorter than k)
 * casts no vote. */
int wm_knn_vote(const float* sim, const int32_t* idx, const int64_t* bank_labels, long long n_labels, int nq, int k,
                int num_classes, float temperature, int64_t* pred_labels, float* scores,
                void* stream);

/* Row-wise L2 normalisation y = x / max(||x||, eps)  (torch.nn.functional.normalize, dim=1),
 * x float32 or bf16 [rows][d] -> y `out_dtype`; inv_norm [rows] float32 optional. */
int wm_l2_normalize(const void* x, int in_dtype, int rows, int d, float eps, void* y,
                    int out_dtype, float* inv_norm, void* stream);
/* Backward of the above: dx = (dy - y * <dy,y>) * inv_norm ; all float32. */
/* dx (out_dtype WM_F32 or WM_BF16) = (dy (dy_dtype WM_F32 or WM_BF16) - y <dy, y>) * inv_norm (* *scale_dev when non-NULL: the device-resident
 * gradient of the scalar loss the normalised rows feed, so that no separate scaling pass is needed). */
int wm_l2_normalize_bwd(const void* dy, int dy_dtype, const float* y, const float* inv_norm, int rows, int d,
                        void* dx, int out_dtype, const float* scale_dev, void* stream);

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
/* The backward splits the columns over ~one block per CU; the splits' partial row gradients go to `workspace`
 * (wm_ntxent_bwd_workspace_bytes) and are summed in split order: no atomics, bit-reproducible. */
size_t wm_ntxent_bwd_workspace_bytes(int b_local, int b_global, int d);
int wm_ntxent_bwd(const float* zn, const float* zall, const float* lse_all, int b_local,
                  int b_global, int rank_offset, int d, float temperature, float grad_scale,
                  float* dzn, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ResNet-18 / projection-head building blocks: bf16 NHWC activations, f32
 * accumulation, f32 parameters.  They replace the cuDNN/ATen kernels behind
 *   timm.create_model("resnet18", num_classes=0)      scripts/WM811k_benchmark.py:231
 *   heads.SimCLRProjectionHead(512, 512, 128)          scripts/WM811k_benchmark.py:233
 *   SimCLR.forward / training_step                     scripts/WM811k_benchmark.py:236-248
 *   torch.optim.SGD(lr, momentum 0.9, wd 5e-4)         scripts/WM811k_benchmark.py:250-255
 * Geometry arguments always describe the FORWARD convolution: input x [N][H][W][C], output
 * y [N][P][Q][K], filter R x S, stride 1|2, zero padding `pad`.
 * ------------------------------------------------------------------------------------------- */

/* y = conv(x, w): w_krsc bf16 [K][R][S][C].  K % 64 == 0; C % 64 == 0, or C == 16 with S == 4
 * (the space-to-depth stem).  A Linear layer is the 1x1 convolution with H = W = P = Q = 1. */
int wm_conv2d_fwd(const void* x, const void* w_krsc, void* y, int N, int H, int W, int C, int K,
                  int R, int S, int P, int Q, int stride, int pad, void* stream);
/* The same convolution, also leaving the per-channel (sum, sum of squares) of its bf16 outputs in stat_part: f32
 * [G][stat_tiles][2 statistics][K] (G = N*P*Q / rows_per_group row groups).  Every 128-row output tile STORES its
 * column sums into its own slot (no atomics, nothing to zero); wm_bn_train_fwd_from_stats adds the slots in slot
 * order, so the statistics -- and everything downstream -- are bit-reproducible from run to run.
 * stat_tiles = wm_conv2d_fwd_stats_tiles(same geometry): rows_per_group / 128, except for the persistent stem
 * kernel, whose workgroups sum over their tiles and write one slot each.  rows_per_group % 128 == 0. */
int wm_conv2d_fwd_stats_tiles(int N, int H, int W, int C, int K, int R, int S, int P, int Q, int stride, int pad,
                              int rows_per_group);
int wm_conv2d_fwd_stats(const void* x, const void* w_krsc, void* y, int N, int H, int W, int C, int K,
                        int R, int S, int P, int Q, int stride, int pad, float* stat_part,
                        int stat_tiles, int rows_per_group, void* stream);
/* dx = conv_transpose(dy, w): w_crsk bf16 [C][R][S][K]; C % 64 == 0. */
int wm_conv2d_dgrad(const void* dy, const void* w_crsk, void* dx, int N, int H, int W, int C, int K,
                    int R, int S, int P, int Q, int stride, int pad, void* stream);
/* dx = conv_transpose(dy, w) + residual: `residual` (bf16, shape of dx) is the gradient that reached
 * the same input through another path (the identity shortcut of a residual block); adding it in the
 * epilogue replaces a separate three-pass elementwise add. */
int wm_conv2d_dgrad_add(const void* dy, const void* w_crsk, const void* residual, void* dx, int N, int H,
                        int W, int C, int K, int R, int S, int P, int Q, int stride, int pad, void* stream);
/* dgrad whose input was relu(BN(bn_y) (+ shortcut)) -- i.e. the convolution that FOLLOWS a BatchNorm + ReLU
 * (timm BasicBlock: conv2 after bn1/act1, and the next block's conv1 after bn2 + shortcut + act2;
 * scripts/WM811k_benchmark.py:231): the epilogue takes the gradient through the ReLU and accumulates the
 * BatchNorm-backward sums, so that BatchNorm's backward needs no reduction pass and no mask:
 *   g  = (conv_transpose(dy, w) (+ residual)) * mask          -> dx (the MASKED gradient, bf16)
 *   stat_part f32 [G][stat_tiles][2][C] = per-tile (sum g, sum g * (bn_y - mean)) per channel, stat_tiles = rows of dx
 *        per group / 128 (plain stores into the tile's slot; wm_bn_train_bwd_from_stats adds the slots in order and
 *        scales the second sum by invstd to sum g * xhat; centred per element, so no cancellation when |mean| >> std)
 * mask = relu_x > 0 when relu_x (the convolution's own forward input, shape of dx) is given; or the bits of relu_mask
 * ([pixels][C / 8] bytes written by wm_bn_train_fwd* for that tensor: 1/16 of its bytes; give one of the two); else
 * recomputed from bn_y as bf16(bn_y * gamma * invstd + beta - mean * gamma * invstd) > 0 (a BatchNorm without shortcut).
 * bn_y has the shape of dx; save_mean / save_invstd [G][C] over G equal groups of images.
 * wm_conv2d_dgrad_bnstat_ok: 1 when the shape is served (every 128-row tile inside one statistics group). */
int wm_conv2d_dgrad_bnstat_ok(int N, int H, int W, int C, int K, int R, int S, int P, int Q, int stride, int pad,
                              int G);
int wm_conv2d_dgrad_bnstat(const void* dy, const void* w_crsk, const void* residual, void* dx, int N, int H, int W,
                           int C, int K, int R, int S, int P, int Q, int stride, int pad, const void* bn_y,
                           const void* relu_x, const void* relu_mask, const float* gamma, const float* beta,
                           const float* save_mean, const float* save_invstd, int G, float* stat_part, int stat_tiles,
                           void* stream);
/* Weight gradient: sum over pixels of dy (x) x, split over pixel ranges.  Split z STORES its partial sums into slab z
 * of dw_slabs (f32 [nsplit][K][R][S][C], nsplit = wm_conv2d_wgrad_splits(same geometry)): no atomics, nothing to zero;
 * wm_wgrad_fold / wm_wgrad_finalize / wm_stem_wgrad_finalize sum the slabs in a fixed order (bit-reproducible). */
int wm_conv2d_wgrad_splits(int N, int H, int W, int C, int K, int R, int S, int P, int Q, int stride, int pad);
int wm_conv2d_wgrad(const void* dy, const void* x, float* dw_slabs, int N, int H, int W, int C, int K,
                    int R, int S, int P, int Q, int stride, int pad, void* stream);
/* Forward with y = conv(x) + bias[k] (+ residual, same shape as y), both optional, added in the
 * epilogue: a Linear layer's bias and the residual add of a transformer block cost no extra pass. */
int wm_conv2d_fwd_bias_res(const void* x, const void* w_krsc, const float* bias, const void* residual, void* y,
                           int N, int H, int W, int C, int K, int R, int S, int P, int Q, int stride, int pad,
                           void* stream);
/* Same, and dbias_slabs [nsplit][K] = per-split sums over pixels of dy (may be NULL): a Linear layer's bias gradient
 * comes out of the weight-gradient launch (one extra MFMA per dY fragment in the first column group's blocks)
 * instead of a separate reduction pass over dy. */
int wm_conv2d_wgrad_bias(const void* dy, const void* x, float* dw_slabs, float* dbias_slabs, int N, int H, int W, int C,
                         int K, int R, int S, int P, int Q, int stride, int pad, void* stream);

/* The two Linear layers around a GELU (ViT MLP: dino vision_transformer.py Mlp, torchvision MLPBlock; reference
 * call sites scripts/WM811k_benchmark.py:548-550, :881-899) with the activation inside the GEMM epilogues:
 *   wm_linear_bias_gelu_fwd: pre = x W^T + bias -> pre_out [rows][K] bf16 (kept for the backward pass),
 *                            gelu(pre) -> y [rows][K]            (x [rows][C], w_krsc [K][C] bf16, bias [K] f32)
 *   wm_linear_dgrad_gelu:    dx = (dy W) * gelu'(pre)            (dy [rows][K], w_crsk [C][K] bf16; pre, dx [rows][C])
 * i.e. the input gradient of the layer that FOLLOWS the activation, already taken through the activation.
 * Replaces a separate bias + GELU pass (forward) and a GELU-gradient + column-sum pass (backward). */
int wm_linear_bias_gelu_fwd(const void* x, const void* w_krsc, const float* bias, void* pre_out, void* y, int rows,
                            int C, int K, void* stream);
int wm_linear_dgrad_gelu(const void* dy, const void* w_crsk, const void* pre, void* dx, int rows, int C, int K,
                         void* stream);

/* The whole transformer MLP block in ONE launch, for forward passes that keep nothing for a backward pass (the DINO
 * teacher, kNN / embedding inference, validation): y = fc2(gelu(fc1(x) + b1)) + b2 (+ residual), with the hidden
 * activation living in LDS only (128 token rows per workgroup, 128 hidden units at a time).  x, y, residual
 * [rows][C] bf16; w1_krsc [H][C], w2_krsc [C][H] bf16 (forward layouts); b1 [H], b2 [C] f32.  Bit-identical to
 * wm_linear_bias_gelu_fwd followed by wm_conv2d_fwd_bias_res.  wm_mlp_fused_fwd_ok: 1 for the served shapes (C = 192,
 * H % 128 == 0, rows <= 32 768: ViT-Tiny; one workgroup per CU -- beyond one round of the chip the two launches win). */
int wm_mlp_fused_fwd_ok(int rows, int C, int H);
int wm_mlp_fused_fwd(const void* x, const void* w1_krsc, const float* b1, const void* w2_krsc, const float* b2,
                     const void* residual, void* y, int rows, int C, int H, void* stream);
/* The same with the block's LayerNorm folded in (forward passes that keep nothing for a backward pass): the normalised
 * rows exist only as register fragments of the workgroup that multiplies them.
 *   wm_ln_linear_fwd     y = LayerNorm(x) W^T + bias                      (norm1 -> qkv; served: C = 192, N >= 384, N % 64 == 0)
 *   wm_ln_mlp_fused_fwd  y = fc2(gelu(fc1(LayerNorm(x)) + b1)) + b2 (+ residual)   (norm2 -> MLP; shapes of wm_mlp_fused_fwd)
 * Same arithmetic as wm_layernorm_fwd followed by the unfused launches (two-pass statistics, bf16 rounding of the
 * normalised rows). */
int wm_ln_linear_fwd_ok(int rows, int C, int N);
int wm_ln_linear_fwd(const void* x, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* w_krsc,
                     const float* bias, void* y, int rows, int C, int N, void* stream);
int wm_ln_mlp_fused_fwd(const void* x, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* w1_krsc,
                        const float* b1, const void* w2_krsc, const float* b2, const void* residual, void* y, int rows,
                        int C, int H, void* stream);

/* f32 OIHW master weights -> bf16 [K][R][S][C] and/or [C][R][S][K] (either may be NULL). */
int wm_weights_prepare(const float* w_oihw, int K, int C, int R, int S, void* w_krsc, void* w_crsk,
                       void* stream);
/* Batched forms of the two layout passes: one launch for every convolution / Linear parameter of a model.
 * descs (DEVICE, n_desc entries, tile0 ascending).
 * wm_layouts_refresh: w = f32 OIHW master weights [K][C][RS]; krsc / crsk = bf16 [K][RS][C] / [C][RS][K] kernel
 *   layouts (either may be NULL); RS = R*S in {1, 9}; tiles_c = ceil(C / 32); tile0 = index of the parameter's first
 *   32 x 32 (k, c) tile in the launch; total_tiles = sum over parameters of ceil(K / 32) * tiles_c.
 * wm_wgrad_fold: ws = f32 weight-gradient slabs [nsplit][K][RS][C] (wm_conv2d_wgrad), nsplit slabs summed in order;
 *   grad = f32 OIHW gradient (grad += sum); nsplit <= 32: a block owns an 8 x 128 (k, c) tile with all its taps
 *   (tiles_c = ceil(C / 128), tiles = ceil(K / 8) * tiles_c); else a 4 x 64 tile of ONE tap (tiles_c = ceil(C / 64),
 *   tiles = ceil(K / 4) * tiles_c * RS) with the slab range cut in four, summed in order; C % 4 == 0.  Optional bias: w = f32 bias slabs [nsplit][K], krsc = (float*) bias gradient [K]
 *   (+=), both NULL when absent (crsk unused).
 * Replaces 24-100 wm_weights_prepare / 19 wm_wgrad_finalize launches per training step. */
typedef struct WmLayoutDesc {
  const float* w;
  uint16_t* krsc;
  uint16_t* crsk;
  float* ws;
  float* grad;
  int32_t K, C, RS, tiles_c, tile0, nsplit;
} WmLayoutDesc; /* 64 bytes */
int wm_layouts_refresh(const WmLayoutDesc* descs_dev, int n_desc, int total_tiles, void* stream);
int wm_wgrad_fold(const WmLayoutDesc* descs_dev, int n_desc, int total_tiles, void* stream);

/* weight-gradient slabs [nsplit][K][R][S][C] f32 -> OIHW gradient (= or += the sum of the slabs, in order). */
int wm_wgrad_finalize(const float* dw_slabs, int nsplit, int K, int C, int R, int S, float* grad_oihw,
                      int accumulate, void* stream);
/* The 7x7/2 pad-3 stem on 3 channels run as a 4x4/1 pad-2 convolution over the 2x2 space-to-depth
 * image [N][H/2][W/2][16] (channel (dh*2+dw)*3+c, 12 used): weights [K][3][7][7] f32 ->
 * [K][4][4][16] bf16, its gradient back, and the image transform (fmt WM_IMG_NCHW_F32 or
 * WM_IMG_NHWC_BF16 with 3 channels). */
int wm_stem_weights_prepare(const float* w_oihw, int K, void* w_s2d, void* stream);
int wm_stem_wgrad_finalize(const float* dw_s2d_slabs, int nsplit, int K, float* grad_oihw, int accumulate, void* stream);
int wm_image_to_s2d(const void* img, int fmt, int N, int H, int W, void* out, void* stream);
int wm_cast_f32_bf16(const float* x, long long n, void* y, void* stream);

/* BatchNorm over a [rows][C] bf16 tensor cut into G equal row groups with independent statistics
 * (G = 2 reproduces the reference's two separate forward(x0), forward(x1) calls on the
 * concatenated batch).  out = relu?(bn(y) (+ residual)).  running_* are updated once per group in
 * order (torch momentum convention, unbiased variance); save_mean/save_invstd are [G][C].
 * num_batches_tracked (torch's int64 buffer, may be NULL) is incremented by G inside the finalize kernel.
 * relu_mask (may be NULL): [rows][C / 8] bytes, bit e of byte (row, chunk) = output channel 8 chunk + e is > 0 -- the
 * ReLU's backward mask for wm_conv2d_dgrad_bnstat at 1/16 of the tensor's bytes.
 * C % 8 == 0, C <= 2048, rows % G == 0. */
size_t wm_bn_workspace_bytes(long long rows, int C, int G);
int wm_bn_train_fwd(const void* y, const void* residual, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, long long* num_batches_tracked, long long rows, int C, int G, float eps,
                    float momentum, int relu, float* save_mean, float* save_invstd, void* out, void* relu_mask,
                    void* workspace, size_t workspace_bytes, void* stream);
/* Training forward whose statistics were accumulated by wm_conv2d_fwd_stats. */
int wm_bn_train_fwd_from_stats(const void* y, const void* residual, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, long long* num_batches_tracked,
                               long long rows, int C, int G, float eps, float momentum, int relu, float* save_mean,
                               float* save_invstd,
                               void* out, void* relu_mask, const float* stat_part, int stat_tiles, void* workspace,
                               size_t workspace_bytes, void* stream);
/* Statistics only: mean / invstd / running stats and the [G][C] scale, shift of the normalisation, for a
 * consumer that applies it itself (the fused stem below).  stat_part NULL: computed from y here. */
int wm_bn_train_stats(const void* y, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, long long* num_batches_tracked, long long rows, int C, int G, float eps, float momentum,
                      float* save_mean, float* save_invstd, float* scale, float* shift, const float* stat_part,
                      int stat_tiles, void* workspace, size_t workspace_bytes, void* stream);
int wm_bn_eval_scale_shift(const float* gamma, const float* beta, const float* running_mean,
                           const float* running_var, int C, float eps, float* scale, float* shift, void* stream);
int wm_bn_eval_fwd(const void* y, const void* residual, const float* gamma, const float* beta,
                   const float* running_mean, const float* running_var, long long rows, int C, float eps,
                   int relu, void* out, void* workspace, size_t workspace_bytes, void* stream);
/* dz = dout * mask; dy = dBN(dz); dgamma/dbeta (= or +=); dz is also written when non-NULL (the
 * gradient of the residual branch).  mask: (out_relu > 0) when out_relu is given; else, when
 * relu_from_y != 0, recomputed from y as bf16(y*gamma*invstd + beta - mean*gamma*invstd) > 0 (valid
 * for a ReLU'd BN without residual; saves reading its output); else no ReLU. */
int wm_bn_train_bwd(const void* y, const void* dout, const void* out_relu, int relu_from_y,
                    const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                    long long rows, int C, int G, float* dgamma, float* dbeta, int accumulate, void* dy,
                    void* dz, void* workspace, size_t workspace_bytes, void* stream);
/* Backward whose per-tile sums were stored by wm_conv2d_dgrad_bnstat: g is the gradient ALREADY taken through the ReLU
 * (it is also the gradient of the shortcut branch, if any): add the slots in order, dgamma / dbeta (= or +=), then one
 * pass dy = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)).  workspace >= (7 + 256) * G * C floats. */
int wm_bn_train_bwd_from_stats(const void* y, const void* g, const float* gamma, const float* beta,
                               const float* save_mean, const float* save_invstd, long long rows, int C, int G,
                               float* dgamma, float* dbeta, int accumulate, void* dy, const float* stat_part,
                               int stat_tiles, void* workspace, size_t workspace_bytes, void* stream);
/* Synchronised BatchNorm (torch.nn.SyncBatchNorm; the reference's `sync_batchnorm` flag, scripts/WM811k_benchmark.py:62,
 * :1103): statistics over the batches of all ranks.  The library holds no communicator, so each pass is cut at the
 * point where the CALLER all-reduces (sum) a [G][2][C] f32 vector: forward (sum y, sum y^2), backward (sum g,
 * sum g * xhat).  group_count = rows per statistics group summed over the ranks (equal per-rank batches).  dgamma /
 * dbeta stay this rank's local sums, as torch's do (the gradient exchange averages them).  C <= 2048.
 * wm_bn_sync_fwd_sums: stat_part NULL -> computed from y (workspace: wm_bn_workspace_bytes), else the slots of
 * wm_conv2d_fwd_stats.  wm_bn_sync_bwd_sums / _apply: mask arguments as wm_bn_train_bwd. */
int wm_bn_sync_fwd_sums(const void* y, long long rows, int C, int G, const float* stat_part, int stat_tiles,
                        float* sums, void* workspace, size_t workspace_bytes, void* stream);
int wm_bn_sync_fwd_apply(const void* y, const void* residual, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, long long* num_batches_tracked, long long rows, int C,
                         int G, long long group_count, float eps, float momentum, int relu, float* save_mean,
                         float* save_invstd, void* out, const float* sums, void* workspace, size_t workspace_bytes,
                         void* stream);
int wm_bn_sync_bwd_sums(const void* y, const void* dout, const void* out_relu, int relu_from_y, const float* gamma,
                        const float* beta, const float* save_mean, const float* save_invstd, long long rows, int C,
                        int G, float* dgamma, float* dbeta, int accumulate, float* sums, void* workspace,
                        size_t workspace_bytes, void* stream);
int wm_bn_sync_bwd_apply(const void* y, const void* dout, const void* out_relu, int relu_from_y, const float* gamma,
                         const float* beta, const float* save_mean, const float* save_invstd, long long rows, int C,
                         int G, long long group_count, const float* sums, void* dy, void* dz, void* workspace,
                         size_t workspace_bytes, void* stream);
/* Backward of the fused stem tail max_pool3x3s2(relu(BN(y))): y [N][H][W][C]; the gradient entering
 * the BN is gathered from pooled_dy / pool_idx [N][P][Q][C] inside the apply pass; with ysel (the
 * inputs at the selected positions, from the forward; may be NULL) the per-channel sums run over the
 * pooled tensors alone. */
int wm_bn_relu_maxpool_bwd(const void* y, const void* ysel, const void* pooled_dy, const void* pool_idx, int N,
                           int H, int W, int C, const float* gamma, const float* beta, const float* save_mean,
                           const float* save_invstd, int G, float* dgamma, float* dbeta, int accumulate,
                           void* dy, void* workspace, size_t workspace_bytes, void* stream);
int wm_add_bf16(const void* a, const void* b, long long n, void* out, void* stream);

/* MaxPool2d(3, stride 2, padding 1) with recorded window positions (uint8, first maximum in scan
 * order as torch does), and global average pooling [N][HW][C] -> [N][C]. */
int wm_maxpool3x3s2_fwd(const void* x, int N, int H, int W, int C, void* y, void* idx, void* stream);
/* ResNet stem: maxpool(relu(x*scale + shift)) in one pass (scale/shift [G][C], G groups of N/G
 * images); the normalised 112x112 activation is never materialised.  Same idx semantics. */
int wm_bn_relu_maxpool3x3s2_fwd(const void* x, const float* scale, const float* shift, int N, int H, int W, int C,
                                int G, void* y, void* idx, void* xsel, void* stream);
int wm_maxpool3x3s2_bwd(const void* dy, const void* idx, int N, int H, int W, int C, void* dx, void* stream);
int wm_gap_fwd(const void* x, int N, int HW, int C, void* y, void* stream);
int wm_gap_bwd(const void* dy, int N, int HW, int C, void* dx, void* stream);

/* torch.optim.SGD step over flat f32 arenas: d = g*hyper[3] + hyper[2]*p; buf = hyper[1]*buf + d;
 * p -= hyper[0]*buf.  hyper is a 4-float DEVICE array {lr, momentum, weight_decay, grad_scale} so a
 * captured graph picks up a new learning rate without re-capture. */
int wm_sgd_step(float* params, const float* grads, float* momentum_buf, long long n, const float* hyper,
                void* stream);

/* torch.nn.CrossEntropyLoss(weight) (= nll_loss(log_softmax)) for the supervised baseline and the linear probe
 * (scripts/WM811k_benchmark.py:211-217; src/ssl_wafermap/models/evals.py:20): logits WM_F32 / WM_BF16 [B][C],
 * labels int64 [B] (outside [0, C): ignored), weight [C] or NULL.  acc2[0] += sum w_y nll, acc2[1] += sum w_y
 * (zero both first; loss = acc2[0] / acc2[1]); dlogits [B][C] f32 = w_y (softmax - onehot), to be divided by
 * acc2[1] by the caller. */
int wm_cross_entropy_fwd_bwd(const void* logits, int dtype, const long long* labels, const float* weight, int B, int C,
                             float* acc2, float* dlogits, void* stream);
/* torch.nn.BCEWithLogitsLoss(pos_weight) (multi-label linear probe, evals.py:93): target f32 [B][C], pos_weight [C]
 * or NULL; loss[0] += mean (zero it first); dlogits [B][C] f32 = d loss / d logits. */
int wm_bce_logits_fwd_bwd(const void* logits, int dtype, const float* target, const float* pos_weight, int B, int C,
                          float* loss, float* dlogits, void* stream);

/* lightly.loss.NegativeCosineSimilarity (BYOL, SimSiam: scripts/WM811k_benchmark.py:446,613):
 * loss[0] += -mean_i cos(x0_i, x1_i) (zero it first; norms clamped at eps as torch.cosine_similarity does);
 * dx0 / dx1 [B][D] float32 gradients, either may be NULL.  x0, x1: WM_F32 or WM_BF16 [B][D]. */
int wm_neg_cosine_fwd_bwd(const void* x0, const void* x1, int dtype, int B, int D, float eps, float* loss, float* dx0,
                          float* dx1, void* stream);

/* lightly.loss.DCLLoss / DCLWLoss (decoupled contrastive learning; the reference's DCLW model,
 * scripts/WM811k_benchmark.py:258-287): z0, z1 L2-normalised [B][D] float32.  Per direction (a, b) and row i:
 * -w_i <a_i,b_i>/T + lse_{k!=i} <a_i,a_k>/T + lse_{k!=i} <a_i,b_k>/T, averaged over rows and both directions;
 * w_i = 1, or 2 - B softmax_i(<z0_i,z1_i>/sigma) when weighted.  loss[0] += the loss (zero it first);
 * dz0, dz1 [B][D] gradients.  Workspace: wm_dcl_workspace_bytes(B). */
size_t wm_dcl_workspace_bytes(int B);
int wm_dcl_fwd_bwd(const float* z0, const float* z1, int B, int D, float temperature, float sigma, int weighted,
                   float* loss, float* dz0, float* dz1, void* workspace, size_t workspace_bytes, void* stream);

/* lightly.loss.BarlowTwinsLoss on the cross-correlation matrix (scripts/WM811k_benchmark.py:364-366):
 * raw_cc [D][D] f32 = sum over the batch of za_norm (x) zb_norm (wm_conv2d_wgrad on the two standardised
 * projections); c = scale * raw_cc; loss[0] += diag_weight sum_i (c_ii - 1)^2 + lambda sum_{i!=j} c_ij^2 (zero it
 * first); draw_cc = d loss / d raw_cc.  Barlow Twins: diag_weight 1; VICReg's covariance term: diag_weight 0,
 * lambda 1/D, scale 1/(N-1) on the covariance of the centred projection. */
int wm_barlow_twins_fwd_bwd(const float* raw_cc, int D, float scale, float lambda, float diag_weight, float* loss,
                            float* draw_cc, void* stream);
/* VICReg pieces (lightly.loss.VICRegLoss, scripts/WM811k_benchmark.py:401): out = z - mean (bf16 [rows][C]);
 * variance term loss[0] += mean_d relu(1 - sqrt(var_biased_d N/(N-1) + eps)) with coef[d] such that
 * d term / d centred[n][d] = coef[d] * centred[n][d]. */
int wm_center_columns(const void* z, const float* mean, long long rows, int C, void* out, void* stream);
int wm_vicreg_variance(const float* var_biased, int N, int D, float eps, float* loss, float* coef, void* stream);

/* lightly's SwaV Sinkhorn-Knopp (scripts/WM811k_benchmark.py:832-834 via lightly.loss.SwaVLoss): out WM_F32 / WM_BF16
 * [B][K] prototype scores; Q [B][K] f32 = the transport plan (rows sum to 1) after `iters` rounds at
 * temperature eps; workspace: B + K floats. */
int wm_sinkhorn(const void* out, int dtype, int B, int K, float eps, int iters, float* Q, float* workspace,
                void* stream);

/* NT-Xent against a memory bank (lightly NTXentLoss(memory_bank_size > 0), the reference's MoCo:
 * scripts/WM811k_benchmark.py:305-307).  q, kpos: L2-normalised [B][D] float32; bank [D][K] float32
 * (lightly's layout, one stored key per column).  logits_i = [<q_i,kpos_i>, <q_i,bank>] / T, label 0.
 * loss[0] += mean CE (zero it first); dq, dk [B][D] = d loss / d q, d kpos.  K + 2 D <= 16384. */
int wm_ntxent_bank_fwd_bwd(const float* q, const float* kpos, const float* bank, int B, int D, int K,
                           float temperature, float* loss, float* dq, float* dk, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Vision-transformer path (SURVEY §8 a13 DINOViT, a14 MAE).
 * Replaces, in the reference: the facebookresearch/dino `dino_vits16` backbone called at
 * scripts/WM811k_benchmark.py:548-550,566-576 (MixedWM38_pretrain.py:141-143), torchvision's
 * vit_b_32 inside lightly's MAEBackbone/MAEDecoder (:881-899, :903-947), lightly.loss.DINOLoss
 * (:564,586), lightly.models.utils.update_momentum (:579-581), torch.optim.AdamW (:591-598) and
 * torch.nn.MSELoss on the masked patches (:900,949-954).
 * Activations are bf16 [rows][C] (rows = tokens, token-major); parameters float32.
 * ------------------------------------------------------------------------------------------- */

/* nn.LayerNorm over the last dim (biased variance, two-pass in registers).  C % 8 == 0, C <= 2048.
 * mean / rstd [rows] are saved for the backward.  dgamma / dbeta are ACCUMULATED (f32 atomics). */
int wm_layernorm_fwd(const void* x, const float* gamma, const float* beta, float eps, long long rows, int C,
                     void* y, float* mean, float* rstd, void* stream);
int wm_layernorm_bwd(const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd,
                     long long rows, int C, void* dx, float* dgamma, float* dbeta, void* stream);
/* Same with the gradient of the residual connection around the normalised branch (x -> LN -> f -> + x; timm / dino
 * Block.forward, the reference's DINOViT backbone, scripts/WM811k_benchmark.py:548-550) added in the same pass:
 * dx = bf16(LN'(dy)) + dres, dres bf16 [rows][C].  Replaces the accumulation pass autograd runs for the two uses of x. */
int wm_layernorm_bwd_add(const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd,
                         long long rows, int C, const void* dres, void* dx, float* dgamma, float* dbeta, void* stream);
/* LayerNorm backward with the parameter-gradient sums as per-block SLOTS instead of f32 atomics (bit-reproducible):
 * part [2][wm_layernorm_bwd_blocks(rows, C)][C] f32 -- the dgamma slots, then the dbeta slots, every slot overwritten; the
 * caller adds them in slot order (wm_wgrad_fold / wm_wgrad_finalize with K = 1, RS = 1).  dres as in
 * wm_layernorm_bwd_add, may be NULL. */
/* Column sums likewise: wm_bias_act_bwd_parts = wm_bias_act_bwd with the bias gradient as per-block slots
 * part [wm_colsum_blocks(rows, C)][C] f32 (act WM_ACT_NONE: the plain column sum of dy; x, bias, dx unused). */
int wm_colsum_blocks(long long rows, int C);
int wm_bias_act_bwd_parts(const void* x, const float* bias, const void* dy, int act, long long rows, int C, void* dx,
                          float* part, void* stream);
int wm_layernorm_bwd_blocks(long long rows, int C);
int wm_layernorm_bwd_parts(const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd,
                           long long rows, int C, const void* dres, void* dx, float* part, void* stream);

/* y = act(x + bias) (+ residual).  act: 0 identity, 1 exact GELU (erf), 2 ReLU (the bias-carrying heads:
 * lightly MoCoProjectionHead).  bias / residual may be NULL. */
#define WM_ACT_NONE 0
#define WM_ACT_GELU 1
#define WM_ACT_RELU 2
int wm_bias_act_fwd(const void* x, const float* bias, const void* residual, int act, long long rows, int C,
                    void* y, void* stream);
/* dx = dy * act'(x + bias) (dx may be NULL for act 0: only the bias gradient is wanted);
 * dbias[C] += column sums of dx (NULL: skipped). */
int wm_bias_act_bwd(const void* x, const float* bias, const void* dy, int act, long long rows, int C, void* dx,
                    float* dbias, void* stream);
/* out[C] (+)= sum over rows of x[rows][C] (bf16 in, f32 out). */
int wm_colsum_bf16(const void* x, long long rows, int C, float* out, int accumulate, void* stream);

/* tokens[n][0] = cls + pos[0]; tokens[n][1+i] = patches[n][i] + pos[1+i]  (cls [D], pos [1+np][D] f32). */
int wm_tokens_assemble(const void* patches, const float* cls, const float* pos, int N, int np, int D, void* tokens,
                       void* stream);
/* images [N][S][S][3] bf16 -> rows [N*(S/p)^2][p*p*3] in (ph, pw, c) order: the patch-embedding
 * convolution (kernel = stride = p) becomes a GEMM against the [D][p][p][3] weight layout. */
int wm_patchify(const void* images, int N, int S, int p, void* rows, void* stream);

/* Multi-head self-attention, head_dim 64 or 32, S <= 256, softmax(scale * q k^T) v.
 * qkv [B][S][3][H][head_dim] (the qkv / in_proj Linear's output as is), out / dout [B][S][H][head_dim],
 * lse [B][H][S] f32.  bwd writes dqkv in the layout of qkv. */
int wm_attention_fwd(const void* qkv, int B, int S, int H, int head_dim, float scale, void* out, float* lse,
                     void* stream);
int wm_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, int B, int S, int H,
                     int head_dim, float scale, void* dqkv, void* stream);

/* Row gather / scatter on [B][S][C] bf16 by per-batch token indices idx [B][K] (int64, as
 * torch.argsort returns): lightly's get_at_index / set_at_index.  scatter writes only the indexed
 * rows (caller pre-fills the rest). */
int wm_gather_rows(const void* x, const long long* idx, int B, int S, int K, int C, void* out, void* stream);
int wm_scatter_rows(const void* src, const long long* idx, int B, int S, int K, int C, void* dst, void* stream);

/* MSE over n elements: loss[0] = mean((pred - target)^2); dpred = 2 (pred - target) / n (bf16).
 * loss must be zeroed by the caller. */
int wm_mse_fwd_bwd(const void* pred, const void* target, long long n, float* loss, void* dpred, void* stream);
/* torch.nn.L1Loss() (SimMIM, scripts/WM811k_benchmark.py:976): loss[0] += mean |pred - target|; dpred = sign / n. */
int wm_l1_fwd_bwd(const void* pred, const void* target, long long n, float* loss, void* dpred, void* stream);

/* DINO loss (lightly.loss.DINOLoss).  teacher [Vt*B][D] bf16 -> probs f32 = softmax((t - center)/temp_t). */
int wm_dino_teacher_probs(const void* teacher, const float* center, float temp_t, long long rows, int D,
                          float* probs, void* stream);
/* student [Vs][B][D] bf16, probs [Vt][B][D]: loss[0] += sum_{t != s} -<probs_t, log_softmax(student_s/temp_s)>
 * / (n_terms * B); dstudent = d loss / d student (bf16).  Views with equal index are the same crop. */
int wm_dino_loss_fwd_bwd(const void* student, const float* probs, int Vs, int Vt, int B, int D, float temp_s,
                         float* loss, void* dstudent, void* stream);
/* Cross-entropy against soft targets (MSN / PMSN, scripts/WM811k_benchmark.py:684,705): student [Vs][B][D] bf16 logits,
 * probs [B][D] f32 targets shared by the Vs views; loss[0] += mean over (view, sample) of
 * -<probs_b, log_softmax(student_vb / temp_s)>; dstudent bf16. */
int wm_soft_cross_entropy_fwd_bwd(const void* student, const float* probs, int Vs, int B, int D, float temp_s,
                                  float* loss, void* dstudent, void* stream);
/* Mean-entropy regulariser of lightly's MSNLoss / PMSNLoss on logits [N][K] bf16: p = softmax(logits / T),
 * m = mean over rows; loss[0] += sum_k m_k (log m_k - log_prior_k) (log_prior NULL: MSN's me-max term
 * sum m log m); dlogits [N][K] f32; mean_ws: K floats of scratch. */
int wm_mean_entropy_reg_fwd_bwd(const void* logits, const float* log_prior, int N, int K, float temperature,
                                float* loss, float* dlogits, float* mean_ws, void* stream);
/* center = momentum * center + (1 - momentum) * mean over rows of teacher[rows][D]. */
int wm_dino_center_update(const void* teacher, long long rows, int D, float momentum, float* center, void* stream);

/* Column statistics and standardisation of a feature matrix x [rows][C] (float32 or bf16: WM_F32 /
 * WM_BF16), the sklearn StandardScaler the reference applies to dumped embeddings
 * (notebooks/3.0-Embeddings-inference.ipynb cell 7): mean[c], var[c] = biased variance (two passes:
 * sums, then squared deviations from the mean).  mean and var must be zeroed by the caller.
 * wm_standardize: out[r][c] = (x[r][c] - mean[c]) * inv_scale[c], float32. */
int wm_colstats(const void* x, int dtype, long long rows, int C, float* mean, float* var, void* stream);
int wm_standardize(const void* x, int dtype, long long rows, int C, const float* mean, const float* inv_scale,
                   float* out, void* stream);

/* torch.optim.AdamW / Adam step over flat f32 arenas; hyper = DEVICE {lr, beta1, beta2, eps, weight_decay,
 * bias_correction1, bias_correction2, grad_scale, l2_mode}: l2_mode 0 = decoupled decay (AdamW), 1 = the
 * decay added to the gradient (torch.optim.Adam). */
int wm_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                  const float* hyper, void* stream);
/* lightly update_momentum: ema = ema * m + p * (1 - m). */
int wm_ema_update(float* ema, const float* params, long long n, float m, void* stream);
/* out[r][:] = ascending argsort of keys[r][0..S) (S <= 256; ties to the lower index): the token permutation of
 * lightly.models.utils.random_token_mask (reference scripts/WM811k_benchmark.py:930, MixedWM38_pretrain.py:208). */
int wm_argsort_rows(const float* keys, int rows, int S, long long* out, void* stream);
/* y = x * g[column] on bf16 [rows][C] with f32 g[C], and its backward (dx = dy * g, dg = / += column sums of dy * x):
 * the trainable gain of the weight-normalised last layer of lightly's DINOProjectionHead(norm_last_layer=False)
 * (reference scripts/WM811k_benchmark.py:553-559). */
int wm_colscale_fwd(const void* x, const float* g, long long rows, int C, void* y, void* stream);
int wm_colscale_bwd(const void* x, const float* g, const void* dy, long long rows, int C, void* dx, float* dg,
                    int accumulate, void* stream);

/* timm.optim.lars.Lars step (BarlowTwins / VICReg in the reference, scripts/WM811k_benchmark.py:383-392,
 * 418-427) over a flat arena: seg_offsets [n_seg + 1] int64 = element offsets of the parameters; per parameter
 * r = trust_coeff |w| / (|g| + wd |w| + eps) (1 if either norm is 0; no adaptation when wd == 0),
 * g <- (g + wd w) r, buf = momentum buf + g, p -= lr buf.  hyper (device) = {lr, momentum, weight_decay,
 * trust_coeff, eps, grad_scale}; norms_ws: 2 * n_seg floats of scratch. */
int wm_lars_step(float* params, const float* grads, float* momentum_buf, const long long* seg_offsets, int n_seg,
                 const float* hyper, float* norms_ws, void* stream);

/* Small float32 matmul c [M][N] = op(a) b, op(a) = a [M][K] (trans_a = 0) or a^T with a stored [K][M] (trans_a = 1);
 * b [K][N]; row-major.  The positional-embedding resize of dino / lightly (interpolate_pos_encoding, reference call
 * sites scripts/WM811k_benchmark.py:548-550, utilities via lightly MAEBackbone) as a fixed matrix product, and its
 * gradient. */
int wm_matmul_f32(const float* a, const float* b, float* c, int M, int N, int K, int trans_a, void* stream);

/* ---- Float32 "parity" preset (csrc/f32path.hip): the FORWARD pass of the SimCLR / DINO / MAE steps with every activation,
 * weight and accumulator in float32 (reference call sites scripts/WM811k_benchmark.py:236-248, :578-588, :902-947).  The
 * production kernels keep activations in bf16; profiles/r04_error_budget_bf16.md shows that storage puts the step losses
 * 0.4e-4 .. 3.4e-4 from the float32 reference (north_star: 1e-4).  These entry points keep them in float32: a validation
 * preset (forward for all three models, backward for the ResNet-18 / head ops), selected by ssl_wafermap_amd.precision("float32").  Activations NHWC float32 / [rows][C]. */
size_t wm_f32_conv2d_workspace_bytes(int C, int K, int R, int S);
/* y = act(conv(x, w) + bias) + residual: x [N][H][W][C], w_oihw [K][C][R][S] (the float32 master layout), bias [K] or NULL,
 * residual [N][P][Q][K] or NULL, act 0 none / 1 GELU (erf) / 2 ReLU.  A Linear layer is the 1x1 case on a 1x1 image. */
int wm_f32_conv2d_fwd(const float* x, const float* w_oihw, const float* bias, const float* residual, float* y, int N, int H,
                      int W, int C, int K, int R, int S, int P, int Q, int stride, int pad, int act, void* workspace,
                      size_t workspace_bytes, void* stream);
size_t wm_f32_bn_workspace_bytes(long long rows, int C, int G);
/* out = relu?(BN(y) (+ residual)) over G equal row groups with their own batch statistics (training != 0: running
 * statistics and the batch counter are updated as by G consecutive nn.BatchNorm calls) or with the running statistics. */
int wm_f32_bn_fwd(const float* y, const float* residual, const float* gamma, const float* beta, float* running_mean,
                  float* running_var, long long* num_batches_tracked, long long rows, int C, int G, int training, float eps,
                  float momentum, int relu, float* save_mean, float* save_invstd, float* out, void* workspace,
                  size_t workspace_bytes, void* stream);
int wm_f32_maxpool3x3s2(const float* x, int N, int H, int W, int C, float* y, void* stream);
int wm_f32_gap(const float* x, int N, int HW, int C, float* y, void* stream);
int wm_f32_layernorm(const float* x, const float* gamma, const float* beta, float eps, long long rows, int C, float* y,
                     void* stream);
/* y = act(x + bias) + residual, element-wise on [rows][C] (bias, residual optional). */
int wm_f32_bias_act(const float* x, const float* bias, const float* residual, int act, long long rows, int C, float* y,
                    void* stream);
/* softmax(q k^T * scale) v per (image, head): qkv [B*S][3][H][HD] -> out [B*S][H*HD]; HD 64 or 32. */
int wm_f32_attention(const float* qkv, int B, int S, int H, int HD, float scale, float* out, void* stream);
/* rows of softmax((x - subtract) * inv_temp) (log_softmax != 0: its logarithm); subtract [D] or NULL. */
int wm_f32_softmax_rows(const float* x, const float* subtract, float inv_temp, int log_softmax, long long rows, int D,
                        float* y, void* stream);
/* pair_loss[(t * SV + s) * B + b] = -sum_d probs[t][b][d] * logq[s][b][d], 0 where t == s (lightly DINOLoss's zeroed diagonal). */
int wm_f32_pair_ce(const float* probs, const float* logq, int T, int SV, int B, int D, float* pair_loss, void* stream);
/* *out = scale * sum_i f(a_i, b_i): mode 0 a_i, 1 (a_i - b_i)^2, 2 |a_i - b_i|; one ordered double-precision sum. */
int wm_f32_reduce(const float* a, const float* b, long long n, int mode, double scale, float* out, void* stream);
/* Backward pieces of the convolution / BatchNorm / pooling / Linear path (the float32 preset can take a whole SimCLR optimiser
 * step: scripts/WM811k_benchmark.py:242-255).  dx [N][H][W][C] = input gradient of wm_f32_conv2d_fwd for dy [N][P][Q][K];
 * dw_oihw [K][C][R][S] = its weight gradient (pixel ranges summed in a fixed order; workspace from
 * wm_f32_conv2d_wgrad_workspace_bytes); wm_f32_colsum: bias gradients. */
int wm_f32_conv2d_dgrad(const float* dy, const float* w_oihw, float* dx, int N, int H, int W, int C, int K, int R, int S, int P,
                        int Q, int stride, int pad, void* workspace, size_t workspace_bytes, void* stream);
size_t wm_f32_conv2d_wgrad_workspace_bytes(int N, int P, int Q, int C, int K, int R, int S);
int wm_f32_conv2d_wgrad(const float* dy, const float* x, float* dw_oihw, int N, int H, int W, int C, int K, int R, int S, int P,
                        int Q, int stride, int pad, void* workspace, size_t workspace_bytes, void* stream);
int wm_f32_colsum(const float* x, long long rows, int C, float* out, void* stream);
/* BatchNorm (training statistics, G groups) backward: gm = dout taken through the ReLU where out_relu (the forward's output)
 * is given; dgamma / dbeta [C] (sums over all groups), dy = gamma * invstd * (gm - mean(gm) - xhat * mean(gm * xhat)), dz (optional)
 * = gm, the gradient of the residual branch.  Workspace: wm_f32_bn_workspace_bytes. */
int wm_f32_bn_bwd(const float* y, const float* dout, const float* out_relu, const float* gamma, const float* save_mean,
                  const float* save_invstd, long long rows, int C, int G, float* dgamma, float* dbeta, float* dy, float* dz,
                  void* workspace, size_t workspace_bytes, void* stream);
int wm_f32_maxpool3x3s2_bwd(const float* x, const float* dy, int N, int H, int W, int C, float* dx, void* stream);
int wm_f32_gap_bwd(const float* dy, int N, int HW, int C, float* dx, void* stream);
/* center = center * momentum + (1 - momentum) * column mean of teacher [rows][D]. */
int wm_f32_center_update(float* center, const float* teacher, int rows, int D, float momentum, void* stream);

/* Debugging probe (no reference counterpart): *slot = max(*slot, max_i |x[i]|), NaN if any x[i] is NaN
 * (+inf stays +inf).  x: n elements of WM_F32 / WM_BF16; *slot must hold a non-negative float (zero it
 * first).  Allocates nothing, so it can sit between the launches of a captured hipGraph
 * (tools/nan_hunt.py; DESIGN.md "NaN at cfg 2"). */
int wm_debug_absmax(const void* x, int dtype, long long n, float* slot, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WAFER_HIP_H */
