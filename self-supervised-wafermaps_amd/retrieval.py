"""Embedding dump and nearest-neighbour retrieval.

The reference's flow (notebooks/3.0-Embeddings-inference.ipynb cell 7; 2.0-Figures-nearest-neighbors
cell 2): `Trainer(inference_mode=True, precision="16-mixed").predict(model, loader(bs=1024,
get_inference_transforms()))` -> `predict_step` returns `backbone(images)`
(src/ssl_wafermap/models/knn.py:135-137) -> torch.cat -> sklearn StandardScaler -> DataFrame; then
nearest neighbours of query embeddings by L2 and by cosine distance.

Here: `embed_dataset` runs the eval-mode backbone over a WaferLoader and keeps the features on the
GPU; `StandardScaler` is two kernels (column statistics, standardise); `nearest_neighbors` is the
streaming top-k kernel.  L2 ranking rides on the same inner-product kernel: argmin ||q - x||^2 =
argmax (q.x - ||x||^2 / 2), so the bank gets one extra column -||x||^2/2 and the query a 1 (padded to
the kernel's 64-float granularity), in float32 so the ranking is exact.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from . import functional as F_hip
from ._lib import check, dtype_code, ptr, require_gpu, stream_ptr


@torch.no_grad()
def embed_dataset(model, loader, out_dtype: Optional[torch.dtype] = torch.float16) -> torch.Tensor:
    """Features [N, D] of every sample of `loader` (a WaferLoader with an inference transform),
    eval-mode `model.predict_step`, concatenated on the device; cast to `out_dtype` (the reference
    stores float16) unless None."""
    was_training = model.training
    model.eval()
    outs = []
    for i, batch in enumerate(loader):
        f = model.predict_step(batch, i)
        outs.append(f.flatten(start_dim=1))
    if was_training:
        model.train()
    feats = torch.cat(outs, dim=0)
    return feats if out_dtype is None else feats.to(out_dtype)


class StandardScaler:
    """sklearn.preprocessing.StandardScaler (with_mean, with_std, biased variance, zero-variance
    columns left unscaled) on device tensors."""

    def __init__(self):
        self.mean_ = self.var_ = self.scale_ = None

    @staticmethod
    def _prep(x: torch.Tensor) -> torch.Tensor:
        require_gpu(x)
        if x.dim() != 2:
            raise ValueError("StandardScaler expects [n_samples, n_features]")
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        return x.contiguous()

    def fit(self, x: torch.Tensor) -> "StandardScaler":
        x = self._prep(x)
        rows, c = x.shape
        mean = torch.zeros(c, dtype=torch.float32, device=x.device)
        var = torch.zeros(c, dtype=torch.float32, device=x.device)
        check(_lib.load().wm_colstats(ptr(x), dtype_code(x), rows, c, ptr(mean), ptr(var), stream_ptr()), "wm_colstats")
        self.mean_, self.var_ = mean, var
        # sklearn's constant-feature rule (_is_constant_feature): variance within the rounding error of
        # the sums -> leave the column unscaled
        eps = torch.finfo(torch.float32).eps
        bound = rows * eps * var + (rows * mean * eps) ** 2
        self.scale_ = torch.where(var <= bound, torch.ones_like(var), var.sqrt())
        return self

    def transform(self, x: torch.Tensor) -> torch.Tensor:
        if self.mean_ is None:
            raise RuntimeError("StandardScaler.transform before fit")
        x = self._prep(x)
        rows, c = x.shape
        if c != self.mean_.numel():
            raise ValueError(f"StandardScaler fitted on {self.mean_.numel()} features, got {c}")
        out = torch.empty((rows, c), dtype=torch.float32, device=x.device)
        inv = (1.0 / self.scale_).contiguous()
        check(_lib.load().wm_standardize(ptr(x), dtype_code(x), rows, c, ptr(self.mean_), ptr(inv), ptr(out),
                                         stream_ptr()), "wm_standardize")
        return out

    def fit_transform(self, x: torch.Tensor) -> torch.Tensor:
        return self.fit(x).transform(x)


def _topk_general(q: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor], k: int):
    lib = _lib.load()
    nq, d = q.shape
    n = b.shape[0]
    need = lib.wm_knn_topk_general_workspace_bytes(nq, n, d, k)
    if need == 0:
        raise ValueError(f"nearest_neighbors: unsupported sizes nq={nq} n={n} d={d} k={k} (k <= 16)")
    ws = torch.empty(need, dtype=torch.uint8, device=q.device)
    sim = torch.empty((nq, k), dtype=torch.float32, device=q.device)
    idx = torch.empty((nq, k), dtype=torch.int32, device=q.device)
    check(lib.wm_knn_topk_general(ptr(q), ptr(b), ptr(bias), nq, n, d, k, 0, ptr(sim), ptr(idx), ptr(ws), need,
                                  stream_ptr()), "wm_knn_topk_general")
    return sim, idx


def nearest_neighbors(query: torch.Tensor, bank: torch.Tensor, k: int, metric: str = "cosine",
                      query_block: int = 1024) -> Tuple[torch.Tensor, torch.Tensor]:
    """k nearest bank rows of every query row (k <= 16).  Returns (distance [nq, k] float32 ascending,
    index [nq, k] int64).  metric "cosine": 1 - cos; "l2": Euclidean distance.

    Cosine with rows the streaming kernel takes (d % 64 == 0, d <= 512, k <= 8 above 256 features)
    runs on wm_knn_topk; everything else, and every L2 query, on the general float32 kernel, which
    scores q.x - ||x||^2/2 for the Euclidean ranking."""
    require_gpu(query, bank)
    if metric not in ("cosine", "l2"):
        raise ValueError("metric must be 'cosine' or 'l2'")
    q, b = query.float().contiguous(), bank.float().contiguous()
    d = q.shape[1]
    if d % 4:
        pad = 4 - d % 4
        q = torch.nn.functional.pad(q, (0, pad)).contiguous()
        b = torch.nn.functional.pad(b, (0, pad)).contiguous()
    bias = None
    if metric == "cosine":
        q, b = F_hip.l2_normalize(q), F_hip.l2_normalize(b)
        fast = q.shape[1] % 64 == 0 and q.shape[1] <= 512 and (k <= 8 or q.shape[1] <= 256)
    else:
        bias = -0.5 * (b * b).sum(dim=1)
        fast = False
    dist, idx = [], []
    for s in range(0, q.shape[0], query_block):
        qs = q[s:s + query_block].contiguous()
        # (the fast path pipelines its query batches over several HIP streams: functional.knn_topk_batched)
        sim, ind = F_hip.knn_topk_batched(qs, b, k, batch=256) if fast else _topk_general(qs, b, bias, k)
        if metric == "cosine":
            dist.append((1.0 - sim).clamp_min_(0.0))
        else:
            qq = (qs * qs).sum(dim=1, keepdim=True)
            dist.append((qq - 2.0 * sim).clamp_min_(0.0).sqrt_())
        idx.append(ind.long())
    return torch.cat(dist), torch.cat(idx)
