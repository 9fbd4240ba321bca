"""Tensor-level entry points over the wafer_hip C ABI (include/wafer_hip.h).

Each function checks shapes on the host (a faulting kernel can take the whole node down), passes
raw device pointers + the current HIP stream to the library and raises on any non-zero return.
There is no CPU fallback.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check, dtype_code, ptr, require_gpu, stream_ptr

# --------------------------------------------------------------------------------------- kNN


def knn_topk(query: torch.Tensor, bank: torch.Tensor, k: int, index_base: int = 0,
             workspace: Optional[torch.Tensor] = None, out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
             ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Top-k inner products of every query row against every bank row.

    query [nq, d], bank [n, d] (both float32 or both bfloat16, row-major, d*itemsize % 256 == 0).
    Returns (sim [nq, k] float32 descending, idx [nq, k] int32).  Replaces the
    `torch.mm(feature, feature_bank)` + `.topk(k)` pair inside lightly's knn_predict
    (reference call site src/ssl_wafermap/models/knn.py:91-98)."""
    require_gpu(query, bank)
    if query.dim() != 2 or bank.dim() != 2 or query.shape[1] != bank.shape[1]:
        raise ValueError(f"knn_topk: query {tuple(query.shape)} vs bank {tuple(bank.shape)}")
    if query.dtype != bank.dtype:
        raise ValueError("knn_topk: query and bank must share a dtype")
    nq, d = query.shape
    n = bank.shape[0]
    lib = _lib.load()
    if k > 16:
        return _knn_topk_paged(query, bank, k, index_base)
    rowbytes = d * query.element_size()
    streaming_ok = rowbytes % 256 == 0 and rowbytes <= (2048 if query.dtype == torch.float32 else 1024) and \
        (k <= 8 or rowbytes <= 1024)
    if not streaming_ok:
        # shapes the streaming kernel does not take (e.g. 768 ViT-B features): the general float32 kernel
        q32, b32 = query.float().contiguous(), bank.float().contiguous()
        if d % 4:
            q32 = torch.nn.functional.pad(q32, (0, 4 - d % 4)).contiguous()
            b32 = torch.nn.functional.pad(b32, (0, 4 - d % 4)).contiguous()
        need = lib.wm_knn_topk_general_workspace_bytes(nq, n, q32.shape[1], k)
        if need == 0:
            raise ValueError(f"knn_topk: unsupported sizes nq={nq} n={n} d={d} k={k} (k <= 16)")
        ws = torch.empty(need, dtype=torch.uint8, device=query.device)
        sim = torch.empty((nq, k), dtype=torch.float32, device=query.device)
        idx = torch.empty((nq, k), dtype=torch.int32, device=query.device)
        check(lib.wm_knn_topk_general(ptr(q32), ptr(b32), 0, nq, n, q32.shape[1], k, index_base, ptr(sim), ptr(idx),
                                      ptr(ws), need, stream_ptr()), "wm_knn_topk_general")
        return sim, idx
    need = lib.wm_knn_topk_workspace_bytes(nq, n, d, k)
    if need == 0:
        raise ValueError(f"knn_topk: unsupported sizes nq={nq} n={n} d={d} k={k}")
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=query.device)
    if out is not None:  # caller-provided [nq, k] float32 / int32 row-major destinations (slices of a larger result)
        sim, idx = out
        if sim.shape != (nq, k) or idx.shape != (nq, k) or sim.dtype != torch.float32 or idx.dtype != torch.int32 \
                or not sim.is_contiguous() or not idx.is_contiguous():
            raise ValueError("knn_topk: out must be contiguous (float32 [nq, k], int32 [nq, k])")
    else:
        sim = torch.empty((nq, k), dtype=torch.float32, device=query.device)
        idx = torch.empty((nq, k), dtype=torch.int32, device=query.device)
    check(lib.wm_knn_topk(ptr(query), ptr(bank), nq, n, d, dtype_code(query), k, index_base, ptr(sim),
                          ptr(idx), ptr(workspace), workspace.numel() * workspace.element_size(),
                          stream_ptr()), "wm_knn_topk")
    return sim, idx


_KNN_LANES = {}


def knn_topk_batched(queries: torch.Tensor, bank: torch.Tensor, k: int, batch: int = 64, lanes: Optional[int] = None,
                     index_base: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """knn_topk for many queries (embedding retrieval, all-pairs): query batches of `batch` rows are issued
    round-robin on `lanes` HIP streams, so the latency-bound selection kernel of one batch runs under the streaming
    kernel of the next (measured on 811 457 x 128 bf16, 64 queries per batch: 51 us per batch on one stream, 39 us
    on three = 0.66 of the HBM roofline end to end; 128 per batch: 72 -> 48 us on two; default lanes: 3 up to 64
    queries per batch, else 2; tools/bench_knn_pipeline.py).
    Returns (sim [nq, k], idx [nq, k]) as knn_topk; the current stream waits for all lanes before returning."""
    require_gpu(queries, bank)
    nq = queries.shape[0]
    if lanes is None:
        lanes = 3 if batch <= 64 else 2
    if nq <= batch or lanes <= 1:
        return knn_topk(queries, bank, k, index_base)
    dev = queries.device
    pool = _KNN_LANES.setdefault(dev.index, [])
    while len(pool) < lanes:
        pool.append(torch.cuda.Stream(device=dev))
    cur = torch.cuda.current_stream(dev)
    sim = torch.empty((nq, k), dtype=torch.float32, device=dev)
    idx = torch.empty((nq, k), dtype=torch.int32, device=dev)
    for s in pool[:lanes]:
        s.wait_stream(cur)
    lib = _lib.load()
    d = queries.shape[1]
    rowbytes = d * queries.element_size()
    direct = (k <= 16 and queries.dtype == bank.dtype and queries.is_contiguous() and rowbytes % 256 == 0
              and rowbytes <= (2048 if queries.dtype == torch.float32 else 1024) and (k <= 8 or rowbytes <= 1024))
    if direct:
        # every launch of every batch from ONE C call (no per-batch host work): wm_knn_topk_many
        need = (lib.wm_knn_topk_workspace_bytes(min(batch, nq), bank.shape[0], d, k) + 255) // 256 * 256
        spaces = [torch.empty(need * lanes, dtype=torch.uint8, device=dev)]
        import ctypes

        arr = (ctypes.c_void_p * lanes)(*[int(s.cuda_stream) for s in pool[:lanes]])
        check(lib.wm_knn_topk_many(ptr(queries), ptr(bank), nq, bank.shape[0], d, dtype_code(queries), k, index_base,
                                   ptr(sim), ptr(idx), batch, ptr(spaces[0]), need, arr, lanes), "wm_knn_topk_many")
    else:
        spaces = []
        for bi, o in enumerate(range(0, nq, batch)):
            with torch.cuda.stream(pool[bi % lanes]):
                ps, pi = knn_topk(queries[o:o + batch], bank, k, index_base)
                sim[o:o + batch].copy_(ps)
                idx[o:o + batch].copy_(pi)
    for s in pool[:lanes]:
        cur.wait_stream(s)
    for w in spaces:
        w.record_stream(cur)  # allocated on `cur`, used on the lanes: not to be recycled before the join above
    return sim, idx


def _knn_topk_paged(query: torch.Tensor, bank: torch.Tensor, k: int, index_base: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """k > 16 (lightly's knn_predict default is 200; the reference uses 5): pages of 16 from the general float32
    kernel, each page starting strictly after the last entry of the previous one in the list order (score
    descending, index ascending) -- exact, ceil(k / 16) passes over the bank."""
    nq, d = query.shape
    n = bank.shape[0]
    if k > n:
        raise ValueError(f"knn_topk: k={k} exceeds the bank size {n}")
    if d > 1024:
        raise ValueError(f"knn_topk: k > 16 needs d <= 1024 (got {d})")
    lib = _lib.load()
    q32, b32 = query.float().contiguous(), bank.float().contiguous()
    if d % 4:
        q32 = torch.nn.functional.pad(q32, (0, 4 - d % 4)).contiguous()
        b32 = torch.nn.functional.pad(b32, (0, 4 - d % 4)).contiguous()
    need = lib.wm_knn_topk_general_workspace_bytes(nq, n, q32.shape[1], 16)
    ws = torch.empty(need, dtype=torch.uint8, device=query.device)
    sim = torch.empty((nq, k), dtype=torch.float32, device=query.device)
    idx = torch.empty((nq, k), dtype=torch.int32, device=query.device)
    cur_s = cur_i = None
    done = 0
    while done < k:
        kk = min(16, k - done)
        ps = torch.empty((nq, kk), dtype=torch.float32, device=query.device)
        pi = torch.empty((nq, kk), dtype=torch.int32, device=query.device)
        check(lib.wm_knn_topk_general_after(ptr(q32), ptr(b32), 0, nq, n, q32.shape[1], kk, index_base, ptr(cur_s),
                                            ptr(cur_i), ptr(ps), ptr(pi), ptr(ws), need, stream_ptr()),
              "wm_knn_topk_general_after")
        sim[:, done:done + kk] = ps
        idx[:, done:done + kk] = pi
        cur_s, cur_i = ps[:, -1].contiguous(), pi[:, -1].contiguous()
        done += kk
    return sim, idx


def knn_merge(sim_parts: torch.Tensor, idx_parts: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge [parts, nq, k] candidate lists (e.g. all-gathered shard results) into [nq, k]."""
    require_gpu(sim_parts, idx_parts)
    parts, nq, k = sim_parts.shape
    if idx_parts.shape != sim_parts.shape or sim_parts.dtype != torch.float32 or idx_parts.dtype != torch.int32:
        raise ValueError("knn_merge: need float32 sims and int32 indices of equal shape")
    sim = torch.empty((nq, k), dtype=torch.float32, device=sim_parts.device)
    idx = torch.empty((nq, k), dtype=torch.int32, device=sim_parts.device)
    check(_lib.load().wm_knn_merge(ptr(sim_parts), ptr(idx_parts), parts, nq, k, ptr(sim), ptr(idx),
                                   stream_ptr()), "wm_knn_merge")
    return sim, idx


def knn_vote(sim: torch.Tensor, idx: torch.Tensor, bank_labels: torch.Tensor, num_classes: int,
             knn_t: float, return_scores: bool = False):
    """exp(sim/t)-weighted class vote; returns pred_labels [nq, num_classes] int64 (classes by
    descending score), as lightly's knn_predict does."""
    require_gpu(sim, idx, bank_labels)
    if bank_labels.dtype != torch.int64:
        raise ValueError("knn_vote: bank_labels must be int64")
    nq, k = sim.shape
    pred = torch.empty((nq, num_classes), dtype=torch.int64, device=sim.device)
    scores = torch.empty((nq, num_classes), dtype=torch.float32, device=sim.device) if return_scores else None
    check(_lib.load().wm_knn_vote(ptr(sim), ptr(idx), ptr(bank_labels), bank_labels.numel(), nq, k, num_classes, float(knn_t),
                                  ptr(pred), ptr(scores), stream_ptr()), "wm_knn_vote")
    return (pred, scores) if return_scores else pred


# --------------------------------------------------------------------------------------- L2 norm


class _L2Normalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps, out_bf16=False):
        require_gpu(x)
        rows, d = x.shape
        y = torch.empty((rows, d), dtype=torch.float32, device=x.device)
        inv = torch.empty((rows,), dtype=torch.float32, device=x.device)
        lib = _lib.load()
        check(lib.wm_l2_normalize(ptr(x), dtype_code(x), rows, d, float(eps), ptr(y), _lib.WM_F32, ptr(inv), stream_ptr()),
              "wm_l2_normalize")
        ctx.save_for_backward(y, inv)
        ctx.in_dtype = x.dtype
        ctx.x_ref = x if x.is_leaf else None   # a PARAMETER (weight-normalised last layer of the DINO head)
        if out_bf16:   # the consumer is an MFMA operand: hand it bf16 rows (own cast kernel; the float32 rows stay saved)
            yb = torch.empty((rows, d), dtype=torch.bfloat16, device=x.device)
            check(lib.wm_cast_f32_bf16(ptr(y), y.numel(), ptr(yb), stream_ptr()), "wm_cast_f32_bf16")
            return yb
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype not in (torch.float32, torch.bfloat16):
            dy = dy.float()
        out_dtype = ctx.in_dtype if ctx.in_dtype in (torch.float32, torch.bfloat16) else torch.float32
        dx = torch.empty(y.shape, dtype=out_dtype, device=y.device)   # written in the input's dtype: no cast pass
        check(_lib.load().wm_l2_normalize_bwd(ptr(dy), dtype_code(dy), ptr(y), ptr(inv), y.shape[0], y.shape[1], ptr(dx),
                                              dtype_code(dx), 0, stream_ptr()), "wm_l2_normalize_bwd")
        x = ctx.x_ref
        slot = x.grad if (x is not None and getattr(x, "_hip_arena_grad", False)) else None
        if slot is not None and slot.is_contiguous() and dx.dtype == torch.float32:
            # the parameter's gradient slot belongs to a fused optimiser: added by the library's own kernel
            check(_lib.load().wm_wgrad_finalize(ptr(dx), 1, 1, dx.numel(), 1, 1, ptr(slot), 1, stream_ptr()), "wm_wgrad_finalize(add)")
            return None, None, None
        return dx.to(ctx.in_dtype), None, None


def l2_normalize(x: torch.Tensor, eps: float = 1e-12, out_dtype: Optional[torch.dtype] = None,
                 differentiable_bf16: bool = False) -> torch.Tensor:
    """torch.nn.functional.normalize(x, dim=1) for a 2-D tensor.  With out_dtype=None the result is
    float32 and differentiable; with out_dtype=torch.bfloat16 it is an inference-only cast (the
    kNN bank build, reference src/ssl_wafermap/models/knn.py:76-80) -- unless differentiable_bf16: bf16 rows WITH a backward
    pass (gradients arrive and leave in bf16 too), for a normalisation that feeds a Linear layer (the DINO head's
    bottleneck -> last layer): the float32 <-> bf16 casts around the GEMM are then the library's, not the framework's."""
    if x.dim() != 2:
        raise ValueError("l2_normalize expects [rows, d]")
    x = x.contiguous()
    if differentiable_bf16:
        return _L2Normalize.apply(x, eps, True)
    if out_dtype is None or out_dtype == torch.float32:
        return _L2Normalize.apply(x, eps)
    require_gpu(x)
    rows, d = x.shape
    y = torch.empty((rows, d), dtype=out_dtype, device=x.device)
    check(_lib.load().wm_l2_normalize(ptr(x), dtype_code(x), rows, d, float(eps), ptr(y), dtype_code(y), 0,
                                      stream_ptr()), "wm_l2_normalize")
    return y


def vector_mean(x: torch.Tensor, scale: float = 1.0, sqrt_of: bool = False) -> torch.Tensor:
    """mean_i f(x_i) of a float32 vector as a 0-d tensor, f = identity or sqrt(scale * x_i): one block, fixed
    summation order (wm_mean_f32) -- keeps torch's reduction kernels out of the captured training step."""
    require_gpu(x)
    x = x.contiguous()
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    check(_lib.load().wm_mean_f32(ptr(x), x.numel(), float(scale), int(bool(sqrt_of)), ptr(out), stream_ptr()), "wm_mean_f32")
    return out[0]


# --------------------------------------------------------------------------------------- NT-Xent


def ntxent_forward(zn: torch.Tensor, zall: torch.Tensor, b_local: int, b_global: int, rank_offset: int,
                   temperature: float):
    """zn [2*b_local, d], zall [2*b_global, d] float32 L2-normalised rows (view-major).
    Returns (lse [2*b_local], loss_rows [2*b_local])."""
    require_gpu(zn, zall)
    d = zn.shape[1]
    if zn.shape != (2 * b_local, d) or zall.shape != (2 * b_global, d) or zn.dtype != torch.float32 \
            or zall.dtype != torch.float32:
        raise ValueError("ntxent_forward: shape/dtype mismatch")
    lse = torch.empty((2 * b_local,), dtype=torch.float32, device=zn.device)
    rows = torch.empty_like(lse)
    check(_lib.load().wm_ntxent_fwd(ptr(zn), ptr(zall), b_local, b_global, rank_offset, d, float(temperature),
                                    ptr(lse), ptr(rows), stream_ptr()), "wm_ntxent_fwd")
    return lse, rows


def ntxent_backward(zn: torch.Tensor, zall: torch.Tensor, lse_all: torch.Tensor, b_local: int, b_global: int,
                    rank_offset: int, temperature: float, grad_scale: float) -> torch.Tensor:
    require_gpu(zn, zall, lse_all)
    d = zn.shape[1]
    if lse_all.shape != (2 * b_global,) or lse_all.dtype != torch.float32:
        raise ValueError("ntxent_backward: lse_all must be float32 [2*b_global]")
    dzn = torch.empty_like(zn)
    lib = _lib.load()
    ws = torch.empty(int(lib.wm_ntxent_bwd_workspace_bytes(b_local, b_global, d)), dtype=torch.uint8, device=zn.device)
    check(lib.wm_ntxent_bwd(ptr(zn), ptr(zall), ptr(lse_all), b_local, b_global, rank_offset, d, float(temperature),
                            float(grad_scale), ptr(dzn), ptr(ws), ws.numel(), stream_ptr()), "wm_ntxent_bwd")
    return dzn
