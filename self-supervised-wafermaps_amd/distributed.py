"""Data parallelism: one process per GPU, torch.distributed over RCCL ("nccl" backend on ROCm).

The reference reaches DDP only through Lightning's `strategy="ddp"` flag, which is off in every
published run (scripts/WM811k_benchmark.py:54,78-85).  Here the gradient exchange is explicit: the
fused optimiser keeps all gradients in one flat float32 arena, which is all-reduced (SUM) in a few
large buckets — xGMI is point-to-point (7 links x ~153 GB/s per GPU), so few, large collectives
beat many small ones — and the 1/world averaging is folded into the SGD kernel's grad_scale.
The sharded kNN exchange is a single all-gather of per-shard top-k lists followed by wm_knn_merge.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # WM_DIST_BACKEND=gloo: rehearse the multi-process path with every rank on ONE GPU (RCCL refuses two
            # ranks per device); together with WM_SINGLE_DEVICE=1 in bench.py
            backend = os.environ.get("WM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class GradSync:
    """All-reduce (SUM) of flat gradient arenas in `bucket_bytes` pieces, asynchronously; call
    `wait()` before the optimiser step.  The sum is turned into the mean by the optimiser's
    grad_scale = 1/world (set here)."""

    def __init__(self, optimizer, bucket_bytes: int = 32 << 20):
        self.optimizer = optimizer
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.handles: List = []
        w = world_size()
        if hasattr(optimizer, "grad_scale"):
            optimizer.grad_scale = 1.0 / w

    def start(self) -> None:
        if world_size() == 1:
            return
        for ai, arena in enumerate(self.optimizer.grad_arenas):
            self.start_range(0, arena.numel(), ai)

    def start_range(self, lo: int, hi: int, arena_index: int = 0) -> None:
        """All-reduce elements [lo, hi) of one gradient arena, asynchronously: the collective runs on the
        process group's own stream once the work enqueued so far on the current stream (the backward stage that
        produced this range) has finished, and overlaps whatever the caller enqueues next."""
        if world_size() == 1 or hi <= lo:
            return
        arena = self.optimizer.grad_arenas[arena_index]
        for o in range(lo, hi, self.bucket_elems):
            self.handles.append(dist.all_reduce(arena[o:min(o + self.bucket_elems, hi)], op=dist.ReduceOp.SUM,
                                                async_op=True))

    def wait(self) -> None:
        for h in self.handles:
            h.wait()
        self.handles.clear()


def broadcast_state(module: torch.nn.Module, optimizer=None, src: int = 0) -> None:
    """Align the replicas before the first step (what DistributedDataParallel's constructor does): rank
    `src`'s parameters -- the optimiser's flat arenas in one broadcast each, then whatever lives outside them
    (frozen EMA teachers, momentum encoders) -- and every buffer (BatchNorm statistics, the MoCo bank, DINO
    centres).  Without it correctness would rest on every rank seeding identically before construction."""
    if world_size() == 1:
        return
    in_arena = set()
    for arena in getattr(optimizer, "_arenas", []) if optimizer is not None else []:
        dist.broadcast(arena.params, src=src)
        lo, hi = arena.params.data_ptr(), arena.params.data_ptr() + arena.params.numel() * 4
        in_arena.update(id(p) for p in module.parameters() if lo <= p.data_ptr() < hi)
    with torch.no_grad():
        for p in module.parameters():
            if id(p) not in in_arena:
                dist.broadcast(p.data, src=src)
        for b in module.buffers():
            if b.numel():
                dist.broadcast(b, src=src)
    if optimizer is not None:
        from . import ops

        ops.bump_weight_epoch()  # the bf16 kernel layouts of the weights are stale now


def sync_bn_buffers(module: torch.nn.Module) -> None:
    """Rank 0's BatchNorm running statistics to everyone (DDP's broadcast_buffers)."""
    if world_size() == 1:
        return
    for b in module.buffers():
        dist.broadcast(b, src=0)


def sharded_knn_topk(query: torch.Tensor, bank_shard: torch.Tensor, k: int, shard_offset: int):
    """Global top-k over a row-sharded bank: local top-k with global indices, ONE all-gather of the
    [nq, k] (sim, idx) lists, merge.  Every rank passes the same queries."""
    from . import functional as F_hip

    sim, idx = F_hip.knn_topk(query, bank_shard, min(k, bank_shard.shape[0]), index_base=shard_offset)
    if sim.shape[1] < k:  # a shard smaller than k: pad with -inf so lists have equal length
        pad = k - sim.shape[1]
        sim = torch.cat([sim, torch.full((sim.shape[0], pad), float("-inf"), device=sim.device)], 1)
        idx = torch.cat([idx, torch.full((idx.shape[0], pad), 2**31 - 1, dtype=torch.int32, device=idx.device)], 1)
    w = world_size()
    if w == 1:
        return sim, idx
    sims = [torch.empty_like(sim) for _ in range(w)]
    idxs = [torch.empty_like(idx) for _ in range(w)]
    dist.all_gather(sims, sim.contiguous())
    dist.all_gather(idxs, idx.contiguous())
    return F_hip.knn_merge(torch.stack(sims), torch.stack(idxs))


_GATHER_DTYPES = [torch.float32, torch.bfloat16, torch.float16, torch.int64, torch.int32, torch.uint8, torch.float64, torch.bool]


def all_gather_rows(t: torch.Tensor) -> torch.Tensor:
    """Concatenation over the ranks (in rank order) of tensors [n_rank, ...] whose first dimension may differ per rank:
    one all-gather of a small descriptor (row count, trailing shape, dtype), one of the rows padded to the longest shard.
    A rank with NO rows (an evaluation loader with fewer batches than ranks) need not know the trailing shape or the
    dtype: it adopts them from the first rank that has rows, so every rank issues the same collective.  world 1: `t`."""
    w = world_size()
    if w == 1:
        return t
    if t.dim() > 5 or t.dtype not in _GATHER_DTYPES:
        raise ValueError(f"all_gather_rows: unsupported tensor {tuple(t.shape)} {t.dtype}")
    desc = torch.zeros(8, dtype=torch.int64, device=t.device)
    meta = [t.shape[0] if t.dim() else 0, t.dim(), _GATHER_DTYPES.index(t.dtype)] + list(t.shape[1:])
    desc[: len(meta)] = torch.tensor(meta, dtype=torch.int64)
    descs = [torch.empty_like(desc) for _ in range(w)]
    dist.all_gather(descs, desc)
    descs = [d.tolist() for d in descs]
    counts = [d[0] for d in descs]
    lead = next((d for d in descs if d[0] > 0), descs[0])
    tail, dtype = tuple(lead[3:3 + max(lead[1] - 1, 0)]), _GATHER_DTYPES[lead[2]]
    if t.shape[0] == 0:
        t = torch.empty((0,) + tail, dtype=dtype, device=t.device)
    elif tuple(t.shape[1:]) != tail or t.dtype != dtype:
        raise ValueError(f"all_gather_rows: rank {rank()} holds {tuple(t.shape)} {t.dtype}, rank with rows holds (n,) + {tail} {dtype}")
    m = max(counts)
    if m == 0:
        return t
    pad = t.new_zeros((m,) + tail)
    pad[: t.shape[0]] = t
    parts = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(parts, pad.contiguous())
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
