"""Fused optimiser over flat parameter / gradient arenas.

`SGD` computes exactly torch.optim.SGD's update (momentum, weight decay, no dampening / nesterov —
what the reference's SimCLR uses, scripts/WM811k_benchmark.py:250-255) in ONE kernel launch over a
contiguous float32 arena.  Parameters keep their identity (their .data become views of the arena,
their .grad views of the gradient arena), so LR schedulers, state_dicts and a flat RCCL all-reduce
of the gradients all work on the same memory.
"""
from __future__ import annotations

from typing import List

import torch

from . import _lib, ops
from ._lib import check, ptr, stream_ptr

_ALIGN = 64  # elements: every parameter starts on a 256-byte boundary


class _Arena:
    def __init__(self, params: List[torch.nn.Parameter], second_moment: bool = False):
        dev = params[0].device
        if dev.type != "cuda":
            raise _lib.WaferHipError("fused SGD needs parameters on the GPU")
        offs, total = [], 0
        for p in params:
            if p.dtype != torch.float32:
                raise TypeError("fused SGD arena holds float32 parameters only")
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = total
        self.params = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(total, dtype=torch.float32, device=dev)
        self.momentum = torch.zeros(total, dtype=torch.float32, device=dev)
        self.second = torch.zeros(total, dtype=torch.float32, device=dev) if second_moment else None
        with torch.no_grad():
            for p, o in zip(params, offs):
                n = p.numel()
                self.params[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.params[o:o + n].view(p.shape)
                p.grad = self.grads[o:o + n].view(p.shape)
                p._hip_arena_grad = True  # backward kernels may accumulate straight into p.grad
        self.offsets = offs


class _HyperUpload:
    """Asynchronous host -> device upload of the few floats an update kernel reads (learning rate, decay, bias corrections).
    `hyper.copy_(torch.tensor(...))` from pageable memory blocks the host until the stream has drained -- with AdamW's
    per-step bias corrections that was one full synchronisation per training step: the host could not prepare step i + 1
    under step i, and the GPU idled 0.7 ms of every 9.5-ms DINO ViT-Tiny step (trace: profiles/r04_experiments.md, section 5).
    Here the values go through a ring of pinned staging rows, each guarded by the event of the copy that last read it; the
    host only ever waits when it is a whole ring (8 steps) ahead of the device."""

    RING = 8

    def __init__(self, n: int):
        self.rows = [torch.zeros(n, dtype=torch.float32).pin_memory() for _ in range(self.RING)]
        self.events = [None] * self.RING
        self.turn = 0

    def __call__(self, dst: torch.Tensor, values) -> None:
        slot = self.turn % self.RING
        self.turn += 1
        if self.events[slot] is not None:
            self.events[slot].synchronize()
        row = self.rows[slot]
        for i, v in enumerate(values):
            row[i] = v
        dst.copy_(row, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[slot] = ev


class _ArenaStateMixin:
    """state_dict / load_state_dict in torch.optim's own format (state[i]["momentum_buffer"] for SGD / LARS,
    state[i]["step" | "exp_avg" | "exp_avg_sq"] for Adam(W)), so that a run resumes with its momentum, moments
    and bias-correction counters and a torch.optim checkpoint of the same parameter list loads here.  The
    tensors live in the flat arenas; torch's `self.state` stays empty."""

    _STATE_KEYS = ("momentum_buffer",)   # arena attribute order: momentum[, second]

    def _param_slices(self):
        """(packed torch index, arena, offset, parameter) for every parameter that owns an arena slot."""
        idx = 0
        for group, arena in zip(self.param_groups, self._arenas):
            offs = iter(arena.offsets)
            for p in group["params"]:
                if p.requires_grad:
                    yield idx, arena, next(offs), p
                idx += 1

    def state_dict(self):
        sd = super().state_dict()
        state = {}
        steps = getattr(self, "_steps", None)
        garena = {id(a): gi for gi, a in enumerate(self._arenas)}
        for idx, arena, off, p in self._param_slices():
            n = p.numel()
            ent = {}
            bufs = (arena.momentum, arena.second)
            for key, buf in zip(self._STATE_KEYS, bufs):
                ent[key] = buf[off:off + n].view(p.shape).clone()
            if steps is not None:
                ent["step"] = torch.tensor(float(steps[garena[id(arena)]]))
            state[idx] = ent
        sd["state"] = state
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict({"state": {}, "param_groups": state_dict["param_groups"]})
        state = state_dict.get("state", {})
        garena = {id(a): gi for gi, a in enumerate(self._arenas)}
        seen_step = {}
        with torch.no_grad():
            for idx, arena, off, p in self._param_slices():
                ent = state.get(idx, state.get(str(idx)))
                if ent is None:
                    continue
                n = p.numel()
                for key, buf in zip(self._STATE_KEYS, (arena.momentum, arena.second)):
                    v = ent.get(key)
                    if v is not None:
                        if v.numel() != n:
                            raise ValueError(f"optimizer state {key} of parameter {idx}: {tuple(v.shape)} vs {tuple(p.shape)}")
                        buf[off:off + n].copy_(v.reshape(-1).to(buf.device, torch.float32))
                if "step" in ent:
                    seen_step.setdefault(garena[id(arena)], int(float(ent["step"])))
        if hasattr(self, "_steps"):
            for gi, t in seen_step.items():
                self._steps[gi] = t
        if hasattr(self, "_hyper_host"):
            self._hyper_host = [None] * len(self._hyper_host)  # re-upload lr / momentum / decay at the next step


class SGD(_ArenaStateMixin, torch.optim.Optimizer):
    def __init__(self, params, lr: float, momentum: float = 0.0, weight_decay: float = 0.0, grad_scale: float = 1.0):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError("invalid SGD hyper-parameter")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.grad_scale = float(grad_scale)
        self._arenas, self._hyper, self._hyper_host = [], [], []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            arena = _Arena(ps)
            self._arenas.append(arena)
            self._hyper.append(torch.zeros(4, dtype=torch.float32, device=arena.params.device))
            self._hyper_host.append(None)
        self._upload = _HyperUpload(4)
        ops.bump_weight_epoch()

    @property
    def grad_arenas(self) -> List[torch.Tensor]:
        """Flat gradient buffers (one per param group): what data-parallel training all-reduces."""
        return [a.grads for a in self._arenas]

    def zero_grad(self, set_to_none: bool = False) -> None:
        ops.drop_pending_folds()  # (a backward pass that raised leaves its queue behind: ADVICE r2)
        for a in self._arenas:
            check(_lib.load().wm_fill_zero(ptr(a.grads), a.grads.numel() * 4, stream_ptr()), "wm_fill_zero")

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        ops.fold_wgrads()  # no-op unless a backward pass ended without its end-of-pass callback
        lib = _lib.load()
        for group, arena, hyper, i in zip(self.param_groups, self._arenas, self._hyper, range(len(self._arenas))):
            h = (float(group["lr"]), float(group["momentum"]), float(group["weight_decay"]), self.grad_scale)
            if h != self._hyper_host[i]:
                self._upload(hyper, h)
                self._hyper_host[i] = h
            check(lib.wm_sgd_step(ptr(arena.params), ptr(arena.grads), ptr(arena.momentum), arena.numel, ptr(hyper),
                                  stream_ptr()), "wm_sgd_step")
        ops.bump_weight_epoch()
        ops.refresh_layouts(p for g in self.param_groups for p in g["params"])  # bf16 kernel layouts: one launch
        return loss


class AdamW(_ArenaStateMixin, torch.optim.Optimizer):
    """torch.optim.AdamW's update (decoupled weight decay, bias-corrected moments, no amsgrad — what
    the reference's DINOViT and MAE use, scripts/WM811k_benchmark.py:591-598, :956-963) in one launch
    per parameter group over flat float32 arenas."""

    _STATE_KEYS = ("exp_avg", "exp_avg_sq")

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 grad_scale: float = 1.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.grad_scale = float(grad_scale)
        self._l2 = False
        self._arenas, self._hyper, self._steps = [], [], []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            arena = _Arena(ps, second_moment=True)
            self._arenas.append(arena)
            self._hyper.append(torch.zeros(9, dtype=torch.float32, device=arena.params.device))
            self._steps.append(0)
        self._upload = _HyperUpload(9)
        ops.bump_weight_epoch()

    @property
    def grad_arenas(self) -> List[torch.Tensor]:
        return [a.grads for a in self._arenas]

    def zero_grad(self, set_to_none: bool = False) -> None:
        ops.drop_pending_folds()  # (a backward pass that raised leaves its queue behind: ADVICE r2)
        for a in self._arenas:
            check(_lib.load().wm_fill_zero(ptr(a.grads), a.grads.numel() * 4, stream_ptr()), "wm_fill_zero")

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        ops.fold_wgrads()  # no-op unless a backward pass ended without its end-of-pass callback
        lib = _lib.load()
        for i, (group, arena, hyper) in enumerate(zip(self.param_groups, self._arenas, self._hyper)):
            self._steps[i] += 1
            t = self._steps[i]
            b1, b2 = group["betas"]
            h = (float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                 1.0 - b1 ** t, 1.0 - b2 ** t, self.grad_scale, 1.0 if self._l2 else 0.0)
            self._upload(hyper, h)
            check(lib.wm_adamw_step(ptr(arena.params), ptr(arena.grads), ptr(arena.momentum), ptr(arena.second),
                                    arena.numel, ptr(hyper), stream_ptr()), "wm_adamw_step")
        ops.bump_weight_epoch()
        ops.refresh_layouts(p for g in self.param_groups for p in g["params"])  # bf16 kernel layouts: one launch
        return loss


class LARS(_ArenaStateMixin, torch.optim.Optimizer):
    """timm.optim.lars.Lars (momentum SGD with a per-parameter trust ratio; the reference's BarlowTwins and
    VICReg optimiser, scripts/WM811k_benchmark.py:383-392): two launches per parameter group over the flat
    arena (per-parameter squared norms, then the update)."""

    def __init__(self, params, lr: float = 1.0, momentum: float = 0.0, weight_decay: float = 0.0,
                 trust_coeff: float = 0.001, eps: float = 1e-8, grad_scale: float = 1.0):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError("invalid LARS hyper-parameter")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, trust_coeff=trust_coeff,
                                      eps=eps))
        self.grad_scale = float(grad_scale)
        self._arenas, self._hyper, self._seg, self._norms = [], [], [], []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            arena = _Arena(ps)
            dev = arena.params.device
            ends = [o + p.numel() for o, p in zip(arena.offsets, ps)]
            # a parameter's segment is [offset, offset + numel): the alignment gaps belong to no segment
            seg = torch.tensor([v for o, e in zip(arena.offsets, ends) for v in (o, e)], dtype=torch.int64)
            self._arenas.append(arena)
            self._hyper.append(torch.zeros(6, dtype=torch.float32, device=dev))
            self._seg.append((seg.to(dev), len(ps)))
            self._norms.append(torch.zeros(4 * len(ps), dtype=torch.float32, device=dev))
        self._upload = _HyperUpload(6)
        ops.bump_weight_epoch()

    @property
    def grad_arenas(self) -> List[torch.Tensor]:
        return [a.grads for a in self._arenas]

    def zero_grad(self, set_to_none: bool = False) -> None:
        ops.drop_pending_folds()  # (a backward pass that raised leaves its queue behind: ADVICE r2)
        for a in self._arenas:
            check(_lib.load().wm_fill_zero(ptr(a.grads), a.grads.numel() * 4, stream_ptr()), "wm_fill_zero")

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        ops.fold_wgrads()  # no-op unless a backward pass ended without its end-of-pass callback
        lib = _lib.load()
        for group, arena, hyper, (seg, n), norms in zip(self.param_groups, self._arenas, self._hyper, self._seg,
                                                        self._norms):
            h = (float(group["lr"]), float(group["momentum"]), float(group["weight_decay"]), float(group["trust_coeff"]),
                 float(group["eps"]), self.grad_scale)
            self._upload(hyper, h)
            # segments are (begin, end) pairs: 2n entries = n "segments" of stride 2 -> pass as 2n-1 boundaries
            check(lib.wm_lars_step(ptr(arena.params), ptr(arena.grads), ptr(arena.momentum), ptr(seg), 2 * n - 1,
                                   ptr(hyper), ptr(norms), stream_ptr()), "wm_lars_step")
        ops.bump_weight_epoch()
        ops.refresh_layouts(p for g in self.param_groups for p in g["params"])  # bf16 kernel layouts: one launch
        return loss


class Adam(AdamW):
    """torch.optim.Adam: the same kernel with the weight decay added to the gradient (L2) instead of decoupled
    (the reference's SwaV: Adam(lr 1e-3 x bs/256, weight_decay 1e-6), scripts/WM811k_benchmark.py:866-871)."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 grad_scale: float = 1.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, grad_scale=grad_scale)
        self._l2 = True
