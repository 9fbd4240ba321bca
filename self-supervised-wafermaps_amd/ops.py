"""Autograd-aware building blocks over the conv / BN / pool kernels (include/wafer_hip.h).

Activations are bf16 tensors of logical shape [N, C, H, W] in torch.channels_last memory format
(physically NHWC) or [B, C] row-major; parameters stay float32 in the torch/timm state_dict layout
(OIHW conv weights), so reference checkpoints load unchanged.  Each Function pairs a forward
kernel with its backward kernels; there is no CPU fallback.
"""
from __future__ import annotations

import contextlib
import os
from typing import Optional

import torch

from . import _lib, f32path
from . import precision as _precision
from ._lib import check, ptr, stream_ptr

_WEIGHT_EPOCH = 0  # bumped by optimisers that update parameters through raw pointers


class KernelTimer:
    """HIP-event bracket around selected launches (bench.py's live roofline measurement): events are recorded on the
    stream the kernel is enqueued on, elapsed times are read after a sync.
    external=True: the events become event-record NODES when the launches are captured into a hipGraph
    (torch.cuda.Event(external=True)): the brackets then time the kernels of a REPLAYED step -- the configuration the
    benchmark times -- without the host gaps an eager step of 10 - 60 us launches has on a slow host.  After every
    replay (and a synchronize) call accumulate()."""

    def __init__(self, external: bool = False):
        self.records = []  # (name, algorithmic work, start event, end event)
        self.external = external
        self.sums = None   # external mode: per-record elapsed ms summed over the replays seen by accumulate()
        self.replays = 0

    def run(self, name, work, fn, *args):
        if self.external:
            a = torch.cuda.Event(enable_timing=True, external=True)
            b = torch.cuda.Event(enable_timing=True, external=True)
        else:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(*args)
        b.record()
        self.records.append((name, work, a, b))
        return rc

    def accumulate(self) -> None:
        """external mode: read every bracket of the replay that has just completed."""
        if self.sums is None:
            self.sums = [0.0] * len(self.records)
        for i, (_, _, a, b) in enumerate(self.records):
            self.sums[i] += a.elapsed_time(b)
        self.replays += 1

    def summary(self, overhead_ms: float = 0.0):
        """Per name: launches, work and bracketed time; `overhead_ms` (see bracket_overhead_ms) is subtracted per launch."""
        out = {}
        for i, (name, work, a, b) in enumerate(self.records):
            d = out.setdefault(name, {"launches": 0, "work": 0.0, "ms": 0.0})
            if self.sums is not None:   # external mode: `replays` executions of the same captured launch
                d["launches"] += self.replays
                d["work"] += work * self.replays
                d["ms"] += max(self.sums[i] - overhead_ms * self.replays, 0.0)
            else:
                d["launches"] += 1
                d["work"] += work
                d["ms"] += max(a.elapsed_time(b) - overhead_ms, 0.0)
        return out

    @staticmethod
    def bracket_overhead_ms(fns, reps: int = 20) -> float:
        """What an event bracket adds to ONE short launch: the launches `fns` (a dependent chain of DIFFERENT kernels, as
        in a training step) `reps` times over, each inside its own bracket, against the same sequence inside one bracket
        (the time a launch takes in an unbroken stream, which is what a kernel trace reports).  A timing event is a barrier
        packet: it keeps the next kernel from ramping up under the tail of the previous one, costs a few microseconds
        itself, and on a slow host the device idles between the event and the launch -- on 10 - 60 us launches that
        inflated the ViT roofline's denominator by 13 % (8.03 ms bracketed against 7.11 ms in the rocprof trace, round 3)
        to 70 % (12.0 ms on a box with a slow host, round 4)."""
        for f in fns:
            f()
        torch.cuda.synchronize()
        n = len(fns) * reps
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        it = iter(pairs)
        for _ in range(reps):
            for f in fns:
                a, b = next(it)
                a.record()
                f()
                b.record()
        torch.cuda.synchronize()
        single = sum(a.elapsed_time(b) for a, b in pairs) / n
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            for f in fns:
                f()
        b.record()
        torch.cuda.synchronize()
        return max(single - a.elapsed_time(b) / n, 0.0)


TIMER = None  # set to a KernelTimer to bracket the conv launches


def _run(name, work, fn, *args):
    if TIMER is None:
        return fn(*args)
    return TIMER.run(name, work, fn, *args)


def bump_weight_epoch() -> None:
    global _WEIGHT_EPOCH
    _WEIGHT_EPOCH += 1


_BN_GROUPS = 1


@contextlib.contextmanager
def bn_groups(g: int):
    """Normalise the batch in `g` equal consecutive groups with independent statistics.
    g = 2 over cat([x0, x1]) equals the reference's two calls forward(x0); forward(x1)
    (scripts/WM811k_benchmark.py:244-245) in one pass."""
    global _BN_GROUPS
    old, _BN_GROUPS = _BN_GROUPS, int(g)
    try:
        yield
    finally:
        _BN_GROUPS = old


# ---- parallel branches of one step (the two views of a siamese step as two streams): objects that a module caches for
# "its" launches -- statistics slots -- are kept per branch
_BRANCH = 0


@contextlib.contextmanager
def branch(i: int):
    global _BRANCH
    old, _BRANCH = _BRANCH, int(i)
    try:
        yield
    finally:
        _BRANCH = old


def current_branch() -> int:
    return _BRANCH


def _bn_fold_buffers(gamma: torch.Tensor, c: int, device):
    """Persistent (dgamma, dbeta) buffers of a BatchNorm whose backward runs on a side branch: stable addresses (the
    fold's descriptor table is cached by them, and a captured graph bakes them in), one pair per use in a pass."""
    idx = getattr(gamma, "_hip_pending", 0)
    bufs = getattr(gamma, "_hip_bn_fold", None)
    if bufs is None:
        bufs = gamma._hip_bn_fold = []
    while len(bufs) <= idx:
        bufs.append(None)
    b = bufs[idx]
    if b is None or b.numel() != 2 * c or b.device != device:
        b = bufs[idx] = torch.empty(2 * c, dtype=torch.float32, device=device)
    return b[:c], b[c:]


def _flat_fold_buffer(owner: torch.Tensor, numel: int, device) -> torch.Tensor:
    """A persistent f32 buffer of `owner` for a gradient that joins the fold as one flat row (see _bn_fold_buffers)."""
    b = getattr(owner, "_hip_flat_fold", None)
    if b is None or b.numel() != numel or b.device != device:
        b = owner._hip_flat_fold = torch.empty(numel, dtype=torch.float32, device=device)
    return b


class _StackRows(torch.autograd.Function):
    """[a; b] for two row blocks of one shape (the outputs of two branches): two device copies into one buffer; the
    gradient is handed back as the two halves (views) of the incoming one."""

    @staticmethod
    def forward(ctx, a, b):
        if a.shape != b.shape or a.dtype != b.dtype or a.dim() != 2:
            raise ValueError(f"stack_rows: {tuple(a.shape)} {a.dtype} vs {tuple(b.shape)} {b.dtype}")
        out = torch.empty((2 * a.shape[0], a.shape[1]), dtype=a.dtype, device=a.device)
        out[: a.shape[0]].copy_(a)
        out[a.shape[0]:].copy_(b)
        ctx.n = a.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        return dout[: ctx.n], dout[ctx.n:]


def stack_rows(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return _StackRows.apply(a.contiguous(), b.contiguous())


def current_bn_groups() -> int:
    return _BN_GROUPS


_CUTS = None  # a list while graph.GraphedTrainStep records the stage boundaries of a step


@contextlib.contextmanager
def record_cuts():
    """Record the stage boundaries (`cut_point` calls) of one forward pass: [(name of the module that follows,
    activation, detached leaf that the rest of the forward consumed)]."""
    global _CUTS
    old, _CUTS = _CUTS, []
    try:
        yield _CUTS
    finally:
        _CUTS = old


def cut_point(x: torch.Tensor, next_module: str) -> torch.Tensor:
    """Stage boundary of the backward pass.  Normally the identity.  While cuts are recorded the forward continues
    from a detached leaf, so `loss.backward()` stops there (with the leaf's .grad = the activation gradient) and
    `x.backward(leaf.grad)` runs the next stage: the backward of a step becomes a few separately captured
    hipGraphs, and the gradient buckets a finished stage has completed are all-reduced on RCCL's stream while
    the next stage computes (distributed.GradSync.start_range)."""
    if _CUTS is None or not (torch.is_grad_enabled() and x.requires_grad):
        return x
    leaf = x.detach().requires_grad_(True)
    link = getattr(x, "_hip_bn", None)
    if link is not None:  # the next stage's first convolution still runs this BatchNorm's backward sums: the gradient
        leaf._hip_bn = link  # it leaves in leaf.grad is handed to x.backward() unchanged
    _CUTS.append((next_module, x, leaf))
    return leaf


def _need_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise _lib.WaferHipError(f"{what}: the HIP path needs a device tensor (no CPU fallback)")


def act_dtype() -> torch.dtype:
    """Storage dtype of activations under the current precision preset (precision.py)."""
    return torch.float32 if _precision.is_f32() else torch.bfloat16


def to_nhwc_bf16(x: torch.Tensor) -> torch.Tensor:
    """[N,C,H,W] any float dtype/layout -> bf16 channels_last (no copy when already so); under the float32 preset the
    activation dtype is float32."""
    if _precision.is_f32():
        return f32path.as_nhwc(x)
    if x.dim() != 4:
        raise ValueError("expected a 4-D [N,C,H,W] tensor")
    if x.dtype != torch.bfloat16:
        x = x.to(torch.bfloat16)
    return x.contiguous(memory_format=torch.channels_last)


def _nhwc_ok(x: torch.Tensor) -> bool:
    n, c, h, w = x.shape
    return x.dtype == torch.bfloat16 and x.stride() == (h * w * c, 1, w * c, c)


def _as_nhwc(x: torch.Tensor) -> torch.Tensor:
    if _nhwc_ok(x):
        return x
    n, c, h, w = x.shape
    out = torch.empty((n, h, w, c), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
    out.copy_(x)
    return out


def _empty_nhwc(n, c, h, w, device) -> torch.Tensor:
    return torch.empty((n, h, w, c), dtype=torch.bfloat16, device=device).permute(0, 3, 1, 2)


class _WeightCache:
    """bf16 kernel-layout copies of a float32 parameter, rebuilt when the parameter changes.
    The copies hang off the parameter object itself (attribute `_hip_layouts`), so they live and
    die with it and a recycled device address can never alias a stale entry."""

    def get(self, w: torch.Tensor, kind: str = "conv", need_crsk: bool = False):
        tag = (w._version, _WEIGHT_EPOCH, w.data_ptr(), kind)
        ent = getattr(w, "_hip_layouts", None)
        if ent is None or ent[0] != tag or (need_crsk and ent[2] is None):
            lib = _lib.load()
            wd = w.detach()
            if not wd.is_contiguous():
                wd = wd.contiguous()
            old_k = ent[1] if ent is not None and ent[0][3] == kind else None   # reuse the buffers:
            old_c = ent[2] if ent is not None and ent[0][3] == kind else None   # static addresses (graphs)
            if kind == "stem":
                k = w.shape[0]
                krsc = old_k if old_k is not None else torch.empty((k, 4, 4, 16), dtype=torch.bfloat16, device=w.device)
                check(lib.wm_stem_weights_prepare(ptr(wd), k, ptr(krsc), stream_ptr()), "wm_stem_weights_prepare")
                crsk = None
            else:
                k, c, r, s = w.shape if kind == "conv" else (w.shape[0], w.shape[1], 1, 1)
                krsc = old_k if old_k is not None else torch.empty((k, r, s, c), dtype=torch.bfloat16, device=w.device)
                need_crsk = need_crsk or old_c is not None
                crsk = old_c if old_c is not None else (
                    torch.empty((c, r, s, k), dtype=torch.bfloat16, device=w.device) if need_crsk else None)
                check(lib.wm_weights_prepare(ptr(wd), k, c, r, s, ptr(krsc), ptr(crsk), stream_ptr()),
                      "wm_weights_prepare")
            ent = (tag, krsc, crsk)
            w._hip_layouts = ent
        return ent[1], ent[2]


_WCACHE = _WeightCache()

# ---- batched layout passes (wm_layouts_refresh / wm_wgrad_fold): one launch for all parameters of a model
import numpy as _np

LAYOUT_DESC = _np.dtype([("w", "<u8"), ("krsc", "<u8"), ("crsk", "<u8"), ("ws", "<u8"), ("grad", "<u8"), ("K", "<i4"),
                         ("C", "<i4"), ("RS", "<i4"), ("tiles_c", "<i4"), ("tile0", "<i4"), ("nsplit", "<i4")])
assert LAYOUT_DESC.itemsize == 64
_TABLES = {}  # key (tuple of pointers and shapes) -> [device table, n_desc, total_tiles, pinned]
_TABLES_MAX = 256


def _desc_table(rows, device, per_tap: bool = False):
    """rows: [(w, krsc, crsk, ws, grad, K, C, RS, nsplit)] of raw pointers / ints -> cached device descriptor table.
    per_tap: a block owns (tile, tap) (wm_wgrad_fold) instead of a tile with all its taps (wm_layouts_refresh).
    A table looked up while a hipGraph is being captured has its device address baked into the graph: it is pinned
    and never evicted (an evicted table's block would return to the caching allocator and later replays would read
    recycled memory as pointers and shapes)."""
    key = (device.index, per_tap, tuple(rows))
    ent = _TABLES.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if ent is None:
        if capturing:
            raise _lib.WaferHipError("a layout descriptor table would have to be built during hipGraph capture: run "
                                     "one eager step (warm-up) with the same model before capturing")
        arr = _np.zeros(len(rows), dtype=LAYOUT_DESC)
        tile0 = 0
        for i, (w, krsc, crsk, ws, grad, k, c, rs, ns) in enumerate(rows):
            if per_tap:   # wm_wgrad_fold: one block per (k, c) tile and tap; 8 x 128 tiles, 4 x 64 beyond 32 slabs
                tk, tc = ((k + 7) // 8, (c + 127) // 128) if ns <= 32 else ((k + 3) // 4, (c + 63) // 64)
                arr[i] = (w, krsc, crsk, ws, grad, k, c, rs, tc, tile0, ns)
                tile0 += tk * tc * (1 if ns <= 32 else rs)   # up to 32 slabs: a block folds all taps of its tile
            else:         # wm_layouts_refresh: (32 x 32) tiles with all their taps
                tc = (c + 31) // 32
                arr[i] = (w, krsc, crsk, ws, grad, k, c, rs, tc, tile0, ns)
                tile0 += ((k + 31) // 32) * tc
        dev_tab = torch.from_numpy(arr.view(_np.uint8).reshape(-1).copy()).to(device)
        if len(_TABLES) >= _TABLES_MAX:
            for old_key in [kk for kk, vv in _TABLES.items() if not vv[3]][: _TABLES_MAX // 2]:
                del _TABLES[old_key]
        ent = [dev_tab, len(rows), tile0, False]
        _TABLES[key] = ent
    if capturing:
        ent[3] = True
    return ent[0], ent[1], ent[2]


def refresh_layouts(params) -> int:
    """Rebuild the bf16 kernel layouts of every parameter in `params` that already has them (i.e. has been used by
    a conv / Linear launch) in ONE launch, and mark them current.  Called by the fused optimisers right after
    their update kernel and by update_momentum for EMA teachers; parameters it does not cover (first use, the 16 x 16
    patch-embedding convolution, the stem's space-to-depth form) keep the lazy per-parameter path of _WeightCache.
    Returns the number of parameters refreshed."""
    rows, todo = [], []
    dev = None
    for p in params:
        ent = getattr(p, "_hip_layouts", None)
        if ent is None or not p.is_cuda:
            continue
        kind = ent[0][3]
        if kind == "conv":
            k, c, r, s = p.shape
        elif kind == "linear":
            (k, c), r, s = p.shape, 1, 1
        else:
            continue
        if r * s not in (1, 9) or not p.is_contiguous() or p.dtype != torch.float32:
            continue
        krsc, crsk = ent[1], ent[2]
        rows.append((p.data_ptr(), ptr(krsc), ptr(crsk), 0, 0, k, c, r * s, 0))
        todo.append((p, kind, krsc, crsk))
        dev = p.device
    if not rows:
        return 0
    tab, n, tiles = _desc_table(rows, dev)
    check(_lib.load().wm_layouts_refresh(ptr(tab), n, tiles, stream_ptr()), "wm_layouts_refresh")
    for p, kind, krsc, crsk in todo:
        p._hip_layouts = ((p._version, _WEIGHT_EPOCH, p.data_ptr(), kind), krsc, crsk)
    return len(rows)


# ---- weight gradients: split-K slabs (no atomics) -> one batched, ordered fold per backward pass
_PENDING_FOLDS = []   # (weight, slabs, nsplit, grad slot, K, C, RS, bias slabs, bias grad slot) of the running pass
_FOLD_QUEUED = False  # the end-of-backward callback of the running pass is registered
_SPLITS = {}


def wgrad_splits(n, h, w, c, k, r, s, p, q, stride, pad) -> int:
    """Number of pixel-range splits (= slabs) wm_conv2d_wgrad uses for this geometry."""
    # (the library reads its split-count switches per call: a changed switch must not meet a cached count, the slab buffer
    # is sized by it)
    key = (n, h, w, c, k, r, s, p, q, stride, pad, os.environ.get("WM_WGRAD_PATCH"), os.environ.get("WM_WGRAD_PATCH_BLOCKS"))
    v = _SPLITS.get(key)
    if v is None:
        v = int(_lib.load().wm_conv2d_wgrad_splits(n, h, w, c, k, r, s, p, q, stride, pad))
        if v <= 0:
            check(v if v < 0 else _lib.WM_EUNSUPPORTED, "wm_conv2d_wgrad_splits")
        _SPLITS[key] = v
    return v


def _slab_buffers(owner: torch.Tensor, nsplit: int, numel: int, bias_k: int = 0):
    """Persistent slab buffers of one parameter: f32 [nsplit * numel] (+ [nsplit * bias_k]).  Every wgrad launch
    overwrites all of its slabs, so nothing is ever zeroed.  A parameter used by several wgrad launches of ONE backward
    pass (shared weights) gets one buffer per launch: the fold at the end of the pass reads them all."""
    idx = getattr(owner, "_hip_pending", 0)
    bufs = getattr(owner, "_hip_wgrad_slabs", None)
    if bufs is None:
        bufs = owner._hip_wgrad_slabs = []
    while len(bufs) <= idx:
        bufs.append(None)
    ent = bufs[idx]
    capturing = owner.is_cuda and torch.cuda.is_current_stream_capturing()
    if (ent is None or ent[0].numel() != nsplit * numel or ent[0].device != owner.device
            or (bias_k > 0 and (ent[1] is None or ent[1].numel() != nsplit * bias_k))):
        if ent is not None and ent[2]:
            # (as StatSlots.get: the slabs a captured graph writes stay alive beside their replacement)
            retired = getattr(owner, "_hip_wgrad_retired", None)
            if retired is None:
                retired = owner._hip_wgrad_retired = []
            retired.append(ent)
        ent = [torch.empty(nsplit * numel, dtype=torch.float32, device=owner.device),
               torch.empty(nsplit * bias_k, dtype=torch.float32, device=owner.device) if bias_k > 0 else None, False]
        bufs[idx] = ent
    if capturing:
        ent[2] = True
    return ent[0], ent[1]


# ---- weight gradients on a SIDE stream.  Nothing in the backward chain waits for a weight gradient: only the fold at
# the end of the pass reads the slabs.  The chain itself alternates MFMA-bound launches (dgrad) with HBM-bound ones
# (BatchNorm / LayerNorm backward), so the MFMA-bound wgrad launches run beside it on a second stream: fork = the side
# stream waits for the point of the main stream where dy exists, join = the main stream waits for the side stream
# before the fold (both are event waits, so a hipGraph capture records two parallel branches).  The operands are kept
# referenced until the join: the caching allocator must not hand dy's memory to a later kernel of the main stream.
# MEASURED SLOWER, hence off by default (WM_WGRAD_SIDE_STREAM=1 enables it): SimCLR 11.82 / 11.85 vs 11.75 ms per
# step, DINO ViT-Tiny 10.48 vs 10.32, MAE ViT-S/16 5.06 vs 4.81 (hipGraph replay, one MI355X).  The wgrad workgroups take
# CU slots and L2 from the chain's kernels; the chain is the critical path and gets slower by more than the weight
# gradients cost when run in line.
_SIDE = {"stream": None, "dirty": False, "keep": []}
_WGRAD_STREAMS = set()
_SIDE_ON = os.environ.get("WM_WGRAD_SIDE_STREAM", "0") == "1"


def _side_fork(*tensors):
    st = _SIDE["stream"]
    if st is None or st.device != tensors[0].device:
        st = _SIDE["stream"] = torch.cuda.Stream(device=tensors[0].device)
    st.wait_stream(torch.cuda.current_stream())
    _SIDE["keep"].extend(tensors)
    _SIDE["dirty"] = True
    return st


def side_join() -> None:
    """The current stream waits for every weight-gradient launch issued on the side stream since the last join."""
    if _SIDE["dirty"]:
        torch.cuda.current_stream().wait_stream(_SIDE["stream"])
        _SIDE["keep"].clear()
        _SIDE["dirty"] = False


def wgrad(dy, x, owner, n, h, w, c, k, r, s, p, q, stride, pad, bias_k: int = 0, name: str = "conv_wgrad"):
    """Launch the weight-gradient kernel of one layer into `owner`'s slab buffer.
    Returns (slabs, bias slabs or None, nsplit)."""
    ns = wgrad_splits(n, h, w, c, k, r, s, p, q, stride, pad)
    slabs, bslabs = _slab_buffers(owner, ns, k * r * s * c, bias_k)
    lib = _lib.load()
    if dy.is_cuda:
        _WGRAD_STREAMS.add(torch.cuda.current_stream(dy.device))   # the fold waits for every stream that wrote slabs
    # the side stream only where the batched fold (which joins) consumes the slabs; wgrad_deliver joins otherwise
    side = _SIDE_ON and dy.is_cuda and _arena_grad(owner) is not None
    ctx = torch.cuda.stream(_side_fork(dy, x, slabs)) if side else contextlib.nullcontext()
    with ctx:
        if bias_k > 0:
            check(_run(name, 2.0 * n * p * q * k * r * s * c, lib.wm_conv2d_wgrad_bias, ptr(dy), ptr(x), ptr(slabs), ptr(bslabs),
                       n, h, w, c, k, r, s, p, q, stride, pad, stream_ptr()), "wm_conv2d_wgrad_bias")
        else:
            check(_run(name, 2.0 * n * p * q * k * r * s * c, lib.wm_conv2d_wgrad, ptr(dy), ptr(x), ptr(slabs), n, h, w, c, k, r,
                       s, p, q, stride, pad, stream_ptr()), "wm_conv2d_wgrad")
    return slabs, bslabs, ns


def rows_outer_product(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """sum over rows n of a[n][:] (x) b[n][:] for bf16 a [N, K], b [N, C] -> float32 [K, C]: the weight-gradient GEMM
    as a plain matrix product (cross-correlation / covariance matrices of the Barlow Twins and VICReg losses)."""
    n, k = a.shape
    c = b.shape[1]
    ns = wgrad_splits(n, 1, 1, c, k, 1, 1, 1, 1, 1, 0)
    slabs = torch.empty(ns * k * c, dtype=torch.float32, device=a.device)
    lib = _lib.load()
    check(lib.wm_conv2d_wgrad(ptr(a), ptr(b), ptr(slabs), n, 1, 1, c, k, 1, 1, 1, 1, 1, 0, stream_ptr()), "wm_conv2d_wgrad")
    out = torch.empty((k, c), dtype=torch.float32, device=a.device)
    check(lib.wm_wgrad_finalize(ptr(slabs), ns, k, c, 1, 1, ptr(out), 0, stream_ptr()), "wm_wgrad_finalize")
    return out


def wgrad_deliver(weight, slabs, ns, k, c, r, s, bias=None, bslabs=None):
    """Turn the slabs of one wgrad launch into gradients: queued for the pass's batched fold when a fused optimiser owns
    the gradient slot (returns None for that gradient), else summed now into a new tensor.  Returns (dw, dbias)."""
    lib = _lib.load()
    slot = _arena_grad(weight)
    bslot = _arena_grad(bias) if bias is not None else None
    dw = db = None
    fold_w = slot is not None and r * s in (1, 9) and c % 4 == 0
    if not (fold_w and (bias is None or bslabs is None or bslot is not None)):
        side_join()   # a finalize launch on this stream reads the slabs now
    if fold_w:
        _queue_fold(weight, slabs, ns, slot, k, c, r * s, bslabs if bslot is not None else None, bslot)
    else:
        tgt = slot
        if tgt is None:
            dw = tgt = torch.empty((k, c, r, s) if weight.dim() == 4 else (k, c), dtype=torch.float32, device=slabs.device)
        check(lib.wm_wgrad_finalize(ptr(slabs), ns, k, c, r, s, ptr(tgt), int(slot is not None), stream_ptr()),
              "wm_wgrad_finalize")
    if bias is not None and bslabs is not None and not (fold_w and bslot is not None):
        tgt = bslot
        if tgt is None:
            db = tgt = torch.empty((k,), dtype=torch.float32, device=slabs.device)
        check(lib.wm_wgrad_finalize(ptr(bslabs), ns, k, 1, 1, 1, ptr(tgt), int(bslot is not None), stream_ptr()),
              "wm_wgrad_finalize(bias)")
    return dw, db


def _queue_fold(weight, slabs, ns, slot, k, c, rs, bslabs=None, bslot=None) -> None:
    global _FOLD_QUEUED
    if not _FOLD_QUEUED:
        # runs when the current backward pass completes, on the caller's stream: after loss.backward() returns,
        # every p.grad is complete -- one fold launch per backward pass instead of one per convolution
        torch.autograd.Variable._execution_engine.queue_callback(fold_wgrads)
        _FOLD_QUEUED = True
    _PENDING_FOLDS.append((weight, slabs, ns, slot, k, c, rs, bslabs, bslot))
    weight._hip_pending = getattr(weight, "_hip_pending", 0) + 1


def drop_pending_folds() -> None:
    """Forget folds queued by a backward pass that did not complete (an exception inside backward: the engine runs no
    end-of-pass callback then).  Called by the optimisers' zero_grad(): the stale entries' slabs are simply overwritten
    by the next pass, and the next pass registers its callback again."""
    global _FOLD_QUEUED
    side_join()
    for ent in _PENDING_FOLDS:
        ent[0]._hip_pending = 0
    _PENDING_FOLDS.clear()
    _FOLD_QUEUED = False


def fold_wgrads() -> None:
    """Sum the pending weight-gradient slabs ([nsplit][K][R][S][C] f32, in slab order) into their OIHW gradient slots
    (and bias slabs into bias gradients): wm_wgrad_fold, one launch."""
    global _FOLD_QUEUED
    _FOLD_QUEUED = False
    side_join()
    if _WGRAD_STREAMS:
        cur = torch.cuda.current_stream()
        for st in _WGRAD_STREAMS:
            if st != cur and st.device == cur.device:
                cur.wait_stream(st)
        _WGRAD_STREAMS.clear()
    if not _PENDING_FOLDS:
        return
    try:
        waves = _plan_fold([(sl.data_ptr(), ns, slot.data_ptr(), k, c, rs, bs.data_ptr() if bs is not None else 0,
                             bg.data_ptr() if bg is not None else 0) for _, sl, ns, slot, k, c, rs, bs, bg in _PENDING_FOLDS])
        dev = _PENDING_FOLDS[0][1].device
    finally:
        for ent in _PENDING_FOLDS:
            ent[0]._hip_pending = 0
        _PENDING_FOLDS.clear()
    for rows in waves:
        tab, n, tiles = _desc_table(tuple(rows), dev, per_tap=True)
        check(_lib.load().wm_wgrad_fold(ptr(tab), n, tiles, stream_ptr()), "wm_wgrad_fold")


def _plan_fold(entries):
    """Pending fold entries (slabs address, nsplit, gradient-slot address, K, C, RS, bias-slabs address or 0, bias-gradient
    address or 0), in order of use -> the launches of the fold: a list of waves, each a list of descriptor rows
    (bias slabs, bias gradient, SECOND slab set or 0, slabs, gradient slot, K, C, RS, nsplit).
    One launch folds every entry whose gradient slot is distinct: its blocks read-modify-write the slots concurrently.  A
    parameter used TWICE in the pass has two entries for one slot:
      * slabs of the same shape and no bias slabs (the two views of a step as parallel branches, nn.ViewBranches): the
        second rides in the first entry's descriptor as its second slab set -- grad = (grad + set 0) + set 1, to the bit
        what a second launch behind the first computes;
      * anything else (the patch-embedding bias of a multi-resolution forward, a third use): a later launch, in order of
        use, so the sums stay reproducible."""
    waves, targets_of, first = [], [], {}
    for sl, ns, slot, k, c, rs, bs, bg in entries:
        row = (bs, bg, 0, sl, slot, k, c, rs, ns)
        targets = {slot} | ({bg} if bg else set())
        prev = first.get(slot) if not bs else None
        if prev is not None:
            wi, ri = prev
            r0 = waves[wi][ri]
            if r0[2] == 0 and r0[5:] == row[5:]:
                waves[wi][ri] = r0[:2] + (sl,) + r0[3:]
                continue
        for wi, wave in enumerate(waves):
            if not (targets_of[wi] & targets):
                wave.append(row)
                targets_of[wi] |= targets
                if not bs and slot not in first:
                    first[slot] = (wi, len(wave) - 1)
                break
        else:
            waves.append([row])
            targets_of.append(set(targets))
            if not bs and slot not in first:
                first[slot] = (len(waves) - 1, 0)
    return waves


def _arena_grad(p: torch.Tensor):
    """The parameter's gradient slot when a fused optimiser owns it (a view of the flat gradient
    arena, zeroed by optimizer.zero_grad): backward kernels then accumulate straight into it and
    autograd receives None, which skips one tiny AccumulateGrad add per parameter per step."""
    if not getattr(p, "_hip_arena_grad", False):   # (also keeps .grad of non-leaf tensors -- e.g. a normalised weight -- untouched)
        return None
    g = p.grad
    if g is not None and g.is_contiguous():
        return g
    return None


_STAT_TILES = {}


def fwd_stat_tiles(n, h, w, c, k, r, s, p, q, stride, pad, rows_per_group) -> int:
    """Statistics slots per group that wm_conv2d_fwd_stats writes for this geometry (one per 128-row tile; the
    persistent stem kernel: one per workgroup)."""
    key = (n, h, w, c, k, r, s, p, q, stride, pad, rows_per_group)
    v = _STAT_TILES.get(key)
    if v is None:
        v = int(_lib.load().wm_conv2d_fwd_stats_tiles(*key))
        if v <= 0:
            check(v if v < 0 else _lib.WM_EUNSUPPORTED, "wm_conv2d_fwd_stats_tiles")
        _STAT_TILES[key] = v
    return v


class StatSlots:
    """Per-tile statistics slots of one BatchNorm, filled by a convolution epilogue with plain stores (no atomics,
    nothing to clear) and summed in slot order by the BatchNorm finalize kernels: float32 [groups, tiles, 2, C],
    (re)allocated when the producing convolution's tile count changes.  `_hip_busy`: a dgrad epilogue has filled it and
    the BatchNorm backward has not consumed it yet."""

    def __init__(self, channels: int):
        self.channels, self.buf, self.tiles, self.groups = channels, None, 0, 0
        self._hip_busy = None

    def get(self, groups: int, tiles: int, device) -> torch.Tensor:
        capturing = torch.cuda.is_current_stream_capturing()
        if self.buf is None or self.tiles != tiles or self.groups != groups or self.buf.device != device:
            if getattr(self, "_captured", False):
                # a captured hipGraph writes / reads the old buffer at every replay: it is kept alive beside the new one
                # (another batch size run eagerly while the graph exists must not hand its memory back to the allocator
                # under the graph)
                self.__dict__.setdefault("_retired", []).append(self.buf)
                self._captured = False
            self.buf = torch.empty((groups, tiles, 2, self.channels), dtype=torch.float32, device=device)
            self.tiles, self.groups = tiles, groups
        if capturing:
            self._captured = True
        return self.buf


def stats_fusable(rows: int, groups: int) -> bool:
    """The conv epilogue can accumulate BatchNorm statistics when every 128-row tile lies inside one
    statistics group."""
    return groups > 0 and rows % groups == 0 and (rows // groups) % 128 == 0


class BnLink:
    """Rides on the OUTPUT tensor of a training-mode BatchNorm + ReLU (attribute `_hip_bn`): what the convolution that
    consumes it needs to run that BatchNorm's backward reduction, and the ReLU's backward, inside its own dgrad
    epilogue (wm_conv2d_dgrad_bnstat).  The convolution's backward fills `g_*` and sets `ready`; the BatchNorm's
    backward checks that the gradient it receives is exactly that tensor (same storage, unmodified: no other consumer
    contributed) and then skips its reduction pass, its mask and the dz copy (wm_bn_train_bwd_from_stats)."""

    __slots__ = ("y", "mean", "invstd", "gamma", "beta", "groups", "has_res", "stats", "ready", "g_ptr", "g_version", "mask")

    def __init__(self, y, mean, invstd, gamma, beta, groups, has_res, stats, mask=None):
        self.y, self.mean, self.invstd, self.gamma, self.beta = y, mean, invstd, gamma, beta
        self.groups, self.has_res, self.stats = groups, has_res, stats
        self.mask = mask   # uint8 [rows][C / 8]: bits of (output > 0), written by the forward apply pass (shortcut case)
        self.ready, self.g_ptr, self.g_version = False, 0, -1


_BN_FUSE_BWD = os.environ.get("WM_BN_FUSE_BWD", "1") != "0"  # A/B switch: 0 keeps the separate reduction pass
_BN_BITMASK = os.environ.get("WM_BN_BITMASK", "1") != "0"    # A/B switch: 0 = the dgrad epilogue re-reads the output tensor


def _out_hw(h, w, r, s, stride, pad):
    return (h + 2 * pad - r) // stride + 1, (w + 2 * pad - s) // stride + 1


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, stride, pad, stats=None, groups=1, pass_input=False, link=None):
        _need_cuda(x, "conv2d")
        x = _as_nhwc(x)
        n, c, h, w = x.shape
        k, c2, r, s = weight.shape
        if c2 != c or weight.dtype != torch.float32:
            raise ValueError(f"conv2d: input channels {c} vs weight {tuple(weight.shape)} ({weight.dtype})")
        p, q = _out_hw(h, w, r, s, stride, pad)
        train = torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)
        krsc, _ = _WCACHE.get(weight, need_crsk=train and x.requires_grad)
        y = _empty_nhwc(n, k, p, q, x.device)
        if stats is not None and stats_fusable(n * p * q, groups):
            tiles = fwd_stat_tiles(n, h, w, c, k, r, s, p, q, stride, pad, n * p * q // groups)
            check(_run("conv_fwd", 2.0 * n * p * q * k * r * s * c, _lib.load().wm_conv2d_fwd_stats, ptr(x), ptr(krsc),
                       y.data_ptr(), n, h, w, c, k, r, s, p, q, stride, pad, ptr(stats.get(groups, tiles, x.device)), tiles,
                       n * p * q // groups, stream_ptr()), "wm_conv2d_fwd_stats")
        else:
            check(_run("conv_fwd", 2.0 * n * p * q * k * r * s * c, _lib.load().wm_conv2d_fwd, ptr(x), ptr(krsc),
                       y.data_ptr(), n, h, w, c, k, r, s, p, q, stride, pad, stream_ptr()), "wm_conv2d_fwd")
        ctx.save_for_backward(x)
        ctx.weight = weight
        ctx.geom = (n, h, w, c, k, r, s, p, q, stride, pad)
        ctx.link = link if (_BN_FUSE_BWD and link is not None and link.y.shape == x.shape) else None
        if pass_input:
            # second output: the input itself (identity).  Its gradient comes back to THIS backward,
            # which adds it inside the dgrad epilogue instead of leaving a separate add to autograd.
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dres=None):
        (x,) = ctx.saved_tensors
        weight = ctx.weight
        n, h, w, c, k, r, s, p, q, stride, pad = ctx.geom
        dy = _as_nhwc(dy)
        lib = _lib.load()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            _, crsk = _WCACHE.get(weight, need_crsk=True)
            dx = _empty_nhwc(n, c, h, w, dy.device)
            if dres is not None:
                dres = _as_nhwc(dres)
            link = ctx.link
            if (link is not None and not link.ready and getattr(link.stats, "_hip_busy", None) is None
                    and lib.wm_conv2d_dgrad_bnstat_ok(n, h, w, c, k, r, s, p, q, stride, pad, link.groups)):
                # the input was relu(BN(link.y) (+ shortcut)): ReLU backward + that BatchNorm's backward sums in the epilogue
                tiles = n * h * w // link.groups // 128
                check(_run("conv_dgrad", 2.0 * n * p * q * k * r * s * c, lib.wm_conv2d_dgrad_bnstat, dy.data_ptr(), ptr(crsk),
                           dres.data_ptr() if dres is not None else 0, dx.data_ptr(), n, h, w, c, k, r, s, p, q, stride, pad,
                           link.y.data_ptr(), x.data_ptr() if (link.has_res and link.mask is None) else 0,
                           ptr(link.mask) if (link.has_res and link.mask is not None) else 0, ptr(link.gamma), ptr(link.beta),
                           ptr(link.mean), ptr(link.invstd), link.groups, ptr(link.stats.get(link.groups, tiles, dy.device)),
                           tiles, stream_ptr()), "wm_conv2d_dgrad_bnstat")
                link.ready, link.g_ptr, link.g_version = True, dx.data_ptr(), dx._version
                link.stats._hip_busy = True
            elif dres is not None:
                check(_run("conv_dgrad", 2.0 * n * p * q * k * r * s * c, lib.wm_conv2d_dgrad_add, dy.data_ptr(),
                           ptr(crsk), dres.data_ptr(), dx.data_ptr(), n, h, w, c, k, r, s, p, q, stride, pad,
                           stream_ptr()), "wm_conv2d_dgrad_add")
            else:
                check(_run("conv_dgrad", 2.0 * n * p * q * k * r * s * c, lib.wm_conv2d_dgrad, dy.data_ptr(), ptr(crsk),
                           dx.data_ptr(), n, h, w, c, k, r, s, p, q, stride, pad, stream_ptr()), "wm_conv2d_dgrad")
        if ctx.needs_input_grad[1]:
            slabs, _, ns = wgrad(dy, x, weight, n, h, w, c, k, r, s, p, q, stride, pad)
            dw, _ = wgrad_deliver(weight, slabs, ns, k, c, r, s)
        return dx, dw, None, None, None, None, None, None


def conv2d(x: torch.Tensor, weight: torch.Tensor, stride: int = 1, padding: int = 0,
           stats: Optional[torch.Tensor] = None, groups: int = 1) -> torch.Tensor:
    """bias-free conv2d; x bf16 NHWC (converted if not), weight float32 [K, C, R, S].
    `stats`: a StatSlots object into whose per-tile slots the epilogue stores the BatchNorm statistics of the output
    (used when `stats_fusable(rows, groups)`)."""
    if _precision.is_f32():
        return f32path.conv2d(x, weight, int(stride), int(padding))
    return _Conv2d.apply(x, weight, int(stride), int(padding), stats, int(groups), False, getattr(x, "_hip_bn", None))


def conv2d_passthrough(x: torch.Tensor, weight: torch.Tensor, stride: int = 1, padding: int = 0,
                       stats: Optional[torch.Tensor] = None, groups: int = 1):
    """conv2d that also hands back its input as a second (identity) output: use that output for the
    residual path of a block, and the gradient of the shortcut is added inside the dgrad kernel's
    epilogue (wm_conv2d_dgrad_add) instead of by a separate elementwise kernel."""
    if _precision.is_f32():
        return f32path.conv2d(x, weight, int(stride), int(padding)), x
    return _Conv2d.apply(x, weight, int(stride), int(padding), stats, int(groups), True, getattr(x, "_hip_bn", None))


class _StemConv(torch.autograd.Function):
    """7x7 stride-2 pad-3 convolution of a 3-channel image, run as a 4x4 convolution over the 2x2
    space-to-depth image (16 channels, 12 used) so every reduction tile is 128 contiguous bytes."""

    @staticmethod
    def forward(ctx, x, weight, stats=None, groups=1):
        _need_cuda(x, "stem_conv")
        n, c, h, w = x.shape
        k = weight.shape[0]
        s2d_input = c == 16 and x.dtype == torch.bfloat16  # already space-to-depth (augment_views fmt "s2d_bf16")
        if (c != 3 and not s2d_input) or tuple(weight.shape[1:]) != (3, 7, 7) or (not s2d_input and (h % 2 or w % 2)):
            raise ValueError("stem_conv: expects [N,3,even,even] (or space-to-depth [N,16,H/2,W/2] bf16) input and "
                             "[K,3,7,7] weights")
        if x.requires_grad:
            raise NotImplementedError("stem_conv: no input gradient (images are data)")
        lib = _lib.load()
        if s2d_input:
            h2, w2 = h, w
            xs = _as_nhwc(x)  # memory [N][H/2][W/2][16]
        else:
            if x.dtype == torch.float32 and x.is_contiguous():
                fmt = _lib.WM_IMG_NCHW_F32
            else:
                x = _as_nhwc(x)
                fmt = _lib.WM_IMG_NHWC_BF16
            h2, w2 = h // 2, w // 2
            xs = torch.empty((n, h2, w2, 16), dtype=torch.bfloat16, device=x.device)
            check(lib.wm_image_to_s2d(x.data_ptr(), fmt, n, h, w, ptr(xs), stream_ptr()), "wm_image_to_s2d")
        ws2d, _ = _WCACHE.get(weight, kind="stem")
        y = _empty_nhwc(n, k, h2, w2, x.device)
        if stats is not None and stats_fusable(n * h2 * w2, groups):
            tiles = fwd_stat_tiles(n, h2, w2, 16, k, 4, 4, h2, w2, 1, 2, n * h2 * w2 // groups)
            check(_run("conv_fwd", 2.0 * n * h2 * w2 * k * 147, lib.wm_conv2d_fwd_stats, ptr(xs), ptr(ws2d), y.data_ptr(),
                       n, h2, w2, 16, k, 4, 4, h2, w2, 1, 2, ptr(stats.get(groups, tiles, xs.device)), tiles,
                       n * h2 * w2 // groups, stream_ptr()), "wm_conv2d_fwd_stats(stem)")
        else:
            check(_run("conv_fwd", 2.0 * n * h2 * w2 * k * 147, lib.wm_conv2d_fwd, ptr(xs), ptr(ws2d), y.data_ptr(), n, h2,
                       w2, 16, k, 4, 4, h2, w2, 1, 2, stream_ptr()), "wm_conv2d_fwd(stem)")
        ctx.save_for_backward(xs)
        ctx.weight = weight
        ctx.geom = (n, h2, w2, k)
        ctx.branch = current_branch()
        return y

    @staticmethod
    def backward(ctx, dy):
        (xs,) = ctx.saved_tensors
        n, h2, w2, k = ctx.geom
        dy = _as_nhwc(dy)
        lib = _lib.load()
        slabs, _, ns = wgrad(dy, xs, ctx.weight, n, h2, w2, 16, k, 4, 4, h2, w2, 1, 2)
        side_join()   # the finalize below reads the slabs on this stream (the stem is the last layer of the pass anyway)
        slot = _arena_grad(ctx.weight)
        if slot is not None and ctx.branch != 0:
            # a side branch must not add into the slot beside the main branch: its gradient goes into a buffer of its own and
            # joins the pass's ordered fold as one flat row
            tmp = _flat_fold_buffer(ctx.weight, k * 147, dy.device)
            check(lib.wm_stem_wgrad_finalize(ptr(slabs), ns, k, ptr(tmp), 0, stream_ptr()), "wm_stem_wgrad_finalize")
            _queue_fold(ctx.weight, tmp, 1, slot, 1, k * 147, 1)
            return None, None, None, None
        if slot is not None:
            check(lib.wm_stem_wgrad_finalize(ptr(slabs), ns, k, ptr(slot), 1, stream_ptr()), "wm_stem_wgrad_finalize")
            return None, None, None, None
        dw = torch.empty((k, 3, 7, 7), dtype=torch.float32, device=dy.device)
        check(lib.wm_stem_wgrad_finalize(ptr(slabs), ns, k, ptr(dw), 0, stream_ptr()), "wm_stem_wgrad_finalize")
        return None, dw, None, None


def stem_conv(x: torch.Tensor, weight: torch.Tensor, stats: Optional[torch.Tensor] = None, groups: int = 1) -> torch.Tensor:
    if _precision.is_f32():
        if x.shape[1] != 3:
            raise ValueError("stem_conv (float32 preset): give [N,3,H,W] images (augment_views fmt \"nchw_f32\"), not the "
                             "space-to-depth bf16 layout")
        return f32path.conv2d(x, weight, 2, 3)
    return _StemConv.apply(x, weight, stats, int(groups))


_BN_WS = {}


def _bn_workspace(rows: int, c: int, g: int, device) -> torch.Tensor:
    need = _lib.load().wm_bn_workspace_bytes(rows, c, g)
    if need == 0:
        raise ValueError(f"batch_norm: unsupported shape rows={rows} C={c} G={g}")
    # one scratch buffer per (device, stream): two forward passes may be in flight on different streams (the DINO
    # teacher runs beside the student's forward pass)
    key = (device, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    ws = _BN_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=device)
        _BN_WS[key] = ws
    return ws


def _rows_c(x: torch.Tensor):
    if x.dim() == 4:
        n, c, h, w = x.shape
        return n * h * w, c
    if x.dim() == 2:
        return x.shape[0], x.shape[1]
    raise ValueError("batch_norm expects [N,C,H,W] or [B,C]")


def _as_act(x: torch.Tensor) -> torch.Tensor:
    if x.dim() == 4:
        return _as_nhwc(x)
    if x.dtype != torch.bfloat16:
        x = x.to(torch.bfloat16)
    return x.contiguous()


class _BatchNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, residual, gamma, beta, running_mean, running_var, training, groups, eps, momentum, relu,
                stats=None, counter=None, bwd_stats=None):
        _need_cuda(y, "batch_norm")
        y = _as_act(y)
        if residual is not None:
            residual = _as_act(residual)
            if residual.shape != y.shape:
                raise ValueError("batch_norm: residual shape mismatch")
        rows, c = _rows_c(y)
        lib = _lib.load()
        out = torch.empty_like(y)
        ws = _bn_workspace(rows, c, groups if training else 1, y.device)
        ctx.link = None
        if training:
            if rows % groups:
                raise ValueError("batch_norm: rows not divisible by groups")
            mean = torch.empty((groups, c), dtype=torch.float32, device=y.device)
            invstd = torch.empty_like(mean)
            will_link = (bwd_stats is not None and relu and y.dim() == 4 and gamma is not None and beta is not None
                         and stats_fusable(rows, groups) and ctx.needs_input_grad[0])
            # shortcut case: the consuming convolution's dgrad epilogue needs (output > 0); as bits that is 1/16 of the
            # bytes of the output tensor it would otherwise re-read
            mask = None
            if will_link and residual is not None and c % 8 == 0 and c <= 2048 and _BN_BITMASK:
                mask = torch.empty((rows, c // 8), dtype=torch.uint8, device=y.device)
            if stats is not None and stats.buf is not None and stats_fusable(rows, groups):
                check(lib.wm_bn_train_fwd_from_stats(y.data_ptr(), ptr(residual) if residual is not None else 0,
                                                     ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                                                     ptr(counter), rows, c,
                                                     groups, eps, momentum, int(relu), ptr(mean), ptr(invstd),
                                                     out.data_ptr(), ptr(mask), ptr(stats.buf), stats.tiles, ptr(ws),
                                                     ws.numel(), stream_ptr()), "wm_bn_train_fwd_from_stats")
            else:
                check(lib.wm_bn_train_fwd(y.data_ptr(), ptr(residual) if residual is not None else 0, ptr(gamma),
                                          ptr(beta), ptr(running_mean), ptr(running_var), ptr(counter), rows, c, groups, eps,
                                          momentum, int(relu), ptr(mean), ptr(invstd), out.data_ptr(), ptr(mask), ptr(ws),
                                          ws.numel(), stream_ptr()), "wm_bn_train_fwd")
            # a ReLU'd BN without residual recomputes its mask from y in the backward: `out` is not needed
            mask_from_y = relu and residual is None and gamma is not None and beta is not None
            ctx.save_for_backward(y, out if (relu and not mask_from_y) else None, mean, invstd)
            ctx.affine = (gamma, beta)
            ctx.meta = (rows, c, groups, relu, residual is not None, mask_from_y)
            ctx.branch = current_branch()
            if will_link:
                ctx.link = BnLink(y, mean, invstd, gamma, beta, groups, residual is not None, bwd_stats, mask)
        else:
            check(lib.wm_bn_eval_fwd(y.data_ptr(), ptr(residual) if residual is not None else 0, ptr(gamma), ptr(beta),
                                     ptr(running_mean), ptr(running_var), rows, c, eps, int(relu), out.data_ptr(),
                                     ptr(ws), ws.numel(), stream_ptr()), "wm_bn_eval_fwd")
            ctx.meta = None
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.meta is None:
            raise NotImplementedError("batch_norm: backward through eval-mode statistics is not implemented")
        y, out, mean, invstd = ctx.saved_tensors
        gamma, beta = ctx.affine
        rows, c, groups, relu, has_res, mask_from_y = ctx.meta
        dout = _as_act(dout)
        lib = _lib.load()
        dy = torch.empty_like(y)
        sg, sb = _arena_grad(gamma), _arena_grad(beta)
        direct = sg is not None and sb is not None
        # a side branch (the second view's pass on its own stream) must not read-modify-write the gradient slots the main
        # branch adds into at the same time: its sums go into their own buffers and join the pass's ordered fold
        via_fold = direct and getattr(ctx, "branch", 0) != 0 and c % 4 == 0
        if via_fold:
            direct = False
            dgamma, dbeta = _bn_fold_buffers(gamma, c, y.device)
        else:
            dgamma = sg if direct else torch.empty((c,), dtype=torch.float32, device=y.device)
            dbeta = sb if direct else torch.empty((c,), dtype=torch.float32, device=y.device)
        ws = _bn_workspace(rows, c, groups, y.device)
        link = ctx.link
        fused = False
        if link is not None and link.ready:
            link.ready = False
            link.stats._hip_busy = None
            if dout.data_ptr() == link.g_ptr and dout._version == link.g_version:
                # the consuming convolution's dgrad epilogue already applied the ReLU mask to this gradient and
                # accumulated its sums: finalize + ONE pass; the shortcut's gradient is the masked gradient itself
                check(lib.wm_bn_train_bwd_from_stats(y.data_ptr(), dout.data_ptr(), ptr(gamma), ptr(beta), ptr(mean),
                                                     ptr(invstd), rows, c, groups, ptr(dgamma), ptr(dbeta), int(direct),
                                                     dy.data_ptr(), ptr(link.stats.buf), link.stats.tiles, ptr(ws),
                                                     ws.numel(), stream_ptr()), "wm_bn_train_bwd_from_stats")
                dz = dout if has_res else None
                fused = True
            else:
                # another consumer's gradient was accumulated on top: the sums in the buffer are not this tensor's.
                # Ignore them (slots are overwritten by their next producer); the general path below re-applies the mask
                # (idempotent) and reduces again.
                pass
        if not fused:
            dz = torch.empty_like(y) if has_res else None
            check(lib.wm_bn_train_bwd(y.data_ptr(), dout.data_ptr(), out.data_ptr() if (relu and not mask_from_y) else 0,
                                      int(mask_from_y), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), rows, c, groups,
                                      ptr(dgamma), ptr(dbeta), int(direct), dy.data_ptr(), dz.data_ptr() if has_res else 0,
                                      ptr(ws), ws.numel(), stream_ptr()), "wm_bn_train_bwd")
        if via_fold:
            _queue_fold(gamma, dgamma, 1, sg, 1, c, 1)
            _queue_fold(beta, dbeta, 1, sb, 1, c, 1)
            return (dy, dz) + (None,) * 12
        if direct:
            return (dy, dz) + (None,) * 12
        return (dy, dz, dgamma, dbeta) + (None,) * 10


def batch_norm(y, gamma, beta, running_mean, running_var, training: bool, residual=None, relu: bool = False,
               eps: float = 1e-5, momentum: float = 0.1, groups: Optional[int] = None, stats=None,
               num_batches_tracked=None, bwd_stats=None):
    """out = relu?(BN(y) (+ residual)) on bf16 [N,C,H,W] (NHWC) or [B,C].  `stats`: the buffer the
    producing conv2d(..., stats=) accumulated into (same `groups`).  `num_batches_tracked` (int64 scalar
    tensor): incremented by `groups` inside the statistics kernel when training.  `bwd_stats`: a second buffer of the
    same kind; when given (training, ReLU, 4-D), the output carries a BnLink and the convolution that consumes it runs
    this BatchNorm's backward reduction in its dgrad epilogue."""
    g = current_bn_groups() if groups is None else groups
    if _precision.is_f32():
        return f32path.batch_norm(y, gamma, beta, running_mean, running_var, bool(training), residual, bool(relu), eps, momentum,
                                  int(g), num_batches_tracked)
    out = _BatchNorm.apply(y, residual, gamma, beta, running_mean, running_var, bool(training), int(g), float(eps),
                           float(momentum), bool(relu), stats, num_batches_tracked, bwd_stats)
    link = getattr(out.grad_fn, "link", None) if out.grad_fn is not None else None
    if link is not None:
        out._hip_bn = link
    return out


class _SyncBatchNorm(torch.autograd.Function):
    """torch.nn.SyncBatchNorm on the HIP kernels: batch statistics over every rank of `group` (the reference's
    `sync_batchnorm=True`, scripts/WM811k_benchmark.py:62,1103 -> Lightning's convert_sync_batchnorm).  Forward: this
    rank's (sum y, sum y^2) -> all-reduce -> normalise with the global mean / variance; backward: (sum g, sum g * xhat)
    -> all-reduce -> dy with the global means.  dgamma / dbeta stay local sums (torch's do; GradSync averages them).
    Equal per-rank batches (drop_last loaders)."""

    @staticmethod
    def forward(ctx, y, residual, gamma, beta, running_mean, running_var, groups, eps, momentum, relu, stats, counter,
                group):
        import torch.distributed as dist

        _need_cuda(y, "sync_batch_norm")
        y = _as_act(y)
        if residual is not None:
            residual = _as_act(residual)
            if residual.shape != y.shape:
                raise ValueError("sync_batch_norm: residual shape mismatch")
        rows, c = _rows_c(y)
        if rows % groups:
            raise ValueError("sync_batch_norm: rows not divisible by groups")
        lib = _lib.load()
        world = dist.get_world_size(group)
        out = torch.empty_like(y)
        ws = _bn_workspace(rows, c, groups, y.device)
        sums = torch.empty((groups, 2, c), dtype=torch.float32, device=y.device)
        fused = stats is not None and stats.buf is not None and stats_fusable(rows, groups)
        check(lib.wm_bn_sync_fwd_sums(y.data_ptr(), rows, c, groups, ptr(stats.buf) if fused else 0,
                                      stats.tiles if fused else 0, ptr(sums), ptr(ws), ws.numel(), stream_ptr()),
              "wm_bn_sync_fwd_sums")
        dist.all_reduce(sums, group=group)
        mean = torch.empty((groups, c), dtype=torch.float32, device=y.device)
        invstd = torch.empty_like(mean)
        count = rows // groups * world
        check(lib.wm_bn_sync_fwd_apply(y.data_ptr(), ptr(residual) if residual is not None else 0, ptr(gamma), ptr(beta),
                                       ptr(running_mean), ptr(running_var), ptr(counter), rows, c, groups, count, eps,
                                       momentum, int(relu), ptr(mean), ptr(invstd), out.data_ptr(), ptr(sums), ptr(ws),
                                       ws.numel(), stream_ptr()), "wm_bn_sync_fwd_apply")
        mask_from_y = relu and residual is None and gamma is not None and beta is not None
        ctx.save_for_backward(y, out if (relu and not mask_from_y) else None, mean, invstd)
        ctx.affine = (gamma, beta)
        ctx.meta = (rows, c, groups, relu, residual is not None, mask_from_y, count, group)
        return out

    @staticmethod
    def backward(ctx, dout):
        import torch.distributed as dist

        y, out, mean, invstd = ctx.saved_tensors
        gamma, beta = ctx.affine
        rows, c, groups, relu, has_res, mask_from_y, count, group = ctx.meta
        dout = _as_act(dout)
        lib = _lib.load()
        dy = torch.empty_like(y)
        sg, sb = _arena_grad(gamma), _arena_grad(beta)
        direct = sg is not None and sb is not None
        dgamma = sg if direct else torch.empty((c,), dtype=torch.float32, device=y.device)
        dbeta = sb if direct else torch.empty((c,), dtype=torch.float32, device=y.device)
        ws = _bn_workspace(rows, c, groups, y.device)
        sums = torch.empty((groups, 2, c), dtype=torch.float32, device=y.device)
        mask = out.data_ptr() if (relu and not mask_from_y) else 0
        check(lib.wm_bn_sync_bwd_sums(y.data_ptr(), dout.data_ptr(), mask, int(mask_from_y), ptr(gamma), ptr(beta),
                                      ptr(mean), ptr(invstd), rows, c, groups, ptr(dgamma), ptr(dbeta), int(direct),
                                      ptr(sums), ptr(ws), ws.numel(), stream_ptr()), "wm_bn_sync_bwd_sums")
        dist.all_reduce(sums, group=group)
        dz = torch.empty_like(y) if has_res else None
        check(lib.wm_bn_sync_bwd_apply(y.data_ptr(), dout.data_ptr(), mask, int(mask_from_y), ptr(gamma), ptr(beta),
                                       ptr(mean), ptr(invstd), rows, c, groups, count, ptr(sums), dy.data_ptr(),
                                       dz.data_ptr() if has_res else 0, ptr(ws), ws.numel(), stream_ptr()),
              "wm_bn_sync_bwd_apply")
        if direct:
            return (dy, dz) + (None,) * 11
        return (dy, dz, dgamma, dbeta) + (None,) * 9


def sync_batch_norm(y, gamma, beta, running_mean, running_var, residual=None, relu: bool = False, eps: float = 1e-5,
                    momentum: float = 0.1, groups: Optional[int] = None, stats=None, num_batches_tracked=None,
                    process_group=None):
    """Training-mode BatchNorm with statistics over all ranks of `process_group` (see _SyncBatchNorm)."""
    g = current_bn_groups() if groups is None else groups
    if _precision.is_f32():
        raise NotImplementedError("sync_batch_norm: not part of the float32 (parity) preset")
    return _SyncBatchNorm.apply(y, residual, gamma, beta, running_mean, running_var, int(g), float(eps), float(momentum),
                                bool(relu), stats, num_batches_tracked, process_group)


class _BnReluMaxPool(torch.autograd.Function):
    """ResNet stem tail: maxpool3x3s2(relu(BN(y))) without materialising the normalised activation."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, training, groups, eps, momentum, stats=None,
                counter=None):
        _need_cuda(y, "bn_relu_maxpool")
        y = _as_nhwc(y)
        n, c, h, w = y.shape
        rows = n * h * w
        lib = _lib.load()
        g = groups if training else 1
        if n % g:
            raise ValueError("bn_relu_maxpool: batch not divisible by the statistics groups")
        scale = torch.empty((g, c), dtype=torch.float32, device=y.device)
        shift = torch.empty_like(scale)
        if training:
            mean, invstd = torch.empty_like(scale), torch.empty_like(scale)
            ws = _bn_workspace(rows, c, g, y.device)
            fused = stats is not None and stats.buf is not None and stats_fusable(rows, g)
            check(lib.wm_bn_train_stats(y.data_ptr(), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                                        ptr(counter), rows, c, g,
                                        eps, momentum, ptr(mean), ptr(invstd), ptr(scale), ptr(shift),
                                        ptr(stats.buf) if fused else 0, stats.tiles if fused else 0, ptr(ws), ws.numel(), stream_ptr()),
                  "wm_bn_train_stats")
            ctx.save_for_backward(y, mean, invstd)
        else:
            check(lib.wm_bn_eval_scale_shift(ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), c, eps,
                                             ptr(scale), ptr(shift), stream_ptr()), "wm_bn_eval_scale_shift")
        p, q = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        out = _empty_nhwc(n, c, p, q, y.device)
        idx = torch.empty((n, p, q, c), dtype=torch.uint8, device=y.device)
        # the conv outputs at the selected positions: the backward's reduction then needs only pooled-size tensors
        ysel = torch.empty((n, p, q, c), dtype=torch.bfloat16, device=y.device) if training and y.requires_grad else None
        check(lib.wm_bn_relu_maxpool3x3s2_fwd(y.data_ptr(), ptr(scale), ptr(shift), n, h, w, c, g, out.data_ptr(), ptr(idx),
                                              ptr(ysel), stream_ptr()), "wm_bn_relu_maxpool3x3s2_fwd")
        ctx.training = training
        ctx.idx = idx
        ctx.ysel = ysel
        ctx.affine = (gamma, beta)
        ctx.meta = (n, c, h, w, g)
        ctx.branch = current_branch()
        return out

    @staticmethod
    def backward(ctx, dpooled):
        if not ctx.training:
            raise NotImplementedError("bn_relu_maxpool: backward through eval-mode statistics is not implemented")
        y, mean, invstd = ctx.saved_tensors
        gamma, beta = ctx.affine
        n, c, h, w, g = ctx.meta
        rows = n * h * w
        lib = _lib.load()
        dpooled = _as_nhwc(dpooled)
        dy = torch.empty_like(y)
        sg, sb = _arena_grad(gamma), _arena_grad(beta)
        direct = sg is not None and sb is not None
        via_fold = direct and getattr(ctx, "branch", 0) != 0 and c % 4 == 0   # (as _BatchNorm.backward)
        if via_fold:
            direct = False
            dgamma, dbeta = _bn_fold_buffers(gamma, c, y.device)
        else:
            dgamma = sg if direct else torch.empty((c,), dtype=torch.float32, device=y.device)
            dbeta = sb if direct else torch.empty((c,), dtype=torch.float32, device=y.device)
        ws = _bn_workspace(rows, c, g, y.device)
        # the pooled gradient is scattered back inside the two BN backward passes (no 112x112 dout tensor)
        check(lib.wm_bn_relu_maxpool_bwd(y.data_ptr(), ptr(ctx.ysel), dpooled.data_ptr(), ptr(ctx.idx), n, h, w, c, ptr(gamma), ptr(beta),
                                         ptr(mean), ptr(invstd), g, ptr(dgamma), ptr(dbeta), int(direct), dy.data_ptr(),
                                         ptr(ws), ws.numel(), stream_ptr()), "wm_bn_relu_maxpool_bwd")
        if via_fold:
            _queue_fold(gamma, dgamma, 1, sg, 1, c, 1)
            _queue_fold(beta, dbeta, 1, sb, 1, c, 1)
            return dy, None, None, None, None, None, None, None, None, None, None
        if direct:
            return dy, None, None, None, None, None, None, None, None, None, None
        return dy, dgamma, dbeta, None, None, None, None, None, None, None, None


def bn_relu_maxpool(y, gamma, beta, running_mean, running_var, training: bool, eps: float = 1e-5,
                    momentum: float = 0.1, groups: Optional[int] = None, stats=None, num_batches_tracked=None):
    """max_pool3x3s2(relu(batch_norm(y))) in one pass over y (ResNet stem)."""
    g = current_bn_groups() if groups is None else groups
    if _precision.is_f32():
        return f32path.max_pool3x3s2(f32path.batch_norm(y, gamma, beta, running_mean, running_var, bool(training), None, True, eps,
                                                         momentum, int(g), num_batches_tracked))
    return _BnReluMaxPool.apply(y, gamma, beta, running_mean, running_var, bool(training), int(g), float(eps),
                                float(momentum), stats, num_batches_tracked)


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need_cuda(x, "max_pool")
        x = _as_nhwc(x)
        n, c, h, w = x.shape
        p, q = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = _empty_nhwc(n, c, p, q, x.device)
        idx = torch.empty((n, p, q, c), dtype=torch.uint8, device=x.device)
        check(_lib.load().wm_maxpool3x3s2_fwd(x.data_ptr(), n, h, w, c, y.data_ptr(), ptr(idx), stream_ptr()),
              "wm_maxpool3x3s2_fwd")
        ctx.save_for_backward(idx)
        ctx.geom = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        n, c, h, w = ctx.geom
        dy = _as_nhwc(dy)
        dx = _empty_nhwc(n, c, h, w, dy.device)
        check(_lib.load().wm_maxpool3x3s2_bwd(dy.data_ptr(), ptr(idx), n, h, w, c, dx.data_ptr(), stream_ptr()),
              "wm_maxpool3x3s2_bwd")
        return dx


def max_pool3x3s2(x: torch.Tensor) -> torch.Tensor:
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1)."""
    if _precision.is_f32():
        return f32path.max_pool3x3s2(x)
    return _MaxPool.apply(x)


class _GlobalAvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need_cuda(x, "global_avg_pool")
        x = _as_nhwc(x)
        n, c, h, w = x.shape
        y = torch.empty((n, c), dtype=torch.bfloat16, device=x.device)
        check(_lib.load().wm_gap_fwd(x.data_ptr(), n, h * w, c, ptr(y), stream_ptr()), "wm_gap_fwd")
        ctx.geom = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w = ctx.geom
        dy = dy.to(torch.bfloat16).contiguous()
        dx = _empty_nhwc(n, c, h, w, dy.device)
        check(_lib.load().wm_gap_bwd(ptr(dy), n, h * w, c, dx.data_ptr(), stream_ptr()), "wm_gap_bwd")
        return dx


def global_avg_pool(x: torch.Tensor) -> torch.Tensor:
    """[N,C,H,W] -> [N,C] (mean over H*W), bf16."""
    if _precision.is_f32():
        return f32path.global_avg_pool(x)
    return _GlobalAvgPool.apply(x)


class _Linear(torch.autograd.Function):
    """y = x @ W^T as the 1x1 convolution on a 1x1 image (same MFMA kernel)."""

    @staticmethod
    def forward(ctx, x, weight):
        _need_cuda(x, "linear")
        if x.dim() != 2 or weight.dim() != 2 or x.shape[1] != weight.shape[1]:
            raise ValueError(f"linear: x {tuple(x.shape)} vs weight {tuple(weight.shape)}")
        x = _as_act(x)
        b, c = x.shape
        k = weight.shape[0]
        train = torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)
        krsc, _ = _WCACHE.get(weight, kind="linear", need_crsk=train and x.requires_grad)
        y = torch.empty((b, k), dtype=torch.bfloat16, device=x.device)
        check(_run("gemm_fwd", 2.0 * b * c * k, _lib.load().wm_conv2d_fwd, ptr(x), ptr(krsc), ptr(y), b, 1, 1, c, k, 1, 1,
                   1, 1, 1, 0, stream_ptr()), "wm_conv2d_fwd(linear)")
        ctx.save_for_backward(x)
        ctx.weight = weight
        ctx.geom = (b, c, k)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        weight = ctx.weight
        b, c, k = ctx.geom
        dy = _as_act(dy)
        lib = _lib.load()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            _, crsk = _WCACHE.get(weight, kind="linear", need_crsk=True)
            dx = torch.empty((b, c), dtype=torch.bfloat16, device=dy.device)
            check(_run("gemm_dgrad", 2.0 * b * c * k, lib.wm_conv2d_dgrad, ptr(dy), ptr(crsk), ptr(dx), b, 1, 1, c, k, 1, 1,
                       1, 1, 1, 0, stream_ptr()), "wm_conv2d_dgrad(linear)")
        if ctx.needs_input_grad[1]:
            slabs, _, ns = wgrad(dy, x, weight, b, 1, 1, c, k, 1, 1, 1, 1, 1, 0, name="gemm_wgrad")
            dw, _ = wgrad_deliver(weight, slabs, ns, k, c, 1, 1)
        return dx, dw


def linear(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """Bias-free nn.Linear: x bf16 [B, C], weight float32 [K, C] -> bf16 [B, K]."""
    if _precision.is_f32():
        return f32path.linear(x, weight)
    return _Linear.apply(x, weight)
