"""hipGraph capture of the training step (launch-bound inner loop -> one graph replay).

A SimCLR step enqueues ~350 kernels; at ~18 ms of device time per step the Python/HIP launch path
costs about as much as the kernels.  `GraphedTrainStep` captures

    zero_grad -> augmentation kernel -> forward -> backward            (one hipGraph)

once, with every input at a static address (the wafer store, a static device buffer of
WmViewParams refreshed by an async copy before each replay, the parameter/gradient arenas), and
replays it per step; the gradient all-reduce (RCCL) and the fused SGD kernel stay eager after the
replay, so the captured graph contains no collective.

Data parallel (`stages=True`, world > 1): the backward is cut at the boundaries the model marks with
`ops.cut_point` (ResNet-18: before layer4 and before layer3) and each stage is its own graph in ONE memory pool:

    G0 = zero_grad, augmentation, forward, backward of head + layer4   -> all-reduce arena[layer4:]   (35 MB, async)
    G1 = backward of layer3                                            -> all-reduce arena[layer3:layer4]  (8 MB)
    G2 = backward of layer2, layer1, stem                              -> all-reduce arena[:layer3]   (3 MB)
    wait, fused SGD

Parameters sit in the flat gradient arena in `model.parameters()` order (stem, layer1..4, head), so the gradients a
stage completes are one contiguous tail range; its all-reduce runs on RCCL's stream under the remaining stages
(three quarters of the bytes are ready after the first ~15 % of the backward), leaving only the last 3 MB exposed.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from . import ops

from .transforms.augmentations import PARAM_DTYPE, validate_params


# WM_STEP_HOST_TIMES=1: host seconds per phase of step() (sample, upload, replay, hook, optimiser), summed in HOST_TIMES --
# a diagnostic for host-bound steps (the unstaged path only)
HOST_TIMES = {} if os.environ.get("WM_STEP_HOST_TIMES") == "1" else None


class GraphedTrainStep:
    def __init__(self, model, optimizer, dataset, batch_size: int, warmup: int = 3, fmt: str = "nhwc_bf16",
                 stages=None):
        self.model, self.opt, self.ds, self.B, self.fmt = model, optimizer, dataset, batch_size, fmt
        tr = dataset.transform
        self.tr = tr
        dev = dataset.store.device
        self.sizes = [(j - i) * batch_size for i, j in tr.groups()]
        self.static = [torch.zeros(n * PARAM_DTYPE.itemsize, dtype=torch.uint8, device=dev) for n in self.sizes]
        # the host runs ahead of the device: a ring of pinned staging buffers, each guarded by the
        # event of the async copy that last read it
        self.RING = 4
        self.pinned = [[torch.zeros(n * PARAM_DTYPE.itemsize, dtype=torch.uint8).pin_memory() for n in self.sizes]
                       for _ in range(self.RING)]
        self.events = [None] * self.RING
        self.turn = 0
        self.staged = bool(stages)
        self.graphs = []          # one graph, or one per backward stage
        self.bounds = []          # staged: arena element offsets [b_k, ..., b_1]; stage s completes arena[bounds[s]:prev)
        self.graph = None
        self.loss = None
        self._last_params = None
        # an EMA teacher's weight layouts come into being lazily in the FIRST eager step and are refreshed as one batched
        # launch (with a descriptor table built on the host) from the SECOND step on: two eager steps before capture
        has_teacher = any(not p.requires_grad for p in model.parameters())
        self._warmup = max(warmup, 2) if has_teacher else warmup

    def _upload(self, params_per_group):
        # the replayed kernel indexes the store and the output with these values: bounds-check every upload
        # on the host (a faulting kernel can reset the GPU), not only the arrays seen at capture time
        for (i, j), p in zip(self.tr.groups(), params_per_group):
            validate_params(p, self.ds.store, self.tr.transforms[i].img_size, (j - i) * self.B)
        slot = self.turn % self.RING
        self.turn += 1
        if self.events[slot] is not None:
            self.events[slot].synchronize()  # the copy that read this staging buffer has finished
        for p, pin, st in zip(params_per_group, self.pinned[slot], self.static):
            pin.numpy()[:] = np.ascontiguousarray(p).view(np.uint8).reshape(-1)
            st.copy_(pin, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[slot] = ev
        self._last_params = params_per_group

    # ---- the step as a list of stage closures: stage 0 = zero_grad .. backward down to the last cut
    def _forward(self):
        self.opt.zero_grad()
        views = self.tr.launch(self.ds.store, self._last_params, self.B, self.fmt, params_dev=self.static)
        if not self.staged:
            return self.model.training_step((views, None), 0), []
        with ops.record_cuts() as cuts:
            loss = self.model.training_step((views, None), 0)
        return loss, cuts

    @staticmethod
    def _cut_groups(cuts):
        """The recorded cut points in backward order, one group per boundary: [(name, [(x, leaf), ...])].  A model that
        runs its two views as parallel branches (nn.ViewBranches) passes every boundary twice, second view first in
        backward order: a stage continues BOTH chains (two backward calls inside one stage graph, each on its branch's
        stream; the gradient slots receive the second view's sums first, then the first view's -- the order of the
        unstaged pass)."""
        order, groups = [], {}
        for name, x, leaf in reversed(cuts):
            if name not in groups:
                groups[name] = []
                order.append(name)
            groups[name].append((x, leaf))
        return [(n, groups[n]) for n in order]

    def _body(self):
        """The whole step eagerly (warm-up; also the unstaged graph's body)."""
        loss, cuts = self._forward()
        self._backward(loss)
        for _, group in self._cut_groups(cuts):
            for x, leaf in group:
                x.backward(leaf.grad)
        return loss

    def _backward(self, loss):
        """loss.backward() with a persistent unit gradient (torch would launch a ones_like fill per step)."""
        one = getattr(self, "_one", None)
        if one is None or one.dtype != loss.dtype or one.device != loss.device or one.shape != loss.shape:
            one = self._one = torch.ones_like(loss.detach())
        loss.backward(one)

    def _stage_bounds(self, cuts):
        """Arena offset of the first parameter of the module that follows each cut (None: not a single flat arena,
        or the parameter order does not make the stages contiguous tail ranges -> run unstaged)."""
        arenas = getattr(self.opt, "_arenas", None)
        if not cuts or arenas is None or len(arenas) != 1:
            return None
        arena = arenas[0]
        ps = [p for g in self.opt.param_groups for p in g["params"] if p.requires_grad]
        off = {id(p): o for p, o in zip(ps, arena.offsets)}
        bounds = []
        for name in dict.fromkeys(name for name, _, _ in cuts):   # (each boundary once, in forward order)
            first = [off[id(p)] for n, p in self.model.named_parameters()
                     if p.requires_grad and id(p) in off and (n.startswith(name + ".") or ("." + name + ".") in n)]
            if not first:
                return None
            bounds.append(min(first))
        return bounds if bounds == sorted(bounds) else None

    def _check_stage_ranges(self, cuts, bounds):
        """One eager pass: after stage s, every gradient BELOW that stage's boundary must still be exactly zero
        (otherwise a later stage would add into a range whose all-reduce is already in flight)."""
        g = self.opt.grad_arenas[0]
        loss, cuts = self._forward()
        loss.backward()
        lows = list(reversed(bounds))
        ok = float(g[: lows[0]].abs().max()) == 0.0
        for k, (_, group) in enumerate(self._cut_groups(cuts)):
            for x, leaf in group:
                x.backward(leaf.grad)
            if k + 1 < len(lows):
                ok = ok and float(g[: lows[k + 1]].abs().max()) == 0.0
        return ok

    def _training_state(self):
        """Every tensor a training step mutates besides the gradients: parameters (arena views and the ones outside,
        e.g. EMA teachers), buffers (BatchNorm statistics and counters, DINO centre, MoCo bank), optimiser moments."""
        ts = [p.data for p in self.model.parameters()] + [b for b in self.model.buffers() if b.numel()]
        for a in getattr(self.opt, "_arenas", []):
            ts += [t for t in (a.momentum, a.second) if t is not None]
        return ts

    def capture(self, sample_idx: np.ndarray, rng: np.random.Generator, sync=None, restore: bool = True, timer=None):
        """Warm up eagerly on a side stream, then record the graph(s) (torch.cuda.graph).  `sync` (a
        distributed.GradSync) keeps the warm-up steps data-parallel: without the gradient exchange the
        replicas' weights would drift apart before the first captured step.
        restore: the warm-up steps (and the capture pass itself) are real optimiser steps on the first batch; with
        restore=True the training state is put back afterwards (parameters, buffers, moments, step counters), so a
        graph-replayed run is step-for-step the eager run -- only torch's RNG stream has advanced.
        timer: an ops.KernelTimer(external=True): its brackets are recorded INSIDE the captured graph(s) as event-record
        nodes (not around the eager warm-up), so every replay times its own MFMA launches."""
        params = self.tr.sample(self.ds.store, np.asarray(sample_idx), rng)
        snap = None
        if restore:
            torch.cuda.synchronize()
            snap = ([t.clone() for t in self._training_state()], list(getattr(self.opt, "_steps", [])))
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._upload(params)
            if self.staged:
                _, cuts = self._forward()
                bounds = self._stage_bounds(cuts)
                if bounds is None or not self._check_stage_ranges(cuts, bounds):
                    self.staged = False
                else:
                    self.bounds = list(reversed(bounds))   # in backward order: [before layer4, before layer3]
                del cuts
            for _ in range(self._warmup):
                self._body()
                if sync is not None:
                    sync.start()
                    sync.wait()
                self.opt.step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        # thread_local: other threads of the process (the RCCL watchdog of torch.distributed polls its
        # events while we record) must not invalidate the capture
        g0 = torch.cuda.CUDAGraph()
        old_timer, ops.TIMER = ops.TIMER, (timer if timer is not None else ops.TIMER)
        try:
            if not self.staged:
                with torch.cuda.graph(g0, capture_error_mode="thread_local"):
                    self.loss = self._body()
                self.graphs = [g0]
            else:
                with torch.cuda.graph(g0, capture_error_mode="thread_local"):
                    self.loss, cuts = self._forward()
                    self._backward(self.loss)
                self.graphs = [g0]
                self._cuts = cuts   # keeps the stage-boundary activations and their gradients alive in the pool
                for _, group in self._cut_groups(cuts):
                    gk = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gk, pool=g0.pool(), capture_error_mode="thread_local"):
                        for x, leaf in group:
                            x.backward(leaf.grad)
                    self.graphs.append(gk)
        finally:
            ops.TIMER = old_timer
        self.graph = g0
        if snap is not None:
            # (the capture pass enqueues nothing, but the warm-up steps ran: undo them)
            torch.cuda.synchronize()
            with torch.no_grad():
                for t, saved in zip(self._training_state(), snap[0]):
                    t.copy_(saved)
            if hasattr(self.opt, "_steps"):
                self.opt._steps[:] = snap[1]
            ops.bump_weight_epoch()
            ops.refresh_layouts(p for g in self.opt.param_groups for p in g["params"])  # bf16 layouts of the restored weights
            teachers = [p for p in self.model.parameters() if not p.requires_grad]
            if teachers:
                ops.refresh_layouts(teachers)
            torch.cuda.synchronize()
        return self

    def _step_timed(self, sample_idx, rng, sync):
        t = [time.perf_counter()]
        params = self.tr.sample(self.ds.store, np.asarray(sample_idx), rng)
        t.append(time.perf_counter())
        self._upload(params)
        t.append(time.perf_counter())
        for g in self.graphs:
            g.replay()
        t.append(time.perf_counter())
        if sync is not None:
            sync.start()
            sync.wait()
        hook = getattr(self.model, "post_graph_step", None)
        if hook is not None:
            hook()
        t.append(time.perf_counter())
        self.opt.step()
        t.append(time.perf_counter())
        for k, a, b in zip(("sample", "upload", "replay", "hook", "optimiser"), t, t[1:]):
            HOST_TIMES[k] = HOST_TIMES.get(k, 0.0) + (b - a)
        HOST_TIMES["steps"] = HOST_TIMES.get("steps", 0) + 1
        return self.loss

    def step(self, sample_idx: np.ndarray, rng: np.random.Generator, sync=None):
        """One training step: fresh decisions -> static buffers -> replay -> (all-reduce) -> SGD."""
        if HOST_TIMES is not None:
            return self._step_timed(sample_idx, rng, sync)
        self._upload(self.tr.sample(self.ds.store, np.asarray(sample_idx), rng))
        if not self.staged:
            self.graph.replay()
            if sync is not None:
                sync.start()
                sync.wait()
        else:
            hi = self.opt.grad_arenas[0].numel()
            for k, g in enumerate(self.graphs):
                g.replay()
                lo = self.bounds[k] if k < len(self.bounds) else 0
                if sync is not None:
                    sync.start_range(lo, hi)   # this stage's gradients: reduced under the next stage's replay
                hi = lo
            if sync is not None:
                sync.wait()
        hook = getattr(self.model, "post_graph_step", None)
        if hook is not None:   # host-side collectives a model keeps out of the captured graph (DINO: the centre)
            hook()
        self.opt.step()
        return self.loss
