"""hipGraph capture of the training step (launch-bound inner loop -> one graph replay).

A SimCLR step enqueues ~350 kernels; at ~18 ms of device time per step the Python/HIP launch path
costs about as much as the kernels.  `GraphedTrainStep` captures

    zero_grad -> augmentation kernel -> forward -> backward            (one hipGraph)

once, with every input at a static address (the wafer store, a static device buffer of
WmViewParams refreshed by an async copy before each replay, the parameter/gradient arenas), and
replays it per step; the gradient all-reduce (RCCL) and the fused SGD kernel stay eager after the
replay, so the captured graph contains no collective.
"""
from __future__ import annotations

import numpy as np
import torch

from .transforms.augmentations import PARAM_DTYPE


class GraphedTrainStep:
    def __init__(self, model, optimizer, dataset, batch_size: int, warmup: int = 3, fmt: str = "nhwc_bf16"):
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
        self.graph = None
        self.loss = None
        self._last_params = None
        self._warmup = warmup

    def _upload(self, params_per_group):
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

    def _body(self):
        self.opt.zero_grad()
        views = self.tr.launch(self.ds.store, self._last_params, self.B, self.fmt, params_dev=self.static)
        loss = self.model.training_step((views, None), 0)
        loss.backward()
        return loss

    def capture(self, sample_idx: np.ndarray, rng: np.random.Generator, sync=None):
        """Warm up eagerly on a side stream, then record the graph (torch.cuda.graph).  `sync` (a
        distributed.GradSync) keeps the warm-up steps data-parallel: without the gradient exchange the
        replicas' weights would drift apart before the first captured step."""
        params = self.tr.sample(self.ds.store, np.asarray(sample_idx), rng)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._upload(params)
            for _ in range(self._warmup):
                self._body()
                if sync is not None:
                    sync.start()
                    sync.wait()
                self.opt.step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # thread_local: other threads of the process (the RCCL watchdog of torch.distributed polls its
        # events while we record) must not invalidate the capture
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self.loss = self._body()
        self.graph = g
        return self

    def step(self, sample_idx: np.ndarray, rng: np.random.Generator, sync=None):
        """One training step: fresh decisions -> static buffers -> replay -> (all-reduce) -> SGD."""
        self._upload(self.tr.sample(self.ds.store, np.asarray(sample_idx), rng))
        self.graph.replay()
        if sync is not None:
            sync.start()
            sync.wait()
        self.opt.step()
        return self.loss
