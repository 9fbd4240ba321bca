"""WaferMapDataset + WaferLoader: the reference's dataset (src/ssl_wafermap/data/dataset.py:5-37)
and its DataLoader usage (scripts/WM811k_benchmark.py:158-195) for a GPU-resident store.

The reference augments one sample at a time in DataLoader worker processes and collates on the
host; here the loader draws the batch's indices and random decisions on the host and the images
are born on the device (one kernel launch per view size), already in the layout the convolutions
read.  Batches have the reference's structure: (views, y) with views a list of [B,3,S,S] tensors
(or a single tensor for single-view transforms with `unwrap_single=True`)."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .store import WaferStore


class WaferMapDataset:
    def __init__(self, X, y=None, transform=None, device: Optional[str] = None):
        self.store = WaferStore(X if isinstance(X, WaferStore) else list(X), device=device)
        if y is not None:
            self.y = torch.as_tensor(np.asarray(list(y)))
        else:
            self.y = torch.zeros(len(self.store), dtype=torch.long)  # SSL does not need labels
        if len(self.y) != len(self.store):
            raise ValueError("X and y differ in length")
        self.transform = transform
        if device is not None:
            self.y = self.y.to(device)

    def to(self, device):
        self.store.to(device)
        self.y = self.y.to(device)
        return self

    def __len__(self):
        return len(self.store)

    def get_batch(self, indices: np.ndarray, rng: np.random.Generator, fmt: str = "nhwc_bf16"):
        if self.transform is None:
            raise ValueError("WaferMapDataset.get_batch needs a transform (images are built by it)")
        views = self.transform(self.store, np.asarray(indices), rng, fmt=fmt)
        idx = torch.as_tensor(np.asarray(indices), device=self.y.device)
        return views, self.y[idx]

    def __getitem__(self, index):
        views, y = self.get_batch(np.array([index]), np.random.default_rng())
        return [v[0] for v in views], y[0]


class WaferLoader:
    """DataLoader(dataset, batch_size, shuffle, drop_last) for a WaferMapDataset.  Deterministic:
    the permutation and every augmentation decision derive from (seed, epoch, batch index) for the
    GLOBAL batch; with `world_size > 1` each rank takes its contiguous slice of that batch, so the
    union over ranks is independent of the number of GPUs."""

    def __init__(self, dataset: WaferMapDataset, batch_size: int, shuffle: bool = False, drop_last: bool = False,
                 seed: int = 0, rank: int = 0, world_size: int = 1, fmt: str = "nhwc_bf16",
                 unwrap_single: bool = True):
        self.dataset, self.batch_size = dataset, int(batch_size)
        self.shuffle, self.drop_last, self.seed = shuffle, drop_last, seed
        self.rank, self.world_size, self.fmt, self.unwrap_single = rank, world_size, fmt, unwrap_single
        self.epoch = 0

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def __len__(self):
        gb = self.batch_size * self.world_size
        n = len(self.dataset)
        return n // gb if self.drop_last else (n + gb - 1) // gb

    def iter_indices(self):
        """(sample indices of this rank's slice, the batch's decision generator) per batch: what a captured
        training step needs (graph.GraphedTrainStep draws the decisions and produces the images itself).
        Same order, same slices and same generators as __iter__."""
        for mine, rng in self._batches():
            yield mine, rng

    def __iter__(self):
        return self.batches_in(0, len(self))

    def batches_in(self, lo: int, hi: int):
        """Batches lo <= index < hi of the epoch, in order: images are produced only for those (an evaluation pass
        sharded over the ranks by contiguous batch ranges: models/knn.py)."""
        for bi, (mine, rng) in enumerate(self._batches()):
            if bi < lo or bi >= hi:
                continue
            views, y = self.dataset.get_batch(mine, rng, fmt=self.fmt)
            if self.unwrap_single and len(views) == 1:
                yield views[0], y
            else:
                yield views, y

    def _batches(self):
        n = len(self.dataset)
        gb = self.batch_size * self.world_size
        order = np.random.default_rng([self.seed, self.epoch]).permutation(n) if self.shuffle else np.arange(n)
        for b in range(len(self)):
            glob = order[b * gb:(b + 1) * gb]
            if self.world_size > 1 and len(glob) < gb:
                # short final global batch (drop_last=False): every rank must still take part in the step and its
                # collectives, with equally many samples -- pad by wrapping around to the start of the epoch's
                # order (what torch's DistributedSampler does) and split evenly
                per = -(-len(glob) // self.world_size)
                glob = np.concatenate([glob, order[: per * self.world_size - len(glob)]])
                mine = glob[self.rank * per:(self.rank + 1) * per]
            else:
                mine = glob[self.rank * self.batch_size:(self.rank + 1) * self.batch_size]
            rng = np.random.default_rng([self.seed, self.epoch, b, self.rank])
            yield mine, rng


class WaferCollateLoader:
    """DataLoader(dataset, batch_size, shuffle, collate_fn=..., drop_last) as scripts/MixedWM38_pretrain.py:117-135
    builds it: the dataset carries NO transform, the collate function (transforms/collate.py: Wafer*CollateFunction)
    turns a list of (image, label, filename) items into `(views, labels, fnames)`.  Items here are (store index,
    label, "index"): the images live in the GPU-resident store the collate function is bound to."""

    def __init__(self, dataset: WaferMapDataset, batch_size: int, collate_fn, shuffle: bool = False,
                 drop_last: bool = False, seed: int = 0, rank: int = 0, world_size: int = 1, fmt: str = "nhwc_bf16"):
        self.dataset, self.collate_fn = dataset, collate_fn
        self._index = WaferLoader(dataset, batch_size, shuffle=shuffle, drop_last=drop_last, seed=seed, rank=rank,
                                  world_size=world_size, fmt=fmt)
        self.batch_size, self.drop_last, self.fmt = int(batch_size), drop_last, fmt
        self._labels = dataset.y.cpu().numpy()

    def set_epoch(self, epoch: int):
        self._index.set_epoch(epoch)

    def __len__(self):
        return len(self._index)

    def __iter__(self):
        for mine, rng in self._index.iter_indices():
            self.collate_fn.bind(self.dataset.store, rng, fmt=self.fmt)
            yield self.collate_fn([(int(i), self._labels[i], str(int(i))) for i in mine])
