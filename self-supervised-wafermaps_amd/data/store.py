"""WaferStore: the ragged wafer-map collection as flat, HBM-resident arrays.

The reference keeps a Python list of variable-size uint8 tensors (src/ssl_wafermap/data/dataset.py:
18-21) and augments one item at a time in DataLoader workers.  Here the whole collection is one
uint8 byte string + offsets/heights/widths on the device (all of WM-811K is ~1.2 GB, a rounding
error in 288 GB of HBM), which the augmentation kernel indexes directly.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch


class WaferStore:
    def __init__(self, wafers: Sequence, device: Optional[torch.device] = None):
        if isinstance(wafers, WaferStore):  # share the host arrays of an existing store
            for k in ("heights_np", "widths_np", "offsets_np", "bytes_np", "max_elems", "n"):
                setattr(self, k, getattr(wafers, k))
            self.device = None
            self.bytes = self.offsets = self.heights = self.widths = None
            if device is not None:
                self.to(device)
            return
        arrays = [np.ascontiguousarray(np.asarray(w), dtype=np.uint8) for w in wafers]
        if not arrays:
            raise ValueError("WaferStore: empty collection")
        for a in arrays:
            if a.ndim != 2 or a.shape[0] < 1 or a.shape[1] < 1 or max(a.shape) > 256:
                raise ValueError(f"WaferStore: wafer maps must be 2-D with sides in [1, 256], got {a.shape}")
        self.heights_np = np.array([a.shape[0] for a in arrays], dtype=np.int32)
        self.widths_np = np.array([a.shape[1] for a in arrays], dtype=np.int32)
        sizes = self.heights_np.astype(np.int64) * self.widths_np.astype(np.int64)
        self.offsets_np = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        self.bytes_np = np.concatenate([a.reshape(-1) for a in arrays])
        self.max_elems = int(sizes.max())
        self.n = len(arrays)
        self.device = None
        self.bytes = self.offsets = self.heights = self.widths = None
        if device is not None:
            self.to(device)

    def to(self, device) -> "WaferStore":
        device = torch.device(device)
        self.bytes = torch.from_numpy(self.bytes_np).to(device)
        self.offsets = torch.from_numpy(self.offsets_np).to(device)
        self.heights = torch.from_numpy(self.heights_np).to(device)
        self.widths = torch.from_numpy(self.widths_np).to(device)
        self.device = device
        return self

    def __len__(self) -> int:
        return self.n

    def wafer(self, i: int) -> np.ndarray:
        h, w, o = int(self.heights_np[i]), int(self.widths_np[i]), int(self.offsets_np[i])
        return self.bytes_np[o : o + h * w].reshape(h, w)

    def subset(self, indices, device: Optional[torch.device] = None) -> "WaferStore":
        """A new store holding wafers `indices` in that order (train / validation splits of one file)."""
        idx = np.asarray(indices, dtype=np.int64)
        if idx.ndim != 1 or len(idx) == 0 or idx.min() < 0 or idx.max() >= self.n:
            raise IndexError("WaferStore.subset: indices outside the store")
        out = WaferStore.__new__(WaferStore)
        out.heights_np, out.widths_np = self.heights_np[idx].copy(), self.widths_np[idx].copy()
        sizes = out.heights_np.astype(np.int64) * out.widths_np.astype(np.int64)
        out.offsets_np = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        out.bytes_np = np.concatenate([self.bytes_np[o:o + s] for o, s in zip(self.offsets_np[idx], sizes)])
        out.max_elems, out.n = int(sizes.max()), len(idx)
        out.device = None
        out.bytes = out.offsets = out.heights = out.widths = None
        if device is not None:
            out.to(device)
        return out

    def nbytes(self) -> int:
        return int(self.bytes_np.nbytes)

    # ---- on-disk form: the flat arrays themselves, so loading is four reads and no
    # per-wafer Python work (the reference unpickles a pandas Series of 2-D arrays: *.pkl.xz)
    def save(self, path, labels=None) -> None:
        extra = {} if labels is None else {"labels": np.asarray(labels)}
        np.savez(path, bytes=self.bytes_np, offsets=self.offsets_np, heights=self.heights_np, widths=self.widths_np,
                 **extra)

    @classmethod
    def load(cls, path, device: Optional[torch.device] = None):
        """-> (store, labels or None) from a file written by save()."""
        with np.load(path) as z:
            self = cls.__new__(cls)
            self.bytes_np = np.ascontiguousarray(z["bytes"], dtype=np.uint8)
            self.offsets_np = z["offsets"].astype(np.int64)
            self.heights_np = z["heights"].astype(np.int32)
            self.widths_np = z["widths"].astype(np.int32)
            labels = z["labels"] if "labels" in z.files else None
        sizes = self.heights_np.astype(np.int64) * self.widths_np.astype(np.int64)
        want = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64) if len(sizes) else np.zeros(0, np.int64)
        if (len(sizes) == 0 or not np.array_equal(want, self.offsets_np) or int(sizes.sum()) != self.bytes_np.size
                or sizes.min() < 1 or max(self.heights_np.max(), self.widths_np.max()) > 256):
            raise ValueError(f"{path}: not a consistent wafer store")
        self.max_elems = int(sizes.max())
        self.n = len(sizes)
        self.device = None
        self.bytes = self.offsets = self.heights = self.widths = None
        if device is not None:
            self.to(device)
        return self, labels
