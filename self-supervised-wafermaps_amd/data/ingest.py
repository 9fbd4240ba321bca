"""Ingest of the reference's processed data files.

The reference reads `pd.read_pickle("../data/processed/WM811K/train_data.pkl.xz")` and takes the
columns `waferMap` (a Series of 2-D uint8 arrays with values {0, 128, 255}) and `failureCode`
(scripts/WM811k_benchmark.py:87-104); MixedWM38 files carry `waferMap` and a multi-label `label`
column (scripts/MixedWM38_pretrain.py:60-75).  `read_wafer_pickle` turns such a file into the flat
WaferStore the augmentation kernel indexes directly, `convert_pickle` writes the flat `.npz` form once so
later runs skip pandas and the per-wafer unpickling.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from .store import WaferStore


def read_wafer_pickle(path, wafer_col: str = "waferMap", label_col: Optional[str] = "failureCode",
                      device=None) -> Tuple[WaferStore, Optional[np.ndarray]]:
    import pandas as pd

    df = pd.read_pickle(path)
    if wafer_col not in df.columns:
        raise KeyError(f"{path}: no column {wafer_col!r} (columns: {list(df.columns)})")
    store = WaferStore(df[wafer_col].tolist(), device=device)
    labels = None
    if label_col is not None:
        if label_col not in df.columns:
            raise KeyError(f"{path}: no column {label_col!r} (columns: {list(df.columns)})")
        col = df[label_col]
        first = col.iloc[0]
        labels = np.stack([np.asarray(v) for v in col]) if np.ndim(first) > 0 else col.to_numpy()
    return store, labels


def convert_pickle(src, dst, wafer_col: str = "waferMap", label_col: Optional[str] = "failureCode") -> WaferStore:
    store, labels = read_wafer_pickle(src, wafer_col, label_col)
    store.save(dst, labels)
    return store
