"""Convert the data files the reference checkout HOLDS into data-only fixtures under tests/golden/.

Run once in the build container (needs /root/reference; never runs on the GPU box):

    python tests/golden/make_reference_data.py

Inputs (reference checkout)                                        -> fixture (arrays only, LZMA-zipped .npz)
  data/processed/WM811K/train_1_split.pkl.xz    623 wafers + failureCode     -> wm811k_train_1_split.npz
  data/processed/WM811K/train_20_split.pkl.xz   12 449 wafers + failureCode  -> wm811k_train_20_split.npz
      (the file the reference's dummy mode trains on: scripts/WM811k_benchmark.py:87-97)
  data/interim/model_preds/SimSiam_preds_subset.pkl.xz  12 449 x 512 float16 backbone features of the SAME wafers
      (row order and failureCode identical to train_20_split) + failureCode  -> simsiam_preds_subset.npz
  data/processed/MixedWM38/train_1_split.pkl.xz  381 maps (52 x 52) + failureType codes + 8-bit labels -> mixedwm38_train_1_split.npz
  data/interim/model_logs/{loss,rep_std,accuracy,f1}/run-SimCLR-tag-*.csv  the reference's own SimCLR curves
      (Step, Value columns)                                                  -> simclr_reference_curves.npz

The wafer fixtures use WaferStore.save()'s keys (bytes, offsets, heights, widths, labels), so
`WaferStore.load(path)` reads them; np.load opens LZMA-compressed zip members transparently.
"""
import io
import zipfile
from pathlib import Path

import numpy as np
import pandas as pd

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def save_npz_lzma(path, **arrays):
    with zipfile.ZipFile(path, "w", compression=zipfile.ZIP_LZMA) as z:
        for k, v in arrays.items():
            b = io.BytesIO()
            np.lib.format.write_array(b, np.ascontiguousarray(v), allow_pickle=False)
            z.writestr(k + ".npy", b.getvalue())


def wafers(name):
    df = pd.read_pickle(REF / "data/processed/WM811K" / f"{name}.pkl.xz")
    maps = [np.ascontiguousarray(w, dtype=np.uint8) for w in df.waferMap]
    h = np.array([m.shape[0] for m in maps], dtype=np.int32)
    w = np.array([m.shape[1] for m in maps], dtype=np.int32)
    sizes = h.astype(np.int64) * w
    save_npz_lzma(OUT / f"wm811k_{name}.npz", bytes=np.concatenate([m.reshape(-1) for m in maps]),
                  offsets=np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64), heights=h, widths=w,
                  labels=df.failureCode.to_numpy().astype(np.int8))
    return df


def mixed(name):
    """MixedWM38 split: 52 x 52 maps + the 8-bit multi-label vector; `failureType` strings factorised as the
    reference does for its (unused) SSL labels (scripts/MixedWM38_pretrain.py:89-91)."""
    df = pd.read_pickle(REF / "data/processed/MixedWM38" / f"{name}.pkl.xz")
    maps = [np.ascontiguousarray(w, dtype=np.uint8) for w in df.waferMap]
    h = np.array([m.shape[0] for m in maps], dtype=np.int32)
    w = np.array([m.shape[1] for m in maps], dtype=np.int32)
    sizes = h.astype(np.int64) * w
    save_npz_lzma(OUT / f"mixedwm38_{name}.npz", bytes=np.concatenate([m.reshape(-1) for m in maps]),
                  offsets=np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64), heights=h, widths=w,
                  labels=df.failureType.factorize(sort=True)[0].astype(np.int64),
                  multilabel=np.stack([np.asarray(v) for v in df.label]).astype(np.int8))


def main():
    mixed("train_1_split")
    wafers("train_1_split")
    t20 = wafers("train_20_split")
    preds = pd.read_pickle(REF / "data/interim/model_preds/SimSiam_preds_subset.pkl.xz")
    emb = preds[list(range(512))].to_numpy().astype(np.float16)
    assert emb.shape == (12449, 512)
    assert np.array_equal(preds.failureCode.to_numpy(), t20.failureCode.to_numpy())
    assert all(np.array_equal(a, b) for a, b in zip(preds.waferMap, t20.waferMap))
    save_npz_lzma(OUT / "simsiam_preds_subset.npz", embeddings=emb, labels=preds.failureCode.to_numpy().astype(np.int8))
    curves = {}
    for tag, sub in (("train_loss_ssl", "loss"), ("rep_std", "rep_std"), ("knn_accuracy", "accuracy"), ("knn_f1", "f1")):
        c = pd.read_csv(REF / "data/interim/model_logs" / sub / f"run-SimCLR-tag-{tag}.csv")
        curves[tag + "_step"] = c["Step"].to_numpy().astype(np.int64)
        curves[tag + "_value"] = c["Value"].to_numpy().astype(np.float64)
    save_npz_lzma(OUT / "simclr_reference_curves.npz", **curves)
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size)


if __name__ == "__main__":
    main()
