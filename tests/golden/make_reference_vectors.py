"""Generate tests/golden/reference_aug_vectors.npz by RUNNING the reference's own augmentation code.

Run once in the build container (needs /root/reference; never runs on the GPU box):

    python tests/golden/make_reference_vectors.py

`ssl_wafermap.transforms.augmentations` imports cv2, torchvision and lightly at module top; none is
installed here.  The functions exercised below (DieNoise, DPWTransform.*, RandomOneOf) use only
torch / numpy / random, so the missing imports are satisfied by EMPTY placeholder modules whose
attributes are never called on these code paths; the arithmetic that runs is the reference's,
unmodified.  MedianFilter (cv2) and the torchvision/PIL compose are NOT exercised (they cannot run).

The output holds data only (inputs, captured random draws, expected outputs).
"""
import random
import sys
import types
from pathlib import Path

import numpy as np
import pandas as pd
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent / "reference_aug_vectors.npz"


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference_augmentations():
    class _Inert:  # attribute sink: anything looked up on it is another inert object
        def __getattr__(self, k):
            return _Inert()

        def __call__(self, *a, **k):
            raise RuntimeError("placeholder module called: this code path needs the real dependency")

    _placeholder("cv2")
    tv = _placeholder("torchvision")
    tvt = _placeholder("torchvision.transforms", Compose=_Inert(), ToPILImage=_Inert())
    tvf = _placeholder("torchvision.transforms.functional", InterpolationMode=_Inert())
    tv.transforms = tvt
    tvt.functional = tvf
    _placeholder("lightly")
    _placeholder("lightly.transforms")
    _placeholder("lightly.transforms.rotation", RandomRotate=_Inert())
    _placeholder("lightly.transforms.multi_view_transform", MultiViewTransform=object)
    sys.path.insert(0, str(REF / "src"))
    import ssl_wafermap.transforms.augmentations as aug  # noqa: E402

    return aug


def main():
    aug = import_reference_augmentations()
    df = pd.read_pickle(REF / "data/processed/WM811K/train_1_split.pkl.xz")
    dm = pd.read_pickle(REF / "data/processed/MixedWM38/train_1_split.pkl.xz")
    # a spread of shapes: small/large/non-square WM-811K wafers + fixed 52x52 MixedWM38
    shapes = np.array([w.shape for w in df.waferMap])
    order = np.argsort(shapes[:, 0] * 1000 + shapes[:, 1])
    pick = order[np.linspace(0, len(order) - 1, 20).astype(int)]
    wafers = [np.array(df.waferMap.iloc[i], dtype=np.uint8) for i in pick]
    wafers += [np.array(dm.waferMap.iloc[i], dtype=np.uint8) for i in (0, 7, 19, 42)]

    store = {}

    def put_ragged(key, arrays):
        store[key + "_data"] = np.concatenate([a.reshape(-1) for a in arrays])
        store[key + "_shape"] = np.array([a.shape for a in arrays], dtype=np.int32)

    put_ragged("wafer", wafers)

    # ---- DieNoise: seed -> (rand field the call draws, output)
    dn_rand, dn_out, dn_p = [], [], []
    for i, w in enumerate(wafers):
        p = [0.03, 0.1, 0.5][i % 3]
        torch.manual_seed(1000 + i)
        rand = torch.rand(*w.shape).numpy().copy()
        torch.manual_seed(1000 + i)
        out = aug.DieNoise(p)(torch.tensor(w.copy())).numpy()
        dn_rand.append(rand.astype(np.float32))
        dn_out.append(out.astype(np.uint8))
        dn_p.append(p)
    put_ragged("dienoise_rand", dn_rand)
    put_ragged("dienoise_out", dn_out)
    store["dienoise_p"] = np.array(dn_p, dtype=np.float64)

    # ---- power law table
    xs = np.arange(15, 231)
    store["powerlaw_x"] = xs
    store["powerlaw_y"] = np.array(
        [aug.DPWTransform.power_law_transform(int(x), 26, 212, 0.4, 0.95, 5.0) for x in xs], dtype=np.float64
    )

    # ---- dpw_transform at fixed scales
    scales = [0.4, 0.55, 0.7, 0.95, 1.0]
    dpw_out = []
    for w in wafers:
        for s in scales:
            dpw_out.append(aug.DPWTransform.dpw_transform(torch.tensor(w.copy()), s).numpy().astype(np.uint8))
    put_ragged("dpw_out", dpw_out)
    store["dpw_scales"] = np.array(scales, dtype=np.float64)

    # ---- DPWTransform.__call__: numpy seed -> (beta draw, output)
    call_beta, call_out = [], []
    t = aug.DPWTransform()
    for i, w in enumerate(wafers):
        np.random.seed(2000 + i)
        b = np.random.beta(0.5, 1.5)
        np.random.seed(2000 + i)
        call_out.append(t(torch.tensor(w.copy())).numpy().astype(np.uint8))
        call_beta.append(b)
    put_ragged("dpwcall_out", call_out)
    store["dpwcall_beta"] = np.array(call_beta, dtype=np.float64)

    # ---- RandomOneOf: python seed -> (uniform used by random.choices, chosen index)
    class _Tag:
        def __init__(self, k):
            self.k = k

        def __call__(self, img):
            return self.k

    for name, weights in (("uniform2", None), ("w3", [0.2, 0.5, 0.3])):
        n = 2 if weights is None else 3
        roo = aug.RandomOneOf([_Tag(k) for k in range(n)], weights=weights)
        us, ks = [], []
        for seed in range(3):
            random.seed(seed)
            draws = [random.random() for _ in range(64)]  # call i uses draws 2i (p test), 2i+1 (choice)
            random.seed(seed)
            for c in range(32):
                ks.append(roo(None))
                us.append(draws[2 * c + 1])
        store[f"oneof_{name}_u"] = np.array(us, dtype=np.float64)
        store[f"oneof_{name}_k"] = np.array(ks, dtype=np.int64)

    np.savez_compressed(OUT, **store)
    print("wrote", OUT, OUT.stat().st_size, "bytes;", len(wafers), "wafers")


if __name__ == "__main__":
    main()
