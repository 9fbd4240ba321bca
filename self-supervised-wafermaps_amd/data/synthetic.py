"""Seeded synthetic wafer maps with WM-811K-like geometry: ragged uint8 maps with
values {0 background, 128 pass, 255 fail}; H, W ~ round(clip(LogNormal(ln 30, 0.35), 22, 212)) drawn
independently (about 80 % non-square); dies inside the inscribed ellipse fail with a per-wafer rate
f ~ U(0.02, 0.3); labels from the WM-811K class prior."""
from __future__ import annotations

import numpy as np

WM811K_PRIOR = np.array([0.069, 0.009, 0.083, 0.156, 0.058, 0.0024, 0.014, 0.019, 0.59])


def synthetic_wafers(n: int, seed: int = 1234, fixed_size: int | None = None):
    rng = np.random.default_rng(seed)
    wafers = []
    for _ in range(n):
        if fixed_size:
            h = w = fixed_size
        else:
            h, w = (int(round(float(np.clip(rng.lognormal(np.log(30.0), 0.35), 22, 212)))) for _ in range(2))
        yy, xx = np.mgrid[0:h, 0:w]
        inside = ((yy + 0.5 - h / 2) / (h / 2)) ** 2 + ((xx + 0.5 - w / 2) / (w / 2)) ** 2 <= 1.0
        f = rng.uniform(0.02, 0.3)
        fail = rng.random((h, w)) < f
        wafers.append(np.where(inside, np.where(fail, 255, 128), 0).astype(np.uint8))
    labels = rng.choice(9, size=n, p=WM811K_PRIOR / WM811K_PRIOR.sum()).astype(np.int64)
    return wafers, labels
