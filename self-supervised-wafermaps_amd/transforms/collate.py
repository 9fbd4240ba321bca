"""Collate functions of the MixedWM38 pre-training script (reference
src/ssl_wafermap/utilities/transforms.py:401-863, selected at scripts/MixedWM38_pretrain.py:106-135).

In the reference a collate function receives a list of (image, label, filename) tuples from a
LightlyDataset and augments every image on the CPU inside the DataLoader worker.  Here the images
live in a GPU-resident WaferStore, so a batch item is (store index, label, filename); the class
keeps the reference's constructor signature, holds the equivalent MultiViewTransform and its
`forward(batch)` returns the reference's triple `(views, labels, fnames)` with the views produced
by the fused augmentation kernel.  `bind(store, rng)` attaches the store (WaferLoader does it).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch

from .augmentations import get_base_transforms, multicrop_view
from .views import MultiViewTransform


class MultiViewCollateFunction:
    def __init__(self, transforms):
        self.transform = MultiViewTransform(transforms)
        self.store = None
        self.rng = np.random.default_rng()
        self.fmt = "nhwc_bf16"

    def bind(self, store, rng=None, fmt: str = "nhwc_bf16"):
        self.store, self.fmt = store, fmt
        if rng is not None:
            self.rng = rng
        return self

    def forward(self, batch: List[tuple]):
        if self.store is None:
            raise RuntimeError("collate function is not bound to a WaferStore: call bind(store) first")
        idx = np.array([int(item[0]) for item in batch], dtype=np.int64)
        labels = torch.as_tensor(np.array([item[1] for item in batch]))
        fnames = [item[2] for item in batch]
        views = self.transform(self.store, idx, self.rng, fmt=self.fmt)
        return views, labels, fnames

    __call__ = forward


class BaseCollateFunction(MultiViewCollateFunction):
    """lightly's BaseCollateFunction: the same transform twice -> two views."""

    def __init__(self, transform):
        super().__init__([transform, transform])

    def forward(self, batch):
        views, labels, fnames = super().forward(batch)
        return (views[0], views[1]), labels, fnames

    __call__ = forward


class WaferImageCollateFunction(BaseCollateFunction):
    """Generic joint-embedding collate (SimCLR, MoCo, BYOL, ...): reference :401-441."""

    def __init__(self, img_size: List[int] = [224, 224], die_noise_prob: float = 0.03, crop: bool = False,
                 denoise: bool = False, hf_prob: float = 0.5, vf_prob: float = 0.5, rr_prob: float = 0.5,
                 normalize: bool = True):
        super().__init__(get_base_transforms(img_size=img_size, die_noise_prob=die_noise_prob, denoise=denoise,
                                             crop=crop, hf_prob=hf_prob, vf_prob=vf_prob, rr_prob=rr_prob,
                                             to_tensor=True, normalize=normalize))


class WaferDINOCOllateFunction(MultiViewCollateFunction):
    """2 global (224, scale 0.6-1.0) + n local (96, scale 0.1-0.4) crops: reference :486-577."""

    def __init__(self, global_crop_size: int = 224, global_crop_scale: Tuple[float, float] = (0.6, 1.0),
                 local_crop_size: int = 96, local_crop_scale: Tuple[float, float] = (0.1, 0.4), n_local_views: int = 6,
                 die_noise_prob: float = 0.03, denoise: bool = False, hf_prob: float = 0.5, vf_prob: float = 0.5,
                 rr_prob: float = 0.5):
        kw = dict(img_size=[global_crop_size, global_crop_size], die_noise_prob=die_noise_prob, denoise=denoise,
                  hf_prob=hf_prob, vf_prob=vf_prob, rr_prob=rr_prob)
        g = multicrop_view(crop_size=global_crop_size, crop_scale=global_crop_scale, **kw)
        l = multicrop_view(crop_size=local_crop_size, crop_scale=local_crop_scale, **kw)
        super().__init__([g] * 2 + [l] * n_local_views)


class WaferMAECollateFunction2(MultiViewCollateFunction):
    """One augmented view (MAE needs a single view): reference :711-742."""

    def __init__(self, img_size: List[int] = [224, 224], die_noise_prob: float = 0.03, denoise: bool = False,
                 crop: bool = False, hf_prob: float = 0.5, vf_prob: float = 0.5, rr_prob: float = 0.5,
                 normalize: bool = True):
        super().__init__([get_base_transforms(img_size=img_size, die_noise_prob=die_noise_prob, denoise=denoise,
                                              crop=crop, hf_prob=hf_prob, vf_prob=vf_prob, rr_prob=rr_prob,
                                              to_tensor=True, normalize=normalize)])

    def forward(self, batch):
        views, labels, fnames = super().forward(batch)
        return views[0], labels, fnames

    __call__ = forward


class WaferMSNCollateFunction(MultiViewCollateFunction):
    """random_views global + focal_views focal crops, no vertical flip by default: reference :580-670."""

    def __init__(self, random_size: int = 224, focal_size: int = 96, random_views: int = 2, focal_views: int = 10,
                 random_crop_scale: Tuple[float, float] = (0.6, 1.0), focal_crop_scale: Tuple[float, float] = (0.1, 0.4),
                 die_noise_prob: float = 0.03, denoise: bool = False, hf_prob: float = 0.5, vf_prob: float = 0.0,
                 rr_prob: float = 0.5):
        kw = dict(img_size=[random_size, random_size], die_noise_prob=die_noise_prob, denoise=denoise, hf_prob=hf_prob,
                  vf_prob=vf_prob, rr_prob=rr_prob)
        r = multicrop_view(crop_size=random_size, crop_scale=random_crop_scale, **kw)
        f = multicrop_view(crop_size=focal_size, crop_scale=focal_crop_scale, **kw)
        super().__init__([r] * random_views + [f] * focal_views)


class WaferSwaVCollateFunction(MultiViewCollateFunction):
    """Multi-crop by category (sizes, counts, min/max scales): reference :810-863."""

    def __init__(self, crop_sizes: List[int] = [224, 96], crop_counts: List[int] = [2, 6],
                 crop_min_scales: List[float] = [0.6, 0.1], crop_max_scales: List[float] = [1.0, 0.4],
                 die_noise_prob: float = 0.03, denoise: bool = False, hf_prob: float = 0.5, vf_prob: float = 0.5,
                 rr_prob: float = 0.5):
        if not (len(crop_sizes) == len(crop_counts) == len(crop_min_scales) == len(crop_max_scales)):
            raise ValueError("Length of crop_sizes, crop_counts, crop_min_scales and crop_max_scales must be equal")
        kw = dict(img_size=[crop_sizes[0], crop_sizes[0]], die_noise_prob=die_noise_prob, denoise=denoise,
                  hf_prob=hf_prob, vf_prob=vf_prob, rr_prob=rr_prob)
        specs = []
        for size, count, lo, hi in zip(crop_sizes, crop_counts, crop_min_scales, crop_max_scales):
            specs += [multicrop_view(crop_size=size, crop_scale=(lo, hi), **kw)] * count
        super().__init__(specs)


def rgb_scale(X, feature_range=[0, 255], data_range=None):
    """Scales an array to the RGB domain [0, 255] (reference :890-910; host-side data preparation)."""
    X = np.asarray(X)
    if data_range is None:
        data_range = [np.min(X), np.max(X)]
    if feature_range is None:
        feature_range = [np.min(X), np.max(X)]
    data_min, data_max = data_range
    feature_min, feature_max = feature_range
    X_std = (X - data_min) / (data_max - data_min)
    return np.round(X_std * (feature_max - feature_min) + feature_min).astype(np.uint8)
