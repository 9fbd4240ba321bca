"""Multi-view transforms with the reference's names and constructor signatures:
BaseViewTransform (src/ssl_wafermap/transforms/wafer_base_transform.py:8-59) and
MultiCropTransform (src/ssl_wafermap/transforms/wafer_multicrop_transform.py:88-171), both
lightly MultiViewTransforms = a list of per-view pipelines.  Here a pipeline is a ViewSpec and a
whole batch of every view is produced by one kernel launch per output size."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch

from ..data.store import WaferStore
from .augmentations import (ViewSpec, augment_views, get_base_transforms, get_inference_transforms, multicrop_view,
                            sample_view_params)
from .utils import NORMALIZE_STATS


class Views(list):
    """List of per-view batch tensors; `.stacked` is the same memory as one [V*B, ...] tensor when
    all views share a size (lets the model run every view in one pass without a concat copy)."""

    stacked = None

    def __init__(self, *a):
        super().__init__(*a)
        self.stacked_groups = {}  # (first view, one past last) -> the launch buffer those views slice


class MultiViewTransform:
    def __init__(self, transforms: Sequence[ViewSpec]):
        self.transforms = list(transforms)

    def groups(self):
        """Runs of consecutive views that share a launch: [(first view, one past last)]."""
        out, i = [], 0
        while i < len(self.transforms):
            spec = self.transforms[i]
            key = (spec.img_size, spec.out_size, spec.normalize)
            j = i
            while j < len(self.transforms) and (self.transforms[j].img_size, self.transforms[j].out_size,
                                                  self.transforms[j].normalize) == key:
                j += 1
            out.append((i, j))
            i = j
        return out

    def sample(self, store: WaferStore, sample_idx: np.ndarray, rng: np.random.Generator):
        """Every random decision of the batch: one PARAM_DTYPE array per launch group."""
        n = len(sample_idx)
        return [np.concatenate([sample_view_params(self.transforms[v], sample_idx, store.heights_np, store.widths_np,
                                                   rng, out_slot_base=(v - i) * n) for v in range(i, j)])
                for i, j in self.groups()]

    def launch(self, store: WaferStore, params_per_group, n: int, fmt: str = "nhwc_bf16", params_dev=None) -> Views:
        """Run the kernels for already-sampled decisions.  `params_dev`: optional list of static
        device buffers already holding the parameters (hipGraph capture)."""
        out = Views()
        groups = self.groups()
        for gi, ((i, j), params) in enumerate(zip(groups, params_per_group)):
            spec = self.transforms[i]
            batch = augment_views(store, params, img_size=spec.img_size, out_size=spec.out_size, fmt=fmt,
                                  normalize=spec.normalize, mean=NORMALIZE_STATS["mean"][0],
                                  std=NORMALIZE_STATS["std"][0], n_slots=(j - i) * n,
                                  params_dev=None if params_dev is None else params_dev[gi])
            for v in range(j - i):
                out.append(batch[v * n:(v + 1) * n])
            out.stacked_groups[(i, j)] = batch  # views i..j-1 are slices of this one buffer
            if len(groups) == 1:
                out.stacked = batch
        return out

    def __call__(self, store: WaferStore, sample_idx: np.ndarray, rng: np.random.Generator,
                 fmt: str = "nhwc_bf16") -> Views:
        return self.launch(store, self.sample(store, np.asarray(sample_idx), rng), len(sample_idx), fmt)


class BaseViewTransform(MultiViewTransform):
    def __init__(self, img_size: List[int] = [224, 224], die_noise_prob: float = 0.03, crop: bool = False,
                 denoise: bool = False, hf_prob: float = 0.5, vf_prob: float = 0.5, rr_prob: float = 0.5,
                 normalize: bool = True, n_views: int = 2):
        assert n_views > 0, "n_views must be greater than 0"
        view = get_base_transforms(img_size=img_size, die_noise_prob=die_noise_prob, denoise=denoise, crop=crop,
                                   hf_prob=hf_prob, vf_prob=vf_prob, rr_prob=rr_prob, to_tensor=True,
                                   normalize=normalize)
        super().__init__([view] * n_views)


class MultiCropTransform(MultiViewTransform):
    def __init__(self, img_size: List[int] = [224, 224], global_crop_size: int = 224,
                 global_crop_scale: Tuple[float, float] = (0.6, 1.0), local_crop_size: int = 96,
                 local_crop_scale: Tuple[float, float] = (0.1, 0.4), n_global_views: int = 2, n_local_views: int = 6,
                 die_noise_prob: float = 0.03, denoise: bool = False, hf_prob: float = 0.5, vf_prob: float = 0.5,
                 rr_prob: float = 0.5, normalize: bool = True):
        assert n_global_views > 0, "n_global_views must be greater than 0"
        assert n_local_views > 0, "n_local_views must be greater than 0"
        kw = dict(img_size=img_size, die_noise_prob=die_noise_prob, denoise=denoise, hf_prob=hf_prob,
                  vf_prob=vf_prob, rr_prob=rr_prob, normalize=normalize)
        g = multicrop_view(crop_size=global_crop_size, crop_scale=global_crop_scale, **kw)
        l = multicrop_view(crop_size=local_crop_size, crop_scale=local_crop_scale, **kw)
        super().__init__([g] * n_global_views + [l] * n_local_views)


class InferenceTransform(MultiViewTransform):
    """get_inference_transforms as a one-view transform (the kNN bank / validation loaders)."""

    def __init__(self, img_size: List[int] = [224, 224], normalize: bool = True):
        super().__init__([get_inference_transforms(img_size, normalize)])
