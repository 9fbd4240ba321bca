"""Host side of the fused augmentation: the reference's transform vocabulary as *specifications*.

The reference composes per-sample callables (src/ssl_wafermap/transforms/augmentations.py).  Here
the same names build a `ViewSpec`; all random choices of a batch are drawn on the host in one
vectorised pass (`sample_view_params`) and executed by ONE kernel launch per output size
(`augment_views` -> wm_augment_views).  Decisions are explicit data, so the CPU oracle can replay
them bit-for-bit.
"""
from __future__ import annotations

from dataclasses import dataclass, replace
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib
from .._lib import check, ptr, stream_ptr
from ..data.store import WaferStore
from .utils import NORMALIZE_STATS

PARAM_DTYPE = np.dtype(
    [("sample", "<i4"), ("out_slot", "<i4"), ("op", "<i4"), ("noise_seed", "<u4"), ("noise_p", "<f4"),
     ("dpw_h", "<i4"), ("dpw_w", "<i4"), ("rot90", "<i4"), ("vflip", "<i4"), ("hflip", "<i4"),
     ("crop", "<i4"), ("crop_i", "<i4"), ("crop_j", "<i4"), ("crop_h", "<i4"), ("crop_w", "<i4"),
     ("reserved", "<i4")]
)
assert PARAM_DTYPE.itemsize == 64


_POWER_LAW_TABLES = {}


# ---- the reference's stage-1 transform names, as parameter holders -------------------------------
@dataclass(frozen=True)
class DieNoise:
    """Flip pass<->fail dies with probability p (reference augmentations.py:14-36)."""

    p: float = 0.03


@dataclass(frozen=True)
class MedianFilter:
    """3x3 median (reference augmentations.py:90-107; only kernel_size 3 is implemented)."""

    kernel_size: int = 3

    def __post_init__(self):
        if self.kernel_size != 3:
            raise NotImplementedError("MedianFilter: only kernel_size=3 has a HIP path")


@dataclass(frozen=True)
class DPWTransform:
    """Lower-die-per-wafer resampling (reference augmentations.py:110-250)."""

    domain_lower: int = 26
    domain_upper: int = 212
    out_lower: float = 0.4
    out_upper: float = 0.95
    alpha: float = 0.5
    beta: float = 1.5
    p: float = 5.0

    def _power_law_one(self, x: int) -> float:
        # plain Python float arithmetic, operation for operation as the reference does it
        # (augmentations.py:152-174): numpy's vectorised pow differs in the last bit for some x
        if x <= self.domain_lower:
            return self.out_upper
        if x >= self.domain_upper:
            return self.out_lower
        domain_range = self.domain_upper - self.domain_lower
        normalized_x = abs(x - self.domain_lower) / domain_range
        y = (1 - normalized_x) ** self.p
        return self.out_lower + y * (self.out_upper - self.out_lower)

    def power_law(self, x: np.ndarray) -> np.ndarray:
        x = np.asarray(x, dtype=np.int64)
        table = _POWER_LAW_TABLES.get(self)
        if table is None:
            table = np.array([self._power_law_one(int(v)) for v in range(0, 513)], dtype=np.float64)
            _POWER_LAW_TABLES[self] = table
        return table[np.clip(x, 0, 512)]

    def scales(self, heights, widths, beta_draws) -> np.ndarray:
        lower = self.power_law(np.maximum(heights, widths))
        return lower + (self.out_upper - lower) * beta_draws


@dataclass(frozen=True)
class RandomOneOf:
    """Apply one of `transforms` (reference augmentations.py:42-87)."""

    transforms: Tuple
    weights: Optional[Tuple[float, ...]] = None
    p: float = 1.0

    def __post_init__(self):
        n = len(self.transforms)
        if self.weights is not None:
            if len(self.weights) != n:
                raise ValueError("The number of weights must match the number of transforms")
            if not all(w >= 0 for w in self.weights):
                raise ValueError("Weights must be non-negative")
            if sum(self.weights) == 0:
                raise ValueError("At least one weight must be greater than 0")
        if self.p is not None and (self.p < 0 or self.p > 1):
            raise ValueError("p must be a float between 0 and 1")

    def normalized_weights(self) -> np.ndarray:
        n = len(self.transforms)
        if self.weights is None:
            return np.full(n, 1.0 / n)
        w = np.asarray(self.weights, dtype=np.float64)
        return w / w.sum()


@dataclass(frozen=True)
class ViewSpec:
    """One view pipeline = what get_base_transforms / get_inference_transforms /
    MultiCropViewTransform build in the reference."""

    img_size: int = 224
    stage1: Optional[RandomOneOf] = None      # None: inference transform (no stage 1, no flips)
    rr_prob: float = 0.0
    vf_prob: float = 0.0
    hf_prob: float = 0.0
    crop_scale: Optional[Tuple[float, float]] = None
    crop_prob: float = 0.0                    # 0.5 for crop=True (RandomApply), 1.0 for multi-crop
    out_size: int = 224
    normalize: bool = True

    # torchvision-Compose-like call on ONE wafer is deliberately not offered on the CPU: the product
    # path is the batched kernel (see WaferLoader / augment_views).


def _stage1_op_code(t) -> int:
    if isinstance(t, DieNoise):
        return _lib.WM_AUG_DIENOISE
    if isinstance(t, DPWTransform):
        return _lib.WM_AUG_DPW
    if isinstance(t, MedianFilter):
        return _lib.WM_AUG_MEDIAN3
    raise TypeError(f"unsupported stage-1 transform {t!r}")


def get_base_transforms(img_size: Sequence[int] = (224, 224), die_noise_prob: float = 0.03,
                        denoise: bool = False, crop: bool = False, rr_prob: float = 0.5,
                        hf_prob: float = 0.5, vf_prob: float = 0.5, to_tensor: bool = True,
                        normalize: bool = True, as_list: bool = False) -> ViewSpec:
    """Same signature as the reference (augmentations.py:253-264).  `to_tensor=False` (PIL output
    for a following crop) has no meaning on the device: the crop is part of the ViewSpec."""
    if img_size[0] != img_size[1]:
        raise NotImplementedError("only square img_size is implemented (the reference always uses 224x224)")
    stage1 = RandomOneOf((DieNoise(die_noise_prob), MedianFilter() if denoise else DPWTransform()))
    return ViewSpec(img_size=int(img_size[0]), stage1=stage1, rr_prob=rr_prob, vf_prob=vf_prob, hf_prob=hf_prob,
                    crop_scale=(0.4, 1.0) if crop else None, crop_prob=0.5 if crop else 0.0,
                    out_size=int(img_size[0]), normalize=normalize)


def get_inference_transforms(img_size: Sequence[int] = (224, 224), normalize: bool = True) -> ViewSpec:
    """Reference augmentations.py:335-357: resize + grayscale + tensor + normalise."""
    if img_size[0] != img_size[1]:
        raise NotImplementedError("only square img_size is implemented")
    return ViewSpec(img_size=int(img_size[0]), stage1=None, out_size=int(img_size[0]), normalize=normalize)


def multicrop_view(img_size=(224, 224), crop_size=224, crop_scale=(0.4, 1.0), die_noise_prob=0.03,
                   denoise=False, hf_prob=0.5, vf_prob=0.5, rr_prob=0.5, normalize=True) -> ViewSpec:
    """MultiCropViewTransform (reference wafer_multicrop_transform.py:16-85): base augment, then an
    unconditional RandomResizedCrop(crop_size, crop_scale, ratio (1,1), NEAREST)."""
    base = get_base_transforms(img_size, die_noise_prob, denoise, False, rr_prob, hf_prob, vf_prob,
                               normalize=normalize)
    return replace(base, crop_scale=tuple(crop_scale), crop_prob=1.0, out_size=int(crop_size))


# ---- batched decision sampling ------------------------------------------------------------------
def sample_view_params(spec: ViewSpec, sample_idx: np.ndarray, heights: np.ndarray, widths: np.ndarray,
                       rng: np.random.Generator, out_slot_base: int = 0) -> np.ndarray:
    """Draw every random choice of `len(sample_idx)` views of one ViewSpec."""
    sample_idx = np.asarray(sample_idx, dtype=np.int64)
    n = sample_idx.shape[0]
    h = heights[sample_idx].astype(np.int64)
    w = widths[sample_idx].astype(np.int64)
    p = np.zeros(n, dtype=PARAM_DTYPE)
    p["sample"] = sample_idx
    p["out_slot"] = out_slot_base + np.arange(n)
    p["dpw_h"], p["dpw_w"] = h, w
    S = spec.img_size
    if spec.stage1 is not None:
        one = spec.stage1
        applied = rng.random(n) < one.p                      # random.random() < self.p
        cum = np.cumsum(one.normalized_weights())
        u = rng.random(n)                                    # random.choices' uniform draw
        which = np.minimum(np.searchsorted(cum, u * cum[-1], side="right"), len(cum) - 1)
        codes = np.array([_stage1_op_code(t) for t in one.transforms], dtype=np.int32)
        p["op"] = np.where(applied, codes[which], _lib.WM_AUG_NONE)
        p["noise_seed"] = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
        for t in one.transforms:
            if isinstance(t, DieNoise):
                p["noise_p"] = np.float32(t.p)
            if isinstance(t, DPWTransform):
                scale = t.scales(h, w, rng.beta(t.alpha, t.beta, size=n))
                is_dpw = p["op"] == _lib.WM_AUG_DPW
                # int(h * scale) in double, like the reference (augmentations.py:201-202)
                p["dpw_h"] = np.where(is_dpw, np.floor(h * scale), h)
                p["dpw_w"] = np.where(is_dpw, np.floor(w * scale), w)
        if (p["dpw_h"] < 1).any() or (p["dpw_w"] < 1).any():
            raise ValueError("DPWTransform would produce an empty wafer map (the reference fails here too)")
        p["rot90"] = rng.random(n) < spec.rr_prob
        p["vflip"] = rng.random(n) < spec.vf_prob
        p["hflip"] = rng.random(n) < spec.hf_prob
    if spec.crop_scale is not None and spec.crop_prob > 0:
        do = rng.random(n) < spec.crop_prob
        s0, s1 = spec.crop_scale
        target = (S * S) * (s0 + (s1 - s0) * rng.random(n))
        side = np.rint(np.sqrt(target)).astype(np.int64)     # ratio (1,1): w == h == round(sqrt(area))
        ok = (side > 0) & (side <= S)
        side = np.where(ok, side, S)
        ci = np.minimum((rng.random(n) * (S - side + 1)).astype(np.int64), S - side)
        cj = np.minimum((rng.random(n) * (S - side + 1)).astype(np.int64), S - side)
        ci, cj = np.where(ok, ci, 0), np.where(ok, cj, 0)
        p["crop"] = do
        p["crop_i"], p["crop_j"] = np.where(do, ci, 0), np.where(do, cj, 0)
        p["crop_h"] = p["crop_w"] = np.where(do, side, S)
    else:
        p["crop_h"] = p["crop_w"] = S
    return p


def validate_params(params: np.ndarray, store: WaferStore, img_size: int, n_slots: int) -> None:
    """Host-side bounds check of everything the kernel indexes with (a faulting kernel can reset the GPU)."""
    if params.dtype != PARAM_DTYPE:
        raise TypeError("params must use PARAM_DTYPE")
    s = params["sample"]
    if s.min() < 0 or s.max() >= len(store):
        raise IndexError("view params reference a wafer outside the store")
    if params["out_slot"].min() < 0 or params["out_slot"].max() >= n_slots:
        raise IndexError("view params reference an output slot outside the output tensor")
    if len(np.unique(params["out_slot"])) != len(params):
        raise ValueError("two views write the same output slot")
    op = params["op"]
    if op.min() < 0 or op.max() > _lib.WM_AUG_MEDIAN3:
        raise ValueError("unknown stage-1 op code")
    dpw = op == _lib.WM_AUG_DPW
    hh, ww = store.heights_np[s], store.widths_np[s]
    if dpw.any() and ((params["dpw_h"][dpw] < 1).any() or (params["dpw_w"][dpw] < 1).any()
                      or (params["dpw_h"][dpw] > hh[dpw]).any() or (params["dpw_w"][dpw] > ww[dpw]).any()):
        raise ValueError("DPW dims must lie in [1, wafer dims]")
    c = params["crop"] != 0
    if c.any():
        ci, cj, ch, cw = (params[k][c] for k in ("crop_i", "crop_j", "crop_h", "crop_w"))
        if (ci < 0).any() or (cj < 0).any() or (ch < 1).any() or (cw < 1).any() \
                or (ci + ch > img_size).any() or (cj + cw > img_size).any():
            raise ValueError("crop box outside the resized image")


_FMT = {"nchw_f32": _lib.WM_IMG_NCHW_F32, "nhwc_bf16": _lib.WM_IMG_NHWC_BF16, "u8": _lib.WM_IMG_HW_U8,
        "s2d_bf16": _lib.WM_IMG_S2D_BF16}


def augment_views(store: WaferStore, params: np.ndarray, img_size: int = 224, out_size: int = 224,
                  fmt: str = "nchw_f32", normalize: bool = True, mean: float = NORMALIZE_STATS["mean"][0],
                  std: float = NORMALIZE_STATS["std"][0], n_slots: Optional[int] = None,
                  out: Optional[torch.Tensor] = None, params_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Run the fused augmentation kernel for `len(params)` views.

    fmt "nchw_f32": float32 [n,3,O,O] (the reference's batch tensor);
        "nhwc_bf16": bfloat16 [n,3,O,O] in channels_last memory format (feeds the conv kernels);
        "u8": uint8 [n,O,O] (the grey image before ToTensor; parity checks);
        "s2d_bf16": bfloat16 [n,16,O/2,O/2] channels_last = the 2x2 space-to-depth image the ResNet stem
        convolution runs on (ops.stem_conv takes it as is: no separate layout pass)."""
    if store.device is None or store.device.type != "cuda":
        raise _lib.WaferHipError("WaferStore must live on the GPU (store.to('cuda')); there is no CPU path")
    n = len(params)
    n_slots = n if n_slots is None else n_slots
    validate_params(params, store, img_size, n_slots)
    if (not np.all(params["crop"] != 0)) and out_size != img_size:
        raise ValueError("out_size != img_size requires every view to crop")
    dev = store.device
    if out is None:
        if fmt == "nchw_f32":
            out = torch.empty((n_slots, 3, out_size, out_size), dtype=torch.float32, device=dev)
        elif fmt == "nhwc_bf16":
            out = torch.empty((n_slots, out_size, out_size, 3), dtype=torch.bfloat16, device=dev).permute(0, 3, 1, 2)
        elif fmt == "u8":
            out = torch.empty((n_slots, out_size, out_size), dtype=torch.uint8, device=dev)
        elif fmt == "s2d_bf16":
            out = torch.empty((n_slots, out_size // 2, out_size // 2, 16), dtype=torch.bfloat16,
                              device=dev).permute(0, 3, 1, 2)
        else:
            raise ValueError(f"unknown fmt {fmt}")
    if params_dev is not None:
        # caller-managed static device buffer (graph capture): it must already hold `params`
        if params_dev.numel() * params_dev.element_size() < n * PARAM_DTYPE.itemsize or params_dev.device != dev:
            raise ValueError("params_dev too small or on the wrong device")
        pdev = params_dev
    else:
        pbytes = torch.from_numpy(np.ascontiguousarray(params).view(np.uint8).reshape(-1))
        pdev = pbytes.to(dev, non_blocking=False)
    check(_lib.load().wm_augment_views(ptr(store.bytes), ptr(store.offsets), ptr(store.heights), ptr(store.widths),
                                       len(store), store.max_elems, ptr(pdev), n, img_size, out_size, _FMT[fmt],
                                       int(bool(normalize)), float(mean), float(std), out.data_ptr(), stream_ptr()),
          "wm_augment_views")
    if params_dev is None:  # pdev must outlive the launch: record it on the current stream
        pdev.record_stream(torch.cuda.current_stream())
    return out
