"""CPU oracle for the wafer-map augmentation path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (self-supervised-wafermaps_amd/) never does.

numpy restatement of the reference's per-sample pipeline, every random choice an explicit input:

  die_noise            src/ssl_wafermap/transforms/augmentations.py:27-36
  random_one_of_index  augmentations.py:82-87
  median3              augmentations.py:103-107 (cv2.medianBlur(x, 3); OpenCV is a third-party
                       dependency absent here: restated as the 3x3 median with replicated border,
                       cross-checked against scipy.ndimage.median_filter(mode="nearest"))
  power_law_transform  augmentations.py:152-174
  skewed_scale         augmentations.py:176-180
  dpw_transform        augmentations.py:182-227
  dpw_call             augmentations.py:229-250
  base_view            augmentations.py:289-330 (+ the torchvision/PIL/lightly semantics of
                       SURVEY.md Appendix A.7, pinned here against PIL itself in tests)
  multicrop_view       src/ssl_wafermap/transforms/wafer_multicrop_transform.py:66-85
  inference_view       augmentations.py:335-357

Parity status: die_noise, power_law_transform, dpw_transform, dpw_call and random_one_of_index are
PINNED by golden vectors produced by running the reference's own code (tests/golden/, generator
tests/golden/make_reference_vectors.py).  The resize / rotate / flip / crop steps live in
torchvision + PIL + lightly (unpinned third-party, absent): pinned against PIL (present) instead.
median3: parity unpinned upstream (cv2 absent), cross-checked against scipy.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional, Tuple

import numpy as np

NORMALIZE_MEAN = 0.4496  # src/ssl_wafermap/transforms/utils.py:1-4
NORMALIZE_STD = 0.2926

OP_NONE, OP_DIENOISE, OP_DPW, OP_MEDIAN3 = 0, 1, 2, 3


# ----------------------------------------------------------------------------- counter RNG
def _lowbias32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x7FEB352D)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x846CA68B)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def rand01(seed: int, n: int) -> np.ndarray:
    """The kernel's counter RNG: float32 uniform [0,1) for idx = 0..n-1 (wafer_hip.h)."""
    with np.errstate(over="ignore"):
        s = _lowbias32(np.array([(seed ^ 0x9E3779B9) & 0xFFFFFFFF], dtype=np.uint32))[0]
        x = _lowbias32(np.arange(n, dtype=np.uint32) ^ s)
    return (x >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


# ----------------------------------------------------------------------------- stage-1 ops
def die_noise(sample: np.ndarray, rand: np.ndarray, p: float = 0.03) -> np.ndarray:
    """augmentations.py:27-36 on a copy (the reference mutates the dataset tensor in place —
    data/dataset.py:30 hands out the stored tensor — which we treat as a reference bug)."""
    sample = np.array(sample, dtype=np.uint8, copy=True)
    mask = (sample == 128) | (sample == 255)
    flip = (rand.astype(np.float32) < np.float32(p)) & mask
    # uint8 arithmetic: 383 wraps to 127; 127-128 -> 255, 127-255 -> 128
    sample[flip] = (np.uint8(127) - sample[flip]).astype(np.uint8)
    return sample


def random_one_of_index(u: float, weights) -> int:
    """random.choices(range(n), weights)[0] given its one uniform draw u = random.random():
    bisect(cumulative_weights, u * total) as CPython does (augmentations.py:84)."""
    import bisect
    import itertools

    cum = list(itertools.accumulate(weights))
    total = cum[-1] + 0.0
    return bisect.bisect(cum, u * total, 0, len(weights) - 1)


def median3(sample: np.ndarray) -> np.ndarray:
    a = np.asarray(sample, dtype=np.uint8)
    p = np.pad(a, 1, mode="edge")
    h, w = a.shape
    stack = np.stack([p[r : r + h, c : c + w] for r in range(3) for c in range(3)], axis=0)
    return np.sort(stack, axis=0)[4].astype(np.uint8)


def power_law_transform(x, domain_lower=26, domain_upper=212, out_lower=0.4, out_upper=0.95, p=5):
    if x <= domain_lower:
        return out_upper
    if x >= domain_upper:
        return out_lower
    domain_range = domain_upper - domain_lower
    inverted_x = abs(x - domain_lower)
    normalized_x = inverted_x / domain_range
    y = (1 - normalized_x) ** p
    out_range = out_upper - out_lower
    return out_lower + y * out_range


def skewed_scale(lower_bound: float, beta_draw: float, upper_bound: float = 0.95) -> float:
    """generate_skewed_random with its np.random.beta(alpha, beta) draw made explicit."""
    return lower_bound + (upper_bound - lower_bound) * beta_draw


def dpw_dims(h: int, w: int, scale: float) -> Tuple[int, int]:
    assert 0.0 < scale <= 1.0, "Scale must be between 0 and 1."
    return int(h * scale), int(w * scale)


def dpw_transform(wafermap: np.ndarray, scale: float) -> np.ndarray:
    a = np.asarray(wafermap, dtype=np.uint8)
    h, w = a.shape
    new_h, new_w = dpw_dims(h, w, scale)
    return dpw_transform_dims(a, new_h, new_w)


def dpw_transform_dims(a: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    h, w = a.shape
    out = np.zeros((new_h, new_w), dtype=np.uint8)
    shape = np.array([h, w], dtype=np.float32)
    new_shape = np.array([new_h, new_w], dtype=np.float32)
    for value in (128, 255):  # passes first, then fails: fails win collisions
        idx = np.argwhere(a == value)
        coords = (idx.astype(np.float32) + np.float32(0.5)) / shape
        new = (coords * new_shape).astype(np.int64)  # .long(): truncation
        out[new[:, 0], new[:, 1]] = value
    return out


def dpw_call(img: np.ndarray, beta_draw: float) -> np.ndarray:
    """DPWTransform.__call__ (defaults 26, 212, 0.4, 0.95, alpha .5, beta 1.5, p 5)."""
    scale_init = power_law_transform(max(img.shape), 26, 212, 0.4, 0.95, 5.0)
    scale = skewed_scale(scale_init, beta_draw, 0.95)
    return dpw_transform(img, scale)


def dpw_scale(shape, beta_draw: float) -> float:
    return skewed_scale(power_law_transform(max(shape), 26, 212, 0.4, 0.95, 5.0), beta_draw, 0.95)


# ----------------------------------------------------------------------------- stage-2 ops
def pil_nearest_map(n_in: int, n_out: int) -> np.ndarray:
    """Source index per destination index of PIL's NEAREST resize: xo starts at 0.5*a and is
    advanced by repeated double addition (Pillow ImagingScaleAffine); int() truncation."""
    a = n_in / n_out
    xo = a * 0.5
    out = np.empty(n_out, dtype=np.int64)
    for k in range(n_out):
        out[k] = min(int(xo), n_in - 1)
        xo += a
    return out


def resize_nearest(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    ym = pil_nearest_map(img.shape[0], out_h)
    xm = pil_nearest_map(img.shape[1], out_w)
    return img[np.ix_(ym, xm)]


def rotate90(img: np.ndarray) -> np.ndarray:
    """lightly RandomRotate -> TF.rotate(img, 90) -> PIL transpose(ROTATE_90) for square images."""
    assert img.shape[0] == img.shape[1]
    return np.rot90(img, 1)


def to_tensor_normalize(img_u8: np.ndarray, normalize=True, mean=NORMALIZE_MEAN, std=NORMALIZE_STD):
    """Grayscale(3) + ToTensor + Normalize: float32 [3,H,W]."""
    x = img_u8.astype(np.float32) / np.float32(255.0)
    if normalize:
        x = (x - np.float32(mean)) / np.float32(std)
    return np.repeat(x[None], 3, axis=0)


def random_resized_crop_params(height, width, scale, u_area, u_i, u_j):
    """torchvision RandomResizedCrop.get_params for ratio=(1,1) with the uniform draws explicit:
    target_area = area * U(scale) ; w = h = round(sqrt(target_area)); i, j ~ randint.
    u_area, u_i, u_j in [0,1)."""
    area = height * width
    target_area = area * (scale[0] + (scale[1] - scale[0]) * u_area)
    w = int(round(math.sqrt(target_area * 1.0)))
    h = int(round(math.sqrt(target_area / 1.0)))
    if 0 < w <= width and 0 < h <= height:
        i = min(int(u_i * (height - h + 1)), height - h)
        j = min(int(u_j * (width - w + 1)), width - w)
        return i, j, h, w
    return 0, 0, height, width  # ratio (1,1) fallback on a square image: the whole image


@dataclass
class ViewDecision:
    """Every random choice of one view (decision-explicit design)."""

    op: int = OP_NONE
    noise_seed: int = 0            # DieNoise via the counter RNG ...
    rand_field: Optional[np.ndarray] = None  # ... or an explicit rand field (golden vectors)
    noise_p: float = 0.03
    dpw_hw: Optional[Tuple[int, int]] = None
    rot90: bool = False
    vflip: bool = False
    hflip: bool = False
    crop: Optional[Tuple[int, int, int, int]] = None  # (i, j, h, w) in the img_size image
    out_size: Optional[int] = None


def stage1(wafer: np.ndarray, d: ViewDecision) -> np.ndarray:
    a = np.asarray(wafer, dtype=np.uint8)
    if d.op == OP_DIENOISE:
        rand = d.rand_field if d.rand_field is not None else rand01(d.noise_seed, a.size).reshape(a.shape)
        return die_noise(a, rand, d.noise_p)
    if d.op == OP_DPW:
        return dpw_transform_dims(a, *d.dpw_hw)
    if d.op == OP_MEDIAN3:
        return median3(a)
    return a


def view_u8(wafer: np.ndarray, d: ViewDecision, img_size: int = 224) -> np.ndarray:
    """The grey uint8 image just before Grayscale(3)/ToTensor."""
    x = stage1(wafer, d)
    x = resize_nearest(x, img_size, img_size)
    if d.rot90:
        x = rotate90(x)
    if d.vflip:
        x = x[::-1, :]
    if d.hflip:
        x = x[:, ::-1]
    if d.crop is not None:
        i, j, h, w = d.crop
        out = d.out_size or img_size
        x = resize_nearest(x[i : i + h, j : j + w], out, out)
    return np.ascontiguousarray(x)


def augment_view(wafer, d: ViewDecision, img_size=224, normalize=True, mean=NORMALIZE_MEAN,
                 std=NORMALIZE_STD) -> np.ndarray:
    """get_base_transforms / MultiCropViewTransform / get_inference_transforms for one view:
    float32 [3, O, O]."""
    return to_tensor_normalize(view_u8(wafer, d, img_size), normalize, mean, std)
