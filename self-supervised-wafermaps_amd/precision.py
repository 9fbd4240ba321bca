"""Precision presets of the path.

"bf16" (default): activations and MFMA operands in bfloat16, float32 accumulation and float32 master weights -- the
production / benchmark preset (the reference trains under fp16 AMP, scripts/WM811k_benchmark.py:57,1107).
"float32" (alias "parity"): the same modules with every activation, weight and accumulator in float32
(csrc/f32path.hip) -- the preset under which the whole-step loss and the embeddings meet north_star's tolerance against
the reference's float32 CPU path (loss 1e-4 relative, embeddings 1e-3 cosine).  The error budget in
profiles/r04_error_budget_bf16.md shows why the bf16 preset cannot: bf16 storage of the inter-layer activations, the
same distance torch's own bf16 autocast lands at.  A validation preset (~50x slower): the ResNet-18 / projection-head
path (convolution, BatchNorm, pooling, Linear, NT-Xent) has its backward pass, so whole SimCLR optimiser steps follow the
oracle to 1e-5; the transformer steps (DINO, MAE) run forward only and raise when differentiated.

    with ssl_wafermap_amd.precision("float32"):
        loss = model.training_step(batch, 0)
"""
from __future__ import annotations

import contextlib

_MODE = "bf16"
_ALIASES = {"bf16": "bf16", "bfloat16": "bf16", "float32": "float32", "f32": "float32", "parity": "float32"}


def current() -> str:
    return _MODE


def is_f32() -> bool:
    return _MODE == "float32"


def set_precision(mode: str) -> str:
    """Select the preset for every following call; returns the previous one."""
    global _MODE
    if mode not in _ALIASES:
        raise ValueError(f"precision: unknown preset {mode!r} (have {sorted(set(_ALIASES))})")
    old, _MODE = _MODE, _ALIASES[mode]
    return old


@contextlib.contextmanager
def precision(mode: str):
    old = set_precision(mode)
    try:
        yield
    finally:
        set_precision(old)
