"""lightly.models.utils equivalents used by the reference's DINO / MAE models
(scripts/WM811k_benchmark.py:561-562,579-581 update_momentum / deactivate_requires_grad;
:911-914,930-947 random_token_mask / get_at_index / set_at_index / repeat_token / patchify)."""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.nn as nn

from .. import _lib
from .._lib import check, ptr, stream_ptr

_ALIGN = 64  # elements; the same rule as optim._Arena so a flattened copy mirrors an arena's gaps


def deactivate_requires_grad(model: nn.Module) -> None:
    for p in model.parameters():
        p.requires_grad = False


def activate_requires_grad(model: nn.Module) -> None:
    for p in model.parameters():
        p.requires_grad = True


def _unique_params(model: nn.Module) -> List[torch.nn.Parameter]:
    return list(model.parameters())


def flatten_parameters(model: nn.Module) -> torch.Tensor:
    """Move the module's float32 parameters into ONE flat buffer (each starting on a 64-element
    boundary, in .parameters() order) and make them views of it.  Returns the buffer."""
    ps = _unique_params(model)
    total, offs = 0, []
    for p in ps:
        offs.append(total)
        total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
    flat = torch.zeros(total, dtype=torch.float32, device=ps[0].device)
    with torch.no_grad():
        for p, o in zip(ps, offs):
            flat[o:o + p.numel()].copy_(p.detach().reshape(-1))
            p.data = flat[o:o + p.numel()].view(p.shape)
    model._hip_flat = flat
    return flat


def _layout(ps) -> Tuple[int, List[int], int]:
    """(start address, offsets in elements, span in elements) when the parameters lie in address
    order inside one allocation-like span; span 0 otherwise."""
    start = ps[0].data_ptr()
    offs, end = [], start
    for p in ps:
        a = p.data_ptr()
        if a < end or (a - start) % 4 or not p.is_contiguous() or p.dtype != torch.float32:
            return start, [], 0
        offs.append((a - start) // 4)
        end = a + p.numel() * 4
    return start, offs, (end - start) // 4


@torch.no_grad()
def update_momentum(model: nn.Module, model_ema: nn.Module, m: float) -> None:
    """model_ema <- m * model_ema + (1 - m) * model, parameter by parameter (lightly semantics).
    One kernel launch over the whole span when both parameter sets are laid out identically in flat
    buffers (the student inside a fused optimiser's arena, the teacher flattened here on first use);
    otherwise one launch per parameter."""
    ps, es = _unique_params(model), _unique_params(model_ema)
    if len(ps) != len(es):
        raise ValueError("update_momentum: the two models have different parameter lists")
    if not ps:
        return
    if ps[0].device.type != "cuda":
        raise _lib.WaferHipError("update_momentum runs on the GPU (no CPU fallback)")
    lib = _lib.load()
    s0, so, sn = _layout(ps)
    if sn and getattr(model_ema, "_hip_flat", None) is None:
        flatten_parameters(model_ema)
    e0, eo, en = _layout(es)
    from .. import ops

    if sn and sn == en and so == eo:
        check(lib.wm_ema_update(e0, s0, sn, float(m), stream_ptr()), "wm_ema_update")
    else:
        for p, e in zip(ps, es):
            check(lib.wm_ema_update(e.data_ptr(), p.data_ptr(), p.numel(), float(m), stream_ptr()), "wm_ema_update")
    # the teacher's bf16 kernel layouts follow in one launch (instead of one lazy wm_weights_prepare per layer in its
    # next forward pass)
    ops.refresh_layouts(es)


def random_token_mask(size: Tuple[int, int], mask_ratio: float = 0.6, mask_class_token: bool = False,
                      device=None, generator: torch.Generator = None):
    """lightly.models.utils.random_token_mask: per-image random permutation of the tokens; the first
    int(S * (1 - ratio)) indices are kept (the class token always, unless mask_class_token)."""
    batch_size, sequence_length = size
    num_keep = int(sequence_length * (1 - mask_ratio))
    noise = torch.rand(batch_size, sequence_length, device=device, generator=generator)
    if not mask_class_token and sequence_length > 0:
        noise[:, 0] = -1
    if noise.is_cuda and 0 < sequence_length <= 256:
        # the permutation on the device: one bitonic network per image (wm_argsort_rows), no library sort
        noise = noise.float().contiguous()
        indices = torch.empty((batch_size, sequence_length), dtype=torch.int64, device=noise.device)
        check(_lib.load().wm_argsort_rows(ptr(noise), batch_size, sequence_length, ptr(indices), stream_ptr()),
              "wm_argsort_rows")
    else:  # host tensors (the CPU-side logic tests) and sequences beyond the kernel's 256 keys
        indices = torch.argsort(noise, dim=1)
    return indices[:, :num_keep], indices[:, num_keep:]


def repeat_token(token: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    batch_size, sequence_length = size
    return token.repeat(batch_size, sequence_length, 1)


def patchify(images: torch.Tensor, patch_size: int) -> torch.Tensor:
    """[N, C, H, W] -> [N, (H/p)(W/p), p*p*C] with the last dim ordered (ph, pw, c) — lightly's
    einsum 'nchpwq->nhwpqc'.  For a channels_last bf16 image batch this is wm_patchify."""
    from .. import ops

    n, c, h, w = images.shape
    if c != 3 or h != w or h % patch_size:
        raise ValueError(f"patchify: unsupported shape {tuple(images.shape)} for patch {patch_size}")
    from .. import precision

    if precision.is_f32():   # float32 preset: pure data movement (lightly's einsum "nchpwq->nhwpqc")
        g = h // patch_size
        x = images.detach().float().reshape(n, c, g, patch_size, g, patch_size)
        return x.permute(0, 2, 4, 3, 5, 1).reshape(n, g * g, patch_size * patch_size * c).contiguous()
    x = ops._as_nhwc(images)
    g = h // patch_size
    rows = torch.empty((n * g * g, patch_size * patch_size * 3), dtype=torch.bfloat16, device=x.device)
    check(_lib.load().wm_patchify(x.data_ptr(), n, h, patch_size, rows.data_ptr(), stream_ptr()), "wm_patchify")
    return rows.view(n, g * g, patch_size * patch_size * 3)


def get_at_index(tokens: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """tokens [B, S, C], index [B, K] -> [B, K, C]."""
    from .. import vit_ops

    b, s, c = tokens.shape
    return vit_ops.gather_rows(tokens.reshape(b * s, c), index, b, s).view(b, index.shape[1], c)


def set_at_index(tokens: torch.Tensor, index: torch.Tensor, value: torch.Tensor) -> torch.Tensor:
    """copy of tokens [B, S, C] with rows index [B, K] replaced by value [B, K, C]."""
    from .. import vit_ops

    b, s, c = tokens.shape
    return vit_ops.scatter_rows(tokens.reshape(b * s, c), value.reshape(-1, c), index, b, s).view(b, s, c)


def mask_at_index(tokens: torch.Tensor, index: torch.Tensor, mask_token: torch.Tensor) -> torch.Tensor:
    b, k = index.shape
    return set_at_index(tokens, index, mask_token.to(tokens.dtype).expand(b, k, tokens.shape[2]))


def _dist_world():
    import torch.distributed as dist

    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def batch_shuffle(batch: torch.Tensor, distributed: bool = False):
    """lightly.models.utils.batch_shuffle: random permutation of the batch dimension (MoCo's guard against
    BatchNorm leaking the positive pair across GPUs; reference scripts/WM811k_benchmark.py:321).  Returns
    (shuffled batch, permutation).  distributed=True (lightly's batch_shuffle_distributed): the batch is gathered
    from all ranks, ONE permutation of the global batch (rank 0's, broadcast) is applied and every rank keeps its
    contiguous share; the second return value is then the inverse permutation for batch_unshuffle."""
    if distributed and _dist_world() > 1:
        import torch.distributed as dist

        from ..distributed import all_gather_rows

        world, rank = dist.get_world_size(), dist.get_rank()
        gathered = all_gather_rows(batch.contiguous())
        n_all = gathered.shape[0]
        idx_shuffle = torch.randperm(n_all, device=batch.device)
        dist.broadcast(idx_shuffle, src=0)
        idx_unshuffle = torch.argsort(idx_shuffle)
        idx_this = idx_shuffle.view(world, -1)[rank]
        return gathered[idx_this], idx_unshuffle
    shuffle = torch.randperm(batch.shape[0], device=batch.device)
    return batch[shuffle], shuffle


def batch_unshuffle(batch: torch.Tensor, shuffle: torch.Tensor, distributed: bool = False):
    """Undo batch_shuffle.  distributed=True: `shuffle` is the inverse permutation of the GLOBAL batch returned by
    batch_shuffle(distributed=True); the ranks' batches are gathered again and this rank takes its original rows."""
    if distributed and _dist_world() > 1:
        import torch.distributed as dist

        from ..distributed import all_gather_rows

        world, rank = dist.get_world_size(), dist.get_rank()
        gathered = all_gather_rows(batch.contiguous())
        idx_this = shuffle.view(world, -1)[rank]
        return gathered[idx_this]
    unshuffle = torch.argsort(shuffle)
    return batch[unshuffle]
