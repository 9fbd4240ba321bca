"""lightly.utils.debug equivalents."""
import torch

from .. import _lib
from .. import functional as F_hip
from .._lib import check, dtype_code, ptr, stream_ptr


def std_of_l2_normalized(z: torch.Tensor) -> torch.Tensor:
    """Mean over dimensions of the per-dimension std of the L2-normalised rows (collapse monitor the
    reference logs as `rep_std`, scripts/WM811k_benchmark.py:239).  Column statistics by wm_colstats (two
    passes, population variance) rescaled to torch.std's unbiased estimate: torch's dim-0 reduction of a
    [256, 512] matrix took 2 x 38 us per training step."""
    if z.dim() != 2:
        raise ValueError(f"Input tensor must have two dimensions but has {z.dim()}!")
    zn = F_hip.l2_normalize(z.detach().contiguous())
    rows, c = zn.shape
    if rows < 2 or not zn.is_cuda:
        return torch.std(zn.float(), dim=0).mean()
    stats = torch.empty(2, c, dtype=torch.float32, device=zn.device)   # (the kernel accumulates: cleared by our own fill)
    check(_lib.load().wm_fill_zero(ptr(stats), stats.numel() * 4, stream_ptr()), "wm_fill_zero")
    check(_lib.load().wm_colstats(ptr(zn), dtype_code(zn), rows, c, ptr(stats[0]), ptr(stats[1]), stream_ptr()),
          "wm_colstats")
    return F_hip.vector_mean(stats[1], scale=rows / (rows - 1.0), sqrt_of=True)
