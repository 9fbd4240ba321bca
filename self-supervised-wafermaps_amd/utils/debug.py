"""lightly.utils.debug equivalents."""
import torch

from .. import functional as F_hip


def std_of_l2_normalized(z: torch.Tensor) -> torch.Tensor:
    """Mean over dimensions of the per-dimension std of the L2-normalised rows (collapse monitor the
    reference logs as `rep_std`, scripts/WM811k_benchmark.py:239)."""
    if z.dim() != 2:
        raise ValueError(f"Input tensor must have two dimensions but has {z.dim()}!")
    zn = F_hip.l2_normalize(z.detach().contiguous())
    return torch.std(zn, dim=0).mean()
