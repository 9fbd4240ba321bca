"""Projection heads with lightly's constructor signatures and `layers.N.*` state_dict keys
(lightly.models.modules.heads; reference use scripts/WM811k_benchmark.py:233)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch.nn as nn

from . import nn as hnn


class ProjectionHead(nn.Module):
    """blocks: (in, out, batch_norm module or None, activation module or None); Linear has a bias
    only without batch norm, as in lightly."""

    def __init__(self, blocks: List[Tuple[int, int, Optional[nn.Module], Optional[nn.Module]]]):
        super().__init__()
        layers = []
        self._plan = []
        for in_dim, out_dim, bn, act in blocks:
            lin = hnn.Linear(in_dim, out_dim, bias=not bool(bn))
            layers.append(lin)
            i_lin = len(layers) - 1
            i_bn = None
            if bn:
                layers.append(bn)
                i_bn = len(layers) - 1
            if act:
                layers.append(act)
            self._plan.append((i_lin, i_bn, act is not None))
        self.layers = nn.Sequential(*layers)

    def forward(self, x):
        for i_lin, i_bn, relu in self._plan:
            x = self.layers[i_lin](x)
            if i_bn is None:
                raise NotImplementedError("heads without batch norm need a bias/activation kernel (not built yet)")
            x = self.layers[i_bn](x, relu=relu)
        return x


class SimCLRProjectionHead(ProjectionHead):
    """Linear-BN-ReLU (x num_layers-1), Linear-BN  (SimCLR v2 form, lightly's default since 1.2;
    SURVEY Appendix A.2 — lightly is unpinned in the reference, the BN-free v1 form is not built)."""

    def __init__(self, input_dim: int = 2048, hidden_dim: int = 2048, output_dim: int = 128, num_layers: int = 2,
                 batch_norm: bool = True):
        if not batch_norm:
            raise NotImplementedError("SimCLRProjectionHead(batch_norm=False) has no HIP path yet")
        blocks = [(input_dim, hidden_dim, hnn.BatchNorm1d(hidden_dim), hnn.ReLU())]
        for _ in range(2, num_layers):
            blocks.append((hidden_dim, hidden_dim, hnn.BatchNorm1d(hidden_dim), hnn.ReLU()))
        blocks.append((hidden_dim, output_dim, hnn.BatchNorm1d(output_dim), None))
        super().__init__(blocks)
