"""DCLW on ResNet-18 (scripts/WM811k_benchmark.py:258-287): SimCLR's encoder and head with the decoupled
contrastive loss."""
from __future__ import annotations

from ..loss import DCLWLoss
from .simclr import SimCLR


class DCLW(SimCLR):
    def __init__(self, dataloader_kNN=None, num_classes=9, **kwargs):
        kwargs.pop("gather_distributed", None)
        super().__init__(dataloader_kNN, num_classes, **kwargs)
        self.criterion = DCLWLoss()
