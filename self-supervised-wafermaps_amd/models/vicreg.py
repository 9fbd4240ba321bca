"""VICReg on ResNet-18 (scripts/WM811k_benchmark.py:394-427): Barlow Twins' encoder and head with VICRegLoss and
LARS (lr 0.3 x bs/256, weight decay 1e-4)."""
from __future__ import annotations

from .. import optim
from ..loss import VICRegLoss
from ..utils import scheduler
from .barlow import BarlowTwins


class VICReg(BarlowTwins):
    def __init__(self, dataloader_kNN=None, num_classes=9, **kwargs):
        super().__init__(dataloader_kNN, num_classes, **kwargs)
        self.criterion = VICRegLoss()

    def configure_optimizers(self):
        opt = optim.LARS(self.parameters(), lr=0.3 * self.lr_factor, weight_decay=1e-4, momentum=0.9)
        return [opt], [scheduler.CosineWarmupScheduler(opt, self.warmup_epochs, self.max_epochs)]
