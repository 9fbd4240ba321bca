"""Barlow Twins on ResNet-18 (scripts/WM811k_benchmark.py:357-392): BarlowTwinsProjectionHead(512, 2048, 2048),
BarlowTwinsLoss, LARS (lr 0.2 x bs/256, weight decay 1.5e-6, momentum 0.9) with cosine warm-up."""
from __future__ import annotations

import torch

from .. import heads, ops, optim
from ..loss import BarlowTwinsLoss
from ..utils import debug, scheduler
from .knn import KNNBenchmarkModule
from .resnet import create_model


class BarlowTwins(KNNBenchmarkModule):
    def __init__(self, dataloader_kNN=None, num_classes=9, batch_size: int = 64, max_epochs: int = 150,
                 log_rep_std: bool = True, **kwargs):
        super().__init__(dataloader_kNN, num_classes, **kwargs)
        self.backbone = create_model("resnet18", num_classes=0, pretrained=False)
        feature_dim = self.backbone.num_features
        self.projection_head = heads.BarlowTwinsProjectionHead(feature_dim, 2048, 2048)
        self.criterion = BarlowTwinsLoss()
        self.warmup_epochs = 40 if max_epochs >= 800 else 20
        self.lr_factor = batch_size / 256
        self.max_epochs = max_epochs
        self.log_rep_std = log_rep_std

    def forward(self, x):
        x = self.backbone(x).flatten(start_dim=1)
        z = self.projection_head(x)
        if self.log_rep_std:
            self.log("rep_std", debug.std_of_l2_normalized(x.detach()[: x.shape[0] // ops.current_bn_groups()]))
        return z

    def training_step(self, batch, batch_index):
        views = batch[0]
        b = views[0].shape[0]
        stacked = getattr(views, "stacked", None)
        x = stacked if stacked is not None else torch.cat([views[0], views[1]], dim=0)
        with ops.bn_groups(2):
            z = self.forward(x)
        loss = self.criterion(z[:b], z[b:])
        self.log("train_loss_ssl", loss)
        return loss

    def configure_optimizers(self):
        opt = optim.LARS(self.parameters(), lr=0.2 * self.lr_factor, weight_decay=1.5e-6, momentum=0.9)
        return [opt], [scheduler.CosineWarmupScheduler(opt, self.warmup_epochs, self.max_epochs)]
