"""SwaV on ResNet-18 (scripts/WM811k_benchmark.py:824-874): multi-crop views, SwaVProjectionHead(512, 2048, 128),
3000 unit-norm prototypes, Sinkhorn assignments of the two 224^2 crops as targets for all other crops, Adam."""
from __future__ import annotations

import torch

from .. import functional as F_hip
from .. import heads, ops, optim
from ..loss import SwaVLoss
from ..utils import debug
from .knn import KNNBenchmarkModule
from .resnet import create_model


class SwaV(KNNBenchmarkModule):
    def __init__(self, dataloader_kNN=None, num_classes=9, batch_size: int = 64, max_epochs: int = 150,
                 log_rep_std: bool = True, **kwargs):
        super().__init__(dataloader_kNN, num_classes, **kwargs)
        self.backbone = create_model("resnet18", num_classes=0, pretrained=False)
        feature_dim = self.backbone.num_features
        self.projection_head = heads.SwaVProjectionHead(feature_dim, 2048, 128)
        self.prototypes = heads.SwaVPrototypes(128, 3000)
        self.criterion = SwaVLoss()
        self.lr_factor = batch_size / 256
        self.max_epochs = max_epochs
        self.log_rep_std = log_rep_std

    def forward(self, x):
        x = self.backbone(x).flatten(start_dim=1)
        if self.log_rep_std:
            self.log("rep_std", debug.std_of_l2_normalized(x.detach()[: x.shape[0] // ops.current_bn_groups()]))
        x = self.projection_head(x)
        x = F_hip.l2_normalize(x)
        return self.prototypes(x)

    def training_step(self, batch, batch_idx):
        self.prototypes.normalize()
        views = batch[0]
        b = views[0].shape[0]
        outs, i = [], 0
        groups = getattr(views, "stacked_groups", None) or {}
        while i < len(views):  # crops of one resolution run as one batch, BatchNorm statistics per crop
            j = i
            while j + 1 < len(views) and views[j + 1].shape == views[i].shape:
                j += 1
            x = groups.get((i, j + 1))
            if x is None:
                x = views[i] if j == i else torch.cat(list(views[i:j + 1]), dim=0)
            with ops.bn_groups(j + 1 - i):
                y = self.forward(x)
            outs.extend(y[k * b:(k + 1) * b] for k in range(j + 1 - i))
            i = j + 1
        loss = self.criterion(outs[:2], outs[2:])
        self.log("train_loss_ssl", loss)
        return loss

    def configure_optimizers(self):
        opt = optim.Adam(self.parameters(), lr=1e-3 * self.lr_factor, weight_decay=1e-6)
        return [opt], [torch.optim.lr_scheduler.CosineAnnealingLR(opt, self.max_epochs)]
