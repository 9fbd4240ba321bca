"""SimCLR on ResNet-18: the reference's model class (scripts/WM811k_benchmark.py:227-255) on the
HIP kernels.  Same attribute names (backbone, projection_head, criterion), same hyper-parameters."""
from __future__ import annotations

import os

import torch

from .. import heads, ops, optim
from ..loss import NTXentLoss, stacked_views
from ..utils import debug
from .knn import KNNBenchmarkModule
from .resnet import create_model


class SimCLR(KNNBenchmarkModule):
    graph_safe = True        # training_step has no host-side scalar that changes from step to step
    stem_takes_s2d = True    # the ResNet stem consumes the augmentation kernel's space-to-depth layout
    backward_stages = True   # the backbone marks stage boundaries (ops.cut_point) for overlapped gradient exchange

    def __init__(self, dataloader_kNN=None, num_classes=9, batch_size: int = 64, max_epochs: int = 150,
                 gather_distributed: bool = False, log_rep_std: bool = True, **kwargs):
        super().__init__(dataloader_kNN, num_classes, **kwargs)
        self.backbone = create_model("resnet18", num_classes=0, pretrained=False)
        feature_dim = self.backbone.num_features
        self.projection_head = heads.SimCLRProjectionHead(feature_dim, feature_dim, 128)
        self.criterion = NTXentLoss(gather_distributed=gather_distributed)
        self.lr_factor = batch_size / 256  # the reference scales the learning rate linearly
        self.max_epochs = max_epochs
        self.log_rep_std = log_rep_std

    def forward(self, x):
        x = self.backbone(x).flatten(start_dim=1)
        z = self.projection_head(x)
        if self.log_rep_std:
            g = ops.current_bn_groups()
            for part in x.detach().chunk(g):  # one value per reference forward() call; the last is kept
                self.log("rep_std", debug.std_of_l2_normalized(part))
        return z

    def training_step(self, batch, batch_index):
        (x0, x1), _ = batch[0], batch[1]
        views = batch[0]
        stacked = getattr(views, "stacked", None)
        if stacked is None:
            stacked = torch.cat([x0, x1], dim=0)
        b = x0.shape[0]
        # both views in one call; BatchNorm statistics per view, exactly as the reference's z0 = forward(x0);
        # z1 = forward(x1) -- which the backbone runs as two parallel branches (models/resnet.py, nn.ViewBranches)
        with ops.bn_groups(2):
            z = self.forward(stacked)
        loss = self.criterion(*stacked_views(z, b))
        self.log("train_loss_ssl", loss)
        return loss

    def configure_optimizers(self):
        optimizer = optim.SGD(self.parameters(), lr=6e-2 * self.lr_factor, momentum=0.9, weight_decay=5e-4)
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, self.max_epochs)
        return [optimizer], [scheduler]
