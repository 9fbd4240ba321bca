"""BYOL, SimSiam and FastSiam on ResNet-18: the reference's classes (scripts/WM811k_benchmark.py:429-488,
605-660) on the HIP kernels.  As for SimCLR, the views run through the encoder as one batch with
BatchNorm statistics per view (`ops.bn_groups`), which is what the reference's per-view forward calls
compute.
"""
from __future__ import annotations

import copy

import torch

from .. import heads, ops, optim
from ..loss import NegativeCosineSimilarity
from ..utils import debug, model_utils
from .knn import KNNBenchmarkModule
from .resnet import create_model


def _stack(views):
    stacked = getattr(views, "stacked", None)
    return stacked if stacked is not None else torch.cat(list(views), dim=0)


class BYOL(KNNBenchmarkModule):
    def __init__(self, dataloader_kNN=None, num_classes=9, batch_size: int = 64, max_epochs: int = 150,
                 log_rep_std: bool = True, **kwargs):
        super().__init__(dataloader_kNN, num_classes, **kwargs)
        self.backbone = create_model("resnet18", num_classes=0, pretrained=False)
        feature_dim = self.backbone.num_features
        self.projection_head = heads.BYOLProjectionHead(feature_dim, 4096, 256)
        self.prediction_head = heads.BYOLPredictionHead(256, 4096, 256)
        self.backbone_momentum = copy.deepcopy(self.backbone)
        self.projection_head_momentum = copy.deepcopy(self.projection_head)
        model_utils.deactivate_requires_grad(self.backbone_momentum)
        model_utils.deactivate_requires_grad(self.projection_head_momentum)
        self.criterion = NegativeCosineSimilarity()
        self.lr_factor = batch_size / 256
        self.max_epochs = max_epochs
        self.log_rep_std = log_rep_std

    def forward(self, x):
        y = self.backbone(x).flatten(start_dim=1)
        z = self.projection_head(y)
        p = self.prediction_head(z)
        if self.log_rep_std:
            self.log("rep_std", debug.std_of_l2_normalized(y.detach()[: y.shape[0] // ops.current_bn_groups()]))
        return p

    @torch.no_grad()
    def forward_momentum(self, x):
        y = self.backbone_momentum(x).flatten(start_dim=1)
        return self.projection_head_momentum(y).detach()

    def training_step(self, batch, batch_idx):
        model_utils.update_momentum(self.backbone, self.backbone_momentum, m=0.99)
        model_utils.update_momentum(self.projection_head, self.projection_head_momentum, m=0.99)
        views = batch[0]
        b = views[0].shape[0]
        x = _stack(views)
        with ops.bn_groups(2):
            p = self.forward(x)
            z = self.forward_momentum(x)
        loss = 0.5 * (self.criterion(p[:b], z[b:]) + self.criterion(p[b:], z[:b]))
        self.log("train_loss_ssl", loss)
        return loss

    def configure_optimizers(self):
        params = (list(self.backbone.parameters()) + list(self.projection_head.parameters())
                  + list(self.prediction_head.parameters()))
        optimizer = optim.SGD(params, lr=6e-2 * self.lr_factor, momentum=0.9, weight_decay=5e-4)
        return [optimizer], [torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, self.max_epochs)]


class SimSiam(KNNBenchmarkModule):
    def __init__(self, dataloader_kNN=None, num_classes=9, max_epochs: int = 150, log_rep_std: bool = True, **kwargs):
        kwargs.pop("batch_size", None)  # the reference applies no lr scaling here
        super().__init__(dataloader_kNN, num_classes, **kwargs)
        self.backbone = create_model("resnet18", num_classes=0, pretrained=False)
        feature_dim = self.backbone.num_features
        self.projection_head = heads.SimSiamProjectionHead(feature_dim, 2048, 2048)
        self.prediction_head = heads.SimSiamPredictionHead(2048, 512, 2048)
        self.criterion = NegativeCosineSimilarity()
        self.max_epochs = max_epochs
        self.log_rep_std = log_rep_std

    def forward(self, x):
        f = self.backbone(x).flatten(start_dim=1)
        z = self.projection_head(f)
        p = self.prediction_head(z)
        if self.log_rep_std:
            self.log("rep_std", debug.std_of_l2_normalized(f.detach()[: f.shape[0] // ops.current_bn_groups()]))
        return z.detach(), p

    def training_step(self, batch, batch_idx):
        views = batch[0]
        b = views[0].shape[0]
        with ops.bn_groups(2):
            z, p = self.forward(_stack(views))
        loss = 0.5 * (self.criterion(z[:b], p[b:]) + self.criterion(z[b:], p[:b]))
        self.log("train_loss_ssl", loss)
        return loss

    def configure_optimizers(self):
        optimizer = optim.SGD(self.parameters(), lr=6e-2, momentum=0.9, weight_decay=5e-4)
        return [optimizer], [torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, self.max_epochs)]


class FastSiam(SimSiam):
    """SimSiam with n views: every prediction is pulled towards the mean of the OTHER views' projections."""

    def training_step(self, batch, batch_idx):
        views = batch[0]
        n, b = len(views), views[0].shape[0]
        with ops.bn_groups(n):
            z, p = self.forward(_stack(views))
        zs = z.float().view(n, b, -1)
        total = zs.sum(dim=0)
        loss = 0.0
        for i in range(n):
            target = (total - zs[i]) / (n - 1)   # mean of the other views' (detached) projections
            loss = loss + self.criterion(p[i * b:(i + 1) * b].float(), target) / n
        self.log("train_loss_ssl", loss)
        return loss
