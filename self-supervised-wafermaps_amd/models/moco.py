"""MoCo on ResNet-18: the reference's MoCo (scripts/WM811k_benchmark.py:289-351): momentum encoder,
MoCoProjectionHead(512, 2048, 128), NTXentLoss(temperature 0.1, memory bank 4096), symmetric loss, SGD.

The reference's `step(x0, x1)` / `step(x1, x0)` run four backbone passes; here the query encoder sees
[x0; x1] and the momentum encoder [x1; x0] as one batch each, with BatchNorm statistics per view
(`ops.bn_groups(2)`), which is what the separate calls compute.  `batch_shuffle` is skipped on one GPU:
with per-view batch statistics a permutation of the batch does not change any sample's output.  The two
loss terms keep the reference's order: the keys of the first term enter the bank before the second term
reads it.
"""
from __future__ import annotations

import copy

import torch

from .. import heads, ops, optim
from ..loss import NTXentLoss
from ..utils import debug, model_utils
from .knn import KNNBenchmarkModule
from .resnet import create_model


class MoCo(KNNBenchmarkModule):
    def __init__(self, dataloader_kNN=None, num_classes=9, batch_size: int = 64, max_epochs: int = 150,
                 memory_bank_size: int = 4096, log_rep_std: bool = True, **kwargs):
        super().__init__(dataloader_kNN, num_classes, **kwargs)
        self.backbone = create_model("resnet18", num_classes=0, pretrained=False)
        feature_dim = self.backbone.num_features
        self.projection_head = heads.MoCoProjectionHead(feature_dim, 2048, 128)
        self.backbone_momentum = copy.deepcopy(self.backbone)
        self.projection_head_momentum = copy.deepcopy(self.projection_head)
        model_utils.deactivate_requires_grad(self.backbone_momentum)
        model_utils.deactivate_requires_grad(self.projection_head_momentum)
        self.criterion = NTXentLoss(temperature=0.1, memory_bank_size=memory_bank_size)
        self.lr_factor = batch_size / 256
        self.max_epochs = max_epochs
        self.log_rep_std = log_rep_std

    def forward(self, x):
        x = self.backbone(x).flatten(start_dim=1)
        return self.projection_head(x)

    def training_step(self, batch, batch_idx):
        views = batch[0]
        x0, x1 = views[0], views[1]
        b = x0.shape[0]
        model_utils.update_momentum(self.backbone, self.backbone_momentum, 0.99)
        model_utils.update_momentum(self.projection_head, self.projection_head_momentum, 0.99)
        stacked = getattr(views, "stacked", None)
        xq = stacked if stacked is not None else torch.cat([x0, x1], dim=0)
        xk = torch.cat([x1, x0], dim=0)
        with ops.bn_groups(2):
            f = self.backbone(xq).flatten(start_dim=1)
            if self.log_rep_std:
                self.log("rep_std", debug.std_of_l2_normalized(f[:b].detach()))
            q = self.projection_head(f)
            with torch.no_grad():
                k = self.projection_head_momentum(self.backbone_momentum(xk).flatten(start_dim=1))
        loss_1 = self.criterion(q[:b], k[:b])   # query x0, key x1
        loss_2 = self.criterion(q[b:], k[b:])   # query x1, key x0 (bank already holds the keys of term 1)
        loss = 0.5 * (loss_1 + loss_2)
        self.log("train_loss_ssl", loss)
        return loss

    def configure_optimizers(self):
        params = list(self.backbone.parameters()) + list(self.projection_head.parameters())
        optimizer = optim.SGD(params, lr=6e-2 * self.lr_factor, momentum=0.9, weight_decay=5e-4)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, self.max_epochs)
        return [optimizer], [sched]
