"""MSN and PMSN on ViT-S/16 (scripts/WM811k_benchmark.py:663-822): a target network (EMA of the anchor network)
sees the first view unmasked; the anchor network sees the second 224^2 view and the focal 96^2 views with 15 %
of the patch tokens dropped; both project onto 1024 learnable prototypes; MSNLoss / PMSNLoss.
"""
from __future__ import annotations

import copy

import torch
import torch.nn as nn

from .. import heads, optim
from ..loss import MSNLoss, PMSNLoss
from ..utils import debug, model_utils, scheduler
from .knn import KNNBenchmarkModule
from .mae import MAEBackbone


class MSN(KNNBenchmarkModule):
    loss_class = MSNLoss

    def __init__(self, dataloader_kNN=None, num_classes=9, batch_size: int = 64, max_epochs: int = 150,
                 log_rep_std: bool = True, **kwargs):
        super().__init__(dataloader_kNN, num_classes, **kwargs)
        self.warmup_epochs = 15
        self.mask_ratio = 0.15
        self.backbone = MAEBackbone(image_size=224, patch_size=16, num_layers=12, num_heads=6, hidden_dim=384,
                                    mlp_dim=384 * 4)
        self.projection_head = heads.MSNProjectionHead(384)
        self.anchor_backbone = copy.deepcopy(self.backbone)
        self.anchor_projection_head = copy.deepcopy(self.projection_head)
        model_utils.deactivate_requires_grad(self.backbone)
        model_utils.deactivate_requires_grad(self.projection_head)
        self.prototypes = nn.Parameter(nn.Linear(256, 1024, bias=False).weight.detach().clone())
        self.criterion = self.loss_class()
        self.lr_factor = batch_size / 256
        self.max_epochs = max_epochs
        self.log_rep_std = log_rep_std

    def encode_masked(self, anchors, generator: torch.Generator = None):
        batch_size, _, _, width = anchors.shape
        seq_length = (width // self.anchor_backbone.patch_size) ** 2
        # as the reference: the mask is drawn over seq_length = number of PATCHES (index 0, always kept, is the
        # class token; the last patch token is never selected)
        idx_keep, _ = model_utils.random_token_mask(size=(batch_size, seq_length), mask_ratio=self.mask_ratio,
                                                    device=anchors.device, generator=generator)
        out = self.anchor_backbone(anchors, idx_keep)
        return self.anchor_projection_head(out)

    def training_step(self, batch, batch_idx, generator: torch.Generator = None):
        model_utils.update_momentum(self.anchor_backbone, self.backbone, 0.996)
        model_utils.update_momentum(self.anchor_projection_head, self.projection_head, 0.996)
        views = batch[0]
        targets, anchors = views[0], views[1]
        anchors_focal = torch.cat(list(views[2:]), dim=0)
        with torch.no_grad():
            targets_out = self.projection_head(self.backbone(targets))
        anchors_out = self.encode_masked(anchors, generator)
        anchors_focal_out = self.encode_masked(anchors_focal, generator)
        anchors_out = torch.cat([anchors_out, anchors_focal_out], dim=0)
        loss = self.criterion(anchors_out, targets_out, self.prototypes.data)
        self.log("train_loss_ssl", loss)
        if self.log_rep_std:
            self.log("rep_std", debug.std_of_l2_normalized(targets_out.detach().flatten(1)))
        return loss

    def configure_optimizers(self):
        params = [*self.anchor_backbone.parameters(), *self.anchor_projection_head.parameters(), self.prototypes]
        opt = optim.AdamW(params, lr=1.5e-4 * self.lr_factor, weight_decay=0.05, betas=(0.9, 0.95))
        return [opt], [scheduler.CosineWarmupScheduler(opt, self.warmup_epochs, self.max_epochs)]


class PMSN(MSN):
    loss_class = PMSNLoss
