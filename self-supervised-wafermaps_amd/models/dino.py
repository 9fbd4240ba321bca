"""DINO on ViT-S/16: the reference's DINOViT (scripts/WM811k_benchmark.py:545-602; the MixedWM38
variant, MixedWM38_pretrain.py:138-199, differs by batch_norm=False in the heads).

One training step = momentum update of the teacher -> teacher on the 2 global crops (no grad) ->
student on all 8 crops -> DINOLoss -> AdamW.  The reference runs ten separate forwards; here the two
224^2 crops go through the ViT as one batch and the six 96^2 crops as another (a ViT has no batch
statistics, so this is the same function), and the projection heads see all views at once with
BatchNorm statistics kept per view (`ops.bn_groups`), which is exactly what per-view calls compute.
"""
from __future__ import annotations

import contextlib
import copy
import os

import torch

from .. import heads, ops, optim
from ..loss import DINOLoss
from ..utils import debug, model_utils, scheduler
from .knn import KNNBenchmarkModule
from .vit import vit_base, vit_small, vit_tiny

_TEACHER_STREAMS = {}   # device -> the side stream of the teacher's pass

_BACKBONES = {"vit_small": vit_small, "vit_tiny": vit_tiny, "vit_base": vit_base}


class DINOViT(KNNBenchmarkModule):
    def __init__(self, dataloader_kNN=None, num_classes=9, batch_size: int = 64, max_epochs: int = 150,
                 batch_norm: bool = True, log_rep_std: bool = True, backbone: str = "vit_small", **kwargs):
        super().__init__(dataloader_kNN, num_classes, **kwargs)
        # "vit_small" = the reference's dino_vits16 (:548-550); "vit_tiny" = BASELINE.json configs[2] (192-d, 3 heads)
        if backbone not in _BACKBONES:
            raise ValueError(f"DINOViT: unknown backbone {backbone!r} (have {sorted(_BACKBONES)})")
        self.backbone = _BACKBONES[backbone](patch_size=16)
        feature_dim = self.backbone.embed_dim
        self.head = heads.DINOProjectionHead(feature_dim, 2048, 256, 2048, batch_norm=batch_norm)
        self.teacher_backbone = copy.deepcopy(self.backbone)
        self.teacher_head = heads.DINOProjectionHead(feature_dim, 2048, 256, 2048, batch_norm=batch_norm)
        model_utils.deactivate_requires_grad(self.teacher_backbone)
        model_utils.deactivate_requires_grad(self.teacher_head)
        self.criterion = DINOLoss(output_dim=2048)
        self.warmup_epochs = 40 if max_epochs >= 800 else 20
        self.lr_factor = batch_size / 256
        self.max_epochs = max_epochs
        self.log_rep_std = log_rep_std

    def forward(self, x):
        y = self.backbone(x).flatten(start_dim=1)
        z = self.head(y)
        if self.log_rep_std:
            self.log("rep_std", debug.std_of_l2_normalized(y))
        return z

    def forward_teacher(self, x):
        y = self.teacher_backbone(x).flatten(start_dim=1)
        return self.teacher_head(y)

    @staticmethod
    def _group_by_size(views):
        """Consecutive views of equal resolution -> (first index, stacked batch)."""
        out, i = [], 0
        while i < len(views):
            j = i
            while j + 1 < len(views) and views[j + 1].shape == views[i].shape:
                j += 1
            out.append((i, j + 1))
            i = j + 1
        return out

    def _stack(self, views, i, j):
        if j - i == 1:
            return views[i]
        stacked = getattr(views, "stacked_groups", None)
        if stacked is not None and (i, j) in stacked:
            return stacked[(i, j)]
        return torch.cat(list(views[i:j]), dim=0)

    def training_step(self, batch, batch_idx):
        views = batch[0]
        b = views[0].shape[0]
        n_views = len(views)
        # The teacher's pass (no gradient; 2 global crops) is independent of the student's forward pass until the loss:
        # it is enqueued on a side stream, forked here and joined in front of the loss (inside a hipGraph capture: two
        # parallel branches).  Its launches fill a fraction of the chip each (197 row tiles on 256 CUs at ViT-Tiny), and
        # so do the student's.  `g` and `teacher_out` stay referenced until this function returns: the caching allocator
        # must not hand the memory one stream still reads to the other.  WM_DINO_TEACHER_STREAM=0: in line.
        g = self._stack(views, 0, 2)
        side = None
        if g.is_cuda and os.environ.get("WM_DINO_TEACHER_STREAM", "1") != "0":
            side = _TEACHER_STREAMS.get(g.device)
            if side is None:
                side = _TEACHER_STREAMS[g.device] = torch.cuda.Stream(device=g.device)
            side.wait_stream(torch.cuda.current_stream(g.device))
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            # (the teacher's momentum update belongs to its branch: only the teacher's pass waits for it; it reads the
            # student's parameters, which nothing writes before the optimiser step behind the join)
            model_utils.update_momentum(self.backbone, self.teacher_backbone, m=0.99)
            model_utils.update_momentum(self.head, self.teacher_head, m=0.99)
            with torch.no_grad(), ops.bn_groups(2):  # statistics per view wherever the networks have BatchNorm
                yt = self.teacher_backbone(g).flatten(start_dim=1)
                teacher_out = self.teacher_head(yt)
        groups = self._group_by_size(views)
        if len(groups) > 1 and hasattr(self.backbone, "forward_multi") and os.environ.get("WM_DINO_MERGE", "1") != "0":
            # ViT backbone: all resolutions through the blocks together (one launch per per-token layer)
            y = self.backbone.forward_multi([self._stack(views, i, j) for i, j in groups]).flatten(start_dim=1)
        else:
            feats = []
            for i, j in groups:
                with ops.bn_groups(j - i):
                    feats.append(self.backbone(self._stack(views, i, j)).flatten(start_dim=1))
            y = feats[0] if len(feats) == 1 else torch.cat(feats, dim=0)
        if self.log_rep_std:
            self.log("rep_std", debug.std_of_l2_normalized(y[-b:]))
        with ops.bn_groups(n_views):
            student_out = self.head(y)
        if side is not None:
            torch.cuda.current_stream(g.device).wait_stream(side)
        loss = self.criterion(teacher_out, student_out, epoch=self.current_epoch, batch=b)
        self.log("train_loss_ssl", loss)
        return loss

    def post_graph_step(self):
        """graph.GraphedTrainStep calls this after the replay: the cross-rank part of the DINO centre update."""
        self.criterion.finish_center_update()

    def configure_optimizers(self):
        param = list(self.backbone.parameters()) + list(self.head.parameters())
        opt = optim.AdamW(param, lr=1.5e-4 * self.lr_factor, weight_decay=0.05, betas=(0.9, 0.95))
        cosine = scheduler.CosineWarmupScheduler(opt, self.warmup_epochs, self.max_epochs)
        return [opt], [cosine]


class DINO(DINOViT):
    """DINO on ResNet-18 (scripts/WM811k_benchmark.py:491-543): the same step with the convolutional
    backbone (BatchNorm statistics per crop), SGD + cosine annealing instead of AdamW + warm-up."""

    def __init__(self, dataloader_kNN=None, num_classes=9, batch_size: int = 64, max_epochs: int = 150,
                 log_rep_std: bool = True, **kwargs):
        from .resnet import create_model

        KNNBenchmarkModule.__init__(self, dataloader_kNN, num_classes, **kwargs)
        self.backbone = create_model("resnet18", num_classes=0, pretrained=False)
        feature_dim = self.backbone.num_features
        self.head = heads.DINOProjectionHead(feature_dim, 2048, 256, 2048, batch_norm=True)
        self.teacher_backbone = copy.deepcopy(self.backbone)
        self.teacher_head = heads.DINOProjectionHead(feature_dim, 2048, 256, 2048, batch_norm=True)
        model_utils.deactivate_requires_grad(self.teacher_backbone)
        model_utils.deactivate_requires_grad(self.teacher_head)
        self.criterion = DINOLoss(output_dim=2048)
        self.lr_factor = batch_size / 256
        self.max_epochs = max_epochs
        self.log_rep_std = log_rep_std

    def configure_optimizers(self):
        param = list(self.backbone.parameters()) + list(self.head.parameters())
        opt = optim.SGD(param, lr=6e-2 * self.lr_factor, momentum=0.9, weight_decay=5e-4)
        return [opt], [torch.optim.lr_scheduler.CosineAnnealingLR(opt, self.max_epochs)]
