"""Supervised baseline and linear probes: the reference's SupervisedR18 (scripts/WM811k_benchmark.py:202-225)
and LinearClassifier / MultilabelLinearClassifier (src/ssl_wafermap/models/evals.py:14-165; drivers
scripts/WM811k_linear_probe.py:286-385, scripts/MixedWM38_evals.py:740-870) — SURVEY 8f.4.

The probes train one Linear layer on frozen features (e.g. `retrieval.embed_dataset` output) with Adam;
`fit_linear_probe` is the loop Lightning's Trainer runs for them: epochs over shuffled mini-batches, then
macro accuracy / F1 (multi-class) or per-label accuracy / macro F1 at threshold 0 (multi-label).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import nn as hnn
from .. import optim
from ..loss import BCEWithLogitsLoss, CrossEntropyLoss
from ..utils import debug
from .knn import KNNBenchmarkModule, macro_metrics
from .resnet import create_model


class SupervisedR18(KNNBenchmarkModule):
    def __init__(self, dataloader_kNN=None, num_classes=9, log_rep_std: bool = True, **kwargs):
        kwargs.pop("batch_size", None)
        kwargs.pop("max_epochs", None)
        super().__init__(dataloader_kNN, num_classes, **kwargs)
        self.backbone = create_model("resnet18", num_classes=0, pretrained=False)
        self.fc = hnn.Linear(self.backbone.num_features, num_classes, bias=True)
        self.criterion = CrossEntropyLoss()
        self.log_rep_std = log_rep_std

    def forward(self, x):
        """Class logits (the reference returns their log-softmax and applies nll_loss: the same loss)."""
        f = self.backbone(x).flatten(start_dim=1)
        if self.log_rep_std:
            self.log("rep_std", debug.std_of_l2_normalized(f.detach()))
        return self.fc(f)

    def training_step(self, batch, batch_idx):
        x, y = batch
        if isinstance(x, (list, tuple)):
            x = x[0]
        loss = self.criterion(self.forward(x), y)
        self.log("train_loss", loss)
        return loss

    def configure_optimizers(self):
        return [optim.AdamW(self.parameters())], []  # torch.optim.AdamW defaults: lr 1e-3, wd 1e-2


class LinearClassifier(nn.Module):
    def __init__(self, num_features: int, num_classes: int = 9, weight: Optional[torch.Tensor] = None):
        super().__init__()
        self.model = hnn.Linear(num_features, num_classes, bias=True)
        self.criterion = CrossEntropyLoss(weight=weight)
        self.num_classes = num_classes

    def forward(self, x):
        return self.model(x)

    def training_step(self, batch, batch_idx):
        x, y = batch
        return self.criterion(self(x), y)

    def configure_optimizers(self):
        return optim.AdamW(self.parameters(), lr=1e-3, weight_decay=0.0)  # == torch.optim.Adam(lr=1e-3)

    @torch.no_grad()
    def evaluate(self, x, y):
        pred = self(x).float().argmax(dim=1)
        acc, f1, cm = macro_metrics(pred, y, self.num_classes)
        return {"acc": acc, "f1": f1, "confusion_matrix": cm}


class MultilabelLinearClassifier(nn.Module):
    def __init__(self, num_features: int, num_classes: int = 8, pos_weight: Optional[torch.Tensor] = None):
        super().__init__()
        self.model = hnn.Linear(num_features, num_classes, bias=True)
        self.criterion = BCEWithLogitsLoss(pos_weight=pos_weight)
        self.num_classes = num_classes

    def forward(self, x):
        return self.model(x)

    def training_step(self, batch, batch_idx):
        x, y = batch
        return self.criterion(self(x), y.float())

    def configure_optimizers(self):
        return optim.AdamW(self.parameters(), lr=1e-3, weight_decay=0.0)

    @torch.no_grad()
    def evaluate(self, x, y):
        pred = (self(x).float() > 0).long()
        y = y.long()
        tp = (pred & y).sum(0).double()
        fp = (pred & (1 - y)).sum(0).double()
        fn = ((1 - pred) & y).sum(0).double()
        f1 = torch.where(2 * tp + fp + fn > 0, 2 * tp / (2 * tp + fp + fn).clamp_min(1), torch.zeros_like(tp))
        return {"acc": float((pred == y).double().mean()), "f1": float(f1.mean())}


def fit_linear_probe(model: nn.Module, features: torch.Tensor, labels: torch.Tensor, epochs: int = 10,
                     batch_size: int = 256, seed: int = 0):
    """Train a (Multilabel)LinearClassifier on frozen features [N, D] (any float dtype, on the GPU)."""
    opt = model.configure_optimizers()
    gen = torch.Generator(device="cpu").manual_seed(seed)
    n = features.shape[0]
    feats = features.to(torch.bfloat16).contiguous()
    losses = []
    for _ in range(epochs):
        perm = torch.randperm(n, generator=gen).to(features.device)
        total, steps = 0.0, 0
        for s in range(0, n, batch_size):
            idx = perm[s:s + batch_size]
            opt.zero_grad()
            loss = model.training_step((feats[idx], labels[idx]), steps)
            loss.backward()
            opt.step()
            total += float(loss.detach())
            steps += 1
        losses.append(total / max(steps, 1))
    return losses
