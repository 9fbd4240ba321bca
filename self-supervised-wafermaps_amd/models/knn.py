"""KNNBenchmarkModule: the reference's kNN-evaluation base class
(src/ssl_wafermap/models/knn.py:28-137) without the Lightning dependency.

Same constructor, same hook names and the same arithmetic: every validation epoch rebuilds an
L2-normalised feature bank from `dataloader_kNN`, predicts each validation batch with the weighted
kNN vote (HIP kernels: wm_l2_normalize, wm_knn_topk, wm_knn_vote), and reports macro accuracy /
macro F1 / a row-normalised confusion matrix (torchmetrics semantics, restated on int tensors).
The bank is kept [N, D] row-major (the reference's [D, N] is handed to knn_predict as a view).
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn as nn

from .. import functional as F_hip
from ..utils.benchmarking import knn_predict


def macro_metrics(preds: torch.Tensor, targets: torch.Tensor, num_classes: int):
    """MulticlassAccuracy(average="macro"), MulticlassF1Score(average="macro") and
    MulticlassConfusionMatrix(normalize="true") of torchmetrics on label tensors: classes without
    any support (and, for F1, without predictions either) are left out of the macro mean."""
    idx = targets.long() * num_classes + preds.long()
    cm = torch.bincount(idx, minlength=num_classes * num_classes).view(num_classes, num_classes).double()
    tp = cm.diag()
    support, predicted = cm.sum(1), cm.sum(0)
    has = support > 0
    acc = (tp[has] / support[has]).mean() if has.any() else torch.tensor(0.0)
    denom = 2 * tp + (predicted - tp) + (support - tp)
    seen = (support + predicted) > 0
    f1 = (2 * tp[seen] / denom[seen]).mean() if seen.any() else torch.tensor(0.0)
    cmn = cm / cm.sum(1, keepdim=True).clamp_min(1)
    return float(acc), float(f1), cmn.float()


class KNNBenchmarkModule(nn.Module):
    def __init__(self, dataloader_kNN, num_classes: int, knn_k: int = 5, knn_t: float = 0.1,
                 knn_dtype: torch.dtype = torch.float32):
        super().__init__()
        self.backbone = nn.Module()
        self.max_accuracy = 0.0
        self.max_f1 = 0.0
        self.dataloader_kNN = dataloader_kNN
        self.num_classes = num_classes
        self.knn_k = knn_k
        self.knn_t = knn_t
        self.knn_dtype = knn_dtype  # float32 = parity preset; bfloat16 = streaming-bandwidth preset
        self.confusion_matrix: List[torch.Tensor] = []
        self.all_preds: List[torch.Tensor] = []
        self.all_targets: List[torch.Tensor] = []
        self.logged: Dict[str, torch.Tensor] = {}
        self.current_epoch = 0

    # ---- minimal stand-ins for the Lightning surface the reference's modules touch
    def log(self, name, value, **_):
        self.logged[name] = value.detach() if torch.is_tensor(value) else value

    @property
    def device(self):
        return next(self.parameters()).device

    def _features(self, img: torch.Tensor) -> torch.Tensor:
        feature = self.backbone(img)
        if feature.dim() > 2:
            feature = feature.flatten(1)
        return F_hip.l2_normalize(feature.contiguous(), out_dtype=self.knn_dtype)

    @torch.no_grad()
    def on_validation_epoch_start(self):
        feats, targets = [], []
        for img, target in self.dataloader_kNN:
            feats.append(self._features(img.to(self.device)))
            targets.append(target.to(self.device))
        self.feature_bank_nd = torch.cat(feats, dim=0).contiguous()   # [N, D]
        self.feature_bank = self.feature_bank_nd.t()                  # [D, N] view, as the reference holds it
        self.targets_bank = torch.cat(targets, dim=0).long().contiguous()

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        images, targets = batch
        feature = self._features(images.to(self.device))
        pred_labels = knn_predict(feature, self.feature_bank, self.targets_bank, self.num_classes, self.knn_k,
                                  self.knn_t)
        self.all_preds.append(pred_labels[:, 0])
        self.all_targets.append(targets.to(self.device))

    def on_validation_epoch_end(self):
        preds, targets = torch.cat(self.all_preds), torch.cat(self.all_targets)
        acc, f1, cm = macro_metrics(preds, targets, self.num_classes)
        self.max_accuracy = max(self.max_accuracy, acc)
        self.max_f1 = max(self.max_f1, f1)
        self.log("knn_accuracy", acc)
        self.log("knn_f1", f1)
        self.confusion_matrix.append(cm.cpu().numpy())
        self.all_preds.clear()
        self.all_targets.clear()

    @torch.no_grad()
    def predict_step(self, batch, batch_idx):
        images, _ = batch
        return self.backbone(images.to(self.device))
