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

from .. import distributed as wdist
from .. import functional as F_hip
from ..utils.benchmarking import knn_predict


def rank_batches(loader):
    """This rank's share of an evaluation loader: with torch.distributed initialised and a loader that is not
    itself rank-sliced, the contiguous range [rank * n / world, (rank + 1) * n / world) of its batches (so that the
    concatenation over the ranks, in rank order, is the single-process order); otherwise every batch."""
    w, r = wdist.world_size(), wdist.rank()
    if w == 1 or getattr(loader, "world_size", 1) > 1:
        yield from loader
        return
    nb = len(loader)
    lo, hi = r * nb // w, (r + 1) * nb // w
    if hasattr(loader, "batches_in"):
        yield from loader.batches_in(lo, hi)   # images are built for the owned batches only
        return
    for bi, batch in enumerate(loader):
        if lo <= bi < hi:
            yield batch


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
        """Rebuild the feature bank (reference :67-85).  Data parallel: every rank embeds only its contiguous share of
        the kNN loader's batches; ONE all-gather of the shards (37 k x 512 floats for WM-811K: few, large collectives
        suit the point-to-point xGMI links) leaves the whole bank, in single-process order, on every rank, so the
        validation batches need no further exchange."""
        feats, targets = [], []
        for img, target in rank_batches(self.dataloader_kNN):
            feats.append(self._features(img.to(self.device)))
            targets.append(target.to(self.device))
        # (a rank that owns no batch -- fewer batches than ranks -- contributes zero rows; all_gather_rows gives it the
        # feature width and dtype of the ranks that have some)
        bank = (torch.cat(feats, dim=0).contiguous() if feats
                else torch.empty((0, 0), dtype=self.knn_dtype, device=self.device))
        tbank = (torch.cat(targets, dim=0).long().contiguous() if targets
                 else torch.empty((0,), dtype=torch.long, device=self.device))
        if wdist.world_size() > 1 and getattr(self.dataloader_kNN, "world_size", 1) == 1:
            bank, tbank = wdist.all_gather_rows(bank), wdist.all_gather_rows(tbank)
        self.feature_bank_nd = bank.contiguous()                      # [N, D]
        self.feature_bank = self.feature_bank_nd.t()                  # [D, N] view, as the reference holds it
        self.targets_bank = tbank

    def shard_eval_batches(self, loader):
        """(batch index, batch) pairs this rank validates (Trainer.validate): its share of the loader's batches."""
        return enumerate(rank_batches(loader))

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        images, targets = batch
        feature = self._features(images.to(self.device))
        pred_labels = knn_predict(feature, self.feature_bank, self.targets_bank, self.num_classes, self.knn_k,
                                  self.knn_t)
        self.all_preds.append(pred_labels[:, 0])
        self.all_targets.append(targets.to(self.device))

    def on_validation_epoch_end(self):
        empty = torch.empty((0,), dtype=torch.long, device=self.device)
        preds = torch.cat(self.all_preds) if self.all_preds else empty
        targets = torch.cat(self.all_targets).long() if self.all_targets else empty
        if wdist.world_size() > 1:   # the ranks validated disjoint shares: the metrics are over all of them
            preds, targets = wdist.all_gather_rows(preds.contiguous()), wdist.all_gather_rows(targets.contiguous())
        acc, f1, cm = macro_metrics(preds, targets, self.num_classes)
        self.max_accuracy = max(self.max_accuracy, acc)
        self.max_f1 = max(self.max_f1, f1)
        self.log("knn_accuracy", acc)
        self.log("knn_f1", f1)
        self.confusion_matrix.append(cm.cpu().numpy())
        self.last_preds, self.last_targets = preds, targets
        self.all_preds.clear()
        self.all_targets.clear()

    @torch.no_grad()
    def predict_step(self, batch, batch_idx):
        images, _ = batch
        return self.backbone(images.to(self.device))
