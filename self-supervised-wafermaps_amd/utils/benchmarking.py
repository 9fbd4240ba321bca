"""kNN evaluation helpers with lightly's names (lightly.utils.benchmarking)."""
from __future__ import annotations

import torch

from .. import functional as F_hip


def knn_predict(feature: torch.Tensor, feature_bank: torch.Tensor, feature_labels: torch.Tensor,
                num_classes: int, knn_k: int = 200, knn_t: float = 0.1) -> torch.Tensor:
    """Drop-in for lightly.utils.benchmarking.knn_predict as the reference calls it
    (src/ssl_wafermap/models/knn.py:91-98).

    feature [B, D]; feature_bank [D, N] (the reference's transposed-contiguous bank: pass
    `bank_nd.t()` of a row-major [N, D] bank to avoid a copy); feature_labels [N] int64.
    Returns pred_labels [B, num_classes]: class ids by descending exp(sim/t)-weighted vote."""
    bank_nd = feature_bank.t()
    if not bank_nd.is_contiguous():
        bank_nd = bank_nd.contiguous()
    feature = feature.contiguous()
    if feature.dtype != bank_nd.dtype:
        feature = feature.to(bank_nd.dtype)
    sim, idx = F_hip.knn_topk(feature, bank_nd, knn_k)
    return F_hip.knn_vote(sim, idx, feature_labels.contiguous(), num_classes, knn_t)


def mean_topk_accuracy(predicted_classes: torch.Tensor, targets: torch.Tensor, k=(1, 5)):
    """lightly.utils.benchmarking.topk.mean_topk_accuracy: {k: fraction of rows whose target is among the first k
    predicted classes} (predicted_classes [B, num_classes] ordered best first)."""
    out = {}
    t = targets.to(predicted_classes.device).long().unsqueeze(1)
    for kk in k:
        kk_eff = min(int(kk), predicted_classes.shape[1])
        out[int(kk)] = (predicted_classes[:, :kk_eff] == t).any(dim=1).float().mean()
    return out


class KNNClassifier(torch.nn.Module):
    """lightly.utils.benchmarking.KNNClassifier (the class form of the evaluation the reference does with
    knn_predict inside its KNNBenchmarkModule, src/ssl_wafermap/models/knn.py:67-101; named by BASELINE.json's
    north_star) without the Lightning dependency: same constructor, same hook names.

        training_step(batch)        features of one batch of the TRAIN set (no gradient) -> the bank being collected
        on_validation_epoch_start() freeze the collected batches into the bank ([D, N] view of a row-major [N, D])
        validation_step(batch)      weighted kNN vote against the bank -> {"val_top1": ..., "val_top5": ...}
        fit_bank(loader) / predict(images)   the two steps as plain calls

    Features are L2-normalised (`normalize=True`) by wm_l2_normalize, ranked by wm_knn_topk and voted by wm_knn_vote;
    with torch.distributed initialised, on_validation_epoch_start all-gathers the ranks' feature batches like lightly's
    concat_all_gather (every rank ends up with the whole bank)."""

    def __init__(self, model: torch.nn.Module, num_classes: int, knn_k: int = 200, knn_t: float = 0.1,
                 topk=(1, 5), feature_dtype: torch.dtype = torch.float32, normalize: bool = True):
        super().__init__()
        self.model = model
        self.num_classes, self.knn_k, self.knn_t = num_classes, knn_k, knn_t
        self.topk, self.feature_dtype, self.normalize = tuple(topk), feature_dtype, normalize
        self._train_features, self._train_targets = [], []
        self._train_features_tensor = None   # [D, N], as lightly holds it
        self._train_targets_tensor = None
        self.logged = {}

    def log_dict(self, d, **_):
        self.logged.update({k: (v.detach() if torch.is_tensor(v) else v) for k, v in d.items()})

    def _features(self, images: torch.Tensor) -> torch.Tensor:
        feats = self.model.forward(images).flatten(start_dim=1)
        if self.normalize:
            return F_hip.l2_normalize(feats.contiguous(), out_dtype=self.feature_dtype)
        return feats.to(self.feature_dtype).contiguous()

    def on_train_epoch_start(self) -> None:
        self._train_features, self._train_targets = [], []
        self._train_features_tensor = self._train_targets_tensor = None

    @torch.no_grad()
    def training_step(self, batch, batch_idx: int = 0) -> None:
        images, targets = batch[0], batch[1]
        self._train_features.append(self._features(images))
        self._train_targets.append(targets.to(images.device).long())

    @torch.no_grad()
    def on_validation_epoch_start(self) -> None:
        if not (self._train_features and self._train_targets):
            return
        feats = torch.cat(self._train_features, dim=0).contiguous()
        targs = torch.cat(self._train_targets, dim=0).contiguous()
        self._train_features, self._train_targets = [], []
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from ..distributed import all_gather_rows

            feats, targs = all_gather_rows(feats), all_gather_rows(targs)
        self._train_features_tensor = feats.t()   # [D, N] view of the row-major bank (no copy in knn_predict)
        self._train_targets_tensor = targs

    @torch.no_grad()
    def predict(self, images: torch.Tensor) -> torch.Tensor:
        """[B, num_classes] class ids, best first."""
        if self._train_features_tensor is None:
            raise RuntimeError("KNNClassifier: no feature bank (run training_step over the train set, then "
                               "on_validation_epoch_start)")
        return knn_predict(self._features(images), self._train_features_tensor, self._train_targets_tensor,
                           self.num_classes, min(self.knn_k, self._train_targets_tensor.numel()), self.knn_t)

    @torch.no_grad()
    def validation_step(self, batch, batch_idx: int = 0):
        if self._train_features_tensor is None or self._train_targets_tensor is None:
            return None
        images, targets = batch[0], batch[1]
        pred = self.predict(images)
        acc = mean_topk_accuracy(pred, targets, k=self.topk)
        log = {f"val_top{k}": v for k, v in acc.items()}
        self.log_dict(log, batch_size=len(targets))
        return pred

    def fit_bank(self, loader) -> "KNNClassifier":
        self.on_train_epoch_start()
        for i, batch in enumerate(loader):
            self.training_step(batch, i)
        self.on_validation_epoch_start()
        return self

    def configure_optimizers(self):
        return None  # nothing is trained
