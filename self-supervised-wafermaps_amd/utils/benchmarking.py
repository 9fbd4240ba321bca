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
