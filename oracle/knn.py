"""CPU oracle for the kNN evaluation path.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates lightly.utils.benchmarking.knn_predict — a third-party dependency (`lightly`, UNPINNED in
the reference's requirements.txt:1, not installed here) — from its published algorithm (Wu et al.
2018, arXiv:1805.01978, weighted kNN), anchored on the reference call site
src/ssl_wafermap/models/knn.py:91-98 and the bank build knn.py:67-81.
Parity status: PARITY UNPINNED upstream (the reference holds no test or golden vector for this
path); cross-checked here against sklearn.neighbors and a float64 closed form (tests/).
"""
import torch
import torch.nn.functional as F


def build_bank(features: torch.Tensor) -> torch.Tensor:
    """knn.py:76-80: F.normalize(dim=1), concatenated, transposed-contiguous -> [D, N]."""
    return F.normalize(features, dim=1).t().contiguous()


def knn_predict(feature, feature_bank, feature_labels, num_classes, knn_k=200, knn_t=0.1):
    """feature [B,D], feature_bank [D,N], feature_labels [N] -> pred_labels [B,C] (+ internals)."""
    sim_matrix = torch.mm(feature, feature_bank)
    sim_weight, sim_indices = sim_matrix.topk(k=knn_k, dim=-1)
    sim_labels = torch.gather(feature_labels.expand(feature.size(0), -1), dim=-1, index=sim_indices)
    sim_weight = (sim_weight / knn_t).exp()
    one_hot_label = torch.zeros(feature.size(0) * knn_k, num_classes)
    one_hot_label = one_hot_label.scatter(dim=-1, index=sim_labels.view(-1, 1), value=1.0)
    pred_scores = torch.sum(one_hot_label.view(feature.size(0), -1, num_classes) * sim_weight.unsqueeze(dim=-1), dim=1)
    pred_labels = pred_scores.argsort(dim=-1, descending=True)
    return pred_labels


def knn_topk(feature, feature_bank, knn_k):
    sim = torch.mm(feature, feature_bank)
    return sim.topk(k=knn_k, dim=-1)


def knn_scores(sim_topk, idx_topk, feature_labels, num_classes, knn_t):
    w = (sim_topk / knn_t).exp()
    labels = feature_labels[idx_topk]
    scores = torch.zeros(sim_topk.shape[0], num_classes, dtype=w.dtype)
    scores.scatter_add_(1, labels, w)
    return scores
