"""kNN / NT-Xent oracles against independent implementations (sklearn, float64 closed forms)."""
import numpy as np
import pytest
import torch

from oracle import knn as ok
from oracle import ntxent as on


def test_ntxent_oracle_matches_closed_form():
    g = torch.Generator().manual_seed(0)
    for b, d in ((4, 8), (32, 128), (64, 128)):
        z0, z1 = torch.randn(b, d, generator=g), torch.randn(b, d, generator=g)
        a = on.ntxent_lightly(z0, z1, 0.5).double()
        c = on.ntxent_closed_form_f64(z0, z1, 0.5)
        assert abs(a - c) / c < 1e-6


def test_ntxent_gathered_equals_mean_of_rank_losses_on_full_batch():
    # two "ranks" of 8: mean over ranks of the gathered loss == loss of the 16-batch
    g = torch.Generator().manual_seed(1)
    z0, z1 = torch.randn(16, 32, generator=g), torch.randn(16, 32, generator=g)
    full = on.ntxent_lightly(z0, z1, 0.5)
    parts = [on.ntxent_lightly(z0[r * 8:(r + 1) * 8], z1[r * 8:(r + 1) * 8], 0.5, z0, z1, rank=r) for r in range(2)]
    assert abs(torch.stack(parts).mean() - full) < 1e-6


def test_knn_oracle_matches_sklearn_neighbours():
    from sklearn.neighbors import NearestNeighbors

    g = torch.Generator().manual_seed(2)
    bank = torch.nn.functional.normalize(torch.randn(2000, 64, generator=g), dim=1)
    q = torch.nn.functional.normalize(torch.randn(50, 64, generator=g), dim=1)
    labels = torch.randint(0, 9, (2000,), generator=g)
    sim, idx = ok.knn_topk(q, bank.t().contiguous(), 5)
    nn = NearestNeighbors(n_neighbors=5, metric="cosine").fit(bank.numpy())
    dist, ind = nn.kneighbors(q.numpy())
    assert np.array_equal(np.sort(ind, 1), np.sort(idx.numpy(), 1))
    np.testing.assert_allclose(1 - dist, sim.numpy(), atol=1e-5)
    pred = ok.knn_predict(q, bank.t().contiguous(), labels, 9, 5, 0.1)
    scores = ok.knn_scores(sim, idx, labels, 9, 0.1)
    assert torch.equal(pred[:, 0], scores.argmax(1))
    assert pred.shape == (50, 9) and sorted(pred[0].tolist()) == list(range(9))
