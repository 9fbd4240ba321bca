"""GPU parity: kNN top-k / vote, L2 normalise and NT-Xent HIP kernels against the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import knn as ok
from oracle import ntxent as on


DEV = "cuda:0"


def _dev():
    return torch.device("cuda:0")


def _check_topk(sim, idx, sim_ref, idx_ref, full_sim_ref, tol):
    """Indices must agree wherever the reference ranking is decided by more than `tol`."""
    sim, idx = sim.cpu(), idx.cpu().long()
    torch.testing.assert_close(sim, sim_ref, atol=tol, rtol=0)
    # every returned index carries the similarity the reference assigns to it
    got = torch.gather(full_sim_ref, 1, idx)
    torch.testing.assert_close(got, sim_ref, atol=tol, rtol=0)
    k = sim_ref.shape[1]
    kth_gap = torch.ones(sim_ref.shape[0], dtype=torch.bool) if full_sim_ref.shape[1] == k else \
        (sim_ref[:, -1] - full_sim_ref.topk(k + 1, dim=1).values[:, -1]) > 4 * tol
    gaps = (sim_ref[:, :-1] - sim_ref[:, 1:]) > 4 * tol
    decided = gaps.all(1) & kth_gap
    assert decided.float().mean() > 0.5
    assert torch.equal(idx[decided], idx_ref[decided])
    assert (sim[:, :-1] >= sim[:, 1:]).all()


@pytest.mark.parametrize("nq,n,d,k", [(64, 9959, 512, 5), (33, 129, 128, 8), (1, 5, 64, 5), (70, 4000, 128, 10)])
def test_knn_topk_f32(nq, n, d, k):
    from ssl_wafermap_amd import functional as F

    g = torch.Generator().manual_seed(nq + n)
    bank = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1)
    q = torch.nn.functional.normalize(torch.randn(nq, d, generator=g), dim=1)
    full = q @ bank.t()
    sim_ref, idx_ref = full.topk(k, dim=1)
    sim, idx = F.knn_topk(q.to(_dev()), bank.to(_dev()), k)
    _check_topk(sim, idx, sim_ref, idx_ref, full, 2e-6)


# (64, 262235, 128, 8): 2 049 chunks on 410 slices (a last round that only some slices take part in)
@pytest.mark.parametrize("nq,n,d,k", [(256, 50000, 128, 8), (64, 12449, 512, 5), (100, 3000, 256, 16), (64, 262235, 128, 8),
                                      (40, 262235, 128, 16)])
def test_knn_topk_bf16(nq, n, d, k):
    from ssl_wafermap_amd import functional as F

    g = torch.Generator().manual_seed(7)
    bank = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1).bfloat16()
    q = torch.nn.functional.normalize(torch.randn(nq, d, generator=g), dim=1).bfloat16()
    full = q.float() @ bank.float().t()  # exact products of the bf16 values, f32 accumulation
    sim_ref, idx_ref = full.topk(k, dim=1)
    sim, idx = F.knn_topk(q.to(_dev()), bank.to(_dev()), k)
    _check_topk(sim, idx, sim_ref, idx_ref, full, 3e-6)
    # and against the f32 features: within the north-star's 1e-3 cosine
    assert (sim.cpu() - sim_ref).abs().max() < 1e-3


def test_knn_duplicates_tie_break_lower_index():
    from ssl_wafermap_amd import functional as F

    g = torch.Generator().manual_seed(3)
    base = torch.nn.functional.normalize(torch.randn(300, 128, generator=g), dim=1)
    bank = torch.cat([base, base, base])  # every row appears three times
    sim, idx = F.knn_topk(base[:40].contiguous().to(_dev()), bank.to(_dev()), 3)
    idx = idx.cpu()
    for i in range(40):
        assert idx[i].tolist() == [i, i + 300, i + 600]


def test_knn_sharded_merge_equals_unsharded():
    from ssl_wafermap_amd import functional as F

    g = torch.Generator().manual_seed(4)
    bank = torch.nn.functional.normalize(torch.randn(4096 + 77, 128, generator=g), dim=1).to(_dev())
    q = torch.nn.functional.normalize(torch.randn(48, 128, generator=g), dim=1).to(_dev())
    sim, idx = F.knn_topk(q, bank, 5)
    bounds = [0, 1000, 2048, 3000, bank.shape[0]]
    parts = [F.knn_topk(q, bank[a:b].contiguous(), 5, index_base=a) for a, b in zip(bounds[:-1], bounds[1:])]
    msim, midx = F.knn_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert torch.equal(midx, idx) and torch.equal(msim, sim)


def test_knn_predict_matches_oracle():
    from ssl_wafermap_amd.utils.benchmarking import knn_predict

    g = torch.Generator().manual_seed(5)
    feats = torch.randn(9959, 512, generator=g)
    labels = torch.randint(0, 9, (9959,), generator=g)
    # clustered features so that the vote is meaningful
    feats += 3 * torch.nn.functional.one_hot(labels, 512).float()
    bank = ok.build_bank(feats)  # [D, N]
    q = torch.nn.functional.normalize(feats[:64] + 0.1 * torch.randn(64, 512, generator=g), dim=1)
    ref = ok.knn_predict(q, bank, labels, 9, 5, 0.1)
    bank_nd = bank.t().contiguous().to(_dev())
    got = knn_predict(q.to(_dev()), bank_nd.t(), labels.to(_dev()), 9, 5, 0.1).cpu()
    assert torch.equal(got[:, 0], ref[:, 0])
    sim, idx = ok.knn_topk(q, bank, 5)
    scores = ok.knn_scores(sim, idx, labels, 9, 0.1)
    from ssl_wafermap_amd import functional as F

    s2, i2 = F.knn_topk(q.to(_dev()), bank_nd, 5)
    _, sc = F.knn_vote(s2, i2, labels.to(_dev()), 9, 0.1, return_scores=True)
    torch.testing.assert_close(sc.cpu(), scores, rtol=2e-4, atol=1e-3)


def test_knn_selection_kernel_forms_agree(monkeypatch):
    """knn_select with 1 024 threads per query (default for k <= 8: one rescoring round) and with 256 (four rounds):
    the same candidates, the same exact rescoring -> identical results, bf16 and float32."""
    from ssl_wafermap_amd import functional as F

    g = torch.Generator().manual_seed(11)
    bank = torch.nn.functional.normalize(torch.randn(70001, 128, generator=g), dim=1)
    q = torch.nn.functional.normalize(torch.randn(96, 128, generator=g), dim=1)
    for dt in (torch.bfloat16, torch.float32):
        b, qq = bank.to(dt).to(_dev()), q.to(dt).to(_dev())
        out = {}
        for th in ("256", "1024"):
            monkeypatch.setenv("WM_KNN_SELECT_THREADS", th)  # (read per call)
            out[th] = F.knn_topk(qq, b, 8)
            torch.cuda.synchronize()
        assert torch.equal(out["256"][0], out["1024"][0]) and torch.equal(out["256"][1], out["1024"][1])


def test_knn_full_size_properties():
    """BASELINE scale (811 457 x 128, bf16): self-retrieval, sortedness, spot check vs CPU."""
    from ssl_wafermap_amd import functional as F

    n, d = 811457, 128
    g = torch.Generator(device="cuda").manual_seed(7)
    bank = torch.nn.functional.normalize(torch.randn(n, d, generator=g, device=_dev()), dim=1).bfloat16()
    rows = torch.tensor([0, 1, 127, 128, 4095, 400000, 811455, 811456], device=_dev())
    q = torch.cat([bank[rows], bank[100000:100056]]).contiguous()
    sim, idx = F.knn_topk(q, bank, 8)
    assert torch.equal(idx[:8, 0].long(), rows)
    assert (sim[:, 0] > 0.99).all() and (sim[:, :-1] >= sim[:, 1:]).all()
    assert int(idx.min()) >= 0 and int(idx.max()) < n
    full = (q[:4].float().cpu() @ bank.float().cpu().t())
    sref, iref = full.topk(8, dim=1)
    torch.testing.assert_close(sim[:4].cpu(), sref, atol=3e-6, rtol=0)
    # every query against a float32 product on the device (the shares of the streaming blocks must cover every bank
    # row exactly once: a dropped or doubled chunk shows here)
    full_all = q.float() @ bank.float().t()
    sref, iref = full_all.topk(8, dim=1)
    _check_topk(sim, idx, sref.cpu(), iref.cpu(), full_all.cpu(), 3e-6)
    del full_all


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_l2_normalize(dtype):
    from ssl_wafermap_amd import functional as F

    g = torch.Generator().manual_seed(0)
    x = (torch.randn(300, 512, generator=g) * 3).to(dtype)
    x[7] = 0  # zero row: eps clamp
    ref = torch.nn.functional.normalize(x.float(), dim=1)
    y = F.l2_normalize(x.to(_dev()))
    torch.testing.assert_close(y.cpu(), ref, atol=1e-6, rtol=1e-6)
    yb = F.l2_normalize(x.to(_dev()), out_dtype=torch.bfloat16)
    torch.testing.assert_close(yb.cpu().float(), ref.bfloat16().float(), atol=1e-2, rtol=0)


def test_l2_normalize_backward():
    from ssl_wafermap_amd import functional as F

    g = torch.Generator().manual_seed(1)
    x = torch.randn(64, 128, generator=g)
    w = torch.randn(64, 128, generator=g)
    xr = x.clone().requires_grad_(True)
    (torch.nn.functional.normalize(xr, dim=1) * w).sum().backward()
    xd = x.to(_dev()).requires_grad_(True)
    (F.l2_normalize(xd) * w.to(_dev())).sum().backward()
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("b,d,t", [(64, 128, 0.5), (256, 128, 0.5), (37, 64, 0.1), (8, 256, 0.07)])
def test_ntxent_loss_and_grad(b, d, t):
    from ssl_wafermap_amd.loss import NTXentLoss

    g = torch.Generator().manual_seed(b)
    z0, z1 = torch.randn(b, d, generator=g), torch.randn(b, d, generator=g)
    r0, r1 = z0.clone().requires_grad_(True), z1.clone().requires_grad_(True)
    ref = on.ntxent_lightly(r0, r1, t)
    ref.backward()
    d0, d1 = z0.to(_dev()).requires_grad_(True), z1.to(_dev()).requires_grad_(True)
    loss = NTXentLoss(temperature=t)(d0, d1)
    loss.backward()
    assert abs(loss.item() - ref.item()) / abs(ref.item()) < 1e-5  # north-star: 1e-4 relative
    c64 = on.ntxent_closed_form_f64(z0, z1, t).item()
    assert abs(loss.item() - c64) / abs(c64) < 1e-5
    scale = r0.grad.abs().max().item()
    torch.testing.assert_close(d0.grad.cpu(), r0.grad, atol=2e-5 * scale, rtol=1e-4)
    torch.testing.assert_close(d1.grad.cpu(), r1.grad, atol=2e-5 * scale, rtol=1e-4)


def test_ntxent_gathered_semantics_single_process():
    """Kernel with b_global > b_local == lightly's gather_distributed branch incl. GatherLayer grads."""
    from ssl_wafermap_amd import functional as F

    g = torch.Generator().manual_seed(9)
    world, b, d, t = 2, 8, 64, 0.5
    z0, z1 = torch.randn(world * b, d, generator=g), torch.randn(world * b, d, generator=g)
    zr0, zr1 = z0.clone().requires_grad_(True), z1.clone().requires_grad_(True)
    losses = [on.ntxent_lightly(zr0[r * b:(r + 1) * b], zr1[r * b:(r + 1) * b], t, zr0, zr1, rank=r) for r in range(world)]
    torch.stack(losses).sum().backward()  # GatherLayer.backward all-reduces (sums) over ranks
    zn_all = torch.nn.functional.normalize(torch.cat([z0, z1]), dim=1).to(_dev())  # [2][Bg]
    zn_all.requires_grad_(False)
    lse_parts, row_parts = [], []
    for r in range(world):
        zn_loc = torch.cat([zn_all[r * b:(r + 1) * b], zn_all[world * b + r * b: world * b + (r + 1) * b]]).contiguous()
        lse, rows = F.ntxent_forward(zn_loc, zn_all, b, world * b, r * b, t)
        assert abs(rows.mean().item() - losses[r].item()) / losses[r].item() < 1e-5
        lse_parts.append(lse)
    lse_all = torch.cat([torch.cat([p[:b] for p in lse_parts]), torch.cat([p[b:] for p in lse_parts])])
    # reference gradient wrt the normalised rows of rank 1
    zn_ref = torch.nn.functional.normalize(torch.cat([z0, z1]), dim=1).clone().requires_grad_(True)

    def total(zn):
        n0, n1 = zn[: world * b], zn[world * b:]
        return sum(on.ntxent_lightly(n0[r * b:(r + 1) * b], n1[r * b:(r + 1) * b], t, n0, n1, rank=r) for r in range(world))

    # normalising already-normalised rows is the identity up to rounding but its Jacobian projects;
    # compare in the tangent space instead: project both gradients
    total(zn_ref).backward()
    r = 1
    zn_loc = torch.cat([zn_all[r * b:(r + 1) * b], zn_all[world * b + r * b: world * b + (r + 1) * b]]).contiguous()
    dzn = F.ntxent_backward(zn_loc, zn_all, lse_all, b, world * b, r * b, t, 1.0 / (2 * b)).cpu()
    zl = zn_loc.cpu()
    dzn_proj = dzn - zl * (dzn * zl).sum(1, keepdim=True)
    gref = torch.cat([zn_ref.grad[r * b:(r + 1) * b], zn_ref.grad[world * b + r * b: world * b + (r + 1) * b]])
    torch.testing.assert_close(dzn_proj, gref, atol=2e-6, rtol=1e-4)


@pytest.mark.parametrize("nq,n,d,k,dtype", [(3, 1, 128, 1, torch.bfloat16), (5, 8, 64, 8, torch.float32),
                                            (130, 127, 128, 5, torch.bfloat16), (2, 300, 512, 3, torch.bfloat16)])
def test_knn_edge_sizes(nq, n, d, k, dtype):
    """Tiny banks (n < one 128-row chunk, n == k), single queries, ragged query tiles, 1-KB bf16 rows."""
    from ssl_wafermap_amd import functional as F

    g = torch.Generator().manual_seed(n * 7 + nq)
    bank = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1).to(dtype)
    q = torch.nn.functional.normalize(torch.randn(nq, d, generator=g), dim=1).to(dtype)
    full = q.float() @ bank.float().t()
    sim_ref, idx_ref = full.topk(k, dim=1)
    sim, idx = F.knn_topk(q.to(_dev()), bank.to(_dev()), k)
    torch.testing.assert_close(sim.cpu(), sim_ref, atol=5e-6, rtol=0)
    assert torch.equal(torch.sort(idx.cpu().long(), 1).values, torch.sort(idx_ref, 1).values) or k < n
    with pytest.raises(Exception):
        F.knn_topk(q.to(_dev()), bank.to(_dev()), n + 1)  # k > n is rejected, not silently clipped


def test_standard_scaler_and_retrieval_match_sklearn():
    """StandardScaler + nearest neighbours (cosine, L2) against sklearn on the same data."""
    from sklearn.neighbors import NearestNeighbors
    from sklearn.preprocessing import StandardScaler as SkScaler

    from ssl_wafermap_amd.retrieval import StandardScaler, nearest_neighbors

    g = torch.Generator().manual_seed(0)
    x = torch.randn(3000, 512, generator=g) * (torch.rand(512, generator=g) * 3 + 0.1) + torch.randn(512, generator=g)
    x[:, 7] = 2.5  # constant feature: sklearn leaves it unscaled
    x = x.half().float()  # the reference stores float16 embeddings
    ref = SkScaler().fit(x.numpy().astype("float64"))
    sc = StandardScaler().fit(x.to("cuda:0"))
    assert np.allclose(sc.mean_.cpu().numpy(), ref.mean_, rtol=1e-5, atol=1e-5)
    assert np.allclose(sc.var_.cpu().numpy(), ref.var_, rtol=1e-4, atol=1e-6)
    z = sc.transform(x.to("cuda:0"))
    zr = ref.transform(x.numpy().astype("float64"))
    assert np.allclose(z.cpu().numpy(), zr, rtol=1e-4, atol=1e-4)  # incl. the constant column: 0 (f32 mean error 1e-6)
    xb = x.bfloat16()  # bf16 features straight from the backbone: same arithmetic on the rounded values
    zb = sc.transform(xb.to("cuda:0"))
    assert np.allclose(zb.cpu().numpy(), ref.transform(xb.float().numpy().astype("float64")), rtol=1e-4, atol=1e-4)
    q = z[:40] + 0.05 * torch.randn(40, 512, generator=g).to(z.device)
    for metric, sk_metric in (("cosine", "cosine"), ("l2", "euclidean")):
        dist, idx = nearest_neighbors(q, z, 10, metric=metric)
        nn_ = NearestNeighbors(n_neighbors=10, metric=sk_metric, algorithm="brute").fit(z.cpu().numpy().astype("float64"))
        dref, iref = nn_.kneighbors(q.cpu().numpy().astype("float64"))
        assert (idx[:, 0].cpu().numpy() == iref[:, 0]).all()
        agree = np.mean([len(set(a.tolist()) & set(b.tolist())) / 10 for a, b in zip(idx.cpu().numpy(), iref)])
        assert agree > 0.99, (metric, agree)
        assert np.allclose(dist.cpu().numpy()[:, :3], dref[:, :3], rtol=2e-3, atol=2e-3)


def test_embed_dataset_runs_the_eval_backbone():
    from ssl_wafermap_amd.data import WaferLoader, WaferMapDataset
    from ssl_wafermap_amd.data.synthetic import synthetic_wafers
    from ssl_wafermap_amd.models import SimCLR
    from ssl_wafermap_amd.retrieval import embed_dataset
    from ssl_wafermap_amd.transforms import InferenceTransform

    wafers, labels = synthetic_wafers(50, seed=2)
    ds = WaferMapDataset(wafers, labels, transform=InferenceTransform(), device="cuda:0")
    torch.manual_seed(0)
    model = SimCLR(None, 9).to("cuda:0").train()
    feats = embed_dataset(model, WaferLoader(ds, 16), out_dtype=torch.float16)
    assert feats.shape == (50, 512) and feats.dtype == torch.float16 and torch.isfinite(feats.float()).all()
    assert model.training  # mode restored
    again = embed_dataset(model, WaferLoader(ds, 25), out_dtype=None)
    assert torch.allclose(again.float(), feats.float(), atol=2e-2, rtol=2e-2)  # batch size does not matter in eval


def test_ntxent_memory_bank_matches_oracle_and_enqueues_like_lightly():
    from ssl_wafermap_amd.loss import NTXentLoss

    g = torch.Generator().manual_seed(3)
    b, d, k = 48, 128, 200
    crit = NTXentLoss(temperature=0.1, memory_bank_size=k).to("cuda:0")
    ref_bank, ref_ptr = None, 0
    for step in range(6):  # 6 x 48 = 288 > 200: exercises the wrap-around
        out0 = torch.randn(b, d, generator=g)
        out1 = out0 + 0.3 * torch.randn(b, d, generator=g)
        o0 = out0.to("cuda:0").requires_grad_(True)
        o1 = out1.to("cuda:0").requires_grad_(True)
        if ref_bank is None:
            crit._init_memory_bank(d, torch.device("cuda:0"))
            ref_bank = crit.bank.cpu().clone()
        loss = crit(o0, o1)
        loss.backward()
        r0, r1 = out0.clone().requires_grad_(True), out1.clone().requires_grad_(True)
        ref = on.ntxent_memory_bank(r0, r1, ref_bank, 0.1)
        ref.backward()
        assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-5 * abs(float(ref.detach())), (step, float(loss), float(ref))
        assert torch.allclose(o0.grad.cpu(), r0.grad, rtol=1e-4, atol=1e-7)
        assert torch.allclose(o1.grad.cpu(), r1.grad, rtol=1e-4, atol=1e-7)
        ref_ptr = on.memory_bank_enqueue(ref_bank, ref_ptr, torch.nn.functional.normalize(out1, dim=1))
        assert int(crit.bank_ptr) == ref_ptr
        assert torch.allclose(crit.bank.cpu(), ref_bank, atol=1e-6)
    # no gradient on out0 -> lightly leaves the bank alone
    before = crit.bank.clone()
    with torch.no_grad():
        crit(torch.randn(b, d, generator=g).to("cuda:0"), torch.randn(b, d, generator=g).to("cuda:0"))
    assert torch.equal(before, crit.bank)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_negative_cosine_similarity_matches_torch(dtype):
    from ssl_wafermap_amd.loss import NegativeCosineSimilarity

    g = torch.Generator().manual_seed(7)
    a = torch.randn(37, 256, generator=g).to(dtype).float()
    b = (a + torch.randn(37, 256, generator=g)).to(dtype).float()
    a[3] = 0  # a zero row: cosine 0, gradient through the eps clamp
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = -torch.nn.functional.cosine_similarity(ar, br, dim=1, eps=1e-8).mean()
    ref.backward()
    ad = a.to("cuda:0").to(dtype).requires_grad_(True)
    bd = b.to("cuda:0").to(dtype).requires_grad_(True)
    loss = NegativeCosineSimilarity()(ad, bd)
    loss.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 1e-6
    tol = 1e-6 if dtype == torch.float32 else 2e-3 * float(ar.grad.abs().max())
    assert torch.allclose(ad.grad.float().cpu(), ar.grad, atol=tol, rtol=1e-4 if dtype == torch.float32 else 1e-2)
    assert torch.allclose(bd.grad.float().cpu()[4:], br.grad[4:], atol=tol, rtol=1e-4 if dtype == torch.float32 else 1e-2)
    # a detached side gets no gradient buffer
    ad2 = a.to("cuda:0").requires_grad_(True)
    NegativeCosineSimilarity()(b.to("cuda:0"), ad2).backward()
    assert torch.allclose(ad2.grad.cpu()[4:], br.grad[4:] * 0 + (-torch.autograd.grad(
        torch.nn.functional.cosine_similarity(b, ar, dim=1).mean(), ar)[0])[4:], atol=1e-6)


def test_cross_entropy_and_bce_match_torch():
    from ssl_wafermap_amd.loss import BCEWithLogitsLoss, CrossEntropyLoss

    g = torch.Generator().manual_seed(11)
    logits = torch.randn(70, 9, generator=g) * 3
    labels = torch.randint(0, 9, (70,), generator=g)
    w = torch.rand(9, generator=g) + 0.2
    for weight in (None, w):
        for dtype in (torch.float32, torch.bfloat16):
            lr = logits.clone().to(dtype).float().requires_grad_(True)
            ref = torch.nn.functional.cross_entropy(lr, labels, weight=weight)
            ref.backward()
            ld = logits.clone().to("cuda:0").to(dtype).requires_grad_(True)
            crit = CrossEntropyLoss(weight=weight).to("cuda:0")
            loss = crit(ld, labels.to("cuda:0"))
            loss.backward()
            assert abs(float(loss.detach()) - float(ref.detach())) <= 2e-6 * abs(float(ref.detach())) + 1e-6
            assert torch.allclose(ld.grad.float().cpu(), lr.grad, atol=1e-6 if dtype == torch.float32 else 1e-3)
    target = (torch.rand(70, 8, generator=g) > 0.7).float()
    ml = torch.randn(70, 8, generator=g) * 2
    pw = torch.rand(8, generator=g) * 3 + 0.5
    for pos_weight in (None, pw):
        lr = ml.clone().requires_grad_(True)
        ref = torch.nn.functional.binary_cross_entropy_with_logits(lr, target, pos_weight=pos_weight)
        ref.backward()
        ld = ml.to("cuda:0").requires_grad_(True)
        loss = BCEWithLogitsLoss(pos_weight=pos_weight).to("cuda:0")(ld, target.to("cuda:0"))
        loss.backward()
        assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-5 * abs(float(ref.detach()))
        assert torch.allclose(ld.grad.cpu(), lr.grad, atol=1e-6, rtol=1e-4)


def test_linear_probes_learn_separable_features():
    """LinearClassifier / MultilabelLinearClassifier on frozen features, trained by the Adam loop."""
    from ssl_wafermap_amd.models import LinearClassifier, MultilabelLinearClassifier, fit_linear_probe

    g = torch.Generator().manual_seed(5)
    centers = torch.randn(9, 512, generator=g) * 2
    y = torch.randint(0, 9, (1800,), generator=g)
    x = centers[y] + torch.randn(1800, 512, generator=g)
    counts = torch.bincount(y, minlength=9).float()
    clf = LinearClassifier(512, 9, weight=counts.sum() / (9 * counts)).to("cuda:0")
    losses = fit_linear_probe(clf, x[:1500].to("cuda:0"), y[:1500].to("cuda:0"), epochs=6, batch_size=128)
    assert losses[-1] < losses[0] * 0.5
    m = clf.evaluate(x[1500:].to("cuda:0").bfloat16(), y[1500:].to("cuda:0"))
    assert m["acc"] > 0.9 and m["f1"] > 0.9
    wdir = torch.randn(8, 512, generator=g)
    xm = torch.randn(1200, 512, generator=g)
    ym = (xm @ wdir.T > 0.5).float()
    mclf = MultilabelLinearClassifier(512, 8, pos_weight=(1 - ym.mean(0)) / ym.mean(0)).to("cuda:0")
    before = mclf.evaluate(xm[1000:].to("cuda:0").bfloat16(), ym[1000:].to("cuda:0"))
    losses = fit_linear_probe(mclf, xm[:1000].to("cuda:0"), ym[:1000].to("cuda:0"), epochs=30, batch_size=100)
    assert losses[-1] < 0.7 * losses[0]
    mm = mclf.evaluate(xm[1000:].to("cuda:0").bfloat16(), ym[1000:].to("cuda:0"))
    assert mm["acc"] > before["acc"] + 0.1 and mm["f1"] > before["f1"]


def test_supervised_r18_step_matches_oracle():
    from oracle import resnet as orn
    from ssl_wafermap_amd import ops
    from ssl_wafermap_amd.models import SupervisedR18

    torch.manual_seed(0)
    model = SupervisedR18(None, 9, log_rep_std=False).to("cuda:0").train()
    (opt,), _ = model.configure_optimizers()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(16, 3, 224, 224, generator=g).bfloat16().float()
    y = torch.randint(0, 9, (16,), generator=g)
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    f = orn.resnet18_features(x, sd, True, prefix="backbone.")
    ref = torch.nn.functional.nll_loss(torch.log_softmax(torch.nn.functional.linear(f, sd["fc.weight"], sd["fc.bias"]), 1), y)
    opt.zero_grad()
    loss = model.training_step((ops.to_nhwc_bf16(x.to("cuda:0")), y.to("cuda:0")), 0)
    loss.backward()
    assert abs(float(loss.detach()) - float(ref)) <= 2e-2 * abs(float(ref)), (float(loss), float(ref))
    first = float(loss.detach())
    for i in range(5):
        opt.step()
        opt.zero_grad()
        loss = model.training_step((ops.to_nhwc_bf16(x.to("cuda:0")), y.to("cuda:0")), i + 1)
        loss.backward()
    assert float(loss.detach()) < first


@pytest.mark.parametrize("weighted", [False, True])
def test_dcl_losses_match_oracle(weighted):
    from ssl_wafermap_amd.loss import DCLLoss, DCLWLoss

    g = torch.Generator().manual_seed(13)
    b, d = 96, 128
    a = torch.randn(b, d, generator=g)
    c = a + 0.5 * torch.randn(b, d, generator=g)
    ar, cr = a.clone().requires_grad_(True), c.clone().requires_grad_(True)
    ref = on.dcl_loss(ar, cr, 0.1, 0.5 if weighted else None)
    ref.backward()
    ad, cd = a.to("cuda:0").requires_grad_(True), c.to("cuda:0").requires_grad_(True)
    crit = (DCLWLoss() if weighted else DCLLoss()).to("cuda:0")
    loss = crit(ad, cd)
    loss.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-5 * abs(float(ref.detach())), (float(loss), float(ref))
    assert torch.allclose(ad.grad.cpu(), ar.grad, rtol=1e-3, atol=2e-6)
    assert torch.allclose(cd.grad.cpu(), cr.grad, rtol=1e-3, atol=2e-6)


def test_barlow_twins_loss_matches_oracle():
    from ssl_wafermap_amd.loss import BarlowTwinsLoss

    g = torch.Generator().manual_seed(17)
    n, d = 64, 256
    a = (torch.randn(n, d, generator=g) * 1.5 + 0.3).bfloat16().float()
    b = (a + 0.7 * torch.randn(n, d, generator=g)).bfloat16().float()
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = on.barlow_twins_loss(ar, br)
    ref.backward()
    ad, bd = a.to("cuda:0").bfloat16().requires_grad_(True), b.to("cuda:0").bfloat16().requires_grad_(True)
    loss = BarlowTwinsLoss().to("cuda:0")(ad, bd)
    loss.backward()
    # standardised projections are stored in bf16 before the correlation GEMM: 2^-9 on each factor
    assert abs(float(loss.detach()) - float(ref.detach())) <= 2e-2 * abs(float(ref.detach())), (float(loss), float(ref))
    ca = torch.nn.functional.cosine_similarity(ad.grad.float().cpu().flatten(), ar.grad.flatten(), dim=0)
    cb = torch.nn.functional.cosine_similarity(bd.grad.float().cpu().flatten(), br.grad.flatten(), dim=0)
    assert float(ca) > 0.995 and float(cb) > 0.995, (float(ca), float(cb))
    scale = float(ar.grad.abs().max())
    assert float((ad.grad.float().cpu() - ar.grad).abs().max()) < 0.05 * scale


def test_lars_matches_timm_restatement():
    from oracle import resnet as orn
    from ssl_wafermap_amd import optim

    torch.manual_seed(5)
    shapes = [(64, 64, 3, 3), (64,), (128, 64), (7,)]
    ps = [torch.nn.Parameter(torch.randn(*s_, device="cuda:0") * 0.1) for s_ in shapes]
    ref = {str(i): p.detach().cpu().clone() for i, p in enumerate(ps)}
    bufs = {}
    opt = optim.LARS(ps, lr=0.2, weight_decay=1.5e-6, momentum=0.9)
    for step in range(4):
        grads = {str(i): torch.randn(*s_) * (0.0 if (step == 1 and i == 3) else 1.0) for i, s_ in enumerate(shapes)}
        for i, p in enumerate(ps):
            p.grad.copy_(grads[str(i)].to("cuda:0"))
        opt.step()
        orn.lars_step(ref, grads, bufs, 0.2, 0.9, 1.5e-6)
    for i, p in enumerate(ps):
        torch.testing.assert_close(p.detach().cpu(), ref[str(i)], rtol=2e-5, atol=1e-7)


def test_vicreg_loss_matches_oracle():
    from ssl_wafermap_amd.loss import VICRegLoss

    g = torch.Generator().manual_seed(19)
    n, d = 64, 256
    a = (torch.randn(n, d, generator=g) * torch.rand(d, generator=g) * 1.5).bfloat16().float()  # some stds below 1
    b = (a + 0.4 * torch.randn(n, d, generator=g)).bfloat16().float()
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = on.vicreg_loss(ar, br)
    ref.backward()
    ad, bd = a.to("cuda:0").bfloat16().requires_grad_(True), b.to("cuda:0").bfloat16().requires_grad_(True)
    loss = VICRegLoss().to("cuda:0")(ad, bd)
    loss.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-2 * abs(float(ref.detach())), (float(loss), float(ref))
    for got, want in ((ad.grad, ar.grad), (bd.grad, br.grad)):
        c = torch.nn.functional.cosine_similarity(got.float().cpu().flatten(), want.flatten(), dim=0)
        assert float(c) > 0.995, float(c)
        assert float((got.float().cpu() - want).abs().max()) < 0.05 * float(want.abs().max())


def test_sinkhorn_and_swav_loss_match_oracle():
    from ssl_wafermap_amd.loss import SwaVLoss, sinkhorn

    g = torch.Generator().manual_seed(23)
    b, k = 48, 3000
    high = [torch.randn(b, k, generator=g).clamp(-1, 1) * 0.9 for _ in range(2)]   # prototype scores are cosines
    low = [torch.randn(b, k, generator=g).clamp(-1, 1) * 0.9 for _ in range(3)]
    q = sinkhorn(high[0].to("cuda:0"))
    qr = on.sinkhorn(high[0])
    assert torch.allclose(q.sum(1).cpu(), torch.ones(b), atol=1e-4)
    assert torch.allclose(q.cpu(), qr, rtol=1e-3, atol=1e-7)
    hr = [t.clone().requires_grad_(True) for t in high]
    lr = [t.clone().requires_grad_(True) for t in low]
    ref = on.swav_loss(hr, lr)
    ref.backward()
    hd = [t.to("cuda:0").requires_grad_(True) for t in high]
    ld = [t.to("cuda:0").requires_grad_(True) for t in low]
    loss = SwaVLoss().to("cuda:0")(hd, ld)
    loss.backward()
    # the loss kernel reads bf16 student scores
    assert abs(float(loss.detach()) - float(ref.detach())) <= 2e-3 * abs(float(ref.detach())), (float(loss), float(ref))
    for got, want in zip(hd + ld, hr + lr):
        c = torch.nn.functional.cosine_similarity(got.grad.float().cpu().flatten(), want.grad.flatten(), dim=0)
        assert float(c) > 0.99, float(c)


def test_adam_with_l2_weight_decay_matches_torch():
    from ssl_wafermap_amd import optim

    torch.manual_seed(9)
    shapes = [(96, 64), (64,), (3, 5, 7)]
    ps = [torch.nn.Parameter(torch.randn(*s_, device="cuda:0")) for s_ in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    topt = torch.optim.Adam(ref, lr=2e-3, weight_decay=1e-2)
    opt = optim.Adam(ps, lr=2e-3, weight_decay=1e-2)
    for _ in range(4):
        for p, r in zip(ps, ref):
            gval = torch.randn_like(r)
            r.grad = gval.clone()
            p.grad.copy_(gval)
        opt.step()
        topt.step()
    for p, r in zip(ps, ref):
        torch.testing.assert_close(p.detach(), r.detach(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("pmsn", [False, True])
def test_msn_losses_match_oracle(pmsn):
    from ssl_wafermap_amd.loss import MSNLoss, PMSNLoss

    g = torch.Generator().manual_seed(29)
    b, v, d, k = 24, 3, 256, 1024
    targets = torch.randn(b, d, generator=g)
    anchors = (targets.repeat(v, 1) + 0.8 * torch.randn(v * b, d, generator=g))
    protos = torch.randn(k, d, generator=g)
    ar = anchors.clone().requires_grad_(True)
    ref = on.msn_loss(ar, targets, protos, power_law_exponent=0.25 if pmsn else None)
    ref.backward()
    ad = anchors.to("cuda:0").requires_grad_(True)
    crit = (PMSNLoss() if pmsn else MSNLoss()).to("cuda:0")
    loss = crit(ad, targets.to("cuda:0"), protos.to("cuda:0"))
    loss.backward()
    # the prototype cosines pass through bf16 before the softmax at T = 0.1 (targets: T * 0.25 = 0.025)
    assert abs(float(loss.detach()) - float(ref.detach())) <= 3e-2 * abs(float(ref.detach())) + 2e-2, (float(loss), float(ref))
    c = torch.nn.functional.cosine_similarity(ad.grad.float().cpu().flatten(), ar.grad.flatten(), dim=0)
    assert float(c) > 0.97, float(c)


@pytest.mark.gpu
def test_std_of_l2_normalized_matches_torch():
    """rep_std (lightly.utils.debug.std_of_l2_normalized): wm_l2_normalize + wm_colstats against torch."""
    from ssl_wafermap_amd.utils.debug import std_of_l2_normalized

    g = torch.Generator().manual_seed(3)
    for rows, c in ((256, 512), (37, 96), (2, 8)):
        z = torch.randn(rows, c, generator=g) * 3 + 0.5
        ref = torch.std(torch.nn.functional.normalize(z, dim=1), dim=0).mean()
        got = std_of_l2_normalized(z.to("cuda:0"))
        assert abs(float(got) - float(ref)) <= 1e-6 + 1e-5 * abs(float(ref)), (rows, c, float(got), float(ref))


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,k,dtype", [(3000, 128, 200, torch.float32), (777, 96, 50, torch.float32),
                                         (2000, 128, 40, torch.bfloat16)])
def test_knn_topk_large_k_pages(n, d, k, dtype):
    """k > 16 (lightly's knn_predict default 200): pages of 16 through wm_knn_topk_general_after are the exact
    top-k in list order, including ties (duplicated bank rows) and an index base."""
    from ssl_wafermap_amd import functional as F_hip

    g = torch.Generator().manual_seed(n + k)
    bank = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1)
    bank[5::7] = bank[3]  # many exact ties
    bank = bank.to(dtype)
    q = bank[:9].clone()
    sim, idx = F_hip.knn_topk(q.to("cuda:0"), bank.to("cuda:0"), k, index_base=1000)
    ref = q.float() @ bank.float().t()
    order = torch.argsort(-ref, dim=1, stable=True)[:, :k]  # descending, lower index first on ties
    rs = torch.gather(ref, 1, order)
    assert torch.allclose(sim.cpu(), rs, atol=2e-6, rtol=0)
    got = idx.cpu().long() - 1000
    # ranking identical wherever the float32 scores are separated; tied groups must contain the same rows
    for r in range(q.shape[0]):
        assert set(got[r].tolist()) == set(order[r].tolist()) or torch.allclose(ref[r, got[r]], rs[r], atol=2e-6)
        assert len(set(got[r].tolist())) == k


def test_knn_topk_batched_pipelined_equals_single_calls():
    """functional.knn_topk_batched (query batches round-robin on several HIP streams) returns exactly what one
    knn_topk call per batch returns, ragged last batch included."""
    from ssl_wafermap_amd import functional as F

    g = torch.Generator().manual_seed(9)
    bank = torch.nn.functional.normalize(torch.randn(20000, 128, generator=g), dim=1).to(_dev()).bfloat16()
    q = bank[100:100 + 64 * 7 + 13].contiguous()
    for lanes in (2, 3):
        sim, idx = F.knn_topk_batched(q, bank, 8, batch=64, lanes=lanes)
        for o in range(0, q.shape[0], 64):
            s1, i1 = F.knn_topk(q[o:o + 64], bank, 8)
            assert torch.equal(sim[o:o + 64], s1) and torch.equal(idx[o:o + 64], i1)
        assert torch.equal(idx[:, 0].long().cpu(), torch.arange(100, 100 + q.shape[0]))


def test_knn_classifier_matches_oracle_knn_predict():
    """utils.benchmarking.KNNClassifier (north_star's API name; lightly's class form of the reference's knn_predict
    evaluation, src/ssl_wafermap/models/knn.py:67-101) on the HIP kernels against the oracle on the same features."""
    from oracle import knn as ok
    from ssl_wafermap_amd.utils.benchmarking import KNNClassifier, mean_topk_accuracy

    g = torch.Generator().manual_seed(11)
    w = torch.randn(48, 64, generator=g)
    centers = torch.randn(6, 48, generator=g) * 1.5

    def make(n):
        y = torch.randint(0, 6, (n,), generator=g)
        return centers[y] + torch.randn(n, 48, generator=g), y

    xb, yb = make(900)
    xv, yv = make(130)

    class Backbone(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(w.clone())

        def forward(self, x):
            return x @ self.w

    clf = KNNClassifier(Backbone().to(DEV), num_classes=6, knn_k=20, knn_t=0.1, topk=(1, 5))
    clf.fit_bank([(xb[i:i + 128].to(DEV), yb[i:i + 128].to(DEV)) for i in range(0, 900, 128)])
    pred = clf.validation_step((xv.to(DEV), yv.to(DEV)))
    fb = torch.nn.functional.normalize(xb @ w, dim=1)
    fq = torch.nn.functional.normalize(xv @ w, dim=1)
    want = ok.knn_predict(fq, fb.t().contiguous(), yb, 6, 20, 0.1)
    assert torch.equal(pred[:, 0].cpu(), want[:, 0])
    acc = mean_topk_accuracy(want, yv, k=(1, 5))
    assert abs(float(clf.logged["val_top1"]) - float(acc[1])) < 1e-6
    assert float(clf.logged["val_top5"]) >= float(clf.logged["val_top1"]) > 0.8


def test_projection_head_variants_match_torch():
    """SimCLRProjectionHead(batch_norm=False) (the v1 form: Linear+bias, ReLU, Linear+bias) and
    DINOProjectionHead(norm_last_layer=False) (trainable weight-norm gain) -- SURVEY Appendix A.2 -- forward and
    gradients against torch float32 on the same (bf16-exact) weights."""
    from ssl_wafermap_amd import heads

    torch.manual_seed(0)
    x = torch.randn(64, 128).bfloat16().float()
    # ---- SimCLR v1
    h = heads.SimCLRProjectionHead(128, 128, 64, batch_norm=False).to(DEV)
    assert [type(m).__name__ for m in h.layers] == ["Linear", "ReLU", "Linear"] and h.layers[0].bias is not None
    with torch.no_grad():
        for p in h.parameters():
            p.copy_(p.bfloat16().float())
    xd = x.to(DEV).bfloat16().requires_grad_(True)
    out = h(xd)
    t = torch.randn(64, 64)
    (out.float() * t.to(DEV)).sum().backward()
    w0, b0, w1, b1 = [p.detach().cpu().clone().requires_grad_(True) for p in h.parameters()]
    xr = x.clone().requires_grad_(True)
    ref = torch.relu(xr @ w0.t() + b0) @ w1.t() + b1
    (ref * t).sum().backward()
    torch.testing.assert_close(out.float().cpu(), ref.detach(), atol=3e-2, rtol=2e-2)
    for p, r in zip(h.parameters(), (w0, b0, w1, b1)):
        torch.testing.assert_close(p.grad.cpu(), r.grad, atol=2e-2 * float(r.grad.abs().max()), rtol=2e-2)
    # ---- DINO head with a trainable last-layer gain
    d = heads.DINOProjectionHead(128, 128, 64, 256, norm_last_layer=False).to(DEV)
    assert d.last_layer.weight_g.requires_grad
    with torch.no_grad():
        d.last_layer.weight_g.copy_(torch.rand(256, 1) + 0.5)
    z = torch.nn.functional.normalize(torch.randn(64, 64), dim=1).bfloat16().float()
    zd = z.to(DEV).bfloat16().requires_grad_(True)
    y = d.last_layer(zd)
    t2 = torch.randn(64, 256)
    (y.float() * t2.to(DEV)).sum().backward()
    v = d.last_layer.weight_v.detach().cpu().clone().requires_grad_(True)
    gq = d.last_layer.weight_g.detach().cpu().clone().requires_grad_(True)
    zr = z.clone().requires_grad_(True)
    wn = gq * v / v.norm(dim=1, keepdim=True)
    yr = zr @ wn.t()
    (yr * t2).sum().backward()
    torch.testing.assert_close(y.float().cpu(), yr.detach(), atol=2e-2, rtol=2e-2)
    torch.testing.assert_close(d.last_layer.weight_g.grad.cpu(), gq.grad, atol=3e-2 * float(gq.grad.abs().max()), rtol=3e-2)
    torch.testing.assert_close(d.last_layer.weight_v.grad.cpu(), v.grad, atol=3e-2 * float(v.grad.abs().max()), rtol=5e-2)
    torch.testing.assert_close(zd.grad.float().cpu(), zr.grad, atol=3e-2 * float(zr.grad.abs().max()), rtol=5e-2)


def test_random_token_mask_on_device_equals_argsort():
    """random_token_mask's permutation is built by wm_argsort_rows (a bitonic network per image), not a library sort:
    same indices as torch.argsort of the same noise, class token first, for the reference's sequence lengths."""
    from ssl_wafermap_amd import _lib
    from ssl_wafermap_amd._lib import check, ptr, stream_ptr
    from ssl_wafermap_amd.utils import random_token_mask

    for b, s in ((7, 50), (5, 197), (3, 256), (4, 1)):
        noise = torch.rand(b, s, device=DEV)
        noise[:, 0] = -1
        noise[:, s // 2] = noise[:, -1]          # a tie: the lower index first
        idx = torch.empty((b, s), dtype=torch.int64, device=DEV)
        check(_lib.load().wm_argsort_rows(ptr(noise), b, s, ptr(idx), stream_ptr()), "wm_argsort_rows")
        want = torch.argsort(noise, dim=1, stable=True)
        assert torch.equal(idx, want), (b, s)
    g = torch.Generator(device=DEV).manual_seed(3)
    keep, mask = random_token_mask((6, 50), 0.75, device=DEV, generator=g)
    assert keep.shape == (6, 12) and mask.shape == (6, 38) and bool((keep[:, 0] == 0).all())
    both = torch.cat([keep, mask], dim=1).sort(dim=1).values
    assert torch.equal(both, torch.arange(50, device=DEV).expand(6, 50))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_ntxent_on_stacked_views_equals_the_two_argument_form(dtype):
    """SimCLR hands NTXentLoss the two halves of ONE stacked projection tensor (loss.stacked_views): the loss then
    works on the stacked tensor as a single autograd node.  Same loss and the same projection gradient, to the bit,
    as passing two unrelated tensors (cat + cast + two separate normalisations' worth of autograd nodes)."""
    from ssl_wafermap_amd.loss import NTXentLoss, stacked_views

    g = torch.Generator().manual_seed(3)
    b, d = 96, 128
    z = torch.randn(2 * b, d, generator=g).to(DEV).to(dtype)
    crit = NTXentLoss(temperature=0.5)
    za = z.clone().requires_grad_(True)
    la = crit(*stacked_views(za, b))
    assert type(la.grad_fn).__name__ == "_NTXentProjectionsBackward"
    la.backward()
    zb = z.clone().requires_grad_(True)
    lb = crit(zb[:b], zb[b:])
    assert type(lb.grad_fn).__name__ != "_NTXentProjectionsBackward"
    lb.backward()
    assert torch.equal(la.detach(), lb.detach())
    assert torch.equal(za.grad, zb.grad)
    # an upstream gradient other than 1 is applied inside the normalisation's backward kernel
    zc = z.clone().requires_grad_(True)
    (crit(*stacked_views(zc, b)) * 0.25).backward()
    torch.testing.assert_close(zc.grad.float(), za.grad.float() * 0.25, rtol=1e-2 if dtype == torch.bfloat16 else 1e-6, atol=1e-9)
