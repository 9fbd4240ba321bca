"""CPU tests of the host-side mirror of the reference interface (no kernels are launched)."""
import numpy as np
import pytest
import torch

from oracle import augment as oa


def test_dpw_scale_and_dims_match_oracle(ref_vectors):
    from ssl_wafermap_amd.transforms import DPWTransform

    t = DPWTransform()
    xs = ref_vectors["powerlaw_x"]
    np.testing.assert_array_equal(t.power_law(xs), ref_vectors["powerlaw_y"])  # pinned to the reference's table
    rng = np.random.default_rng(0)
    h, w = rng.integers(18, 213, 500), rng.integers(18, 213, 500)
    beta = rng.beta(0.5, 1.5, 500)
    sc = t.scales(h, w, beta)
    for i in range(500):
        assert sc[i] == oa.dpw_scale((int(h[i]), int(w[i])), float(beta[i]))
        assert (int(np.floor(h[i] * sc[i])), int(np.floor(w[i] * sc[i]))) == oa.dpw_dims(int(h[i]), int(w[i]), float(sc[i]))


def test_sample_view_params_ranges_and_determinism():
    from ssl_wafermap_amd.transforms import get_base_transforms, multicrop_view, sample_view_params

    h, w = np.random.default_rng(1).integers(22, 213, (2, 300))
    idx = np.arange(300)
    a = sample_view_params(get_base_transforms(crop=True), idx, h, w, np.random.default_rng(7))
    b = sample_view_params(get_base_transforms(crop=True), idx, h, w, np.random.default_rng(7))
    assert np.array_equal(a, b)
    assert set(np.unique(a["op"])) <= {1, 2} and 0.35 < (a["op"] == 1).mean() < 0.65
    for k in ("rot90", "vflip", "hflip", "crop"):
        assert 0.35 < a[k].mean() < 0.65
    dpw = a["op"] == 2
    assert (a["dpw_h"][dpw] <= h[dpw]).all() and (a["dpw_h"][dpw] >= np.floor(0.4 * h[dpw])).all()
    assert (a["dpw_h"][~dpw] == h[~dpw]).all()
    c = a["crop"] == 1
    side = a["crop_h"][c]
    assert (side == a["crop_w"][c]).all() and (side >= round((0.4 * 224 * 224) ** 0.5) - 1).all() and (side <= 224).all()
    assert (a["crop_i"][c] + side <= 224).all() and (a["crop_j"][c] + side <= 224).all()
    loc = sample_view_params(multicrop_view(crop_size=96, crop_scale=(0.1, 0.4)), idx, h, w, np.random.default_rng(2))
    assert loc["crop"].all() and (loc["crop_h"] <= round((0.4 * 224 * 224) ** 0.5) + 1).all()
    # crop boxes agree with the oracle's restatement of RandomResizedCrop.get_params for the same area draw
    for area_u in (0.0, 0.3, 0.999):
        i, j, hh, ww = oa.random_resized_crop_params(224, 224, (0.1, 0.4), area_u, 0.5, 0.5)
        assert hh == ww == int(round((224 * 224 * (0.1 + 0.3 * area_u)) ** 0.5))


def test_transform_classes_have_the_reference_structure():
    from ssl_wafermap_amd.transforms import BaseViewTransform, InferenceTransform, MultiCropTransform, RandomOneOf, DieNoise

    b = BaseViewTransform()
    assert len(b.transforms) == 2 and b.transforms[0].out_size == 224 and b.transforms[0].crop_scale is None
    assert len(BaseViewTransform(n_views=4).transforms) == 4
    with pytest.raises(AssertionError):
        BaseViewTransform(n_views=0)
    m = MultiCropTransform()
    assert [t.out_size for t in m.transforms] == [224, 224] + [96] * 6
    assert m.transforms[0].crop_scale == (0.6, 1.0) and m.transforms[2].crop_scale == (0.1, 0.4)
    assert InferenceTransform().transforms[0].stage1 is None
    with pytest.raises(ValueError):
        RandomOneOf((DieNoise(),), weights=(1.0, 2.0))
    with pytest.raises(ValueError):
        RandomOneOf((DieNoise(),), weights=(0.0,))
    with pytest.raises(ValueError):
        RandomOneOf((DieNoise(),), p=1.5)


def test_validate_params_rejects_out_of_range_views():
    from ssl_wafermap_amd.data import WaferStore
    from ssl_wafermap_amd.transforms import get_inference_transforms, sample_view_params
    from ssl_wafermap_amd.transforms.augmentations import validate_params

    store = WaferStore([np.zeros((30, 40), np.uint8), np.zeros((22, 22), np.uint8)])
    p = sample_view_params(get_inference_transforms(), np.array([0, 1]), store.heights_np, store.widths_np, np.random.default_rng(0))
    validate_params(p, store, 224, 2)
    for field, val, exc in (("sample", 2, IndexError), ("out_slot", 5, IndexError), ("op", 9, ValueError)):
        bad = p.copy()
        bad[field][0] = val
        with pytest.raises(exc):
            validate_params(bad, store, 224, 2)
    bad = p.copy()
    bad["op"][0], bad["dpw_h"][0] = 2, 31
    with pytest.raises(ValueError):
        validate_params(bad, store, 224, 2)
    with pytest.raises(ValueError):
        WaferStore([np.zeros((300, 10), np.uint8)])
    assert store.max_elems == 1200 and store.nbytes() == 1200 + 484 and store.wafer(1).shape == (22, 22)


def test_resnet18_state_dict_has_timm_keys():
    from ssl_wafermap_amd.heads import SimCLRProjectionHead
    from ssl_wafermap_amd.models import create_model

    sd = create_model("resnet18", num_classes=0).state_dict()
    expect = ["conv1.weight"] + [f"bn1.{k}" for k in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")]
    assert list(sd)[:6] == expect
    for layer, planes, cin in ((1, 64, 64), (2, 128, 64), (3, 256, 128), (4, 512, 256)):
        assert sd[f"layer{layer}.0.conv1.weight"].shape == (planes, cin, 3, 3)
        assert sd[f"layer{layer}.1.conv2.weight"].shape == (planes, planes, 3, 3)
        assert (f"layer{layer}.0.downsample.0.weight" in sd) == (layer > 1)
        assert float(sd[f"layer{layer}.0.bn2.weight"].abs().sum()) == 0.0  # zero_init_last
    assert sd["conv1.weight"].shape == (64, 3, 7, 7)
    assert sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k) == 11176512
    h = SimCLRProjectionHead(512, 512, 128).state_dict()
    assert h["layers.0.weight"].shape == (512, 512) and h["layers.3.weight"].shape == (128, 512)
    assert "layers.0.bias" not in h and "layers.4.running_var" in h
    with pytest.raises(NotImplementedError):
        create_model("resnet50")
    # the BN-free (SimCLR v1) form: Linear layers with a bias, no BatchNorm keys (lightly's batch_norm=False)
    h1 = SimCLRProjectionHead(512, 512, 128, batch_norm=False).state_dict()
    assert set(h1) == {"layers.0.weight", "layers.0.bias", "layers.2.weight", "layers.2.bias"}


def test_ntxent_constructor_checks_and_scheduler():
    from ssl_wafermap_amd.loss import NTXentLoss
    from ssl_wafermap_amd.utils.scheduler import cosine_warmup_factor

    with pytest.raises(ValueError):
        NTXentLoss(temperature=0.0)
    with pytest.raises(ValueError):
        NTXentLoss(memory_bank_size=-1)
    crit = NTXentLoss(temperature=0.1, memory_bank_size=4096)
    assert crit.size == 4096 and crit.bank.numel() == 0  # the bank is created on first use (lightly)
    assert cosine_warmup_factor(0, 20, 150) == pytest.approx(1 / 20)
    assert cosine_warmup_factor(19, 20, 150) == pytest.approx(1.0)
    assert cosine_warmup_factor(20, 20, 150) == pytest.approx(1.0)
    assert cosine_warmup_factor(150, 20, 150) == pytest.approx(0.0, abs=1e-12)


def test_macro_metrics_match_sklearn():
    from sklearn.metrics import confusion_matrix, f1_score, recall_score

    from ssl_wafermap_amd.models import macro_metrics

    rng = np.random.default_rng(0)
    t = rng.integers(0, 7, 500)  # classes 7, 8 never occur
    p = np.where(rng.random(500) < 0.6, t, rng.integers(0, 9, 500))
    acc, f1, cm = macro_metrics(torch.tensor(p), torch.tensor(t), 9)
    labs = sorted(set(t))
    assert abs(acc - recall_score(t, p, labels=labs, average="macro")) < 1e-6
    seen = sorted(set(t) | set(p))
    assert abs(f1 - f1_score(t, p, labels=seen, average="macro")) < 1e-6
    ref = confusion_matrix(t, p, labels=list(range(9))).astype(float)
    ref = ref / np.maximum(ref.sum(1, keepdims=True), 1)
    np.testing.assert_allclose(cm.numpy(), ref, atol=1e-6)


def test_oracle_resnet_matches_torch_modules():
    """The functional oracle equals an nn.Module ResNet-18 assembled from torch layers."""
    import torch.nn as nn

    from oracle import resnet as orn
    from ssl_wafermap_amd.models import create_model

    torch.manual_seed(0)
    ours = create_model("resnet18", num_classes=0)
    sd = {k: v.clone() for k, v in ours.state_dict().items()}

    def block(p, cin, cout, stride):
        conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        conv1.weight.data = sd[p + ".conv1.weight"]
        return conv1

    x = torch.randn(2, 3, 64, 64)
    f = orn.resnet18_features(x, {k: v.clone() for k, v in sd.items()}, training=True)
    assert f.shape == (2, 512) and torch.isfinite(f).all()
    # stem + first conv spot check against module code
    y = nn.functional.conv2d(x, sd["conv1.weight"], None, 2, 3)
    assert y.shape == (2, 64, 32, 32)
    assert block("layer2.0", 64, 128, 2)(torch.randn(1, 64, 16, 16)).shape == (1, 128, 8, 8)


def test_collate_functions_have_the_reference_view_structure():
    from ssl_wafermap_amd import transforms as T

    assert len(T.WaferImageCollateFunction().transform.transforms) == 2
    d = T.WaferDINOCOllateFunction()
    assert [v.out_size for v in d.transform.transforms] == [224, 224] + [96] * 6
    assert d.transform.transforms[0].crop_scale == (0.6, 1.0) and d.transform.transforms[-1].crop_scale == (0.1, 0.4)
    assert len(T.WaferMAECollateFunction2(denoise=True).transform.transforms) == 1
    m = T.WaferMSNCollateFunction()
    assert [v.out_size for v in m.transform.transforms] == [224] * 2 + [96] * 10 and m.transform.transforms[0].vf_prob == 0.0
    s = T.WaferSwaVCollateFunction(crop_counts=[2, 4])
    assert [v.out_size for v in s.transform.transforms] == [224, 224, 96, 96, 96, 96]
    with pytest.raises(ValueError):
        T.WaferSwaVCollateFunction(crop_sizes=[224])
    with pytest.raises(RuntimeError):
        T.WaferImageCollateFunction()([(0, 1, "a")])
    denoise = T.WaferImageCollateFunction(denoise=True).transform.transforms[0].stage1.transforms
    assert isinstance(denoise[1], T.MedianFilter)  # denoise=True -> RandomOneOf{DieNoise | MedianFilter}
    x = T.rgb_scale(np.array([[0.0, 0.5], [1.0, 0.25]]))
    assert x.dtype == np.uint8 and x.tolist() == [[0, 128], [255, 64]]


def test_ingest_of_reference_pickles_and_flat_store_roundtrip(tmp_path):
    """pandas *.pkl.xz with waferMap / failureCode columns (the reference's processed files) -> flat
    store -> .npz -> identical store; a dataset built on the loaded store sees the same wafers."""
    import pandas as pd

    from ssl_wafermap_amd.data import WaferMapDataset, WaferStore, convert_pickle, read_wafer_pickle
    from ssl_wafermap_amd.data.synthetic import synthetic_wafers

    wafers, labels = synthetic_wafers(37, seed=4)
    df = pd.DataFrame({"waferMap": pd.Series(list(wafers)), "failureCode": np.asarray(labels).astype(np.int8),
                       "label": [np.arange(8) % 2 for _ in wafers]})
    src = tmp_path / "train_1_split.pkl.xz"
    df.to_pickle(src)
    store, y = read_wafer_pickle(src)
    assert len(store) == 37 and np.array_equal(y, np.asarray(labels))
    for i in (0, 5, 36):
        assert np.array_equal(store.wafer(i), wafers[i])
    _, multi = read_wafer_pickle(src, label_col="label")
    assert multi.shape == (37, 8)
    with pytest.raises(KeyError):
        read_wafer_pickle(src, label_col="nope")
    dst = tmp_path / "train_1_split.npz"
    convert_pickle(src, dst)
    loaded, y2 = WaferStore.load(dst)
    assert np.array_equal(loaded.bytes_np, store.bytes_np) and np.array_equal(loaded.offsets_np, store.offsets_np)
    assert np.array_equal(y2, y) and loaded.max_elems == store.max_elems
    ds = WaferMapDataset(loaded, y2)
    assert len(ds) == 37 and np.array_equal(ds.store.wafer(7), wafers[7])
    bad = dict(np.load(dst))
    bad["heights"] = bad["heights"].copy()
    bad["heights"][3] += 1
    np.savez(tmp_path / "bad.npz", **bad)
    with pytest.raises(ValueError):
        WaferStore.load(tmp_path / "bad.npz")


_REF_SPLIT = "/root/reference/data/processed/WM811K/train_1_split.pkl.xz"


@pytest.mark.skipif(not __import__("os").path.exists(_REF_SPLIT), reason="reference checkout not present")
def test_ingest_reads_a_real_reference_split():
    from ssl_wafermap_amd.data import read_wafer_pickle

    store, y = read_wafer_pickle(_REF_SPLIT)
    assert len(store) == len(y) > 100
    vals = np.unique(store.bytes_np)
    assert set(vals.tolist()) <= {0, 128, 255}
    assert 1 <= store.heights_np.min() and store.heights_np.max() <= 256


def test_multiply_high_division_identity():
    """common.h WmDiv: floor(x / d) == (umulhi(x, m) + x) >> l with l = ceil(log2 d), m = ceil(2^(32+l) / d) - 2^32,
    for every x < 2^31 -- the index decode of the tile kernels and the pooled BatchNorm kernels relies on it.  Checked
    here with the device's 32-bit wrap-around arithmetic on the divisors those kernels see and on random ones."""
    rng = np.random.default_rng(0)
    divisors = [1, 2, 3, 5, 7, 8, 12, 14, 16, 24, 28, 49, 56, 64, 96, 112, 196, 224, 256, 784, 3136, 12544, 50176,
                65535, 65536, 1000003, (1 << 30) + 1] + [int(v) for v in rng.integers(1, 1 << 20, 40)]
    xs = np.concatenate([np.arange(0, 70000, dtype=np.uint64), rng.integers(0, 1 << 31, 200000).astype(np.uint64),
                         np.array([(1 << 31) - 1, (1 << 31) - 2], dtype=np.uint64)])
    for d in divisors:
        l = 0
        while (1 << l) < d:
            l += 1
        m = ((1 << (32 + l)) + d - 1) // d - (1 << 32)
        assert 0 <= m < (1 << 32)
        hi = (xs * np.uint64(m)) >> np.uint64(32)
        q = ((hi + xs) & np.uint64(0xFFFFFFFF)) >> np.uint64(l)   # the device adds in 32 bits
        edge = np.array([d * 1000 - 1, d * 1000], dtype=np.uint64)
        edge = edge[edge < (1 << 31)]
        assert np.array_equal(q, xs // np.uint64(d)), d
        hi_e = (edge * np.uint64(m)) >> np.uint64(32)
        assert np.array_equal(((hi_e + edge) & np.uint64(0xFFFFFFFF)) >> np.uint64(l), edge // np.uint64(d)), d


def test_fold_plan_merges_the_second_branch_and_orders_everything_else():
    """ops._plan_fold (host logic of the weight-gradient fold): entries with distinct gradient slots share one launch; the
    second use of a parameter with slabs of the same shape -- the second of two parallel branches (nn.ViewBranches) -- rides
    in the first entry's descriptor as its second slab set; a different slab count, a third use or bias slabs go into a
    later launch, in order of use."""
    from ssl_wafermap_amd import ops

    plan = ops._plan_fold
    # (slabs, nsplit, slot, K, C, RS, bias slabs, bias gradient)
    assert plan([(100, 8, 1, 64, 64, 9, 0, 0), (200, 8, 2, 64, 64, 9, 0, 0)]) == \
        [[(0, 0, 0, 100, 1, 64, 64, 9, 8), (0, 0, 0, 200, 2, 64, 64, 9, 8)]]
    assert plan([(100, 8, 1, 64, 64, 9, 0, 0), (200, 8, 1, 64, 64, 9, 0, 0)]) == [[(0, 0, 200, 100, 1, 64, 64, 9, 8)]]
    assert plan([(100, 8, 1, 64, 64, 9, 0, 0), (200, 4, 1, 64, 64, 9, 0, 0)]) == \
        [[(0, 0, 0, 100, 1, 64, 64, 9, 8)], [(0, 0, 0, 200, 1, 64, 64, 9, 4)]]
    third = plan([(100, 8, 1, 64, 64, 9, 0, 0), (200, 8, 1, 64, 64, 9, 0, 0), (300, 8, 1, 64, 64, 9, 0, 0)])
    assert third == [[(0, 0, 200, 100, 1, 64, 64, 9, 8)], [(0, 0, 0, 300, 1, 64, 64, 9, 8)]]
    biased = plan([(100, 8, 1, 64, 64, 1, 500, 9), (200, 8, 1, 64, 64, 1, 600, 9)])
    assert biased == [[(500, 9, 0, 100, 1, 64, 64, 1, 8)], [(600, 9, 0, 200, 1, 64, 64, 1, 8)]]
    # the order of use survives interleaving: two parameters, each used by both branches
    both = plan([(10, 4, 1, 8, 8, 1, 0, 0), (20, 4, 2, 8, 8, 1, 0, 0), (11, 4, 1, 8, 8, 1, 0, 0), (21, 4, 2, 8, 8, 1, 0, 0)])
    assert both == [[(0, 0, 11, 10, 1, 8, 8, 1, 4), (0, 0, 21, 20, 2, 8, 8, 1, 4)]]


def test_view_branches_and_parallel_branch_switches_fail_loudly_without_a_gpu():
    """nn.ViewBranches is GPU state (a side stream, flat device buffers): a CPU model keeps the single-stream path, and the
    ResNet-18 backbone's branch test is False off the GPU (no silent CPU emulation of the branches)."""
    import torch

    from ssl_wafermap_amd import ops
    from ssl_wafermap_amd.models.resnet import create_model

    m = create_model("resnet18").train()
    x = torch.zeros(4, 3, 32, 32)
    with ops.bn_groups(2):
        assert m._branch_ok(x) is False
    assert getattr(m, "_branches", None) is None
    assert ops.current_branch() == 0
    with ops.branch(1):
        assert ops.current_branch() == 1
    assert ops.current_branch() == 0
