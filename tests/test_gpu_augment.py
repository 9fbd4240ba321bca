"""GPU parity: the fused augmentation kernel against the numpy oracle, bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_ragged
from oracle import augment as oa


def _decision(p, out_size):
    return oa.ViewDecision(op=int(p["op"]), noise_seed=int(p["noise_seed"]), noise_p=float(p["noise_p"]),
                           dpw_hw=(int(p["dpw_h"]), int(p["dpw_w"])), rot90=bool(p["rot90"]), vflip=bool(p["vflip"]),
                           hflip=bool(p["hflip"]),
                           crop=(int(p["crop_i"]), int(p["crop_j"]), int(p["crop_h"]), int(p["crop_w"])) if p["crop"] else None,
                           out_size=out_size)


def _store(wafers):
    from ssl_wafermap_amd.data import WaferStore

    return WaferStore(wafers, device="cuda:0")


@pytest.mark.parametrize("denoise", [False, True])
def test_base_views_bit_exact(ref_vectors, denoise):
    from ssl_wafermap_amd.transforms import augment_views, get_base_transforms, sample_view_params

    wafers = load_ragged(ref_vectors, "wafer")
    store = _store(wafers)
    spec = get_base_transforms(denoise=denoise)
    rng = np.random.default_rng(11)
    idx = np.tile(np.arange(len(wafers)), 6)  # every wafer, many decision combinations
    params = sample_view_params(spec, idx, store.heights_np, store.widths_np, rng)
    assert set(np.unique(params["op"])) == ({1, 3} if denoise else {1, 2})
    out = augment_views(store, params, fmt="nchw_f32").cpu().numpy()
    u8 = augment_views(store, params, fmt="u8").cpu().numpy()
    bf = augment_views(store, params, fmt="nhwc_bf16")
    assert bf.shape == out.shape and bf.is_contiguous(memory_format=torch.channels_last)
    for v, p in enumerate(params):
        d = _decision(p, 224)
        assert np.array_equal(u8[v], oa.view_u8(wafers[p["sample"]], d)), (v, p)
        assert np.array_equal(out[v], oa.augment_view(wafers[p["sample"]], d)), (v, p)
    assert torch.equal(bf.cpu().float(), torch.from_numpy(out).bfloat16().float())


@pytest.mark.parametrize("out_size,scale", [(224, (0.6, 1.0)), (96, (0.1, 0.4))])
def test_multicrop_views_bit_exact(ref_vectors, out_size, scale):
    from ssl_wafermap_amd.transforms import augment_views, multicrop_view, sample_view_params

    wafers = load_ragged(ref_vectors, "wafer")
    store = _store(wafers)
    spec = multicrop_view(crop_size=out_size, crop_scale=scale)
    params = sample_view_params(spec, np.tile(np.arange(len(wafers)), 4), store.heights_np, store.widths_np,
                                np.random.default_rng(out_size))
    assert params["crop"].all()
    out = augment_views(store, params, out_size=out_size, fmt="nchw_f32").cpu().numpy()
    for v, p in enumerate(params):
        assert np.array_equal(out[v], oa.augment_view(wafers[p["sample"]], _decision(p, out_size))), (v, p)


def test_base_crop_and_inference_views(ref_vectors):
    from ssl_wafermap_amd.transforms import augment_views, get_base_transforms, get_inference_transforms, sample_view_params

    wafers = load_ragged(ref_vectors, "wafer")
    store = _store(wafers)
    rng = np.random.default_rng(3)
    spec = get_base_transforms(crop=True, normalize=False)
    params = sample_view_params(spec, np.tile(np.arange(len(wafers)), 4), store.heights_np, store.widths_np, rng)
    assert 0 < params["crop"].mean() < 1
    out = augment_views(store, params, fmt="nchw_f32", normalize=False).cpu().numpy()
    for v, p in enumerate(params):
        assert np.array_equal(out[v], oa.augment_view(wafers[p["sample"]], _decision(p, 224), normalize=False))
    inf = get_inference_transforms()
    params = sample_view_params(inf, np.arange(len(wafers)), store.heights_np, store.widths_np, rng)
    assert not params["op"].any() and not params["rot90"].any()
    out = augment_views(store, params, fmt="nchw_f32").cpu().numpy()
    for v, p in enumerate(params):
        assert np.array_equal(out[v], oa.augment_view(wafers[p["sample"]], oa.ViewDecision()))


def test_die_noise_kernel_reproduces_reference_given_its_rand_field(ref_vectors):
    """The kernel's RNG differs from torch.rand, so check the flip RULE on the kernel's own field:
    oracle.die_noise is pinned to the reference (test_oracle_augment) and must reproduce the kernel."""
    from ssl_wafermap_amd.transforms import augment_views, sample_view_params, get_base_transforms

    wafers = load_ragged(ref_vectors, "wafer")
    store = _store(wafers)
    spec = get_base_transforms(die_noise_prob=0.5, rr_prob=0, hf_prob=0, vf_prob=0)
    params = sample_view_params(spec, np.arange(len(wafers)), store.heights_np, store.widths_np, np.random.default_rng(0))
    params["op"] = 1
    params["dpw_h"], params["dpw_w"] = store.heights_np, store.widths_np
    u8 = augment_views(store, params, fmt="u8").cpu().numpy()
    flips = 0
    for v, p in enumerate(params):
        w = wafers[v]
        noisy = oa.die_noise(w, oa.rand01(int(p["noise_seed"]), w.size).reshape(w.shape), 0.5)
        flips += int((noisy != w).sum())
        assert np.array_equal(u8[v], oa.resize_nearest(noisy, 224, 224))
    assert flips > 1000


def test_full_batch_properties():
    """BASELINE batch (256 samples x 2 views, synthetic WM-811K-like wafers): value set, R==G==B,
    flips are involutions of the inference view."""
    from ssl_wafermap_amd.data.synthetic import synthetic_wafers
    from ssl_wafermap_amd.transforms import augment_views, get_base_transforms, get_inference_transforms, sample_view_params

    wafers, _ = synthetic_wafers(256, seed=1234)
    store = _store(wafers)
    spec = get_base_transforms()
    rng = np.random.default_rng(0)
    idx = np.arange(256)
    p0 = sample_view_params(spec, idx, store.heights_np, store.widths_np, rng, out_slot_base=0)
    p1 = sample_view_params(spec, idx, store.heights_np, store.widths_np, rng, out_slot_base=256)
    params = np.concatenate([p0, p1])
    out = augment_views(store, params, fmt="nchw_f32")
    assert out.shape == (512, 3, 224, 224)
    lut = torch.from_numpy(oa.to_tensor_normalize(np.array([[0, 128, 255]], dtype=np.uint8))[0, 0]).cuda()
    assert torch.isin(out, lut).all()
    assert torch.equal(out[:, 0], out[:, 1]) and torch.equal(out[:, 1], out[:, 2])
    pinf = sample_view_params(get_inference_transforms(), idx, store.heights_np, store.widths_np, rng)
    base = augment_views(store, pinf, fmt="u8")
    pf = pinf.copy()
    pf["hflip"] = 1
    pf["vflip"] = 1
    assert torch.equal(augment_views(store, pf, fmt="u8"), base.flip(1).flip(2))
    pr = pinf.copy()
    pr["rot90"] = 1
    assert torch.equal(augment_views(store, pr, fmt="u8"), torch.rot90(base, 1, (1, 2)))


def test_bad_params_are_rejected_on_the_host(ref_vectors):
    from ssl_wafermap_amd.transforms import augment_views, get_inference_transforms, sample_view_params

    wafers = load_ragged(ref_vectors, "wafer")
    store = _store(wafers)
    p = sample_view_params(get_inference_transforms(), np.arange(4), store.heights_np, store.widths_np, np.random.default_rng(0))
    bad = p.copy(); bad["sample"][0] = len(wafers)
    with pytest.raises(IndexError):
        augment_views(store, bad)
    bad = p.copy(); bad["crop"][1] = 1; bad["crop_i"][1] = 200; bad["crop_h"][1] = 100
    with pytest.raises(ValueError):
        augment_views(store, bad)
    bad = p.copy(); bad["out_slot"][2] = bad["out_slot"][3]
    with pytest.raises(ValueError):
        augment_views(store, bad)


def test_collate_function_returns_reference_triple(ref_vectors):
    from ssl_wafermap_amd.transforms import WaferDINOCOllateFunction, WaferImageCollateFunction, WaferMAECollateFunction2

    wafers = load_ragged(ref_vectors, "wafer")
    store = _store(wafers)
    batch = [(i, i % 3, f"w{i}.png") for i in range(6)]
    (x0, x1), labels, fnames = WaferImageCollateFunction(denoise=True).bind(store, np.random.default_rng(0), fmt="nchw_f32")(batch)
    assert x0.shape == x1.shape == (6, 3, 224, 224) and labels.tolist() == [0, 1, 2, 0, 1, 2] and fnames[3] == "w3.png"
    views, _, _ = WaferDINOCOllateFunction().bind(store, np.random.default_rng(1))(batch)
    assert [tuple(v.shape) for v in views] == [(6, 3, 224, 224)] * 2 + [(6, 3, 96, 96)] * 6
    x, _, _ = WaferMAECollateFunction2().bind(store, np.random.default_rng(2))(batch)
    assert x.shape == (6, 3, 224, 224)


def test_extreme_wafer_sizes():
    """Largest supported wafer (256 x 256: both LDS images at their maximum) and a 1 x 1 map."""
    from ssl_wafermap_amd.data import WaferStore
    from ssl_wafermap_amd.transforms import augment_views, get_base_transforms, get_inference_transforms, sample_view_params

    rng = np.random.default_rng(0)
    big = rng.choice(np.array([0, 128, 255], dtype=np.uint8), size=(256, 256))
    tiny = np.array([[255]], dtype=np.uint8)
    thin = rng.choice(np.array([128, 255], dtype=np.uint8), size=(3, 200))
    wafers = [big, tiny, thin]
    store = WaferStore(wafers, device="cuda:0")
    p = sample_view_params(get_inference_transforms(), np.arange(3), store.heights_np, store.widths_np, rng)
    out = augment_views(store, p, fmt="u8").cpu().numpy()
    for i, w in enumerate(wafers):
        assert np.array_equal(out[i], oa.resize_nearest(w, 224, 224))
    spec = get_base_transforms(denoise=True)
    p = sample_view_params(spec, np.array([0, 0, 2, 2]), store.heights_np, store.widths_np, rng)
    out = augment_views(store, p, fmt="u8").cpu().numpy()
    for v, q in enumerate(p):
        assert np.array_equal(out[v], oa.view_u8(wafers[q["sample"]], _decision(q, 224)))
    with pytest.raises(ValueError):  # DPW of a 1 x 1 wafer would be empty (the reference fails too)
        sample_view_params(get_base_transforms(), np.array([1] * 64), store.heights_np, store.widths_np, rng)


@pytest.mark.parametrize("denoise", [False, True])
def test_wafers_larger_than_the_lds_images_take_the_on_demand_path(denoise):
    """The kernel's LDS images hold 8 K elements (~90 x 90); a larger wafer (WM-811K goes up to 212 x 204) computes its
    stage-1 pixels on demand from the store.  Same bit-exact contract, every stage-1 op, base views and crops, and the
    s2d layout against the nhwc one."""
    from ssl_wafermap_amd import _lib
    from ssl_wafermap_amd._lib import check, ptr, stream_ptr
    from ssl_wafermap_amd.data.synthetic import synthetic_wafers
    from ssl_wafermap_amd.transforms import augment_views, get_base_transforms, multicrop_view, sample_view_params

    rng = np.random.default_rng(5)
    small, _ = synthetic_wafers(6, seed=3)
    big = []
    for h, w in ((212, 204), (120, 150), (200, 45), (91, 91), (256, 33)):
        yy, xx = np.mgrid[0:h, 0:w]
        inside = ((yy + 0.5 - h / 2) / (h / 2)) ** 2 + ((xx + 0.5 - w / 2) / (w / 2)) ** 2 <= 1.0
        big.append(np.where(inside, np.where(rng.random((h, w)) < 0.2, 255, 128), 0).astype(np.uint8))
    wafers = small + big
    store = _store(wafers)
    assert store.max_elems > 8192
    idx = np.tile(np.arange(len(wafers)), 5)
    for spec, out_size in ((get_base_transforms(denoise=denoise), 224), (multicrop_view(crop_size=96, crop_scale=(0.1, 0.4),
                                                                                    denoise=denoise), 96)):
        params = sample_view_params(spec, idx, store.heights_np, store.widths_np, rng)
        out = augment_views(store, params, out_size=out_size, fmt="nchw_f32").cpu().numpy()
        for v, p in enumerate(params):
            assert np.array_equal(out[v], oa.augment_view(wafers[p["sample"]], _decision(p, out_size))), (v, p)
        nhwc = augment_views(store, params, out_size=out_size, fmt="nhwc_bf16")
        s2d = augment_views(store, params, out_size=out_size, fmt="s2d_bf16")
        ref = torch.empty((len(params), out_size // 2, out_size // 2, 16), dtype=torch.bfloat16, device="cuda:0")
        check(_lib.load().wm_image_to_s2d(nhwc.data_ptr(), _lib.WM_IMG_NHWC_BF16, len(params), out_size, out_size, ptr(ref),
                                          stream_ptr()), "wm_image_to_s2d")
        assert torch.equal(s2d.permute(0, 2, 3, 1).contiguous(), ref)
