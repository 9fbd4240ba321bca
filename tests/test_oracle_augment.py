"""The augmentation oracle against (a) golden vectors produced by the reference's own code
(tests/golden/make_reference_vectors.py) and (b) PIL / scipy for the third-party steps."""
import numpy as np
import pytest
from PIL import Image

from conftest import load_ragged
from oracle import augment as oa


def test_die_noise_matches_reference(ref_vectors):
    wafers = load_ragged(ref_vectors, "wafer")
    rands = load_ragged(ref_vectors, "dienoise_rand")
    outs = load_ragged(ref_vectors, "dienoise_out")
    flipped = 0
    for w, r, o, p in zip(wafers, rands, outs, ref_vectors["dienoise_p"]):
        got = oa.die_noise(w, r, p)
        assert np.array_equal(got, o)
        flipped += int((got != w).sum())
        assert np.array_equal(got == 0, w == 0)  # background untouched
    assert flipped > 100  # the vectors do exercise flips


def test_power_law_matches_reference(ref_vectors):
    for x, y in zip(ref_vectors["powerlaw_x"], ref_vectors["powerlaw_y"]):
        assert oa.power_law_transform(int(x)) == y  # bit-identical double


def test_dpw_transform_matches_reference(ref_vectors):
    wafers = load_ragged(ref_vectors, "wafer")
    outs = load_ragged(ref_vectors, "dpw_out")
    scales = ref_vectors["dpw_scales"]
    k = 0
    for w in wafers:
        for s in scales:
            got = oa.dpw_transform(w, float(s))
            assert got.shape == outs[k].shape
            assert np.array_equal(got, outs[k])
            k += 1


def test_dpw_call_matches_reference(ref_vectors):
    wafers = load_ragged(ref_vectors, "wafer")
    outs = load_ragged(ref_vectors, "dpwcall_out")
    for w, o, b in zip(wafers, outs, ref_vectors["dpwcall_beta"]):
        got = oa.dpw_call(w, float(b))
        assert got.shape == o.shape and np.array_equal(got, o)


@pytest.mark.parametrize("name,weights", [("uniform2", [0.5, 0.5]), ("w3", [0.2, 0.5, 0.3])])
def test_random_one_of_matches_reference(ref_vectors, name, weights):
    us, ks = ref_vectors[f"oneof_{name}_u"], ref_vectors[f"oneof_{name}_k"]
    # the reference normalises weights by their sum (augmentations.py:69-74)
    tot = sum(weights)
    wn = [w / tot for w in weights]
    got = [oa.random_one_of_index(float(u), wn) for u in us]
    assert got == list(ks)
    assert len(set(got)) == len(weights)


@pytest.mark.parametrize("n_out", [224, 96])
def test_nearest_map_matches_pil(n_out):
    for n_in in list(range(1, 257)):
        row = np.arange(n_in, dtype=np.uint8)[None, :].repeat(2, 0) if n_in <= 256 else None
        pil = np.asarray(Image.fromarray(row).resize((n_out, 2), Image.NEAREST))[0]
        assert np.array_equal(pil, oa.pil_nearest_map(n_in, n_out).astype(np.uint8)), n_in


def test_resize_rotate_flip_match_pil(ref_vectors):
    for w in load_ragged(ref_vectors, "wafer"):
        img = Image.fromarray(w)  # ToPILImage on a 2-D uint8 tensor -> mode "L"
        r = img.resize((224, 224), Image.NEAREST)
        assert np.array_equal(np.asarray(r), oa.resize_nearest(w, 224, 224))
        assert np.array_equal(np.asarray(r.rotate(90)), oa.rotate90(oa.resize_nearest(w, 224, 224)))
        assert np.array_equal(np.asarray(r.transpose(Image.FLIP_TOP_BOTTOM)), oa.resize_nearest(w, 224, 224)[::-1])
        assert np.array_equal(np.asarray(r.transpose(Image.FLIP_LEFT_RIGHT)), oa.resize_nearest(w, 224, 224)[:, ::-1])


def test_full_view_matches_pil_pipeline(ref_vectors):
    rng = np.random.default_rng(5)
    for w in load_ragged(ref_vectors, "wafer"):
        for out_size, scale in ((224, (0.6, 1.0)), (96, (0.1, 0.4))):
            d = oa.ViewDecision(op=oa.OP_DIENOISE, noise_seed=int(rng.integers(1 << 31)), noise_p=0.03,
                                rot90=bool(rng.integers(2)), vflip=bool(rng.integers(2)),
                                hflip=bool(rng.integers(2)),
                                crop=oa.random_resized_crop_params(224, 224, scale, *rng.random(3)),
                                out_size=out_size)
            img = Image.fromarray(oa.stage1(w, d)).resize((224, 224), Image.NEAREST)
            if d.rot90:
                img = img.rotate(90)
            if d.vflip:
                img = img.transpose(Image.FLIP_TOP_BOTTOM)
            if d.hflip:
                img = img.transpose(Image.FLIP_LEFT_RIGHT)
            i, j, h, ww = d.crop
            img = img.crop((j, i, j + ww, i + h)).resize((out_size, out_size), Image.NEAREST)
            assert np.array_equal(np.asarray(img), oa.view_u8(w, d))


def test_median3_matches_scipy(ref_vectors):
    from scipy.ndimage import median_filter

    for w in load_ragged(ref_vectors, "wafer"):
        assert np.array_equal(oa.median3(w), median_filter(w, size=3, mode="nearest"))


def test_to_tensor_normalize_values():
    x = oa.to_tensor_normalize(np.array([[0, 128, 255]], dtype=np.uint8))
    assert x.shape == (3, 1, 3) and x.dtype == np.float32
    # the three possible pixel values
    np.testing.assert_allclose(x[0, 0], [-1.5366, 0.1790, 1.8811], atol=1e-4)
    assert np.array_equal(x[0], x[1]) and np.array_equal(x[1], x[2])


def test_rand01_is_uniform_and_deterministic():
    a, b = oa.rand01(123, 100000), oa.rand01(123, 100000)
    assert np.array_equal(a, b) and a.dtype == np.float32
    assert 0.0 <= a.min() and a.max() < 1.0
    assert abs(a.mean() - 0.5) < 5e-3 and abs((a < 0.03).mean() - 0.03) < 2e-3
    assert not np.array_equal(a, oa.rand01(124, 100000))
