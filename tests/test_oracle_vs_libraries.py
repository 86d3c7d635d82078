"""
CPU tests: each third-party operation the oracle restates, against the library call
the reference makes (scipy.ndimage / numpy are present in the build image and on the
GPU box; the tests skip where scipy is not importable).
"""
import numpy as np
import pytest

from oracle import nellie_oracle as orc

ndi = pytest.importorskip("scipy.ndimage")


@pytest.mark.parametrize("shape,sig", [((9, 20, 23), (1.25, 1.25, 1.25)), ((5, 7, 40), (0.4167, 1.25, 1.25)),
                                        ((3, 4, 5), (2.9, 2.9, 2.9)), ((12, 16, 16), (0.0, 1.1, 0.9))])
def test_gaussian_matches_scipy_bitwise(shape, sig):
    rng = np.random.default_rng(1)
    a = rng.normal(100, 5, shape).astype(np.float32)
    ref = a.copy()
    ndi.gaussian_filter(ref, sigma=sig, output=ref, mode="reflect", cval=0.0, truncate=3.0)
    assert np.array_equal(orc.gaussian_filter_f32(a, sig), ref)


def test_gradient_matches_numpy_bitwise():
    rng = np.random.default_rng(2)
    a = rng.normal(0, 1, (6, 9, 11)).astype(np.float32)
    for axis, h in enumerate((0.3, 0.1, 0.1)):
        assert np.array_equal(orc.gradient_axis(a, h, axis), np.gradient(a, h, axis=axis))
    two = a[:2]
    assert np.array_equal(orc.gradient_axis(two, 0.3, 0), np.gradient(two, 0.3, axis=0))
    with pytest.raises(ValueError):
        orc.gradient_axis(a[:1], 0.3, 0)


@pytest.mark.parametrize("seed", range(4))
def test_histogram_matches_numpy(seed):
    rng = np.random.default_rng(seed)
    v = (rng.gamma(2.0, 3.0, 50000) if seed % 2 else rng.normal(100, 5, 50000)).astype(np.float32)
    v[:256] = np.linspace(v.min(), v.max(), 256, dtype=np.float32)   # values on/near bin edges
    c, e = orc.histogram_f32(v, 256)
    cr, er = np.histogram(v, bins=256, range=(v.min(), v.max()))
    assert np.array_equal(c, cr) and np.array_equal(e, er) and e.dtype == er.dtype
    one = np.full(10, 3.0, np.float32)
    c, e = orc.histogram_f32(one)
    cr, er = np.histogram(one, bins=256, range=(one.min(), one.max()))
    assert np.array_equal(c, cr) and np.array_equal(e, er)


def test_percentile_matches_numpy():
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 100, 101, 5000, 33333):
        v = rng.gamma(1.0, 0.01, n).astype(np.float32)
        assert orc.percentile_linear_f32(v, 1) == np.percentile(v, 1)
        assert type(np.percentile(v, 1)) is np.float32


def test_eigvalsh_closed_form_matches_lapack():
    rng = np.random.default_rng(4)
    m = 200000
    h = [rng.normal(0, s, m).astype(np.float32) for s in (300, 50, 40, 200, 60, 100)]
    H = np.stack([np.stack([h[0], h[1], h[2]], -1), np.stack([h[1], h[3], h[4]], -1),
                  np.stack([h[2], h[4], h[5]], -1)], -2)
    ref = np.linalg.eigvalsh(H)
    assert ref.dtype == np.float32
    got = orc.eigvalsh3_f32(*h)
    assert np.mean(got == ref) > 0.9999
    assert np.allclose(got, ref, rtol=0, atol=1e-4 * 300)
    assert np.array_equal(orc.sort_by_abs(ref),
                          np.take_along_axis(ref, np.argsort(np.abs(ref), axis=1), axis=1))


def test_morphology_and_labels_match_scipy():
    rng = np.random.default_rng(5)
    for p in (0.3, 0.55, 0.8):
        m = rng.random((9, 14, 13)) < p
        assert np.array_equal(orc.binary_dilation6(orc.binary_erosion6(m)), ndi.binary_opening(m))
        assert np.array_equal(orc.fill_holes6(m), ndi.binary_fill_holes(m))
        lab, n = ndi.label(m, structure=np.ones((3, 3, 3), bool))
        assert np.array_equal(orc.label26(m), lab) and lab.dtype == np.int32
        assert np.array_equal(orc.majority3(m), ndi.uniform_filter(m.astype(np.float32), size=3) > 0.5)
