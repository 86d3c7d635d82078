"""CPU tests of the boundary: the library builds for gfx950, loads, exports every symbol
include/nellie_amd.h declares, and fails loudly (no CPU fallback) without a GPU."""
import os
import re
import shutil

import numpy as np
import pytest

from conftest import REPO


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "nellie_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nl_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        from nellie_amd import hipnative
        if not os.path.exists(hipnative.LIB_PATH):
            pytest.skip("no hipcc and no prebuilt libnellie_hip.so")
    else:
        from nellie_amd import build
        build.build(verbose=False)
    from nellie_amd import hipnative
    return hipnative.load()


def test_header_and_binding_agree(lib):
    from nellie_amd import hipnative
    declared = _declared_symbols()
    assert declared == hipnative.ALL_SYMBOLS, set(declared) ^ set(hipnative.ALL_SYMBOLS)
    for name in declared:
        assert hasattr(lib.cdll, name), f"{name} declared in include/nellie_amd.h but not exported"
    assert "gfx950" in lib.version()


def test_no_cpu_fallback_without_gpu(lib):
    from nellie_amd import hipnative
    if lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="GPU backend requested"):
        hipnative.Context((4, 8, 8))
    from types import SimpleNamespace
    from nellie_amd.segmentation.filtering import Filter
    from nellie_amd.segmentation.labelling import Label
    im = SimpleNamespace(no_t=True, no_z=False, shape=(1, 4, 8, 8), axes="TZYX",
                         dim_res={"X": .1, "Y": .1, "Z": .1, "T": 1.0})
    for cls in (Filter, Label):
        with pytest.raises(RuntimeError, match="GPU backend requested"):
            cls(im, device="gpu")
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            cls(im, device="cpu")
        with pytest.raises(ValueError):
            cls(im, device="tpu")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "nellie_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f"{f} mentions the oracle"


def test_host_thresholds_match_reference_formulas():
    """nellie_amd.utils.gpu_functions on numpy histograms == the oracle's restatement."""
    from nellie_amd.utils import gpu_functions as gf
    from oracle import nellie_oracle as orc
    rng = np.random.default_rng(0)
    for seed in range(3):
        v = rng.gamma(2.0, 3.0, 20000).astype(np.float32)
        c, e = orc.histogram_f32(v)
        assert np.array_equal(gf.histogram_edges(v.min(), v.max()), e)
        assert gf.min_triangle_otsu(c, e) == orc.min_tri_otsu(v)
        assert gf.otsu_threshold(v)[0] == orc.otsu_threshold(v)
        assert gf.triangle_threshold(v) == orc.triangle_threshold(v)


def test_host_thresholds_keep_numpys_edge_dtype():
    """Label's intensity thresholds run on the original image (labelling.py:414-431): uint16 / float64 data give float64
    histogram edges and float64 bin centres; float32 data float32 ones.  Value, dtype and the variance gpu_functions.py:50
    returns beside the threshold all equal the numpy formulas."""
    from nellie_amd.utils import gpu_functions as gf
    from oracle import nellie_oracle as orc
    rng = np.random.default_rng(5)
    for k, data in enumerate((rng.gamma(2.0, 300.0, 30000).astype(np.uint16), rng.gamma(2.0, 0.3, 30000), rng.random(5000).astype(np.float32),
                              rng.integers(1, 4096, 40000).astype(np.uint16))):
        flat = data.reshape(-1)
        counts, edges = np.histogram(flat, bins=256, range=(flat.min(), flat.max()))
        want = orc.otsu_from_hist(counts, edges)
        got, var = gf.otsu_threshold(data)
        assert got == want and got.dtype == want.dtype == edges.dtype, (k, got, want)
        c = counts / np.sum(counts)
        centres = (edges[:-1] + edges[1:]) / 2.0
        w1 = np.cumsum(c); m1 = np.cumsum(c * centres) / w1
        w2 = np.cumsum(c[::-1])[::-1]; m2 = (np.cumsum((c * centres)[::-1]) / w2[::-1])[::-1]
        v12 = w1[:-1] * w2[1:] * (m1[:-1] - m2[1:]) ** 2
        assert var == v12[np.argmax(v12)]
        tri = gf.triangle_threshold(data)
        assert tri == orc.triangle_from_hist(counts, edges) and tri.dtype == edges.dtype


def test_host_parameters_match_oracle():
    from nellie_amd import pipeline as pl
    from oracle import nellie_oracle as orc
    for dr in ({"X": .1, "Y": .1, "Z": .1}, {"X": .1, "Y": .1, "Z": .3}, {"X": .065, "Y": .065, "Z": .2}):
        assert pl.default_sigmas(dr) == orc.default_sigmas(dr)
        s = pl.default_sigmas(dr)
        assert pl.cascade_deltas(s, pl.z_ratio_of(dr)) == orc.cascade_deltas(s, orc.z_ratio(dr))
        assert pl.min_area_pixels_of(dr) == orc.min_area_pixels(dr)
    for shape in ((24, 48, 48), (256, 512, 512), (1024, 1024, 1024), (1024, 2048, 2048), (50, 150, 141)):
        assert pl.sample_strides(shape, int(1e6)) == orc.sample_strides(shape)
    for sd in (1.25, 0.4166, 2.9, 1e-16):
        w = pl.gaussian_weights(sd)
        if sd <= 1e-15:
            assert w is None
        else:
            assert np.array_equal(w, orc.gaussian_kernel1d(sd, orc.gaussian_radius(sd)))


def test_host_histogram_thresholds_equal_numpy_then_library():
    """nl_host_hist_thresholds_f32 = np.histogram of float32 data + nl_hist_thresholds_ex, bit for bit (labelling.py:448-455)."""
    import numpy as np
    from nellie_amd import hipnative
    rng = np.random.default_rng(7)
    cases = [np.log10(rng.random(n, dtype=np.float32) + np.float32(1e-6)) for n in (2, 3, 17, 1000, 30011, 70001)]
    cases.append(np.log10(rng.lognormal(0.0, 2.0, 20000).astype(np.float32)))
    cases.append(np.array([0.25, 0.25, 0.25, 0.5], np.float32))                      # values on bin edges
    cases.append(np.linspace(-3.0, 2.0, 257, dtype=np.float32))                     # every value on or near an edge
    cases.append(np.float32(1e-30) * rng.random(5000, dtype=np.float32))            # tiny range
    for v in cases:
        counts, edges = np.histogram(v, bins=256, range=(v.min(), v.max()))
        assert edges.dtype == np.float32
        try:
            want = hipnative.hist_thresholds(counts, edges)
        except ValueError:
            want = None
        if want is None:
            with pytest.raises(ValueError):
                hipnative.host_hist_thresholds(v, 256)
            continue
        tri, otsu, c2, e2 = hipnative.host_hist_thresholds(v, 256, with_histogram=True)
        assert np.array_equal(c2, counts) and np.array_equal(e2, edges)
        assert (tri, otsu) == want and type(tri) is type(want[0])
    same = np.full(100, 0.5, np.float32)                                             # empty range: numpy widens by +-0.5
    counts, edges = np.histogram(same, bins=256, range=(same.min(), same.max()))
    _, _, c2, e2 = _with_hist_or_none(hipnative, same)
    assert np.array_equal(c2, counts) and np.array_equal(e2, edges)
    with pytest.raises(ValueError):
        hipnative.host_hist_thresholds(np.array([1.0, np.inf], np.float32), 256)


def _with_hist_or_none(hipnative, v):
    import numpy as np
    import ctypes as C
    # the thresholds of a one-bin histogram are degenerate (ValueError): fetch the histogram alone through the C entry point
    counts, edges = np.zeros(256, np.int64), np.zeros(257, np.float32)
    tri, otsu, st = C.c_double(0), C.c_double(0), C.c_int(0)
    hipnative.load().call("nl_host_hist_thresholds_f32", v.ctypes.data_as(C.c_void_p), int(v.size), 256, C.byref(tri), C.byref(otsu),
                          C.byref(st), counts.ctypes.data_as(C.c_void_p), edges.ctypes.data_as(C.c_void_p))
    return tri.value, otsu.value, counts, edges


def test_percentile_shortcut_is_numpy_percentile():
    """pipeline.percentile_of_samples: the shortcut passes its own probe with the installed numpy and equals np.percentile."""
    import numpy as np
    from nellie_amd import pipeline as pl
    assert pl._shortcut_matches_numpy()
    rng = np.random.default_rng(11)
    for n in (1, 2, 5, 99, 100, 101, 199, 200, 201, 12345, 50000):
        for make in (lambda: rng.random(n, dtype=np.float32), lambda: np.round(rng.random(n, dtype=np.float32), 1) + np.float32(0.1),
                     lambda: (rng.random(n, dtype=np.float32) * np.float32(1e-20)).astype(np.float32)):
            v = make()
            keep = v.copy()
            got, want = pl.percentile_of_samples(v, 1), np.percentile(v, 1)
            assert type(got) is type(want) and got == want, (n, got, want)
            assert np.array_equal(v, keep)                        # the caller's samples are left alone
    assert pl.percentile_of_samples(np.arange(10.0), 1) == np.percentile(np.arange(10.0), 1)      # other dtypes: numpy itself


def test_percentile_falls_back_to_numpy_when_the_probe_fails(monkeypatch):
    """A numpy whose percentile the shortcut does not reproduce (the probe says so) gets np.percentile itself."""
    import numpy as np
    from nellie_amd import pipeline as pl
    calls = []
    monkeypatch.setattr(pl, "_FAST_PERCENTILE", None)
    monkeypatch.setattr(pl, "_shortcut_matches_numpy", lambda: False)
    monkeypatch.setattr(pl, "_percentile_shortcut", lambda v, q: calls.append(1) or np.float32(-1))
    v = np.random.default_rng(3).random(1000, dtype=np.float32)
    assert pl.percentile_of_samples(v, 1) == np.percentile(v, 1) and not calls and pl._FAST_PERCENTILE is False
