import glob
import os

# every range + histogram chain of the GPU tests also checks that the bin edges the device built are numpy's, bit for bit
os.environ.setdefault("NELLIE_CHECK_EDGES", "1")
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _hip_device_count():
    try:
        from nellie_amd import hipnative
        return hipnative.load().device_count()
    except Exception:
        return 0


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests/` on a box without a GPU skips the gpu-marked tests instead of failing them.  Selecting them
    (`-m gpu`, what the GPU box runs) keeps the loud failure: a missing device or library must not read as green."""
    expr = config.getoption("markexpr") or ""
    if "gpu" in expr and "not gpu" not in expr:
        return
    if os.environ.get("NELLIE_REQUIRE_GPU") == "1" or not any("gpu" in it.keywords for it in items):
        return
    if _hip_device_count() > 0:
        return
    skip = pytest.mark.skip(reason="no HIP device (run with -m gpu on the GPU box to make this an error)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def golden_names(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def load_golden(name):
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))
    if "input" not in g and "input_shape" in g:
        import zlib
        from nellie_amd.synthetic import make_image_2d, make_volume
        shape = tuple(int(s) for s in g["input_shape"])
        vol = (make_image_2d if len(shape) == 2 else make_volume)(shape, int(g["input_seed"]))
        assert np.uint32(zlib.crc32(vol.tobytes())) == g["input_crc"], "synthetic generator drifted"
        g["input"] = vol
    if "dim_res" in g:
        z, y, x = (float(v) for v in g["dim_res"])
        g["dim_res_dict"] = {"X": x, "Y": y, "Z": None if np.isnan(z) else z, "T": 1.0}
    g["kwargs"] = {}
    for k in list(g):
        if k.startswith("kw_"):
            v = float(g[k])
            name_ = k[3:]
            if np.isnan(v):
                g["kwargs"][name_] = None
            elif name_ == "frob_thresh_division":
                g["kwargs"][name_] = int(v)
            else:
                g["kwargs"][name_] = v
    # the `mask` argument of Filter.run() (filtering.py:1033) is a property of the run, not of the constructor: kept apart
    g["run_mask"] = bool(g["kwargs"].pop("mask", 1.0))
    return g


FILTER_CASES = [n for n in golden_names() if not n.startswith(("labelonly", "labelintensity", "removeedges", "twod", "markers", "network"))]
MARKERS_CASES = golden_names("markers")
NETWORK_CASES = golden_names("network")
FILTER_2D_CASES = golden_names("twod")
LABEL_INTENSITY_CASES = golden_names("labelintensity")
LABEL_ONLY_CASES = golden_names("labelonly")


@pytest.fixture(scope="session")
def hip():
    """The product's HIP library handle; GPU tests fail loudly when it is missing."""
    from nellie_amd import hipnative
    return hipnative.load()
