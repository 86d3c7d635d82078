"""
CPU tests: the oracle (oracle/nellie_oracle.py + oracle/ccl_oracle.c) against the
golden vectors captured from the imported reference (tests/golden/make_golden.py).
This is what pins the oracle; everything up to the final Frangi product must be
bit-equal, and on these fixtures the final product is bit-equal too (same numpy exp).
"""
import zlib

import numpy as np
import pytest

from conftest import FILTER_2D_CASES, FILTER_CASES, LABEL_INTENSITY_CASES, LABEL_ONLY_CASES, MARKERS_CASES, load_golden
from oracle import nellie_oracle as orc


def _filter_kwargs(g):
    kw = dict(g["kwargs"])
    sig_kw = {k: kw.pop(k) for k in ("min_radius_um", "max_radius_um") if k in kw}
    return kw, sig_kw


@pytest.mark.parametrize("name", FILTER_CASES)
def test_filter_matches_reference(name):
    g = load_golden(name)
    dr = g["dim_res_dict"]
    kw, sig_kw = _filter_kwargs(g)
    vol = g["input"]
    if "error_type" in g:
        with pytest.raises(ValueError, match=str(g["error_msg"])[:30]) as ei:       # (numpy.linalg.LinAlgError is a ValueError)
            orc.filter_frame(vol, dr, mask=g["run_mask"], **kw)
        assert type(ei.value).__name__ == str(g["error_type"])
        return
    sigmas = orc.default_sigmas(dr, **sig_kw)
    assert np.array_equal(np.array(sigmas), g["sigmas"])
    trace = []
    vess, masks = orc.compute_vesselness(vol, dr, sigmas=sigmas, trace=trace, mask=g["run_mask"], **kw)
    assert len(trace) == len(g["gamma"])
    for s, rec in enumerate(trace):
        assert rec["gamma"] == g["gamma"][s]
        assert np.uint32(zlib.crc32(rec["gauss"].tobytes())) == g["gauss_crc"][s], f"gauss scale {s}"
        assert np.array_equal(rec["gauss"][vol.shape[0] // 2], g["gauss_mid_planes"][s])
        assert rec["max_abs"] == g["max_abs"][s]
        if g["run_mask"] and not np.isnan(g["frob_thr"][s]):      # (mask=False never derives a threshold)
            assert rec["frob_thr"] == g["frob_thr"][s]
        assert rec["mask_count"] == g["mask_count"][s]
        if not g["run_mask"]:
            assert rec["mask_count"] == vol.size          # h_mask = ones_like(image) (filtering.py:566-567)
    fr = vess * masks
    assert np.array_equal(fr, g["run_frame"])
    if float(fr.sum()) > 0:
        out, thr = orc.mask_volume(fr, return_thr=True)
        assert float(thr) == float(g["percentile_thr"])
    else:
        out = fr
    assert out.dtype == np.float32
    assert np.array_equal(out, g["frangi"])


@pytest.mark.parametrize("name", [n for n in FILTER_CASES])
def test_label_matches_reference_given_reference_frangi(name):
    g = load_golden(name)
    if "error_type" in g:
        pytest.skip("reference raises on this input")
    labels, thr = orc.label_frame(g["frangi"], g["dim_res_dict"], return_thr=True)
    if np.isnan(g["label_thr"]):
        assert thr is None
    else:
        assert float(thr) == float(g["label_thr"])
    assert orc.min_area_pixels(g["dim_res_dict"]) == int(g["min_area_pixels"])
    assert labels.dtype == np.int32
    assert np.array_equal(labels, g["labels"])


@pytest.mark.parametrize("name", FILTER_2D_CASES)
def test_filter_and_label_2d_match_reference(name):
    """2-D images (im_info.no_z): per-scale intermediates, _run_frame (with the LoG blob response), _mask_volume and
    Label, all bit-equal to the imported reference."""
    g = load_golden(name)
    dr = g["dim_res_dict"]
    kw = dict(g["kwargs"])
    rm = bool(kw.pop("remove_edges", False))
    img = g["input"]
    assert img.ndim == 2
    sigmas = orc.default_sigmas_2d(dr)
    assert np.array_equal(np.array(sigmas), g["sigmas"])
    trace = []
    orc.compute_vesselness_2d(img, dr, sigmas=sigmas, trace=trace, mask=g["run_mask"], **kw)
    assert len(trace) == len(g["gamma"])
    for s, rec in enumerate(trace):
        assert rec["gamma"] == g["gamma"][s]
        assert np.uint32(zlib.crc32(rec["gauss"].tobytes())) == g["gauss_crc"][s], f"gauss scale {s}"
        assert rec["max_abs"] == g["max_abs"][s]
        if g["run_mask"] and not np.isnan(g["frob_thr"][s]):      # (mask=False never derives a threshold)
            assert rec["frob_thr"] == g["frob_thr"][s]
        assert rec["mask_count"] == g["mask_count"][s]
    fr = orc.run_frame_2d(img, dr, remove_edges_flag=rm, mask=g["run_mask"], **kw)
    assert np.array_equal(fr, g["run_frame"])
    if float(fr.sum()) > 0:
        out, thr = orc.mask_volume_2d(fr, return_thr=True)
        assert float(thr) == float(g["percentile_thr"])
    else:
        out = fr
    assert out.dtype == np.float32 and np.array_equal(out, g["frangi"])
    labels, lthr = orc.label_frame_2d(g["frangi"], dr, return_thr=True)
    if np.isnan(g["label_thr"]):
        assert lthr is None
    else:
        assert float(lthr) == float(g["label_thr"])
    assert orc.min_area_pixels_2d(dr) == int(g["min_area_pixels"])
    assert np.array_equal(labels, g["labels"])


@pytest.mark.parametrize("name", MARKERS_CASES)
def test_markers_match_reference(name):
    """Markers stage (mocap_marking.py): marker, distance and border images bit-equal to the imported reference."""
    g = load_golden(name)
    dr = g["dim_res_dict"]
    kw = {k: (int(v) if k in ("peak_min_distance", "num_sigma") else v) for k, v in g["kwargs"].items()}
    sig, _ = orc.marker_sigmas(dr, num_sigma=kw.get("num_sigma", 5))
    assert np.array_equal(np.array(sig), g["sigmas"])
    marker, dist, border = orc.markers_frame(g["input"], g["labels_in"], dr, frangi=g.get("frangi_in"), **kw)
    assert marker.dtype == np.uint8 and dist.dtype == np.float32 and border.dtype == np.uint8
    assert np.array_equal(dist, g["distance"])
    assert np.array_equal(border, g["border"])
    assert np.array_equal(marker, g["marker"])


@pytest.mark.parametrize("name", LABEL_ONLY_CASES)
def test_label_only_cases(name):
    g = load_golden(name)
    labels, thr = orc.label_frame(g["frangi"], g["dim_res_dict"], return_thr=True)
    assert float(thr) == float(g["label_thr"])
    assert np.array_equal(labels, g["labels"])
    assert labels.max() >= 4


def test_reference_toy_label_semantics():
    """tests/test_labelling.py:25-53 restated for 3-D: ids restart at 1 per call, subset of {0,1}."""
    fr = np.zeros((5, 7, 7), np.float32)
    fr[1:4, 1:6, 1:6] = 1.0
    dr = {"X": 1.0, "Y": 1.0, "Z": 1.0, "T": 1.0}
    for _ in range(2):
        _, labels = orc.get_labels(fr, 0.5, orc.min_area_pixels(dr))
        assert labels.max() == 1 and set(np.unique(labels)) <= {0, 1}


def test_remove_edges_matches_reference():
    g = load_golden("removeedges_16x96x40_s8")
    fr = orc.run_frame(g["input"], g["dim_res_dict"], remove_edges_flag=True)
    assert np.array_equal(fr, g["run_frame"])
    assert np.array_equal(orc.mask_volume(fr), g["frangi"])
    assert (g["run_frame"] > 0).any()


@pytest.mark.parametrize("name", LABEL_INTENSITY_CASES)
def test_label_with_intensity_threshold_matches_reference(name):
    g = load_golden(name)
    kw = dict(otsu_thresh_intensity=True) if int(g["otsu"]) else dict(threshold=float(g["threshold"]))
    labels, thr = orc.label_frame(g["frangi"], g["dim_res_dict"], return_thr=True, original=g["input"], **kw)
    assert float(thr) == float(g["label_thr"])
    assert np.array_equal(labels, g["labels"])


@pytest.mark.parametrize("name", __import__("conftest").NETWORK_CASES)
def test_network_steps_match_reference(name):
    """Network's pixel classes and branch labels (networking.py:672-683, 758-800) against the reference's outputs."""
    g = load_golden(name)
    pc = orc.network_pixel_class(g["skel"])
    assert str(pc.dtype) == str(g["pixel_class_dtype"]) and np.array_equal(pc, g["pixel_class"])
    bl = orc.network_branch_skel_labels(pc)
    assert str(bl.dtype) == str(g["branch_labels_dtype"]) and np.array_equal(bl, g["branch_labels"])


def _given_scales(gamma, max_abs, frob_thr, mask_count):
    return [dict(gamma=float(a), max_abs=float(b), frob_thr=None if np.isnan(c) else float(c), skipped=int(d) == 0)
            for a, b, c, d in zip(gamma, max_abs, frob_thr, mask_count)]


def crop_valid_slices(box, shape, margin):
    """Within a crop `box` = ((z0, z1), (y0, y1), (x0, x1)) of a volume of `shape`: the part that `filter_frame_crop` reproduces
    exactly -- `margin` voxels in from every face of the box that is not a face of the volume."""
    return tuple(slice(0 if a == 0 else m, (b - a) - (0 if b == n else m)) for (a, b), n, m in zip(box, shape, margin))


@pytest.mark.parametrize("name,boxes", [
    ("strided_50x150x141_s5", [((0, 50), (20, 130), (0, 100)), ((0, 50), (0, 150), (30, 141)), ((0, 50), (40, 150), (41, 141))]),
    ("someempty_24x48x48_s0", [((0, 24), (0, 48), (0, 48))]),
    ("aniso_20x40x44_s3", [((0, 20), (0, 40), (0, 44))]),
])
def test_crop_mode_reproduces_the_reference_on_crop_interiors(name, boxes):
    """The crop mode of the oracle (`filter_frame_crop`: volume-wide thresholds supplied, per-voxel arithmetic on a box) against the
    REFERENCE's own outputs: the golden case's thresholds in, the golden `run_frame` / `frangi` arrays on the crop's valid interior
    out, bit for bit.  This is what licenses the voxel-level checks of 1024^3 and 128x2048x2048 runs in test_hip_full_size.py."""
    g = load_golden(name)
    dr = g["dim_res_dict"]
    kw, sig_kw = _filter_kwargs(g)
    vol = g["input"]
    sigmas = orc.default_sigmas(dr, **sig_kw)
    given = _given_scales(g["gamma"], g["max_abs"], g["frob_thr"], g["mask_count"])
    margin = orc.crop_margin(dr, sigmas)
    masked = float(g["run_frame"].sum()) > 0
    for box in boxes:
        sl = tuple(slice(a, b) for a, b in box)
        valid = crop_valid_slices(box, vol.shape, margin)
        if any(v.stop - v.start <= 0 for v in valid):
            pytest.fail(f"box {box} has no valid interior with margin {margin}")
        raw = orc.filter_frame_crop(vol[sl], dr, given, None, sigmas=sigmas, **kw)
        assert np.array_equal(raw[valid], g["run_frame"][sl][valid])
        if masked:
            out = orc.filter_frame_crop(vol[sl], dr, given, float(g["percentile_thr"]), sigmas=sigmas, **kw)
            assert np.array_equal(out[valid], g["frangi"][sl][valid])
    # the margin matters: close to an artificial face the box's own reflect padding / one-sided differences show
    if name.startswith("strided"):
        box = ((0, 50), (20, 130), (0, 100))
        sl = tuple(slice(a, b) for a, b in box)
        valid = crop_valid_slices(box, vol.shape, (3, 3, 3))
        raw = orc.filter_frame_crop(vol[sl], dr, given, None, sigmas=sigmas, **kw)
        assert not np.array_equal(raw[valid], g["run_frame"][sl][valid])
