"""
Per-frame orchestration of the HIP engine: the host half of Nellie's Filter -> Label hot path.

The device does every full-volume pass (libnellie_hip.so, include/nellie_amd.h); the host
decides the data-dependent thresholds between passes exactly where the reference does
(nellie/segmentation/filtering.py:806-853, 952-967; labelling.py:385-455), from 256-bin
histograms / <= 1e6 samples the device hands back.  No array math on full volumes happens
here, and there is no CPU fallback: without the HIP library this module cannot run.

`FramePipeline` keeps one frame resident in HBM from upload to label download; the
`Filter` / `Label` stage classes (nellie_amd/segmentation) wrap it behind the reference's
stage API and on-disk layout.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import os

import numpy as np

from nellie_amd import hipnative
from nellie_amd.hipnative import FIELD_FRANGI, FIELD_FROB, FIELD_GAUSS, FIELD_VESSELNESS
from nellie_amd.utils.gpu_functions import (histogram_edges, min_triangle_otsu, otsu_threshold,
                                            triangle_threshold)

_EPS32 = float(np.finfo(np.float32).eps)
_TRUNCATE = 3.0


# ----------------------------------------------------------------------------- parameters
def z_ratio_of(dim_res) -> float:
    """filtering.py:75-78."""
    z_res = dim_res.get("Z") or dim_res.get("X") or 1.0
    x_res = dim_res.get("X") or 1.0
    return float(z_res) / float(x_res)


def spacing_of(dim_res):
    """filtering.py:265-275."""
    z = dim_res.get("Z") or dim_res.get("X") or 1.0
    y = dim_res.get("Y") or 1.0
    x = dim_res.get("X") or 1.0
    return (float(z), float(y), float(x))


def default_sigmas(dim_res, min_radius_um=0.25, max_radius_um=1.0):
    """filtering.py:88-89, 288-316."""
    min_radius_px = min_radius_um / dim_res["X"]
    max_radius_px = max_radius_um / dim_res["X"]
    min_sigma_step_size = 0.2
    num_sigma = 5
    sigma_1 = min_radius_px / 2.0
    sigma_2 = max_radius_px / 3.0
    sigma_min = min(sigma_1, sigma_2)
    sigma_max = max(sigma_1, sigma_2)
    if sigma_max <= sigma_min:
        sigma_max = sigma_min + min_sigma_step_size
    sigma_step_size_calculated = (sigma_max - sigma_min) / float(num_sigma)
    sigma_step_size = max(min_sigma_step_size, sigma_step_size_calculated)
    sigmas = list(np.arange(sigma_min, sigma_max, sigma_step_size, dtype=float))
    sigmas.sort()
    return sigmas


def sample_strides(shape, max_samples):
    """filtering.py:328-340."""
    if max_samples is None or max_samples <= 0:
        return (1,) * len(shape)
    total = int(np.prod(shape))
    if total <= max_samples:
        return (1,) * len(shape)
    ndim = len(shape)
    stride = int(np.ceil((total / max_samples) ** (1.0 / ndim)))
    strides = [max(1, stride) for _ in range(ndim)]
    while int(np.prod([int(np.ceil(s / st)) for s, st in zip(shape, strides)])) > max_samples:
        idx = int(np.argmax([s / st for s, st in zip(shape, strides)]))
        strides[idx] += 1
    return tuple(strides)


_WEIGHTS_CACHE = {}


def gaussian_weights(sigma: float):
    """
    scipy.ndimage `gaussian_filter1d` + `_gaussian_kernel1d` (order 0, truncate 3.0):
    lw = int(truncate*sd + 0.5); w = exp(-0.5/sd^2 * x^2) / sum.  None when the axis is
    skipped (scipy `gaussian_filter` skips sigma <= 1e-15).  Computed with numpy exactly as
    scipy does, so the float64 weights carry scipy's bits.  The (read-only) array of a sigma is kept: a stack's frames
    ask for the same fifteen kernels again and again, 70 us of numpy per frame with the GPU waiting for the first of them.
    """
    sd = float(sigma)
    if not sd > 1e-15:
        return None
    got = _WEIGHTS_CACHE.get(sd)
    if got is not None:
        return got
    lw = int(_TRUNCATE * sd + 0.5)
    sigma2 = sd * sd
    x = np.arange(-lw, lw + 1)
    phi_x = np.exp(-0.5 / sigma2 * x ** 2)
    phi_x = phi_x / phi_x.sum()
    w = np.ascontiguousarray(phi_x[::-1])
    w.setflags(write=False)
    if len(_WEIGHTS_CACHE) < 4096:
        _WEIGHTS_CACHE[sd] = w
    return w


def gaussian_derivative_weights(sigma: float, order: int, truncate: float = 4.0):
    """scipy.ndimage `_gaussian_kernel1d(sigma, order, radius)` reversed as `gaussian_filter1d` does, radius =
    int(truncate*sd + 0.5) -- `gaussian_laplace` runs with scipy's default truncate = 4.0 (filtering.py:781)."""
    sd = float(sigma)
    radius = int(truncate * sd + 0.5)
    exponent_range = np.arange(order + 1)
    sigma2 = sd * sd
    x = np.arange(-radius, radius + 1)
    phi_x = np.exp(-0.5 / sigma2 * x ** 2)
    phi_x = phi_x / phi_x.sum()
    if order == 0:
        return np.ascontiguousarray(phi_x[::-1])
    q = np.zeros(order + 1)
    q[0] = 1
    D = np.diag(exponent_range[1:], 1)
    P = np.diag(np.ones(order) / -sigma2, -1)
    Q_deriv = D + P
    for _ in range(order):
        q = Q_deriv.dot(q)
    q = (x[:, None] ** exponent_range).dot(q)
    return np.ascontiguousarray((q * phi_x)[::-1])


def marker_sigmas(dim_res, min_radius_um=0.20, max_radius_um=1, num_sigma=5):
    """mocap_marking.py:121-134, 329-362: (sigmas, max_radius_px)."""
    x_res = dim_res.get("X") or 1.0
    min_r = max(min_radius_um, float(x_res)) / float(x_res)
    max_r = max_radius_um / float(x_res)
    sigma_min, sigma_max = min_r / 2.0, max_r / 3.0
    rng = sigma_max - sigma_min
    if rng <= 0:
        return [sigma_min], max_r
    step = max(0.2, rng / max(num_sigma, 1))
    sig = list(np.arange(sigma_min, sigma_max, step))
    return (sig if len(sig) else [sigma_min]), max_r


_DELTAS_CACHE = {}


def cascade_deltas(sigmas, z_ratio):
    """filtering.py:814-825 (kept per (sigmas, z_ratio): the same for every frame of a stack)."""
    key = (tuple(float(s) for s in sigmas), float(z_ratio))
    got = _DELTAS_CACHE.get(key)
    if got is None:
        got = _cascade_deltas(sigmas, z_ratio)
        if len(_DELTAS_CACHE) < 256:
            _DELTAS_CACHE[key] = got
    return list(got)


def _cascade_deltas(sigmas, z_ratio):
    out = []
    prev = 0.0
    for sigma in sigmas:
        vp = (float(prev) / z_ratio, float(prev), float(prev))
        vc = (float(sigma) / z_ratio, float(sigma), float(sigma))
        delta = []
        for sp, sc in zip(vp, vc):
            diff = max(0.0, float(sc) ** 2 - float(sp) ** 2)
            delta.append(np.sqrt(diff))
        out.append(tuple(delta))
        prev = sigma
    return out


def min_area_pixels_of(dim_res, min_radius_um=0.25, no_z=False):
    """labelling.py:95-97, 209-219."""
    x_res = dim_res.get("X") or 1.0
    y_res = dim_res.get("Y") or x_res
    r = max(float(min_radius_um), float(x_res))
    if no_z:
        area_px = (np.pi * (r ** 2)) / (float(x_res) * float(y_res))
        return max(1, int(np.ceil(area_px)))
    z_res = dim_res.get("Z") or x_res
    volume_um3 = (4.0 / 3.0) * np.pi * (r ** 3)
    volume_px = volume_um3 / (float(x_res) * float(y_res) * float(z_res))
    return max(1, int(np.ceil(volume_px)))


@dataclass
class FilterParams:
    dim_res: dict
    min_radius_um: float = 0.25
    max_radius_um: float = 1.0
    alpha_sq: float = 0.5
    beta_sq: float = 0.5
    frob_thresh: Optional[float] = None
    frob_thresh_division: object = 2
    max_threshold_samples: int = int(1e6)
    sigmas: Optional[list] = None

    def resolved_sigmas(self):
        if self.sigmas is not None:
            return list(self.sigmas)
        return default_sigmas(self.dim_res, self.min_radius_um, self.max_radius_um)


@dataclass
class ScaleTrace:
    sigma: float
    gamma: float
    max_abs: float
    frob_thr: Optional[float]
    mask_count: int
    skipped: bool
    one_pass: bool = False          # the scale needed a single walk over the Hessian (nl_vesselness_spec hit)


@dataclass
class FrameTrace:
    scales: list = field(default_factory=list)
    n_positive: int = 0
    percentile_thr: Optional[float] = None
    label_thr: Optional[float] = None
    n_labels: int = 0


# ----------------------------------------------------------------------------- the pipeline
class FramePipeline:
    """One (Z, Y, X) frame on one GPU, resident in HBM across Filter and Label."""

    def __init__(self, shape, device: int = 0, ctx=None):
        self.shape = tuple(int(s) for s in shape)      # the GLOBAL frame shape (thresholds sample its lattice)
        self.two_d = len(self.shape) == 2              # (Y, X) image, im_info.no_z: held as one plane
        if self.two_d:
            self.shape = (1,) + self.shape
        if len(self.shape) != 3:
            raise ValueError("FramePipeline takes a (Z, Y, X) or a (Y, X) shape")
        self.ctx = ctx if ctx is not None else hipnative.Context(self.shape, device=device)
        if self.two_d:
            self.ctx.set_ndim(2)
        self.trace = FrameTrace()
        # One walk over the Hessian per scale instead of two (statistics, then masks): the mask threshold is
        # predicted from the sample lattice and bracketed by a relative margin; a scale whose exact threshold
        # falls outside the bracket is redone the two-pass way, so results never depend on it.
        self.one_pass = bool(getattr(self.ctx, "one_pass_available", lambda: False)()) and not self.two_d
        self.one_pass_margin = 1e-3
        # range + histogram of a threshold in one device round trip (a Z-slab pipeline reduces the range across ranks
        # between the two passes and keeps them apart)
        self._chain_hist = hasattr(self.ctx, "sample_range_hist") and os.environ.get("NELLIE_CHAIN_HIST", "1") != "0"
        self._pair_hist = hasattr(self.ctx, "sample_range_hist2") and os.environ.get("NELLIE_PAIR_HIST", "1") != "0"
        self.check_device_edges = os.environ.get("NELLIE_CHECK_EDGES", "0") == "1"
        self._one_pass_test_scale = 1.0      # tests: shifts the prediction to force a miss
        try:
            self._yx_max_r = int(self.ctx.info("gauss_yx_max_r"))
        except (KeyError, AttributeError):
            self._yx_max_r = 0

    def _step_fits_ahead(self, ws) -> bool:
        """nl_gauss_step_ahead takes steps that write at most two of the three ping-pong volumes (a Z pass and a fused Y+X pass):
        the third pass of a one-kernel-per-axis step lands in the volume the current scale still reads."""
        wz, wy, wx = ws
        n = 0 if wz is None else 1
        if wy is not None and wx is not None and len(wy) == len(wx) and 1 <= (len(wy) - 1) // 2 <= min(self._yx_max_r, self.shape[1]):
            n += 1
        else:
            n += (wy is not None) + (wx is not None)
        return n <= 2

    def close(self):
        self.ctx.close()

    def _strides(self, max_samples):
        """filtering.py:328-340 on the frame's own dimensionality (a 2-D image samples a 2-D lattice)."""
        if self.two_d:
            return (1,) + tuple(sample_strides(self.shape[1:], max_samples))
        return sample_strides(self.shape, max_samples)

    def _positive_lattice_samples(self, fld, strides):
        """arr[::sz, ::sy, ::sx][... > 0] of a device field (filtering.py:348-363), compacted on the device."""
        f = getattr(self.ctx, "sample_gather_positive", None)
        if f is not None:
            return self._gather(f(fld, strides))
        sample = self._gather(self.ctx.sample_gather(fld, strides))
        return sample[sample > 0]

    def _positive_flat_samples(self, fld, offset, step):
        """flat[offset::step][... > 0] (labelling.py:418-433), compacted on the device."""
        f = getattr(self.ctx, "flat_sample_gather_positive", None)
        if f is not None:
            return self._gather(f(fld, offset, step))
        sample = self._gather(self.ctx.flat_sample_gather(fld, offset, step))
        return sample[sample > 0]

    def _as_frame(self, a):
        a = np.asarray(a)
        return a[None] if (self.two_d and a.ndim == 2) else a

    # ---- hooks a Z-slab pipeline overrides (nellie_amd/sharded.py); identity on a single GPU -------------
    def _after_load(self, p):
        pass

    def _gauss_range(self, rz):
        return 0, self.shape[0]

    def _after_cascade_step(self, k):
        pass

    def _vess_range(self):
        return -1, -1

    def _reduce_minmax(self, mn, mx, npos):
        return mn, mx, npos

    def _reduce_counts(self, counts):
        return counts

    def _reduce_stats(self, max_abs, max_fsq, any_inf, overflow=0, fused_call=False):
        return max_abs, max_fsq, any_inf, overflow

    def _reduce_sum(self, n):
        return n

    def _reduce_mask_count(self, n):
        """h_mask count of a scale the one-pass walk completed: reporting only (ScaleTrace), nothing downstream waits for it."""
        return self._reduce_sum(n)

    def _settle_mask_counts(self):
        """Called once the scale loop is over: Z slabs turn the per-rank counts kept so far into global ones here."""

    def _gather(self, samples):
        return samples

    def _reduce_fused_count(self, n):
        return self._reduce_sum(n)      # (a Z-slab pipeline with a fused communicator gets the global count from the call itself)

    # ------------------------------------------------------------------ Filter
    def load_input(self, frame):
        """Keep the raw frame resident in HBM; `filter(None, ...)` then starts from device memory."""
        self.ctx.input_load(self._as_frame(frame))

    def _normalised_range(self, spec, max_abs):
        """(min, max) of the positive NL_FIELD_FROB samples for the exact normalisation, derived from the range the
        bracket round measured with max_abs := 1: x -> x / max_abs is monotone in float32, so min and max commute with
        it.  None (measure it) unless the one-pass walk is still valid (no inf) and nothing underflows to 0."""
        rng = getattr(self, "_raw_frob_range", None)
        if not spec or rng is None:
            return None
        with np.errstate(all="ignore"):
            mn, mx = rng[0] / np.float32(max_abs), rng[1] / np.float32(max_abs)
        return (mn, mx) if (mn > 0 and np.isfinite(mx)) else None

    def _threshold_from_field(self, fld, strides, known_range=None, pre=None):
        """min(triangle, otsu) over the positive lattice samples of a device field, or None if none.
        pre: the field's sample_range_hist result when it was fetched together with another field's."""
        if known_range is not None:
            mn, mx, npos = known_range[0], known_range[1], 1
        elif self._chain_hist:
            mn, mx, counts, npos, edges = self._range_hist(fld, strides, pre)
            if npos == 0:
                return None
            return float(min_triangle_otsu(counts, edges))
        else:
            mn, mx, npos = self._reduce_minmax(*self.ctx.sample_minmax(fld, strides))
        if npos == 0:
            return None
        edges = histogram_edges(mn, mx, 256)
        counts = self._reduce_counts(self.ctx.sample_hist(fld, strides, edges))
        return float(min_triangle_otsu(counts, edges))

    def _range_hist(self, fld, strides, pre=None):
        """Range and 256-bin histogram of the positive lattice samples in one device round trip (single GPU: no
        cross-rank reduction sits between the two passes).  The device builds numpy's float32 edges itself and they serve
        the threshold arithmetic too (the GPU tests run with NELLIE_CHECK_EDGES=1: every such chain compares them with
        numpy's, bit for bit); a degenerate range goes through numpy on the host, which raises numpy's errors."""
        mn, mx, npos, counts, dev_edges, valid = pre if pre is not None else self.ctx.sample_range_hist(fld, strides, 256)
        if npos and valid == 1 and self.check_device_edges:
            assert np.array_equal(dev_edges, histogram_edges(mn, mx, 256)), "device-built histogram edges differ from numpy's"
        edges = dev_edges if (npos and valid == 1) else (histogram_edges(mn, mx, 256) if npos else None)
        return mn, mx, counts, npos, edges

    def _fsq_bracket(self, strides, division, pre=None):
        """Predicted [lo, hi] for the un-normalised frob_sq threshold of the current scale, or None.
        Normalising by 1 instead of the (unknown) global max |H| rescales samples and threshold alike
        (filtering.py:421-444), so the histogram threshold of sqrt(frob_sq) predicts sqrt(fsq_min) up to float32
        rounding -- unless the rounding moves the histogram argmax to another bin, which the bracket then misses."""
        if pre is None:
            self.ctx.set_frob_norm(1.0, 0.0)
        self._raw_frob_range = None
        if self._chain_hist:
            mn, mx, counts, npos, edges = self._range_hist(FIELD_FROB, strides, pre)
            if npos == 0 or not np.isfinite(mx):
                return None
            self._raw_frob_range = (np.float32(mn), np.float32(mx))
        else:
            mn, mx, npos = self._reduce_minmax(*self.ctx.sample_minmax(FIELD_FROB, strides))
            if npos == 0 or not np.isfinite(mx):
                return None
            self._raw_frob_range = (np.float32(mn), np.float32(mx))
            edges = histogram_edges(mn, mx, 256)
            counts = self._reduce_counts(self.ctx.sample_hist(FIELD_FROB, strides, edges))
        t = float(min_triangle_otsu(counts, edges)) * self._one_pass_test_scale / division
        lo, hi = np.float32(t * t * (1.0 - self.one_pass_margin)), np.float32(t * t * (1.0 + self.one_pass_margin))
        if not (np.isfinite(lo) and np.isfinite(hi) and hi > 0):
            return None
        return float(lo), float(hi)

    def compute_vesselness(self, frame, p: FilterParams, mask: bool = True, finish: bool = True):
        """filtering.py:806-853 + 926: leaves `vesselness * masks` on the device; returns #voxels > 0.
        finish=False stops before the product is materialised (see filter()); returns None then."""
        ctx = self.ctx
        max_samples = int(p.max_threshold_samples) if p.max_threshold_samples is not None else 0
        if max_samples <= 0:
            raise ValueError("max_threshold_samples must be a positive integer")
        self.trace = FrameTrace()
        if frame is None:
            ctx.filter_begin()          # restart from the frame kept resident by load_input()
        else:
            self._load(frame)
        self._after_load(p)
        if self._chain_usable(p, mask):
            if self._scales_on_the_device(p, max_samples):
                return self._finish_frame(finish, p, mask)
            # a condition the device-resident chain leaves to the host turned up: the frame is redone, synchronously
            self.chain_fallbacks += 1
            self.trace = FrameTrace()
            if frame is None:
                ctx.filter_begin()
            else:
                self._load(frame)
            self._after_load(p)
        zr = z_ratio_of(p.dim_res)
        spacing = spacing_of(p.dim_res)
        sigmas = p.resolved_sigmas()
        strides = self._strides(max_samples)
        alpha_sq = float(p.alpha_sq)
        beta_sq = float(p.beta_sq)
        pending = None       # trace entry whose h_mask count is still being produced on the side stream

        def settle():
            nonlocal pending
            if pending is not None:
                pending.mask_count = self._reduce_mask_count(ctx.vesselness_count())
                pending = None

        deltas = list(cascade_deltas(sigmas, zr))
        if self.two_d:
            deltas = [(0.0, d[1], d[2]) for d in deltas]          # sigma_vec = (s, s): no Z axis (filtering.py:281-282)
        ahead = False        # the cascade step of the current scale was enqueued during the previous one

        def cascade_step(k, run_ahead):
            delta = deltas[k]
            if not any(s > 0 for s in delta):
                return False
            ws = [gaussian_weights(d) for d in delta]
            if run_ahead and not self._step_fits_ahead(ws):
                return False
            z0, z1 = self._gauss_range(0 if ws[0] is None else (len(ws[0]) - 1) // 2)
            ctx.gauss_step(*ws, z0=z0, z1=z1, **({"ahead": True} if run_ahead else {}))
            return True

        for k, sigma in enumerate(sigmas):
            if ahead:
                ctx.gauss_commit()
            else:
                cascade_step(k, False)
            self._after_cascade_step(k)      # Z slabs: the ghost planes the NEXT step needs start travelling now
            # the next cascade step only reads the Gaussian of THIS scale: enqueue it now, on the side stream, so that
            # it runs beside this scale's Hessian walk (filtering.py:814-835 has no such dependency either)
            ahead = self._gauss_ahead and k + 1 < len(sigmas) and cascade_step(k + 1, True)
            # gamma (filtering.py:365-380, 839-840)
            # the gamma samples and the raw Frobenius samples of the bracket need nothing from each other: one round trip
            want_bracket = bool(self.one_pass and mask and p.frob_thresh_division and p.frob_thresh is None)
            pair = None
            if want_bracket and self._chain_hist and self._pair_hist:
                ctx.set_spacing(spacing)
                ctx.set_frob_norm(1.0, 0.0)
                pair = ctx.sample_range_hist2(FIELD_GAUSS, FIELD_FROB, strides, 256)
            gamma = self._threshold_from_field(FIELD_GAUSS, strides, pre=None if pair is None else pair[0])
            if gamma is None or gamma <= 0:
                gamma = _EPS32
            gamma_sq = 2.0 * (float(gamma) ** 2)
            # Hessian statistics (filtering.py:555-562) -- taken together with the vesselness candidates when the
            # mask threshold can be bracketed beforehand (nl_vesselness_spec), by a pass of their own otherwise
            spec = False
            stats = None
            if want_bracket:
                if pair is None:
                    ctx.set_spacing(spacing)
                bracket = self._fsq_bracket(strides, float(p.frob_thresh_division), None if pair is None else pair[1])
                if bracket is not None:
                    settle()
                    vz0, vz1 = self._vess_range()
                    ma, mf, inf_, ovf = ctx.vesselness_spec(spacing, bracket[0], bracket[1], z0=vz0, z1=vz1)
                    stats = self._reduce_stats(ma, mf, inf_, ovf, fused_call=True)    # nl_vesselness_spec reduces them itself on a fused communicator
                    spec = not stats[2] and not stats[3]
                    if stats[2]:
                        stats = None     # a +inf frob_sq turned up: the one-pass walk only flags it (the largest finite
                                         # value comes from the statistics pass below), and the scale goes the two-pass way
            if stats is None:
                stats = self._reduce_stats(*ctx.hessian_stats(spacing))
            max_abs32, max_fsq32, any_inf = stats[:3]
            max_abs = float(max_abs32)
            if max_abs <= 0:
                max_abs = 1.0
            with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
                max_frob = np.sqrt(np.float32(max_fsq32)) / np.float32(max_abs)   # float32, as the volume op
            ctx.set_frob_norm(max_abs, float(max_frob) if any_inf else 0.0)
            thr = None
            if not mask:
                # h_mask = ones_like(image) (filtering.py:566-567)
                thr_cmp, nonempty = np.float32(-np.inf), True
            elif not p.frob_thresh_division:
                thr_cmp = None                                   # mask = frob > 0 (filtering.py:428-430)
                nonempty = bool(max_frob > 0)
            else:
                if p.frob_thresh is None:
                    t = self._threshold_from_field(FIELD_FROB, strides, self._normalised_range(spec, max_abs))
                    thr = 0.0 if t is None else t                # filtering.py:433-439
                else:
                    thr = float(p.frob_thresh)
                thr_cmp = np.float32(thr / p.frob_thresh_division)   # weak python scalar vs float32 array
                nonempty = bool(max_frob > thr_cmp)
            count = 0
            hit = False
            if nonempty:                                         # filtering.py:843-844
                if spec:
                    hit = ctx.vesselness_resolve(gamma_sq, alpha_sq, beta_sq, thr_cmp)
                if not hit:
                    settle()
                    vz0, vz1 = self._vess_range()
                    count = self._reduce_sum(ctx.vesselness_step(gamma_sq, alpha_sq, beta_sq, thr_cmp, z0=vz0, z1=vz1))
            self.trace.scales.append(ScaleTrace(float(sigma), float(gamma), max_abs, thr, count, not nonempty, hit))
            if hit:
                settle()
                pending = self.trace.scales[-1]      # its kernel overlaps the next scale's Gaussian; count read later
        settle()
        if not mask and not self.two_d:
            # filtering.py:581: with h_mask all ones EVERY Hessian goes to numpy.linalg.eigvalsh, and LAPACK gives up on a matrix with a
            # NaN entry -- the reference's run ends in LinAlgError("Eigenvalues did not converge") (pinned: tests/golden nomask_nan_*).
            # The device solves in closed form and would return 0 there; it reports such a voxel instead and the same exception is raised.
            if self._reduce_sum(int(ctx.info("nan_hessian"))):
                raise np.linalg.LinAlgError("Eigenvalues did not converge")
        return self._finish_frame(finish, p, mask)

    # Device-resident threshold chain (csrc/chain.inc): the scale loop without a host round trip; NELLIE_DEVICE_CHAIN=0: off
    _device_chain = os.environ.get("NELLIE_DEVICE_CHAIN", "1") == "1"
    chain_fallbacks = 0

    def _chain_usable(self, p: FilterParams, mask: bool) -> bool:
        return bool(self._device_chain and (self.one_pass or self.two_d) and mask and p.frob_thresh is None and p.frob_thresh_division
                    and hasattr(self.ctx, "chain_begin") and getattr(self.ctx, "chain_available", lambda: True)()
                    and self._chain_reductions_on_device())

    def _chain_reductions_on_device(self) -> bool:
        return True            # one GPU: nothing to reduce (a Z-slab pipeline needs its fused communicator)

    def _scales_on_the_device(self, p: FilterParams, max_samples: int) -> bool:
        """The sigma loop of compute_vesselness with every threshold decided on the device (nl_chain_*): kernels only, ONE
        wait at the end.  True: the trace is filled and the frame stands; False: the caller redoes the frame synchronously."""
        ctx = self.ctx
        zr = z_ratio_of(p.dim_res)
        spacing = spacing_of(p.dim_res)
        sigmas = p.resolved_sigmas()
        if not 1 <= len(sigmas) <= 16:
            return False
        strides = self._strides(max_samples)
        deltas = list(cascade_deltas(sigmas, zr))
        if self.two_d:
            deltas = [(0.0, d[1], d[2]) for d in deltas]          # sigma_vec = (s, s): no Z axis (filtering.py:281-282)
        ctx.chain_begin(len(sigmas))
        # No cascade step runs ahead here: that device (sharded.py) fills the GPU while the host waits for a scale's collectives,
        # and the chain has no such waits -- beside the walk the step only competes with it (measured on a 128 x 2048 x 2048 slab:
        # 34.1 ms/step with a step running ahead, 29.9 without; the synchronous path: 31.6 / 30.4).
        # ... except on frames that stream from the caches (below 2^26 voxels: a config-5 frame).  There a scale's threshold kernels --
        # one wave or a 10^6-point lattice each, ten of them per scale -- leave the GPU idle for 0.12 ms of every 0.5 ms, and the next
        # cascade step, enqueued on the side stream, fills those holes (round 5: 2.70 -> see profiles/r05_c5_chain_ahead.txt).
        run_ahead = self._chain_ahead(int(np.prod(self.shape))) and not self.two_d
        ahead = False
        # images: two passes per scale, the raw round's "bracket" decides nothing (margin 1: nl_chain_scale)
        margin = 1.0 if self.two_d else self.one_pass_margin

        def cascade_step(k, on_side):
            delta = deltas[k]
            if not any(s > 0 for s in delta):
                return False
            ws = [gaussian_weights(d) for d in delta]
            if on_side and not self._step_fits_ahead(ws):
                return False                 # (it runs in order at the top of the next trip)
            z0, z1 = self._gauss_range(0 if ws[0] is None else (len(ws[0]) - 1) // 2)
            ctx.gauss_step(*ws, z0=z0, z1=z1, **({"ahead": True} if on_side else {}))
            return True

        for k in range(len(sigmas)):
            if ahead:
                ctx.gauss_commit()
            else:
                cascade_step(k, False)
            self._after_cascade_step(k)
            ahead = run_ahead and k + 1 < len(sigmas) and cascade_step(k + 1, True)
            vz0, vz1 = self._vess_range()
            ctx.chain_scale(spacing, strides, float(p.alpha_sq), float(p.beta_sq), float(p.frob_thresh_division), margin,
                            self._one_pass_test_scale, z0=vz0, z1=vz1)
        # filter()'s percentile threshold reads lattice samples of the result: their compaction is enqueued BEFORE the wait, so it
        # runs while the host waits for the chain's records and repeats its decisions (one round trip less per frame)
        # Round 4: with the percentile selected on the device (nl_tail_enqueue) the WHOLE epilogue is enqueued before the wait -- the
        # host repeats the chain's decisions while the GPU masks the frame -- and no sample ever travels.
        dev_tail = self._tail_field is not None and self._device_tail_usable()
        tail = not dev_tail and self._tail_ok and self._tail_field is not None and hasattr(ctx, "sample_gather_positive_begin")
        self._tail_samples = None
        self._tail_dev_pending = False
        if dev_tail:
            ctx.chain_flush()
            ctx.tail_enqueue(strides, 1.0)
        elif tail:
            ctx.chain_flush()
            ctx.sample_gather_positive_begin(self._tail_field, strides)
        flags, gamma, max_abs, thr, counts = ctx.chain_finish()
        samples = ctx.sample_gather_positive_end() if tail else None
        self.last_chain_flags = [int(f) for f in flags]
        if not self._all_ranks_agree(not flags.any()):
            if dev_tail:
                ctx.tail_finish(commit=False)              # the frame is redone: what the epilogue computed from it is dropped
            return False
        self._tail_samples = samples
        self._tail_dev_pending = dev_tail
        if self.check_device_edges:
            for k in range(len(sigmas)):
                for which in range(3):
                    _, edges, rng, _ = ctx.chain_log(k, which)
                    assert np.array_equal(edges, histogram_edges(rng[0], rng[1], 256)), "device-built histogram edges differ from numpy's"
        for k, sigma in enumerate(sigmas):
            self.trace.scales.append(ScaleTrace(float(sigma), float(gamma[k]), float(max_abs[k]), float(thr[k]),
                                                self._reduce_mask_count(int(counts[k])), False, not self.two_d))
        return True

    def _all_ranks_agree(self, ok: bool) -> bool:
        return ok

    # NELLIE_CHAIN_AHEAD: unset = frames below 2^26 voxels on a single context; 0 / 1: never / always.  Read per frame (a test or
    # a tool that sets it after the import is heard); an instance attribute `_chain_ahead_env` overrides the environment.
    _chain_ahead_env = None

    def _chain_ahead(self, n_voxels: int) -> bool:
        env = self._chain_ahead_env if self._chain_ahead_env is not None else os.environ.get("NELLIE_CHAIN_AHEAD")
        if env is not None:
            return env == "1"
        return n_voxels < (1 << 26)

    def _finish_frame(self, finish: bool, p: FilterParams, mask: bool):
        """The product `vesselness * masks` (filtering.py:926) once every scale is in."""
        ctx = self.ctx
        sigmas = p.resolved_sigmas()
        self._settle_mask_counts()
        if not finish:
            return None
        vz0, vz1 = self._vess_range()
        self.trace.n_positive = self._reduce_sum(ctx.filter_finish(vz0, vz1))
        if self.two_d:
            # filtering.py:927-930: blob response of the (by now fully blurred) frame, maximum with the vesselness
            for i, s in enumerate(sigmas):
                w2, w0 = gaussian_derivative_weights(s, 2), gaussian_derivative_weights(s, 0)
                ctx.log2d_step(w2, w0, w2, w0, np.float32(float(s) ** 2), first=(i == 0), use_mask=mask)
            self.trace.n_positive = ctx.log2d_finish()
        return self.trace.n_positive

    def _load(self, frame):
        self.ctx.filter_load(self._as_frame(frame))

    def mask_volume(self, p: FilterParams):
        """filtering.py:952-967 on the device-resident frame."""
        strides = self._strides(int(p.max_threshold_samples))
        if self._device_percentile_usable():
            rec = self.ctx.mask_volume_dev(strides, 1.0)
            if rec["n_samples"] == 0:
                return None
            self._check_device_percentile(rec)
            self.trace.percentile_thr = float(rec["thr"])
            return rec["thr"]
        positive = self._positive_lattice_samples(FIELD_FRANGI, strides)
        if positive.size == 0:
            return None
        thr = percentile_of_samples(positive, 1)
        self.ctx.mask_volume(thr)
        self.trace.percentile_thr = float(thr)
        return thr

    # the device chain may gather the samples of the percentile threshold under its own wait (a single context: a slab pipeline
    # gathers them across ranks)
    _tail_ok = True
    _tail_field = None
    _tail_samples = None
    _tail_dev_pending = False
    _device_tail = os.environ.get("NELLIE_DEVICE_TAIL", "1") == "1"      # 0: the percentile threshold on the host (round 3)

    def _device_percentile_usable(self) -> bool:
        """The percentile selected on the device: needs the entry points and the host's repetition of numpy's interpolation to be
        the installed numpy's (percentile_of_samples checks that once per process)."""
        global _FAST_PERCENTILE
        if _FAST_PERCENTILE is None:
            percentile_of_samples(np.array([1.0, 2.0], np.float32), 1)
        return bool(self._device_tail and _FAST_PERCENTILE and hasattr(self.ctx, "mask_volume_dev") and self._tail_reductions_on_device())

    def _device_tail_usable(self) -> bool:
        """... and the whole fused epilogue enqueued without a wait (nl_tail_enqueue): 3-D frames."""
        return bool(self._device_percentile_usable() and not self.two_d and hasattr(self.ctx, "tail_enqueue"))

    def _check_device_percentile(self, rec):
        thr, gamma = percentile_from_order_statistics(rec["n_samples"], rec["a"], rec["b"], 1)
        if not (np.float32(thr) == rec["thr"] and np.float32(gamma) == rec["gamma"] and rec["a"] <= rec["b"]):
            raise RuntimeError(f"device percentile {rec} differs from numpy's rule ({thr}, {gamma})")

    def _tail_reductions_on_device(self) -> bool:
        return True            # one GPU: nothing to reduce (a Z-slab pipeline needs its fused communicator)

    def _finish_device_tail(self):
        """Waits for the epilogue nl_tail_enqueue started, checks the device's interpolation against numpy's rule, fills the trace.
        -> positive voxels, or None when there was no positive sample (the caller takes the plain path)."""
        rec = self.ctx.tail_finish(commit=True)
        if rec["n_samples"] == 0:
            return None
        self._check_device_percentile(rec)
        self.trace.percentile_thr = float(rec["thr"])
        self.trace.n_positive = self._reduce_fused_count(rec["n_positive"])
        return self.trace.n_positive
    _fused_epilogue = True      # (a Z-slab pipeline on a context without nl_mask_volume_fused keeps the two-step epilogue)
    # Enqueue the cascade step of scale s+1 on the side stream beside the Hessian walk of scale s.  Exact either way.
    # Off by default: at 1024^3 it buys ~1 % (two full-GPU kernels mostly take turns) and it blurs per-kernel timings.
    _gauss_ahead = os.environ.get("NELLIE_GAUSS_AHEAD", "0") == "1"

    def filter(self, frame, p: FilterParams, mask: bool = True, remove_edges: bool = False):
        """filtering.py:1012-1018: _run_frame, then _mask_volume when the frame has signal.
        Common case in one go: the percentile threshold only needs lattice samples of `vesselness * masks`, which
        can be read through the mask bits, so the product is never written just to be thresholded and rewritten
        (nl_mask_volume_fused).  No positive sample, no evaluated scale, 2-D or a slab: the two plain steps."""
        if remove_edges:                     # filtering.py:931-932: between the product and _mask_volume
            self.compute_vesselness(frame, p, mask=mask)
            npos = self.trace.n_positive = self._reduce_sum(self.ctx.remove_edges(15))
            if npos > 0:
                self.mask_volume(p)
            return npos
        if self._fused_epilogue and not self.two_d:
            self._tail_field = FIELD_VESSELNESS if (self._tail_ok or self._device_tail_usable()) else None
            self._tail_dev_pending = False
            try:
                self.compute_vesselness(frame, p, mask=mask, finish=False)
            finally:
                self._tail_field = None
            positive, self._tail_samples = self._tail_samples, None
            pending, self._tail_dev_pending = self._tail_dev_pending, False
            if any(not sc.skipped for sc in self.trace.scales):
                strides = self._strides(int(p.max_threshold_samples))
                if pending or (positive is None and self._device_tail_usable()):
                    if not pending:
                        self.ctx.tail_enqueue(strides, 1.0)
                    npos = self._finish_device_tail()
                    if npos is not None:
                        return npos
                    positive = np.zeros(0, np.float32)          # no positive sample: the plain path below
                    pending = False
                if positive is None:
                    positive = self._positive_lattice_samples(FIELD_VESSELNESS, strides)
                if positive.size > 0:
                    thr = percentile_of_samples(positive, 1)
                    self.trace.percentile_thr = float(thr)
                    self.trace.n_positive = self._reduce_fused_count(self.ctx.mask_volume_fused(thr))
                    return self.trace.n_positive
            if pending:                                         # every scale was skipped: nothing to commit
                self.ctx.tail_finish(commit=False)
            vz0, vz1 = self._vess_range()
            npos = self.trace.n_positive = self._reduce_sum(self.ctx.filter_finish(vz0, vz1))
        else:
            npos = self.compute_vesselness(frame, p, mask=mask)
        if npos > 0:      # float(sum(frame)) > 0 for a non-negative frame
            self.mask_volume(p)
        return npos

    def download_frangi(self, out=None):
        return self.ctx.filter_store(out=out)

    # ------------------------------------------------------------------ Label
    def upload_frangi(self, frangi):
        self.ctx.label_load_frangi(self._as_frame(np.asarray(frangi, dtype=np.float32)))

    def frangi_threshold(self, max_samples=1_000_000, nbins=256):
        """labelling.py:385-455 on the device-resident Frangi frame (no mask arguments)."""
        n = int(np.prod(self.shape))
        if n == 0:
            return None
        max_samples = max(1, int(max_samples))
        step = max(n // max_samples, 1)
        offsets = (0, step // 2) if step > 1 and step // 2 > 0 else (0,)
        values = np.zeros(0, np.float32)
        found = False
        for offset in offsets:
            values = self._positive_flat_samples(FIELD_FRANGI, offset, step)
            if values.size > 0 or step == 1:
                found = True
                break
        if not found:
            full = self._gather(self.ctx.flat_sample_gather(FIELD_FRANGI, 0, 1))     # rare: every strided sample empty
            if full.size == 0 or float(full.max()) <= 0:
                values = values[:0]
            else:
                values = full[full > 0]
        if values.size == 0:
            return None
        return log10_min_triangle_otsu(values, nbins)

    def label(self, frangi_thresh, min_area, fill_holes=True):
        """labelling.py:467-509; returns the number of labels.  Labels stay on the device."""
        self.trace.label_thr = None if frangi_thresh is None else float(frangi_thresh)
        self.trace.n_labels = self.ctx.label_run(frangi_thresh, int(min_area), fill_holes)
        return self.trace.n_labels

    # ------------------------------------------------------------------ Markers (the stage after Label)
    def markers(self, dim_res, labels=None, intensity=None, min_radius_um=0.20, max_radius_um=1, num_sigma=5,
                peak_min_distance=2, use_image=None):
        """mocap_marking.py:648-703 on the device; returns the number of markers.  labels / intensity default to what
        Label and load_input left on the device.  use_image: the float32 image the LoG runs on (use_im='frangi',
        :675-679); None = the distance image (use_im='distance').  2-D pipelines follow the reference's `no_z`
        branch (sigma_vec = (s, s), :323-324).  Products: download_markers()."""
        sigmas, max_r_px = marker_sigmas(dim_res, min_radius_um, max_radius_um, num_sigma)
        ctx = self.ctx
        ctx.markers_begin(None if labels is None else self._as_frame(labels), None if intensity is None else self._as_frame(intensity))
        n_mask = ctx.markers_distance(np.float32(max_r_px * 2.0))
        if n_mask > 0:                                  # empty mask: no markers, zero distance and border (:662-667)
            if use_image is not None:
                ctx.markers_use_image(self._as_frame(use_image))
            zr = None if self.two_d else z_ratio_of(dim_res)
            for s in sigmas:
                sv = float(s)
                w2, w0 = gaussian_derivative_weights(sv, 2), gaussian_derivative_weights(sv, 0)
                if self.two_d:
                    ctx.markers_log_step(None, None, w2, w0, w2, w0, np.float32(sv ** 2))
                else:
                    wz2, wz0 = gaussian_derivative_weights(sv / zr, 2), gaussian_derivative_weights(sv / zr, 0)
                    ctx.markers_log_step(wz2, wz0, w2, w0, w2, w0, np.float32(sv ** 2))
        return ctx.markers_finish(int(peak_min_distance))

    def download_markers(self):
        """(marker uint8, distance float32, border uint8)."""
        return self.ctx.markers_store()

    def download_labels(self, out=None):
        return self.ctx.label_store(out=out)


_FAST_PERCENTILE = None     # None: not checked yet; True / False: the shortcut below reproduces this numpy's np.percentile / does not


def _percentile_shortcut(values, q):
    """np.percentile(values, q) (method 'linear') of a 1-D float32 array without NaN, as numpy >= 2.0 evaluates it for float32
    input -- q / float32(100), virtual index (n - 1) * q in float32, _lerp in float32 -- with ONE selection instead of numpy's
    four-point partition plus its wrappers (0.4 ms on 3 * 10^4 samples, during which the GPU waits; filtering.py:957-962)."""
    n = values.size
    qf = np.true_divide(q, np.float32(100))
    vi = (n - 1) * qf
    if vi >= n - 1:
        lo = hi = n - 1
    else:
        lo = int(np.floor(vi))
        hi = lo + 1
    gamma = np.float32(np.float64(vi) - lo)
    part = np.partition(values, lo)
    a = part[lo]
    b = part[lo + 1:].min() if hi != lo else a          # the next order statistic: the smallest of what lies above the pivot
    diff = b - a
    if gamma >= 0.5:
        return b - diff * (1 - gamma)
    return a + diff * gamma


def percentile_from_order_statistics(n, a, b, q):
    """np.percentile of n float32 values whose order statistics at floor((n - 1) q / 100) and the next index are a and b: the
    interpolation of _percentile_shortcut (numpy's float32 'linear' rule).  -> (thr, gamma); the host's check of the device's."""
    qf = np.true_divide(q, np.float32(100))
    vi = (n - 1) * qf
    lo = n - 1 if vi >= n - 1 else int(np.floor(vi))
    gamma = np.float32(np.float64(vi) - lo)
    a, b = np.float32(a), np.float32(b)
    diff = b - a
    return (b - diff * (1 - gamma) if gamma >= 0.5 else a + diff * gamma), gamma


def _shortcut_matches_numpy():
    rng = np.random.default_rng(12345)
    for n in (1, 2, 3, 7, 100, 101, 1000, 4097):
        for scale in (1.0, 1e-6):
            v = (rng.random(n, dtype=np.float32) * np.float32(scale)).astype(np.float32)
            for q in (1, 50, 99):
                x, y = _percentile_shortcut(v, q), np.percentile(v, q)
                if type(x) is not type(y) or not (x == y):
                    return False
    return True


def percentile_of_samples(values, q):
    """np.percentile(values, q) for the positive lattice samples (1-D float32, no NaN): by the shortcut when it reproduces the
    installed numpy bit for bit and in type on a set of probes (checked once per process), by numpy otherwise."""
    global _FAST_PERCENTILE
    if _FAST_PERCENTILE is None:
        try:
            _FAST_PERCENTILE = bool(_shortcut_matches_numpy())
        except Exception:
            _FAST_PERCENTILE = False
    if _FAST_PERCENTILE and isinstance(values, np.ndarray) and values.dtype == np.float32 and values.ndim == 1 and values.size:
        return _percentile_shortcut(values, q)
    return np.percentile(values, q)


def log10_min_triangle_otsu(values, nbins=256):
    """labelling.py:448-455: thresholds in the log10 domain, mapped back, minimum of the two.  The reference histograms the same
    log values twice (once per threshold); one histogram serves both here -- same counts, same edges, same results."""
    log_values = np.log10(values)
    if log_values.dtype == np.float32 and log_values.size:
        triangle, otsu = hipnative.host_hist_thresholds(log_values, nbins)      # numpy's histogram arithmetic, one library call
    else:
        from nellie_amd.utils.gpu_functions import _host_histogram
        counts, edges = _host_histogram(log_values, nbins)
        triangle, otsu = hipnative.hist_thresholds(counts, edges)
    triangle = 10 ** triangle
    otsu = 10 ** otsu
    return min(triangle, otsu)
