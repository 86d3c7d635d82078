"""
CPU ORACLE for the Nellie `nellie/segmentation` hot path (Filter -> Label).

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it, and there only as the checker / timed CPU baseline -- never as the
thing shipped.  The product (`nellie_amd/`) never imports this module and fails
loudly when its HIP library is missing.

What it is: a step-by-step numpy restatement (plus a small plain-C helper,
`oracle/ccl_oracle.c`, for the integer connected-component / flood-fill work)
of the reference algorithm, each function citing the reference `file:line`
(paths relative to the reference checkout root) it follows.  Third-party
arithmetic the reference reaches through `scipy.ndimage` (scipy==1.15.3) and
`numpy` (numpy==2.2.6, both pinned in the reference's uv.lock) is restated from
its published algorithm:

  * scipy.ndimage.gaussian_filter / gaussian_filter1d / correlate1d
      separable Gaussian, radius int(truncate*sd+0.5), float64 accumulation in
      the symmetric-kernel order `c*w0 + sum_j (in[-j]+in[+j])*w[j]`, result
      stored to float32 after each axis, 'reflect' = (d c b a | a b c d | d c b a).
  * numpy.gradient            central differences, one-sided at the faces.
  * numpy.histogram           uniform-bin fast path incl. the +-1 edge fix-up.
  * numpy.linalg.eigvalsh     float32 in -> float64 LAPACK -> float32 out; restated as
                              the float64 trigonometric closed form rounded to float32.
  * numpy.percentile          method='linear'.
  * scipy.ndimage.binary_opening / binary_fill_holes / label / uniform_filter.
  * scipy.ndimage.gaussian_laplace (2-D images only: `_gaussian_kernel1d` of order 2, truncate 4.0,
                              one second-derivative pass per axis, summed in float32).

Also restated: the Markers stage (mocap_marking.py) and the two dense steps of the Network stage
(networking.py: pixel classes, branch labels).

Both dimensionalities of the stage are covered: (Z, Y, X) volumes and (Y, X) images (im_info.no_z: 2x2
closed-form eigenvalues, two-eigenvalue Frangi, the multi-scale LoG blob response, 4-connected opening).

Pinning (parity is PINNED): the reference's own tests hold no vector for Filter
and two toy cases for Label (tests/test_labelling.py:25-77).  The oracle is
therefore pinned against golden vectors produced by importing the reference
itself in the build container (tests/golden/make_golden.py, committed with the
.npz files) and, where scipy/numpy are importable, against the library calls
directly (tests/test_oracle_*.py).
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from types import SimpleNamespace

import numpy as np

F32 = np.float32
_HERE = os.path.dirname(os.path.abspath(__file__))


# =============================================================================
# C helper (integer work: CCL, flood fill)
# =============================================================================
_LIB = None


def build_c_helper(force: bool = False) -> str:
    """Compile oracle/ccl_oracle.c -> oracle/libccl_oracle.so with gcc."""
    src = os.path.join(_HERE, "ccl_oracle.c")
    out = os.path.join(_HERE, "libccl_oracle.so")
    if force or (not os.path.exists(out)) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", out, src])
    return out


def _lib():
    global _LIB
    if _LIB is None:
        path = build_c_helper()
        lib = ctypes.CDLL(path)
        i64 = ctypes.c_int64
        p = ctypes.c_void_p
        lib.orc_label.restype = i64
        lib.orc_label.argtypes = [p, p, i64, i64, i64, ctypes.c_int]
        lib.orc_fill_holes.restype = None
        lib.orc_fill_holes.argtypes = [p, p, i64, i64, i64]
        _LIB = lib
    return _LIB


# =============================================================================
# Filter: parameters
# =============================================================================
def z_ratio(dim_res) -> float:
    """filtering.py:75-78."""
    z_res = dim_res.get("Z") or dim_res.get("X") or 1.0
    x_res = dim_res.get("X") or 1.0
    return float(z_res) / float(x_res)


def spacing3(dim_res):
    """filtering.py:265-275 (3-D branch)."""
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
    step_calc = (sigma_max - sigma_min) / float(num_sigma)
    step = max(min_sigma_step_size, step_calc)
    sigmas = list(np.arange(sigma_min, sigma_max, step, dtype=float))
    sigmas.sort()
    return sigmas


def sigma_vec(sigma, zr):
    """filtering.py:277-286 (3-D branch)."""
    return (float(sigma) / zr, float(sigma), float(sigma))


def cascade_deltas(sigmas, zr):
    """filtering.py:814-825: per-scale incremental sigma per axis."""
    out = []
    prev = 0.0
    for s in sigmas:
        vp = sigma_vec(prev, zr)
        vc = sigma_vec(s, zr)
        d = []
        for sp, sc in zip(vp, vc):
            diff = max(0.0, float(sc) ** 2 - float(sp) ** 2)
            d.append(float(np.sqrt(diff)))
        out.append(tuple(d))
        prev = s
    return out


# =============================================================================
# scipy.ndimage.gaussian_filter restated
# =============================================================================
def gaussian_kernel1d(sigma: float, radius: int) -> np.ndarray:
    """scipy/ndimage/_filters.py `_gaussian_kernel1d` (order 0)."""
    sigma2 = sigma * sigma
    x = np.arange(-radius, radius + 1)
    phi_x = np.exp(-0.5 / sigma2 * x ** 2)
    phi_x = phi_x / phi_x.sum()
    return phi_x


def gaussian_radius(sigma: float, truncate: float = 3.0) -> int:
    """scipy `gaussian_filter1d`: lw = int(truncate * sd + 0.5)."""
    return int(truncate * float(sigma) + 0.5)


def _reflect_index(idx: np.ndarray, n: int) -> np.ndarray:
    """scipy NI_EXTEND_REFLECT: (d c b a | a b c d | d c b a)."""
    period = 2 * n
    m = np.mod(idx, period)
    return np.where(m >= n, period - 1 - m, m)


def correlate1d_reflect_f32(a: np.ndarray, w: np.ndarray, axis: int) -> np.ndarray:
    """
    scipy ni_filters.c NI_Correlate1D, symmetric branch, mode=reflect, origin 0:
        tmp = in[0]*w[0];  for j = -r..-1: tmp += (in[j] + in[-j]) * w[j]
    in float64, one rounding per operation, result stored as float32.
    """
    assert a.dtype == np.float32
    r = (len(w) - 1) // 2
    n = a.shape[axis]
    a64 = np.moveaxis(a, axis, 0).astype(np.float64)
    idx = np.arange(n)
    tmp = a64 * w[r]
    for j in range(-r, 0):
        lo = a64[_reflect_index(idx + j, n)]
        hi = a64[_reflect_index(idx - j, n)]
        tmp = tmp + (lo + hi) * w[r + j]
    return np.ascontiguousarray(np.moveaxis(tmp.astype(np.float32), 0, axis))


def gaussian_filter_f32(a: np.ndarray, sigmas, truncate: float = 3.0) -> np.ndarray:
    """scipy gaussian_filter(mode='reflect'): axes in order, skipping sigma <= 1e-15."""
    out = a
    for axis, sd in enumerate(sigmas):
        if sd > 1e-15:
            r = gaussian_radius(sd, truncate)
            w = gaussian_kernel1d(float(sd), r)
            out = correlate1d_reflect_f32(out, w, axis)
    return out


# =============================================================================
# threshold helpers
# =============================================================================
def sample_strides(shape, max_samples=int(1e6)):
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


def subsample_positive(arr, max_samples=int(1e6)):
    """filtering.py:342-363."""
    if arr.size == 0:
        return arr
    st = sample_strides(arr.shape, max_samples)
    if not all(s == 1 for s in st):
        arr = arr[tuple(slice(None, None, s) for s in st)]
    arr = arr[arr > 0]
    if arr.size == 0:
        return arr
    if arr.size > max_samples:
        stride = max(1, arr.size // max_samples)
        arr = arr[::stride]
    return arr


def linspace_f32(first: np.float32, last: np.float32, num: int) -> np.ndarray:
    """numpy/_core/function_base.py `linspace` for float32 endpoints, endpoint=True."""
    first = F32(first)
    last = F32(last)
    div = num - 1
    delta = F32(last - first)
    y = np.arange(0, num, dtype=np.float32)
    step = F32(delta / F32(div))
    if step == 0:
        y = y / F32(div)
        y = y * delta
    else:
        y = y * step
    y = y + first
    y[-1] = last
    return y.astype(np.float32)


def histogram_f32(flat: np.ndarray, nbins: int = 256):
    """
    numpy/lib/_histograms_impl.py `histogram(a, bins=n, range=(a.min(), a.max()))`
    for float32 data: float32 bin edges from linspace, index = trunc((a-first)/(last-first)*n)
    in float32, then the +-1 edge correction against the float32 edges.
    """
    flat = np.ascontiguousarray(flat, dtype=np.float32).reshape(-1)
    first = flat.min()
    last = flat.max()
    if first == last:
        first = F32(first - F32(0.5))
        last = F32(last + F32(0.5))
    edges = linspace_f32(first, last, nbins + 1)
    if np.any(edges[:-1] >= edges[1:]):
        raise ValueError(
            f"Too many bins for data range. Cannot create {nbins} finite-sized bins.")
    norm_denom = F32(last - first)
    f_idx = ((flat - first) / norm_denom) * F32(nbins)
    idx = f_idx.astype(np.intp)
    idx[idx == nbins] -= 1
    dec = flat < edges[idx]
    idx[dec] -= 1
    inc = (flat >= edges[idx + 1]) & (idx != nbins - 1)
    idx[inc] += 1
    counts = np.bincount(idx, minlength=nbins).astype(np.intp)
    return counts, edges


def otsu_from_hist(counts, edges):
    """gpu_functions.py:36-50 (everything after the histogram call)."""
    bin_centers = (edges[:-1] + edges[1:]) / 2.0
    c = counts / np.sum(counts)
    weight1 = np.cumsum(c)
    mean1 = np.cumsum(c * bin_centers) / weight1
    weight2 = np.cumsum(c[::-1])[::-1]
    mean2 = (np.cumsum((c * bin_centers)[::-1]) / weight2[::-1])[::-1]
    variance12 = weight1[:-1] * weight2[1:] * (mean1[:-1] - mean2[1:]) ** 2
    idx = np.argmax(variance12)
    return bin_centers[idx]


def triangle_from_hist(counts, edges):
    """gpu_functions.py:64-94 (everything after the histogram call)."""
    nbins = len(counts)
    bin_centers = (edges[:-1] + edges[1:]) / 2.0
    hist = counts / np.sum(counts)
    arg_peak_height = np.argmax(hist)
    peak_height = hist[arg_peak_height]
    arg_low_level, arg_high_level = np.flatnonzero(hist)[[0, -1]]
    flip = arg_peak_height - arg_low_level < arg_high_level - arg_peak_height
    if flip:
        hist = np.flip(hist, axis=0)
        arg_low_level = nbins - arg_high_level - 1
        arg_peak_height = nbins - arg_peak_height - 1
    width = arg_peak_height - arg_low_level
    x1 = np.arange(width)
    y1 = hist[x1 + arg_low_level]
    norm = np.sqrt(peak_height ** 2 + width ** 2)
    peak_height = peak_height / norm
    width = width / norm
    length = peak_height * x1 - width * y1
    arg_level = np.argmax(length) + arg_low_level
    if flip:
        arg_level = nbins - arg_level - 1
    return bin_centers[arg_level]


def otsu_threshold(values, nbins=256):
    """gpu_functions.py:23-50."""
    counts, edges = histogram_f32(values, nbins)
    return otsu_from_hist(counts, edges)


def triangle_threshold(values, nbins=256):
    """gpu_functions.py:53-94."""
    counts, edges = histogram_f32(values, nbins)
    return triangle_from_hist(counts, edges)


def min_tri_otsu(values, nbins=256):
    """The recurring `min(triangle, otsu)` on one sample set (one histogram serves both)."""
    counts, edges = histogram_f32(values, nbins)
    return min(triangle_from_hist(counts, edges), otsu_from_hist(counts, edges))


def calculate_gamma(gauss, max_samples=int(1e6)) -> float:
    """filtering.py:365-380."""
    positive = subsample_positive(gauss, max_samples)
    if positive.size == 0:
        return float(np.finfo(np.float32).eps)
    gamma = float(min_tri_otsu(positive))
    if gamma <= 0:
        gamma = float(np.finfo(np.float32).eps)
    return gamma


# =============================================================================
# Hessian (numpy.gradient applied twice), Frobenius mask
# =============================================================================
def gradient_axis(f: np.ndarray, h: float, axis: int) -> np.ndarray:
    """
    numpy/lib/_function_base_impl.py `gradient`, uniform spacing, edge_order=1, float32:
      interior (f[i+1]-f[i-1]) / (2.*h); faces (f[1]-f[0])/h and (f[-1]-f[-2])/h.
    `h` is a python float => weak scalar => float32 arithmetic with divisor float32(2h)/float32(h).
    """
    assert f.dtype == np.float32
    n = f.shape[axis]
    if n < 2:
        raise ValueError(
            "Shape of array too small to calculate a numerical gradient, "
            "at least (edge_order + 1) elements are required.")
    fm = np.moveaxis(f, axis, 0)
    out = np.empty_like(fm)
    out[1:-1] = (fm[2:] - fm[:-2]) / F32(2.0 * h)
    out[0] = (fm[1] - fm[0]) / F32(h)
    out[-1] = (fm[-1] - fm[-2]) / F32(h)
    return np.ascontiguousarray(np.moveaxis(out, 0, axis))


def hessian_components(img: np.ndarray, spacing):
    """filtering.py:518-536: (hxx,hxy,hxz,hyy,hyz,hzz) with 'x' = axis 0 (naming only)."""
    g0 = gradient_axis(img, spacing[0], 0)
    g1 = gradient_axis(img, spacing[1], 1)
    g2 = gradient_axis(img, spacing[2], 2)
    hxx = gradient_axis(g0, spacing[0], 0)
    hxy = gradient_axis(g0, spacing[1], 1)
    hxz = gradient_axis(g0, spacing[2], 2)
    hyy = gradient_axis(g1, spacing[1], 1)
    hyz = gradient_axis(g1, spacing[2], 2)
    hzz = gradient_axis(g2, spacing[2], 2)
    return hxx, hxy, hxz, hyy, hyz, hzz


def frobenius(h6):
    """filtering.py:538-562: frob_sq, max_abs, frob = sqrt(frob_sq)/max_abs."""
    hxx, hxy, hxz, hyy, hyz, hzz = h6
    frob_sq = hxx ** 2 + hyy ** 2 + hzz ** 2 + F32(2.0) * (hxy ** 2 + hxz ** 2 + hyz ** 2)
    max_abs = 0.0
    for comp in h6:
        if comp.size > 0:
            max_abs = max(max_abs, float(np.max(np.abs(comp))))
    if max_abs <= 0:
        max_abs = 1.0
    with np.errstate(invalid="ignore"):
        frob = np.sqrt(frob_sq) / F32(max_abs)
    return frob_sq, max_abs, frob


def frob_mask(frob, frob_thresh=None, frob_thresh_division=2, max_samples=int(1e6)):
    """filtering.py:407-444.  Returns (mask, threshold_used_or_None)."""
    inf_mask = np.isinf(frob)
    if np.any(inf_mask):
        finite_vals = frob[~inf_mask]
        max_finite = float(np.max(finite_vals)) if finite_vals.size > 0 else 0.0
        frob = frob.copy()
        frob[inf_mask] = max_finite
    if not frob_thresh_division:
        return frob > 0, None
    if frob_thresh is None:
        positive = subsample_positive(frob, max_samples)
        if positive.size == 0:
            thr = 0.0
        else:
            thr = float(min_tri_otsu(positive))
    else:
        thr = float(frob_thresh)
    with np.errstate(invalid="ignore"):
        mask = frob > (thr / frob_thresh_division)
    return mask, thr


# =============================================================================
# numpy.linalg.eigvalsh restated (3x3 symmetric, float32 in/out, float64 inside)
# =============================================================================
def eigvalsh3_f32(hxx, hxy, hxz, hyy, hyz, hzz):
    """
    numpy.linalg.eigvalsh on float32 (M,3,3): numpy/linalg/_linalg.py computes in
    float64 (LAPACK dsyevd) and casts the ascending eigenvalues back to float32.
    Restated as the float64 trigonometric closed form (Smith 1961):
        q = tr/3, B = A - qI, p = sqrt(||B||_F^2 / 6), r = det(B)/(2 p^3) clipped to [-1,1],
        phi = acos(r)/3, lambda_k = q + 2 p cos(phi + 2 pi k / 3)
    rounded to float32.  Returns (M,3) ascending.
    """
    a00 = hxx.astype(np.float64); a01 = hxy.astype(np.float64); a02 = hxz.astype(np.float64)
    a11 = hyy.astype(np.float64); a12 = hyz.astype(np.float64); a22 = hzz.astype(np.float64)
    q = (a00 + a11 + a22) / 3.0
    b00 = a00 - q; b11 = a11 - q; b22 = a22 - q
    p2 = (b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * (a01 * a01 + a02 * a02 + a12 * a12)) / 6.0
    p = np.sqrt(p2)
    det = (b00 * (b11 * b22 - a12 * a12)
           - a01 * (a01 * b22 - a12 * a02)
           + a02 * (a01 * a12 - b11 * a02))
    with np.errstate(divide="ignore", invalid="ignore"):
        r = det / (2.0 * p2 * p)
    r = np.where(p2 > 0, r, 0.0)
    r = np.clip(r, -1.0, 1.0)
    phi = np.arccos(r) / 3.0
    e_max = q + 2.0 * p * np.cos(phi)
    e_min = q + 2.0 * p * np.cos(phi + (2.0 * np.pi / 3.0))
    e_mid = 3.0 * q - e_max - e_min
    ev = np.stack([e_min, e_mid, e_max], axis=1)
    return ev.astype(np.float32)


def sort_by_abs(ev):
    """filtering.py:583-584: argsort(|ev|) (stable for the 3-element insertion sort) + take."""
    order = np.argsort(np.abs(ev), axis=1, kind="stable")
    return np.take_along_axis(ev, order, axis=1)


def frangi_response(ev, alpha_sq, beta_sq, gamma_sq):
    """filtering.py:744-766 (3-D branch).  float32 throughout; python floats are weak scalars."""
    l1 = ev[:, 0]
    l2 = ev[:, 1]
    l3 = ev[:, 2]
    with np.errstate(all="ignore"):
        ra_sq = (np.abs(l2) / (np.abs(l3) + 1e-12)) ** 2
        rb_sq = (np.abs(l2) / (np.sqrt(np.abs(l2 * l3)) + 1e-12)) ** 2
        s_sq = l1 ** 2 + l2 ** 2 + l3 ** 2
        v = ((1.0 - np.exp(-(ra_sq / alpha_sq)))
             * np.exp(-(rb_sq / beta_sq))
             * (1.0 - np.exp(-(s_sq / gamma_sq))))
    v[l3 > 0] = 0.0
    v[l2 > 0] = 0.0
    v = np.nan_to_num(v, nan=0.0, posinf=0.0, neginf=0.0)
    return v


# =============================================================================
# Filter frame
# =============================================================================
def compute_vesselness(frame, dim_res, sigmas=None, alpha_sq=0.5, beta_sq=0.5,
                       frob_thresh=None, frob_thresh_division=2,
                       max_samples=int(1e6), mask=True, trace=None, given_scales=None):
    """
    filtering.py:806-853 (+ 651-715 for the masked evaluation).
    `trace`, if a list, receives one dict per scale with the intermediates the
    golden vectors pin.

    `given_scales` (crop mode, see `filter_frame_crop`): one dict per scale with the volume-wide quantities of a
    run over the WHOLE volume -- `gamma`, `max_abs`, `frob_thr` (filtering.py:839, 555-562, 432-441) and `skipped`
    (the scale's mask was empty everywhere, :843-844) -- used instead of deriving them from `frame`, which is then
    a crop of that volume.  Everything else is the same per-voxel arithmetic.
    """
    frame = np.asarray(frame, dtype=np.float32)
    zr = z_ratio(dim_res)
    spacing = spacing3(dim_res)
    if sigmas is None:
        sigmas = default_sigmas(dim_res)
    vesselness = np.zeros_like(frame, dtype=np.float32)
    masks = np.ones_like(frame, dtype=bool)
    gauss = frame.copy()  # the reference blurs in place (hazard B.1); results are identical
    for k, (sigma, delta) in enumerate(zip(sigmas, cascade_deltas(sigmas, zr))):
        if any(s > 0 for s in delta):
            gauss = gaussian_filter_f32(gauss, delta, 3.0)
        giv = None if given_scales is None else given_scales[k]
        gamma = calculate_gamma(gauss, max_samples) if giv is None else float(giv["gamma"])
        gamma_sq = 2.0 * (float(gamma) ** 2)
        h6 = hessian_components(gauss, spacing)
        if mask and giv is not None:
            # the volume's max|H| and threshold: frob = sqrt(frob_sq) / float32(max_abs), mask = frob > thr / division
            frob_sq = frobenius(h6)[0]
            max_abs = float(giv["max_abs"])
            with np.errstate(invalid="ignore"):
                frob = np.sqrt(frob_sq) / F32(max_abs)
            h_mask, thr = frob_mask(frob, giv["frob_thr"] if frob_thresh_division else None, frob_thresh_division, max_samples)
            if giv.get("skipped"):
                h_mask = np.zeros_like(frame, dtype=bool)
        elif mask:
            _, max_abs, frob = frobenius(h6)
            h_mask, thr = frob_mask(frob, frob_thresh, frob_thresh_division, max_samples)
        else:
            # filtering.py:555-567: the normalisation is still computed, then h_mask = ones_like(image)
            max_abs = (frobenius(h6)[1] if giv is None else float(giv["max_abs"]))
            thr = None
            h_mask = np.ones_like(frame, dtype=bool)
        rec = dict(sigma=float(sigma), delta=delta, gamma=gamma, gamma_sq=gamma_sq,
                   max_abs=max_abs, frob_thr=thr, mask_count=int(h_mask.sum()))
        if trace is not None:
            rec["gauss"] = gauss.copy()
            trace.append(rec)
        if not np.any(h_mask):
            continue
        coords = np.where(h_mask)
        if not mask and any(np.isnan(c[coords]).any() for c in h6):
            # filtering.py:581 calls numpy.linalg.eigvalsh on every masked Hessian; with mask=False that includes the ones holding a NaN,
            # on which LAPACK does not converge (observed with the reference itself: tests/golden/nomask_nan_*.npz pins the exception).
            # With mask=True a NaN Hessian has a NaN Frobenius norm and never passes `frobenius_norm > threshold`.
            raise np.linalg.LinAlgError("Eigenvalues did not converge")
        ev = eigvalsh3_f32(*[c[coords] for c in h6])
        ev = sort_by_abs(ev)
        v = frangi_response(ev, alpha_sq, beta_sq, gamma_sq).astype(np.float32, copy=False)
        vessel_scale = np.zeros_like(frame, dtype=np.float32)
        vessel_scale[coords] = v
        if trace is not None:
            rec["vessel_scale"] = vessel_scale
        vesselness = np.maximum(vesselness, vessel_scale)
        masks &= h_mask
    return vesselness, masks


def remove_edges(frangi_frame):
    """filtering.py:969-1000 (3-D branch): per Z slice, zero 15 rows at both ends of the row bounding box."""
    frangi_frame = frangi_frame.copy()
    margin = 15
    for z_idx in range(frangi_frame.shape[0]):
        slice_im = frangi_frame[z_idx]
        rows = np.any(slice_im, axis=1)
        cols = np.any(slice_im, axis=0)
        if (not rows.any()) or (not cols.any()):
            continue
        rmin, rmax = np.where(rows)[0][[0, -1]]
        height = max(0, int(rmax) - int(rmin) + 1)
        if height <= 0:
            continue
        use = min(margin, height)
        frangi_frame[z_idx, rmin:rmin + use, :] = 0
        frangi_frame[z_idx, rmax - use + 1:rmax + 1, :] = 0
    return frangi_frame


def run_frame(frame, dim_res, remove_edges_flag=False, **kw):
    """filtering.py:910-933 (3-D): vesselness * masks, then the optional edge removal."""
    vesselness, masks = compute_vesselness(frame, dim_res, **kw)
    out = vesselness * masks
    if remove_edges_flag:
        out = remove_edges(out)
    return out


def percentile_linear_f32(values: np.ndarray, q: float):
    """
    numpy.percentile(values, q) (method='linear') for a float32 1-D array, numpy 2.2.6
    (numpy/lib/_function_base_impl.py:4257 `percentile`, `_quantile`, `_get_indexes`,
    `_get_gamma`, `_lerp`): for float input the quantile is q / float32(100), so the
    virtual index (n-1)*q, its fractional part and the interpolation are ALL float32.
    """
    a = np.sort(np.asarray(values, dtype=np.float32).reshape(-1))
    n = a.size
    q32 = np.true_divide(q, F32(100))                # float32
    virtual = F32((n - 1) * q32)                     # python int * float32 -> float32
    if virtual >= n - 1:
        lo = hi = n - 1
        prev_f = F32(-1)
    elif virtual < 0:
        lo = hi = 0
        prev_f = F32(0)
    else:
        prev_f = np.floor(virtual)
        lo = int(prev_f)
        hi = lo + 1
    # `_get_gamma` subtracts the intp index (after the -1 / 0 clamps) from the f32 virtual index
    prev_idx = np.intp(lo if lo != n - 1 or virtual < n - 1 else -1)
    gamma = F32(np.asanyarray(virtual - prev_idx, dtype=np.float32))
    av = a[lo]
    bv = a[hi]
    diff = np.subtract(bv, av)
    res = np.add(av, diff * gamma)
    if gamma >= 0.5:
        res = np.subtract(bv, diff * (1 - gamma))
    return F32(res)


def binary_erosion6(m: np.ndarray) -> np.ndarray:
    """scipy.ndimage.binary_erosion, default 6-connected cross, border_value=0."""
    p = np.pad(m, 1, mode="constant", constant_values=False)
    c = p[1:-1, 1:-1, 1:-1]
    return (c & p[:-2, 1:-1, 1:-1] & p[2:, 1:-1, 1:-1]
            & p[1:-1, :-2, 1:-1] & p[1:-1, 2:, 1:-1]
            & p[1:-1, 1:-1, :-2] & p[1:-1, 1:-1, 2:])


def binary_dilation6(m: np.ndarray) -> np.ndarray:
    """scipy.ndimage.binary_dilation, default 6-connected cross, border_value=0."""
    p = np.pad(m, 1, mode="constant", constant_values=False)
    c = p[1:-1, 1:-1, 1:-1]
    return (c | p[:-2, 1:-1, 1:-1] | p[2:, 1:-1, 1:-1]
            | p[1:-1, :-2, 1:-1] | p[1:-1, 2:, 1:-1]
            | p[1:-1, 1:-1, :-2] | p[1:-1, 1:-1, 2:])


def mask_volume(frangi_frame, max_samples=int(1e6), return_thr=False, given_thr=None):
    """filtering.py:952-967.  `given_thr`: the percentile threshold of the whole volume (crop mode)."""
    if given_thr is None:
        positive = subsample_positive(frangi_frame, max_samples)
        if positive.size == 0:
            return (frangi_frame, None) if return_thr else frangi_frame
        thr = percentile_linear_f32(positive, 1)
    else:
        thr = F32(given_thr)
    m = frangi_frame > thr
    m = binary_dilation6(binary_erosion6(m))      # binary_opening, 1 iteration
    out = frangi_frame * m
    return (out, thr) if return_thr else out


def filter_frame(frame, dim_res, **kw):
    """filtering.py:1012-1020: _run_frame then _mask_volume when the frame has signal."""
    fr = run_frame(frame, dim_res, **kw)
    if float(np.sum(fr)) > 0.0:
        fr = mask_volume(fr, kw.get("max_samples", int(1e6)))
    return fr


def filter_frame_crop(crop, dim_res, given_scales, percentile_thr, **kw):
    """
    Voxel-level check of a volume too large for this oracle (1024^3 needs ~45 min and ~96 GB): `crop` is a box cut
    out of the volume, `given_scales` / `percentile_thr` the volume-wide thresholds of the run under test
    (filtering.py:839, 555-562, 432-441, 843-844, 963 -- everything the algorithm derives from ALL voxels).
    The result equals the whole-volume result on every voxel farther than
        sum_s int(3 * delta_sigma_s + 0.5)   (the cascade, filtering.py:816-835)
        + 2                                   (np.gradient applied twice, :518-536)
        + 2                                   (binary_opening = erosion + dilation, :965)
    voxels (per axis, with that axis' sigma) from a face of the crop that is NOT a face of the volume; a face the crop
    shares with the volume needs no margin: reflect padding, one-sided differences and the opening's zero border are then
    the volume's own.  `percentile_thr=None`: `_mask_volume` is not applied (the `_run_frame` product).
    """
    fr = run_frame(np.asarray(crop, np.float32), dim_res, given_scales=given_scales, **kw)
    if percentile_thr is not None:
        fr = mask_volume(fr, given_thr=percentile_thr)
    return fr


def crop_margin(dim_res, sigmas=None, with_mask_volume=True):
    """Planes / rows / columns of a crop (per axis) that an artificial face invalidates: see `filter_frame_crop`."""
    zr = z_ratio(dim_res)
    if sigmas is None:
        sigmas = default_sigmas(dim_res)
    reach = [0, 0, 0]
    for delta in cascade_deltas(sigmas, zr):
        for a in range(3):
            if delta[a] > 1e-15:
                reach[a] += gaussian_radius(delta[a], 3.0)
    return tuple(r + 2 + (2 if with_mask_volume else 0) for r in reach)


# =============================================================================
# Filter, 2-D images (im_info.no_z): filtering.py:461-490, 675-690, 732-741, 772-796, 927-930
# =============================================================================
def default_sigmas_2d(dim_res, min_radius_um=0.25, max_radius_um=1.0):
    """filtering.py:288-311; same arithmetic as 3-D (only X enters the pixel radii)."""
    return default_sigmas(dim_res, min_radius_um, max_radius_um)


def cascade_deltas_2d(sigmas):
    """filtering.py:816-825 with sigma_vec = (s, s) (filtering.py:281-282)."""
    out, prev = [], 0.0
    for s in sigmas:
        d = float(np.sqrt(max(0.0, float(s) ** 2 - float(prev) ** 2)))
        out.append((d, d))
        prev = s
    return out


def gaussian_kernel1d_order(sigma: float, order: int, radius: int) -> np.ndarray:
    """scipy/ndimage/_filters.py `_gaussian_kernel1d` for any derivative order (float64)."""
    exponent_range = np.arange(order + 1)
    sigma2 = sigma * sigma
    x = np.arange(-radius, radius + 1)
    phi_x = np.exp(-0.5 / sigma2 * x ** 2)
    phi_x = phi_x / phi_x.sum()
    if order == 0:
        return phi_x
    q = np.zeros(order + 1)
    q[0] = 1
    D = np.diag(exponent_range[1:], 1)
    P = np.diag(np.ones(order) / -sigma2, -1)
    Q_deriv = D + P
    for _ in range(order):
        q = Q_deriv.dot(q)
    q = (x[:, None] ** exponent_range).dot(q)
    return q * phi_x


def gaussian_filter_orders_f32(a, sigmas, orders, truncate):
    """scipy gaussian_filter with a derivative order per axis: axes in order, float32 between the axes.
    `gaussian_filter1d` correlates with the REVERSED kernel; orders 0 and 2 give symmetric kernels."""
    out = a
    for axis, (sd, order) in enumerate(zip(sigmas, orders)):
        if sd > 1e-15:
            r = gaussian_radius(sd, truncate)
            w = gaussian_kernel1d_order(float(sd), order, r)[::-1]
            out = correlate1d_reflect_f32(out, np.ascontiguousarray(w), axis)
    return out


def gaussian_laplace_f32(a, sigmas, truncate=4.0):
    """scipy.ndimage.gaussian_laplace (generic_laplace): the second derivative along axis 0 goes to the output,
    the ones along the other axes are added to it in float32.  Default truncate = 4.0."""
    nd = a.ndim
    out = None
    for ax in range(nd):
        orders = [0] * nd
        orders[ax] = 2
        t = gaussian_filter_orders_f32(a, sigmas, orders, truncate)
        out = t if out is None else (out + t)
    return out


def hessian_components_2d(img, spacing):
    """filtering.py:461-490: (hxx, hxy, hyy), 'x' = axis 0 (naming only)."""
    g0 = gradient_axis(img, spacing[0], 0)
    g1 = gradient_axis(img, spacing[1], 1)
    return gradient_axis(g0, spacing[0], 0), gradient_axis(g0, spacing[1], 1), gradient_axis(g1, spacing[1], 1)


def frobenius_2d(h3):
    """filtering.py:488-489, 555-562."""
    hxx, hxy, hyy = h3
    frob_sq = hxx ** 2 + hyy ** 2 + F32(2.0) * (hxy ** 2)
    max_abs = 0.0
    for comp in h3:
        if comp.size > 0:
            max_abs = max(max_abs, float(np.max(np.abs(comp))))
    if max_abs <= 0:
        max_abs = 1.0
    with np.errstate(invalid="ignore"):
        frob = np.sqrt(frob_sq) / F32(max_abs)
    return frob_sq, max_abs, frob


def eig2_sorted_abs_f32(hxx, hxy, hyy):
    """filtering.py:675-690: closed-form 2x2 eigenvalues in float32, smaller |.| first."""
    with np.errstate(all="ignore"):
        trace = hxx + hyy
        diff = hxx - hyy
        delta = np.sqrt(diff * diff + F32(4.0) * (hxy * hxy))
        l1 = F32(0.5) * (trace - delta)
        l2 = F32(0.5) * (trace + delta)
    swap = np.abs(l1) > np.abs(l2)
    return np.where(swap, l2, l1), np.where(swap, l1, l2)


def frangi_response_2d(e1, e2, beta_sq, gamma_sq):
    """filtering.py:732-741, 759-766 (2-D branch)."""
    with np.errstate(all="ignore"):
        rb_sq = (np.abs(e1) / (np.abs(e2) + 1e-12)) ** 2
        s_sq = e1 ** 2 + e2 ** 2
        v = np.exp(-(rb_sq / beta_sq)) * (1.0 - np.exp(-(s_sq / gamma_sq)))
    v[e2 > 0] = 0.0
    return np.nan_to_num(v, nan=0.0, posinf=0.0, neginf=0.0)


def compute_vesselness_2d(frame, dim_res, sigmas=None, beta_sq=0.5, frob_thresh=None, frob_thresh_division=2,
                          max_samples=int(1e6), mask=True, trace=None):
    """filtering.py:806-853 on a (Y, X) image.  Returns (vesselness, masks, gauss): `gauss` is the image after
    the last cascade step -- the reference blurs `frame` itself in place, and `_filter_log` then runs on it."""
    frame = np.asarray(frame, dtype=np.float32)
    spacing = (float(dim_res.get("Y") or 1.0), float(dim_res.get("X") or 1.0))
    if sigmas is None:
        sigmas = default_sigmas_2d(dim_res)
    vesselness = np.zeros_like(frame, dtype=np.float32)
    masks = np.ones_like(frame, dtype=bool)
    gauss = frame.copy()
    for sigma, delta in zip(sigmas, cascade_deltas_2d(sigmas)):
        if any(s > 0 for s in delta):
            gauss = gaussian_filter_f32(gauss, delta, 3.0)
        gamma = calculate_gamma(gauss, max_samples)
        gamma_sq = 2.0 * (float(gamma) ** 2)
        h3 = hessian_components_2d(gauss, spacing)
        if mask:
            _, max_abs, frob = frobenius_2d(h3)
            h_mask, thr = frob_mask(frob, frob_thresh, frob_thresh_division, max_samples)
        else:
            # filtering.py:555-567: the normalisation is still computed, then h_mask = ones_like(image)
            max_abs = frobenius_2d(h3)[1]
            thr = None
            h_mask = np.ones_like(frame, dtype=bool)
        rec = dict(sigma=float(sigma), delta=delta, gamma=gamma, gamma_sq=gamma_sq, max_abs=max_abs, frob_thr=thr,
                   mask_count=int(h_mask.sum()))
        if trace is not None:
            rec["gauss"] = gauss.copy()
            trace.append(rec)
        if not np.any(h_mask):
            continue
        coords = np.where(h_mask)
        e1, e2 = eig2_sorted_abs_f32(*[c[coords] for c in h3])
        v = frangi_response_2d(e1, e2, beta_sq, gamma_sq).astype(np.float32, copy=False)
        vessel_scale = np.zeros_like(frame, dtype=np.float32)
        vessel_scale[coords] = v
        vesselness = np.maximum(vesselness, vessel_scale)
        masks &= h_mask
    return vesselness, masks, gauss


def filter_log_2d(frame, mask, sigmas):
    """filtering.py:772-796: multi-scale -LoG * sigma^2, masked, maximum over scales, clipped at 0, scaled to
    [0, 0.1]."""
    frame = np.asarray(frame, dtype=np.float32)
    lapofg = None
    for i, s in enumerate(sigmas):
        cur = -gaussian_laplace_f32(frame, (float(s), float(s))) * (float(s) ** 2)
        cur = cur * mask
        if i == 0:
            lapofg = cur
        else:
            sel = cur > lapofg
            lapofg[sel] = cur[sel]
    lapofg[lapofg < 0] = 0.0
    lapofg_max = np.max(lapofg)
    lapofg = lapofg / (lapofg_max + 1e-12)
    return lapofg / 10.0


def remove_edges_2d(frangi_frame):
    """filtering.py:974-985."""
    frangi_frame = frangi_frame.copy()
    rows = np.any(frangi_frame, axis=1)
    cols = np.any(frangi_frame, axis=0)
    if (not rows.any()) or (not cols.any()):
        return frangi_frame
    rmin, rmax = np.where(rows)[0][[0, -1]]
    height = max(0, int(rmax) - int(rmin) + 1)
    if height <= 0:
        return frangi_frame
    margin = min(15, height)
    frangi_frame[rmin:rmin + margin, :] = 0
    frangi_frame[rmax - margin + 1:rmax + 1, :] = 0
    return frangi_frame


def run_frame_2d(frame, dim_res, remove_edges_flag=False, mask=True, **kw):
    """filtering.py:910-933 for a (Y, X) image: vesselness * masks, then the maximum with the blob response."""
    sigmas = kw.get("sigmas")
    if sigmas is None:
        sigmas = default_sigmas_2d(dim_res)
        kw = dict(kw, sigmas=sigmas)
    vesselness, masks, gauss = compute_vesselness_2d(frame, dim_res, mask=mask, **kw)
    out = vesselness * masks
    blob = filter_log_2d(gauss, masks if mask else np.ones_like(gauss, bool), sigmas)
    blob = np.maximum(blob, 0)
    out = np.maximum(out, blob)
    if remove_edges_flag:
        out = remove_edges_2d(out)
    return out


def binary_opening4(m):
    """scipy.ndimage.binary_opening of a 2-D mask: default 4-connected cross, one iteration, border_value 0."""
    p = np.pad(m, 1, mode="constant", constant_values=False)
    c = p[1:-1, 1:-1]
    er = c & p[:-2, 1:-1] & p[2:, 1:-1] & p[1:-1, :-2] & p[1:-1, 2:]
    p = np.pad(er, 1, mode="constant", constant_values=False)
    c = p[1:-1, 1:-1]
    return c | p[:-2, 1:-1] | p[2:, 1:-1] | p[1:-1, :-2] | p[1:-1, 2:]


def mask_volume_2d(frangi_frame, max_samples=int(1e6), return_thr=False):
    """filtering.py:952-967 on a (Y, X) image."""
    positive = subsample_positive(frangi_frame, max_samples)
    if positive.size == 0:
        return (frangi_frame, None) if return_thr else frangi_frame
    thr = percentile_linear_f32(positive, 1)
    out = frangi_frame * binary_opening4(frangi_frame > thr)
    return (out, thr) if return_thr else out


def filter_frame_2d(frame, dim_res, **kw):
    """filtering.py:1012-1020 for a (Y, X) image."""
    fr = run_frame_2d(frame, dim_res, **kw)
    if float(np.sum(fr)) > 0.0:
        fr = mask_volume_2d(fr, kw.get("max_samples", int(1e6)))
    return fr


# =============================================================================
# Label
# =============================================================================
def min_area_pixels(dim_res, min_radius_um=0.25):
    """labelling.py:95-97, 209-219 (3-D branch)."""
    x_res = dim_res.get("X") or 1.0
    y_res = dim_res.get("Y") or x_res
    z_res = dim_res.get("Z") or x_res
    r = max(float(min_radius_um), float(x_res))
    volume_um3 = (4.0 / 3.0) * np.pi * (r ** 3)
    volume_px = volume_um3 / (float(x_res) * float(y_res) * float(z_res))
    return max(1, int(np.ceil(volume_px)))


def sample_nonzero(frame, max_samples=1_000_000, mask_frame=None, mask_thresh=None):
    """labelling.py:385-438: strided positives, optionally only where `mask_frame > mask_thresh`."""
    flat = frame.reshape(-1)
    if flat.size == 0:
        return flat
    mflat = None if mask_frame is None or mask_thresh is None else mask_frame.reshape(-1)
    max_samples = max(1, int(max_samples))
    step = max(int(flat.size) // max_samples, 1)
    offsets = (0, step // 2) if step > 1 and step // 2 > 0 else (0,)
    values = flat[:0]
    for offset in offsets:
        sample = flat[offset::step]
        if mflat is None:
            values = sample[sample > 0]
        else:
            values = sample[(sample > 0) & (mflat[offset::step] > mask_thresh)]
        if values.size > 0 or step == 1:
            return values
    if float(flat.max()) <= 0:
        return values
    if mflat is None:
        return flat[flat > 0]
    return flat[(flat > 0) & (mflat > mask_thresh)]


def intensity_otsu(original, max_samples=1_000_000, nbins=256):
    """labelling.py:457-465: Otsu threshold of the strided positive intensities (None if there are none)."""
    values = sample_nonzero(original, max_samples)
    if values.size == 0:
        return None
    flat = np.asarray(values).reshape(-1)
    counts, edges = np.histogram(flat, bins=nbins, range=(flat.min(), flat.max()))
    return otsu_from_hist(counts, edges)


def frangi_threshold(frangi, max_samples=1_000_000, nbins=256, mask_frame=None, mask_thresh=None):
    """labelling.py:440-455: log10-domain min(triangle, otsu); None when no positive sample."""
    values = sample_nonzero(frangi, max_samples, mask_frame, mask_thresh)
    if values.size == 0:
        return None
    log_values = np.log10(values)
    counts, edges = histogram_f32(log_values, nbins)
    triangle = 10 ** triangle_from_hist(counts, edges)
    otsu = 10 ** otsu_from_hist(counts, edges)
    return min(triangle, otsu)


def label26(mask: np.ndarray) -> np.ndarray:
    """scipy.ndimage.label(structure=ones(3,3,3)): int32 ids in raster order of first voxel."""
    return _label(mask, 26)


def _label(mask, conn):
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.zeros(m.shape, dtype=np.int32)
    nz, ny, nx = m.shape
    _lib().orc_label(m.ctypes.data, out.ctypes.data, nz, ny, nx, conn)
    return out


def fill_holes6(mask: np.ndarray) -> np.ndarray:
    """scipy.ndimage.binary_fill_holes (default 6-connected structure)."""
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.empty(m.shape, dtype=np.uint8)
    nz, ny, nx = m.shape
    _lib().orc_fill_holes(m.ctypes.data, out.ctypes.data, nz, ny, nx)
    return out.astype(bool)


def majority3(mask: np.ndarray) -> np.ndarray:
    """
    labelling.py:503-505: uniform_filter(float32 mask, size=3, mode='reflect') > 0.5
    == at least 14 of the 27 reflect-padded neighbours set (13/27 < 0.5 < 14/27).
    """
    p = np.pad(mask.astype(np.int32), 1, mode="symmetric")
    s = p[:-2] + p[1:-1] + p[2:]
    s = s[:, :-2] + s[:, 1:-1] + s[:, 2:]
    s = s[:, :, :-2] + s[:, :, 1:-1] + s[:, :, 2:]
    return s >= 14


def get_labels(frangi, frangi_thresh, min_area):
    """labelling.py:467-509 (3-D)."""
    if frangi_thresh is None:
        mask = np.zeros_like(frangi, dtype=bool)
    else:
        mask = frangi > frangi_thresh
    mask = fill_holes6(mask)
    labels = label26(mask)
    if labels.size == 0:
        return mask, labels
    areas = np.bincount(labels.ravel())
    if areas.size <= 1:
        return mask, labels
    areas[0] = 0
    keep = areas >= min_area
    mask = keep[labels]
    mask = majority3(mask)
    labels = label26(mask)
    return mask, labels


def label_frame(frangi, dim_res, min_radius_um=0.25, max_samples=1_000_000, nbins=256,
                return_thr=False, original=None, otsu_thresh_intensity=False, threshold=None):
    """labelling.py:511-532 + 538-556, including the optional intensity masking of the Frangi frame."""
    intensity_thresh = None
    if otsu_thresh_intensity:
        intensity_thresh = intensity_otsu(original, max_samples, nbins)
        if intensity_thresh is None:
            intensity_thresh = 0
    elif threshold is not None:
        intensity_thresh = threshold
    if intensity_thresh is not None:
        thr = frangi_threshold(frangi, max_samples, nbins, mask_frame=original, mask_thresh=intensity_thresh)
        frangi = frangi * (np.asarray(original) > intensity_thresh)
    else:
        thr = frangi_threshold(frangi, max_samples, nbins)
    _, labels = get_labels(frangi, thr, min_area_pixels(dim_res, min_radius_um))
    return (labels, thr) if return_thr else labels


def min_area_pixels_2d(dim_res, min_radius_um=0.25):
    """labelling.py:95-97, 209-216 (no_z branch)."""
    x_res = dim_res.get("X") or 1.0
    y_res = dim_res.get("Y") or x_res
    r = max(float(min_radius_um), float(x_res))
    return max(1, int(np.ceil(np.pi * (r ** 2) / (float(x_res) * float(y_res)))))


def get_labels_2d(frangi, frangi_thresh, min_area):
    """labelling.py:467-509 on a (Y, X) image: no hole filling, 8-connected labels, 3x3 majority."""
    if frangi_thresh is None:
        mask = np.zeros_like(frangi, dtype=bool)
    else:
        mask = frangi > frangi_thresh
    labels = _label(mask[None], 26)[0]                 # 26-connectivity inside one plane = 8-connectivity
    areas = np.bincount(labels.ravel())
    if labels.size == 0 or areas.size <= 1:
        return mask, labels
    areas[0] = 0
    mask = (areas >= min_area)[labels]
    p = np.pad(mask.astype(np.int32), 1, mode="symmetric")
    s9 = p[:-2] + p[1:-1] + p[2:]
    s9 = s9[:, :-2] + s9[:, 1:-1] + s9[:, 2:]
    mask = s9 >= 5                                      # uniform_filter(size=3) > 0.5: 4/9 < 0.5 < 5/9
    return mask, _label(mask[None], 26)[0]


def label_frame_2d(frangi, dim_res, min_radius_um=0.25, max_samples=1_000_000, nbins=256, return_thr=False):
    """labelling.py:538-556 on a (Y, X) image."""
    thr = frangi_threshold(frangi, max_samples, nbins)
    _, labels = get_labels_2d(frangi, thr, min_area_pixels_2d(dim_res, min_radius_um))
    return (labels, thr) if return_thr else labels


# =============================================================================
# Markers (nellie/segmentation/mocap_marking.py) -- the stage after Label
# =============================================================================
def marker_sigmas(dim_res, min_radius_um=0.20, max_radius_um=1, num_sigma=5):
    """mocap_marking.py:121-134, 329-362."""
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


def distance_transform_edt_exact(mask):
    """scipy.ndimage.distance_transform_edt(mask) (unit sampling): Euclidean distance of every True voxel to the
    nearest False voxel, sqrt of the exact integer squared distance in float64; 0 on False voxels.  Separable
    min-plus passes; a volume without any False voxel is outside what the stage feeds it."""
    m = np.asarray(mask, dtype=bool)
    big = np.int64(1) << 40
    d2 = np.where(m, big, np.int64(0))
    for axis in range(m.ndim):
        n = m.shape[axis]
        a = np.moveaxis(d2, axis, 0)
        out = np.full_like(a, big)
        k = np.arange(n, dtype=np.int64)
        for i in range(n):                                  # out[i] = min_j a[j] + (i-j)^2
            w = ((k - i) ** 2).reshape((n,) + (1,) * (a.ndim - 1))
            out[i] = np.min(a + w, axis=0)
        d2 = np.moveaxis(out, 0, axis)
    return np.sqrt(d2.astype(np.float64))


def marker_distance_and_border(mask, max_radius_px):
    """mocap_marking.py:419-450."""
    if mask.ndim == 3:
        border = binary_dilation6(mask) ^ mask
    else:                                           # 2-D: the default structuring element is the 4-connected cross
        p = np.pad(mask, 1, mode="constant", constant_values=False)
        border = (p[1:-1, 1:-1] | p[:-2, 1:-1] | p[2:, 1:-1] | p[1:-1, :-2] | p[1:-1, 2:]) ^ mask
    dist = distance_transform_edt_exact(mask).astype(np.float32)
    np.minimum(dist, max_radius_px * 2.0, out=dist)
    return dist, border


def maximum_filter_nearest(a, size):
    """scipy.ndimage.maximum_filter(a, size=size, mode='nearest'): separable running maximum, edge replicated."""
    r = size // 2
    out = a
    for axis in range(a.ndim):
        p = np.pad(out, [(r, r) if ax == axis else (0, 0) for ax in range(a.ndim)], mode="edge")
        n = out.shape[axis]
        acc = None
        for k in range(size):
            sl = [slice(None)] * a.ndim
            sl[axis] = slice(k, k + n)
            acc = p[tuple(sl)] if acc is None else np.maximum(acc, p[tuple(sl)])
        out = acc
    return out


def marker_local_max_peaks(use_im, mask, distance_im, sigmas, z_ratio_):
    """mocap_marking.py:452-511: multi-scale -LoG * sigma^2, local maxima, best response across scales."""
    valid = mask & (distance_im > 0)
    best = np.zeros_like(use_im, dtype=np.float32)
    peak = np.zeros(use_im.shape, dtype=bool)
    for s in sigmas:
        sv = float(s)
        vec = (sv / z_ratio_, sv, sv) if use_im.ndim == 3 else (sv, sv)
        log_resp = -gaussian_laplace_f32(use_im, vec)
        log_resp = (log_resp * (sv ** 2)).astype(np.float32, copy=False)
        log_resp[log_resp < 0] = 0
        local_max = log_resp == maximum_filter_nearest(log_resp, 3)
        local_max &= valid
        better = local_max & (log_resp > best)
        peak[better] = True
        best[better] = log_resp[better]
    return np.argwhere(peak)


def marker_remove_close_peaks(coords, intensity_im, peak_min_distance=2):
    """mocap_marking.py:569-606."""
    if coords.size == 0:
        return coords
    score = np.zeros_like(intensity_im, dtype=np.float32)
    score[tuple(coords.T)] = intensity_im[tuple(coords.T)]
    mx = maximum_filter_nearest(score, 2 * int(peak_min_distance) + 1)
    return np.argwhere((score == mx) & (score > 0))


def markers_frame(intensity, labels, dim_res, min_radius_um=0.20, max_radius_um=1, num_sigma=5, peak_min_distance=2,
                  frangi=None):
    """mocap_marking.py:648-703, 3-D volumes and 2-D images: (marker uint8, distance float32, border uint8).
    frangi=None: use_im='distance'; a float32 image: use_im='frangi' (:675-679, the LoG runs on it)."""
    mask = np.asarray(labels) > 0
    if not mask.any():
        return (np.zeros(mask.shape, np.uint8), np.zeros(mask.shape, np.float32), np.zeros(mask.shape, np.uint8))
    sigmas, max_r = marker_sigmas(dim_res, min_radius_um, max_radius_um, num_sigma)
    dist, border = marker_distance_and_border(mask, max_r)
    base = dist if frangi is None else np.asarray(frangi)
    coords = marker_local_max_peaks(base, mask, dist, sigmas, z_ratio(dim_res) if mask.ndim == 3 else None)
    coords = marker_remove_close_peaks(coords, np.asarray(intensity), peak_min_distance)
    marker = np.zeros(mask.shape, np.uint8)
    if coords.size:
        marker[tuple(coords.T)] = 1
    return marker, dist, border.astype(np.uint8)


# ---------------------------------------------------------------------------------------------------------------
# Network stage: the two dense per-voxel steps (nellie/segmentation/networking.py)
# ---------------------------------------------------------------------------------------------------------------
def network_pixel_class(skel):
    """networking.py:672-683 (`_get_pixel_class_impl`): occupancy of the 3x3x3 (2-D: 3x3) neighbourhood, centre included,
    zero outside the image (`mode="constant", cval=0`), kept on skeleton voxels, clipped at 4; uint8.
    Restated as a sum of shifted copies of the zero-padded mask (exact: integer counts <= 27)."""
    m = (np.asarray(skel) > 0).astype(np.uint8)
    pad = np.pad(m, 1)
    total = np.zeros(m.shape, dtype=np.uint8)
    for off in np.ndindex(*[3] * m.ndim):
        total += pad[tuple(slice(o, o + n) for o, n in zip(off, m.shape))]
    total *= m
    total[total > 4] = 4
    return total


def network_branch_skel_labels(pixel_class):
    """networking.py:758-800 (`_get_branch_skel_labels`): label((pc > 0) & (pc != 4), structure=ones(3,...)),
    int32 ids in raster order of each component's first voxel (26-connected; 8-connected for a 2-D image)."""
    pc = np.asarray(pixel_class)
    nj = (pc > 0) & (pc != 4)
    if nj.ndim == 2:
        return label26(nj[None])[0]
    return label26(nj)



def segment_frame(frame, dim_res):
    """Filter then Label on one 3-D frame: (im_preprocessed float32, im_instance_label int32)."""
    fr = filter_frame(frame, dim_res)
    return fr, label_frame(fr, dim_res)


def fake_im_info(shape_zyx, dim_res):
    """The duck-typed ImInfo the reference's own tests use (tests/test_labelling.py:7-22)."""
    z, y, x = shape_zyx
    return SimpleNamespace(no_t=True, no_z=False, shape=(1, z, y, x), axes="TZYX",
                           dim_res=dict(dim_res))
