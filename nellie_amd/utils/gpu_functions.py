"""
Host side of the histogram thresholds (reference: nellie/utils/gpu_functions.py:23-94).

The reference calls `xp.histogram` and then does 256-element arithmetic.  Here the
histogram COUNTS come from the device (nl_sample_hist reproduces numpy's float32 binning),
the float32 bin edges come from numpy.linspace exactly as numpy.histogram builds them, and
the 256-element arithmetic below is the reference's, operation for operation.
"""
from __future__ import annotations

import numpy as np


def histogram_edges(first, last, nbins=256):
    """
    numpy/lib/_histograms_impl.py `_get_outer_edges` + `_get_bin_edges` for float32 data:
    expand an empty range by +-0.5, float32 linspace, strictly increasing or ValueError.
    """
    first = np.float32(first)
    last = np.float32(last)
    if first > last:
        raise ValueError("max must be larger than min in range parameter.")
    if not (np.isfinite(first) and np.isfinite(last)):
        raise ValueError(f"supplied range of [{first}, {last}] is not finite")
    if first == last:
        first = first - 0.5
        last = last + 0.5
    edges = np.linspace(first, last, nbins + 1, endpoint=True, dtype=np.float32)
    if np.any(edges[:-1] >= edges[1:]):
        raise ValueError(
            f"Too many bins for data range. Cannot create {nbins} finite-sized bins.")
    return edges


def otsu_from_hist(counts, bin_edges):
    """gpu_functions.py:36-50."""
    bin_centers = (bin_edges[:-1] + bin_edges[1:]) / 2.0
    counts = counts / np.sum(counts)
    weight1 = np.cumsum(counts)
    mean1 = np.cumsum(counts * bin_centers) / weight1
    weight2 = np.cumsum(counts[::-1])[::-1]
    mean2 = (np.cumsum((counts * bin_centers)[::-1]) / weight2[::-1])[::-1]
    variance12 = weight1[:-1] * weight2[1:] * (mean1[:-1] - mean2[1:]) ** 2
    idx = np.argmax(variance12)
    return bin_centers[idx], variance12[idx]


def triangle_from_hist(counts, bin_edges):
    """gpu_functions.py:64-94."""
    nbins = len(counts)
    bin_centers = (bin_edges[:-1] + bin_edges[1:]) / 2.0
    hist = counts / np.sum(counts)
    arg_peak_height = np.argmax(hist)
    peak_height = hist[arg_peak_height]
    arg_low_level, arg_high_level = np.flatnonzero(hist)[[0, -1]]
    flip = arg_peak_height - arg_low_level < arg_high_level - arg_peak_height
    if flip:
        hist = np.flip(hist, axis=0)
        arg_low_level = nbins - arg_high_level - 1
        arg_peak_height = nbins - arg_peak_height - 1
    del arg_high_level
    width = arg_peak_height - arg_low_level
    x1 = np.arange(width)
    y1 = hist[x1 + arg_low_level]
    norm = np.sqrt(peak_height ** 2 + width ** 2)
    peak_height = peak_height / norm
    width = width / norm
    length = peak_height * x1 - width * y1
    arg_level = np.argmax(length) + arg_low_level   # ValueError on an empty sequence, as in the reference
    if flip:
        arg_level = nbins - arg_level - 1
    return bin_centers[arg_level]


def min_triangle_otsu(counts, bin_edges):
    """The recurring `min(triangle_threshold(x), otsu_threshold(x)[0])` (filtering.py:374-376, 437-439)."""
    tri = triangle_from_hist(counts, bin_edges)
    otsu, _ = otsu_from_hist(counts, bin_edges)
    return min(tri, otsu)


def otsu_threshold(matrix, nbins=256, xp=None):
    """gpu_functions.py:23-50 for host arrays (Label's sparse log10 samples live on the host)."""
    flat = np.asarray(matrix).reshape(-1)
    counts, bin_edges = np.histogram(flat, bins=nbins, range=(flat.min(), flat.max()))
    return otsu_from_hist(counts, bin_edges)


def triangle_threshold(matrix, nbins=256, xp=None):
    """gpu_functions.py:53-94 for host arrays."""
    flat = np.asarray(matrix).reshape(-1)
    hist, bin_edges = np.histogram(flat, bins=nbins, range=(np.min(flat), np.max(flat)))
    return triangle_from_hist(hist, bin_edges)
