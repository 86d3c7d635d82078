"""
Histogram thresholds (reference: nellie/utils/gpu_functions.py:23-94), HIP backend.

The reference builds a 256-bin histogram with `xp.histogram` and reduces it to the Otsu and the triangle
threshold.  Here the histogram of a device field is built on the device (nl_sample_range_hist reproduces
numpy's float32 binning and bin edges), and the reduction of the finished histogram to the two bin centres is
one call into the library (`nl_hist_thresholds`, C++ on the host with numpy's float64 operation order).  The two
functions at the bottom serve host arrays (Label's log10 samples): numpy builds that histogram, the library
reduces it.
"""
from __future__ import annotations

import numpy as np

from nellie_amd import hipnative


def histogram_edges(first, last, nbins=256):
    """The float32 bin edges numpy.histogram uses for range=(first, last) (numpy/lib/_histograms_impl.py
    `_get_outer_edges` + `_get_bin_edges`): a degenerate range widens by +-0.5, edges are a float32 linspace and
    must increase strictly; numpy's ValueErrors are raised for the ranges numpy rejects."""
    first, last = np.float32(first), np.float32(last)
    if first > last:
        raise ValueError("max must be larger than min in range parameter.")
    if not (np.isfinite(first) and np.isfinite(last)):
        raise ValueError(f"supplied range of [{first}, {last}] is not finite")
    if first == last:
        first, last = first - 0.5, last + 0.5
    edges = np.linspace(first, last, nbins + 1, endpoint=True, dtype=np.float32)
    if np.any(edges[:-1] >= edges[1:]):
        raise ValueError(f"Too many bins for data range. Cannot create {nbins} finite-sized bins.")
    return edges


def min_triangle_otsu(counts, bin_edges):
    """`min(triangle_threshold(x), otsu_threshold(x)[0])` (filtering.py:374-376, 437-439) of a finished histogram."""
    tri, otsu = hipnative.hist_thresholds(counts, bin_edges)
    return min(tri, otsu)


def _host_histogram(matrix, nbins):
    flat = np.asarray(matrix).reshape(-1)
    return np.histogram(flat, bins=nbins, range=(flat.min(), flat.max()))


def otsu_threshold(matrix, nbins=256, xp=None):
    """gpu_functions.py:23-50 for a host array: (threshold, between-class variance at it), in the dtype numpy's histogram
    edges have for the data (float32 for float32, float64 for integer and float64 images)."""
    counts, edges = _host_histogram(matrix, nbins)
    _, otsu, var = hipnative.hist_thresholds(counts, edges, with_variance=True)
    return otsu, var


def triangle_threshold(matrix, nbins=256, xp=None):
    """gpu_functions.py:53-94 for a host array."""
    counts, edges = _host_histogram(matrix, nbins)
    return hipnative.hist_thresholds(counts, edges)[0]
