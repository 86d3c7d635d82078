"""
Z-slab sharding of ONE frame across the GPUs of a node (the reference has no
distributed code, its closest precedent is the Z-chunked labelling of labelling.py:585-691).

Rank r owns a contiguous range of Z planes and holds H ghost planes on each interior side.  The raw
(float32-converted) ghost planes are exchanged ONCE per frame -- over RCCL/xGMI in production
(`RcclComm` -> nl_halo_exchange) -- and every cascade step then simply computes a Z range that shrinks by
its radius: H = 2 (Hessian) + 2 (opening of _mask_volume) + sum_s r_z(s), i.e. 24 planes at 0.1 um
isotropic.  All data-dependent scalars are made global exactly: min / max / positive-count of the lattice
samples, the 256 histogram counts, max|H|, the largest finite frob_sq and the inf flag are all-reduced
(sums of integers, mins and maxes: order independent), so every rank derives bit-identical thresholds and
the sharded Frangi frame equals the single-GPU frame bit for bit.  The <= 1e6 samples of the two
percentile / log-domain thresholds are gathered (order does not matter: histogram and order statistics).

Label across slabs: the thresholded mask is 1 bit/voxel, so instead of stitching per-slab labellings every
rank packs the mask bits of its own planes into a GLOBAL bit mask, the bit planes are all-gathered (RCCL
broadcasts, 1/32 of the float traffic), the run-level labelling -- whose cost scales with the number of runs,
not voxels -- runs redundantly on the global mask on every rank, and each rank paints only its own planes.
That IS the single-volume algorithm, so the labels equal the single-GPU labels bit for bit.
"""
from __future__ import annotations

import numpy as np

from nellie_amd import hipnative
from nellie_amd.hipnative import FIELD_FRANGI, FIELD_GAUSS
from nellie_amd.pipeline import (FilterParams, FramePipeline, cascade_deltas, gaussian_weights, z_ratio_of)


def slab_range(gnz: int, world: int, rank: int):
    """Even split of the global planes: [o0, o1) owned by `rank`."""
    base, rem = divmod(int(gnz), int(world))
    o0 = rank * base + min(rank, rem)
    return o0, o0 + base + (1 if rank < rem else 0)


def halo_depth(p: FilterParams) -> int:
    """Ghost planes a slab needs on an interior side for the whole Filter: see the module docstring."""
    sig = p.resolved_sigmas()
    rz = 0
    for delta in cascade_deltas(sig, z_ratio_of(p.dim_res)):
        w = gaussian_weights(delta[0])
        rz += 0 if w is None else (len(w) - 1) // 2
    return rz + 4


def slab_geometry(gshape, world, rank, halo):
    """(local_shape, gz0, own_lo, own_hi) of rank's slab including its ghost planes."""
    gnz, ny, nx = (int(s) for s in gshape)
    o0, o1 = slab_range(gnz, world, rank)
    if world > 1 and (o1 - o0) < halo:
        raise ValueError(f"slab of {o1 - o0} planes is thinner than the {halo}-plane halo: use fewer ranks")
    lo = min(halo, o0)
    hi = min(halo, gnz - o1)
    return (o1 - o0 + lo + hi, ny, nx), o0 - lo, lo, lo + (o1 - o0)


class RcclComm:
    """Production communicator: ghost planes and scalar all-reduces over RCCL on the context's stream.
    `host_gather(array) -> concatenated array` moves the <= 1e6 threshold samples through the control plane."""

    def __init__(self, ctx, world, rank, uid: bytes, host_gather):
        self.world, self.rank = world, rank
        self.ctx = ctx
        self.host_gather = host_gather
        ctx.comm_init(world, rank, uid)

    def exchange_halo(self, ctx, field, depth):
        ctx.halo_exchange(field, depth)

    def allreduce(self, arr, op):
        return self.ctx.allreduce(arr, op)

    def allgather(self, arr):
        return self.host_gather(arr)

    def allgather_mask_bits(self, ctx, slab_plane0):
        ctx.label_bits_allgather(slab_plane0)


class ShardedFramePipeline(FramePipeline):
    def __init__(self, gshape, rank, world, comm_factory, params: FilterParams, device: int = 0, ctx_factory=None):
        """
        comm_factory(ctx) -> communicator with exchange_halo / allreduce / allgather.
        ctx_factory(local_shape, device, gz0, gnz, own) -> context (default: the HIP context).
        """
        self.rank, self.world = int(rank), int(world)
        self.halo = halo_depth(params)
        lshape, gz0, own_lo, own_hi = slab_geometry(gshape, world, rank, self.halo)
        self.lshape, self.gz0, self.own = lshape, gz0, (own_lo, own_hi)
        make = ctx_factory or (lambda shp, dev, g0, gn, own: hipnative.Context(shp, device=dev, gz0=g0, gnz=gn, own=own))
        ctx = make(lshape, device, gz0, int(gshape[0]), (own_lo, own_hi))
        super().__init__(gshape, device=device, ctx=ctx)
        self._chain_hist = False           # the sample range is reduced across the ranks before the histogram pass
        self.comm = comm_factory(ctx)
        self.params = params
        self._valid = (0, lshape[0])

    # ---- loading: own planes in, ghost planes from the neighbours ---------------------------------------
    def _load(self, frame):
        """`frame` = this rank's OWN planes (own, Y, X)."""
        lo, hi = self.own
        self.ctx.filter_load(np.asarray(frame), z0=lo, z1=hi)

    def load_input(self, frame):
        lo, hi = self.own
        self.ctx.input_load(np.asarray(frame), z0=lo, z1=hi)

    def _after_load(self, p):
        lo, hi = self.own
        nzl = self.lshape[0]
        depth = max(lo, nzl - hi)
        if depth:
            self.comm.exchange_halo(self.ctx, FIELD_GAUSS, depth)
        self._valid = (0, nzl)

    # ---- shrinking Z ranges -------------------------------------------------------------------------------
    _fused_epilogue = False
    _gauss_ahead = False

    def _gauss_range(self, rz):
        v0, v1 = self._valid
        nzl = self.lshape[0]
        z0 = v0 if self.gz0 + v0 == 0 else v0 + rz              # a true face reflects, no shrink
        z1 = v1 if self.gz0 + v1 == self.shape[0] else v1 - rz
        z0 = max(z0, 0)
        z1 = min(z1, nzl)
        self._valid = (z0, z1)
        return z0, z1

    def _vess_range(self):
        lo, hi = self.own
        return max(lo - 2, 0), min(hi + 2, self.lshape[0])

    # ---- exact global scalars -----------------------------------------------------------------------------
    def _reduce_minmax(self, mn, mx, npos):
        n = int(self.comm.allreduce(np.array([npos], np.int64), "sum")[0])
        if n == 0:
            return mn, mx, 0
        big = np.float32(np.inf)
        gmn = self.comm.allreduce(np.array([mn if npos else big], np.float32), "min")[0]
        gmx = self.comm.allreduce(np.array([mx if npos else -big], np.float32), "max")[0]
        return np.float32(gmn), np.float32(gmx), n

    def _reduce_counts(self, counts):
        return self.comm.allreduce(np.ascontiguousarray(counts, dtype=np.int64), "sum")

    def _reduce_stats(self, max_abs, max_fsq, any_inf):
        r = self.comm.allreduce(np.array([max_abs, max_fsq, 1.0 if any_inf else 0.0], np.float32), "max")
        return np.float32(r[0]), np.float32(r[1]), bool(r[2] > 0)

    def _reduce_sum(self, n):
        return int(self.comm.allreduce(np.array([n], np.int64), "sum")[0])

    def _gather(self, samples):
        return self.comm.allgather(np.ascontiguousarray(samples, dtype=np.float32))

    # ---- outputs ------------------------------------------------------------------------------------------
    def download_frangi(self, out=None):
        """This rank's OWN planes of the Filter output."""
        lo, hi = self.own
        return self.ctx.filter_store(z0=lo, z1=hi, out=out)

    def upload_frangi(self, frangi):
        """`frangi` = this rank's OWN planes (Label run stand-alone on slabs)."""
        lo, hi = self.own
        self.ctx.label_load_frangi(np.asarray(frangi, dtype=np.float32), z0=lo, z1=hi)

    def label(self, frangi_thresh, min_area, fill_holes=True):
        """labelling.py:467-509 across slabs (see the module docstring); returns the GLOBAL label count."""
        self.trace.label_thr = None if frangi_thresh is None else float(frangi_thresh)
        self.ctx.label_pack(frangi_thresh)
        plane0 = [slab_range(self.shape[0], self.world, r)[0] for r in range(self.world)] + [self.shape[0]]
        self.comm.allgather_mask_bits(self.ctx, plane0)
        self.trace.n_labels = self.ctx.label_run_global(int(min_area), fill_holes)
        return self.trace.n_labels

    def download_labels(self, out=None):
        """This rank's OWN planes of the int32 label volume."""
        lo, hi = self.own
        return self.ctx.label_store(z0=lo, z1=hi, out=out)
