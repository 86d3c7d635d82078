"""
Z-slab sharding of ONE frame across the GPUs of a node (the reference has no
distributed code, its closest precedent is the Z-chunked labelling of labelling.py:585-691).

Rank r owns a contiguous range of Z planes and holds ghost planes on each interior side.  Two exchange schemes, both
exact (the sharded Frangi frame equals the single-GPU frame bit for bit):

  "steps" (default)  every rank evaluates each scale on its owned planes +- 4 (2 for the double difference of the Hessian,
      2 for the opening of _mask_volume) and fetches, per cascade step, only the r_z(s) planes beyond that from the
      neighbour that owns them (`exchange_halo(..., offset=4, depth=r_z)`): 4 + max r_z ghost planes per side (9 at 0.1 um
      isotropic), 6 % redundant Gaussian work on 128-plane slabs, and the exchange for step s+1 is issued right after step
      s -- on a stream and an RCCL communicator of its own -- so it travels while scale s's thresholds, Hessian walk and
      eigen-solves run; only the first exchange of a frame (4 + r_z(1) raw planes) is exposed.
  "fat"  the raw ghost planes are exchanged ONCE per frame, H = 4 + sum_s r_z(s) deep (24 planes at 0.1 um isotropic), and
      every cascade step computes a Z range that shrinks by its radius: one message, up to 37 % redundant Gaussian work at
      128-plane slabs, nothing to overlap it with.

All data-dependent scalars are made global exactly: min / max / any-positive of the lattice samples (one max-reduction),
the 256 histogram counts, max|H|, the largest finite frob_sq, the inf and queue-overflow flags (one max-reduction), the
mask counts (sums of integers): order independent, so every rank derives bit-identical thresholds.  The <= 1e6 samples of
the two percentile / log-domain thresholds are gathered (order does not matter: histogram and order statistics).

Label across slabs (no replication): every rank labels its own planes plus ONE ghost bit plane per interior side
(the bit planes travel like the float ghost planes: RCCL send/recv between neighbours).  A component that crosses an
interface appears on both ranks as a tree containing runs of the two planes both ranks see, and the k-th run of such
a plane is the same voxels on both sides.  Per phase (hole filling, area filter, numbering) the ranks all-gather the
(tree, quantity) tables of those planes -- a few thousand int32 pairs -- join the trees of neighbouring ranks (a
connected-components problem on that small graph, solved identically on every rank) and patch the global answer
back into the device arrays: "touches a face of the volume" (OR), object size (sum), owner and label of a component
(the lowest rank holding voxels of it numbers it; ids are offset by the exclusive sum of the lower ranks' counts).
The result is the single-volume labelling bit for bit, numbering included (labelling.py:467-509; the reference's
own chunked mode, labelling.py:585-691, stitches per-chunk labellings and is NOT equivalent to its full-volume mode).
"""
from __future__ import annotations

import os

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


def step_radii(p: FilterParams):
    """Z radius of every cascade step (0 where the step has no Z pass)."""
    out = []
    for delta in cascade_deltas(p.resolved_sigmas(), z_ratio_of(p.dim_res)):
        w = gaussian_weights(delta[0])
        out.append(0 if w is None else (len(w) - 1) // 2)
    return out


def halo_depth_steps(p: FilterParams) -> int:
    """Ghost planes per interior side of the per-step exchange scheme: 4 + the largest Z radius of a cascade step."""
    return 4 + max(step_radii(p) + [0])


def slab_geometry(gshape, world, rank, halo):
    """(local_shape, gz0, own_lo, own_hi) of rank's slab including its ghost planes."""
    gnz, ny, nx = (int(s) for s in gshape)
    o0, o1 = slab_range(gnz, world, rank)
    if world > 1 and (o1 - o0) < halo:
        raise ValueError(f"slab of {o1 - o0} planes is thinner than the {halo}-plane halo: use fewer ranks")
    lo = min(halo, o0)
    hi = min(halo, gnz - o1)
    return (o1 - o0 + lo + hi, ny, nx), o0 - lo, lo, lo + (o1 - o0)


SLAB_FILL, SLAB_AREA, SLAB_NUMBER = 0, 1, 2


SLAB_BLOB_HEADER = 8


def pack_slab_tables(roots, values, nruns=0):
    """The blob a rank publishes per phase (nl_slab_phase's layout): [n0 n1 n2 n3 | ints | overflow | runs | 0] then the roots of
    the four planes, then their values (int32).  Contexts that build their tables on the host (the CPU double of the tests) use it."""
    counts = [int(np.asarray(r).size) for r in roots]
    total = sum(counts)
    head = np.array(counts + [SLAB_BLOB_HEADER + 2 * total, 0, int(nruns), 0], np.int32)
    return np.concatenate([head] + [np.asarray(r, np.int32) for r in roots] + [np.asarray(v, np.int32) for v in values])


def unpack_slab_tables(blob):
    """A rank's blob -> (roots[4], values[4]), planes in the order ghost-low, own-first, own-last, ghost-high."""
    blob = np.asarray(blob, np.int32)
    counts = [int(c) for c in blob[:4]]
    total = sum(counts)
    cuts = np.cumsum([0] + counts)
    h = SLAB_BLOB_HEADER
    r, v = blob[h:h + total], blob[h + total:h + 2 * total]
    return [r[cuts[k]:cuts[k + 1]] for k in range(4)], [v[cuts[k]:cuts[k + 1]] for k in range(4)]


class _Joined:
    """One node per (rank, tree) that appears in a table: rank, root, val; comp = its component after joining the ranks."""
    __slots__ = ("rank", "root", "val", "comp", "ncomp")


def join_slab_blobs(blobs):
    """The joined view of every rank's tables.  With the library loaded: its host routine (nl_host_slab_join, C++: the numpy
    model below takes ~1 ms per rank on tables of a 2048 x 2048 plane, which an 8-rank step would pay three times); without it
    (CPU tests on a machine without the built library) the model itself.  Same nodes, same components, same numbering
    (tests/test_sharded_cpu.py compares them)."""
    from nellie_amd import hipnative
    try:
        hipnative.load()
    except (hipnative.LibraryUnavailable, OSError):      # only "there is no library": an error the library REPORTS (tables that
        return join_slab_tables([unpack_slab_tables(b) for b in blobs])      # disagree, a table beyond its block) propagates
    rank, root, val, comp, ncomp = hipnative.host_slab_join([np.ascontiguousarray(b, np.int32) for b in blobs])
    j = _Joined()
    j.rank, j.root, j.val, j.comp, j.ncomp = rank, root, val, comp, ncomp
    return j


def join_slab_tables(tables):
    """tables[r] = (roots[4], values[4]) of rank r, planes in the order ghost-low, own-first, own-last, ghost-high.
    Rank r's last owned plane is rank r+1's low ghost plane and rank r's high ghost plane is rank r+1's first owned
    plane: the k-th entry (a run, or a segment component: whatever unit both ranks cut the plane into alike) of such a plane
    names the same voxels on both sides, which joins the two trees."""
    ranks, roots, vals, ids, n = [], [], [], [], 0
    for r, (rt, vt) in enumerate(tables):
        allr = np.concatenate([np.asarray(x, np.int64) for x in rt]) if rt else np.zeros(0, np.int64)
        allv = np.concatenate([np.asarray(x, np.int64) for x in vt]) if vt else np.zeros(0, np.int64)
        u, first, inv = np.unique(allr, return_index=True, return_inverse=True)
        cuts = np.cumsum([0] + [len(x) for x in rt])
        ids.append([n + inv[cuts[k]:cuts[k + 1]] for k in range(4)])
        ranks.append(np.full(u.size, r, np.int64)); roots.append(u); vals.append(allv[first])
        n += u.size
    ea, eb = [], []
    for r in range(len(tables) - 1):
        for mine, theirs in ((2, 0), (3, 1)):
            a, b = ids[r][mine], ids[r + 1][theirs]
            if a.size != b.size:
                raise RuntimeError(f"slab tables of ranks {r} and {r + 1} disagree ({a.size} vs {b.size} runs): the ghost bit planes are stale")
            ea.append(a); eb.append(b)
    j = _Joined()
    j.rank = np.concatenate(ranks) if ranks else np.zeros(0, np.int64)
    j.root = np.concatenate(roots).astype(np.int32) if roots else np.zeros(0, np.int32)
    j.val = np.concatenate(vals) if vals else np.zeros(0, np.int64)
    if n == 0:
        j.comp, j.ncomp = np.zeros(0, np.int64), 0
        return j
    ea = np.concatenate(ea) if ea else np.zeros(0, np.int64)
    eb = np.concatenate(eb) if eb else np.zeros(0, np.int64)
    j.ncomp, j.comp = _components(n, ea, eb)
    return j


def _components(n, ea, eb):
    """Connected components of the graph (n nodes, undirected edges ea[i] -- eb[i]) by min-label propagation with pointer
    jumping (numpy only: the product path carries no scipy).  -> (count, component id per node); ids number the components
    by their smallest node, so every rank derives the same numbering from the same tables."""
    lab = np.arange(n, dtype=np.int64)
    if ea.size:
        while True:
            m = np.minimum(lab[ea], lab[eb])
            new = lab.copy()
            np.minimum.at(new, ea, m)
            np.minimum.at(new, eb, m)
            new = new[new]                       # labels are node indices: jump to the label's own label
            if np.array_equal(new, lab):
                break
            lab = new
    u, comp = np.unique(lab, return_inverse=True)
    return int(u.size), comp.astype(np.int64)


class RcclComm:
    """Production communicator: everything the path exchanges travels over RCCL on the context's stream -- float ghost
    planes and bit planes (ncclSend / ncclRecv between Z neighbours), scalar reductions (ncclAllReduce) and the
    variable-size gathers of threshold samples and slab tables (ncclAllGather on padded staging)."""

    def __init__(self, ctx, world, rank, uid: bytes, host_gather=None, uid2: bytes = None):
        self.world, self.rank = world, rank
        self.ctx = ctx
        ctx.comm_init(world, rank, uid)
        self.has_side_channel = uid2 is not None          # a second communicator: asynchronous ghost-plane exchanges
        if uid2 is not None:
            ctx.comm_init2(world, rank, uid2)
        # the sampling / statistics entry points return GLOBAL values (nl_comm_fuse): ShardedFramePipeline then makes the
        # single-GPU sequence of calls and skips the host-level all-reduce behind each of them
        self.fused = os.environ.get("NELLIE_FUSE_REDUCE", "1") == "1"
        if self.fused:
            ctx.comm_fuse(True)

    def exchange_halo(self, ctx, field, depth, offset=0, run_async=False):
        ctx.halo_exchange_at(field, offset, depth, run_async and self.has_side_channel)

    def exchange_bits(self, ctx, which):
        ctx.slab_bits_exchange(which)

    def allreduce(self, arr, op):
        return self.ctx.allreduce(arr, op)

    def positive_samples_world(self, ctx, field, mode, a, b, c, block_items):
        return ctx.positive_samples_world(field, mode, a, b, c, block_items, self.world)

    def slab_phase_gather(self, ctx, phase):
        """A Label phase up to the tables of ALL ranks: built, all-gathered (fixed blocks, ncclAllGather on the context stream)
        and fetched inside one library call -- one wait, no size negotiation through the host."""
        return ctx.slab_phase(phase, gather_world=self.world)

    def allgather_list(self, arr):
        return self.ctx.allgather_var(np.ascontiguousarray(arr), self.world)

    def allgather(self, arr):
        return np.concatenate(self.allgather_list(arr))

    def allgather_mask_bits(self, ctx, slab_plane0):
        ctx.label_bits_allgather(slab_plane0)


class ShardedFramePipeline(FramePipeline):
    _tail_ok = False            # the percentile samples are gathered across the ranks (pipeline.py: _scales_on_the_device)

    def __init__(self, gshape, rank, world, comm_factory, params: FilterParams, device: int = 0, ctx_factory=None, halo=None,
                 halo_mode=None):
        """
        comm_factory(ctx) -> communicator with exchange_halo / exchange_bits / allreduce / allgather.
        ctx_factory(local_shape, device, gz0, gnz, own) -> context (default: the HIP context).
        halo_mode: "steps" (per-step exchange, default; NELLIE_HALO overrides) or "fat" (one exchange per frame).
        halo: ghost planes per interior side (default: what the scheme needs; Label alone needs 1).
        """
        import os
        self.rank, self.world = int(rank), int(world)
        self.halo_mode = halo_mode or os.environ.get("NELLIE_HALO", "steps")
        if self.halo_mode not in ("steps", "fat"):
            raise ValueError(f"halo_mode must be 'steps' or 'fat', not {self.halo_mode!r}")
        self._rz = step_radii(params)
        need = halo_depth_steps(params) if self.halo_mode == "steps" else halo_depth(params)
        self.halo = need if halo is None else int(halo)
        self._filter_ok = self.halo >= need                # a Label-only pipeline may hold a single ghost plane
        lshape, gz0, own_lo, own_hi = slab_geometry(gshape, world, rank, self.halo)
        self.lshape, self.gz0, self.own = lshape, gz0, (own_lo, own_hi)
        make = ctx_factory or (lambda shp, dev, g0, gn, own: hipnative.Context(shp, device=dev, gz0=g0, gnz=gn, own=own))
        ctx = make(lshape, device, gz0, int(gshape[0]), (own_lo, own_hi))
        super().__init__(gshape, device=device, ctx=ctx)
        self.comm = comm_factory(ctx)
        # the sample range is reduced across the ranks before the histogram pass: by the library itself between the two
        # kernels (a "fused" communicator), or by a host-level all-reduce between two calls
        # the next cascade step is enqueued beside this scale's Hessian walk (pipeline.py): on slabs it fills the GPU while
        # the host waits for the collectives of the scale's thresholds (31.3 -> 30.8 ms/step on one rank; NELLIE_GAUSS_AHEAD=0: off)
        self._gauss_ahead = hasattr(ctx, "gauss_commit") and os.environ.get("NELLIE_GAUSS_AHEAD", "1") == "1"
        self._fused_reduce = bool(getattr(self.comm, "fused", False))
        self._chain_hist = self._fused_reduce
        self.params = params
        self._valid = (0, lshape[0])
        self._raw_ghosts_loaded = False
        # threshold + opening + product in one go, ghost planes included (nl_mask_volume_fused works on owned +- 2)
        self._fused_epilogue = hasattr(ctx, "mask_volume_fused") and hasattr(ctx, "sample_gather_positive")

    # ---- loading: own planes in, ghost planes from the neighbours ---------------------------------------
    def raw_ghost_needed(self):
        """(low, high) raw planes beyond the owned ones that the first cascade step reads: what `load_input` / `filter`
        accept in addition to the owned planes.  Whoever holds the whole input (a file, a host array) hands them over with
        the frame and the first -- the only exposed -- ghost-plane exchange of the frame is not needed at all."""
        if self.halo_mode == "fat":
            return self.own[0], self.lshape[0] - self.own[1]
        lo, hi = self.own
        d = 4 + self._rz[0]
        return min(d, lo), min(d, self.lshape[0] - hi)

    def _place(self, frame):
        """`frame` = this rank's OWN planes, or the owned planes with `raw_ghost_needed()` ghost planes on each side:
        -> (first local plane, last + 1) it fills."""
        lo, hi = self.own
        n = np.shape(frame)[0]
        g_lo, g_hi = self.raw_ghost_needed()
        if n == hi - lo:
            self._raw_ghosts_loaded = (g_lo == 0 and g_hi == 0)
            return lo, hi
        if n == hi - lo + g_lo + g_hi:
            self._raw_ghosts_loaded = True
            return lo - g_lo, hi + g_hi
        raise ValueError(f"a frame of {n} planes is neither the {hi - lo} owned planes nor those plus the {g_lo} + {g_hi} raw ghost planes")

    def _load(self, frame):
        z0, z1 = self._place(frame)
        self.ctx.filter_load(np.asarray(frame), z0=z0, z1=z1)

    def load_input(self, frame):
        z0, z1 = self._place(frame)
        self.ctx.input_load(np.asarray(frame), z0=z0, z1=z1)

    def _after_load(self, p):
        if not self._filter_ok:
            raise ValueError(f"this slab holds {self.halo} ghost planes, the {self.halo_mode!r} exchange scheme of Filter needs more")
        lo, hi = self.own
        nzl = self.lshape[0]
        have = self._raw_ghosts_loaded            # every rank is called the same way: all of them have their ghosts or none has
        if self.halo_mode == "fat":
            depth = max(lo, nzl - hi)
            if depth and not have:
                self.comm.exchange_halo(self.ctx, FIELD_GAUSS, depth)
            self._valid = (0, nzl)
        elif self.world > 1 and not have:
            self.comm.exchange_halo(self.ctx, FIELD_GAUSS, 4 + self._rz[0], 0, False)     # raw planes: nothing to hide them behind

    # ---- Z ranges of the cascade ------------------------------------------------------------------------------

    def _gauss_range(self, rz):
        nzl = self.lshape[0]
        if self.halo_mode == "steps":      # owned planes +- 4, clipped where the slab ends at a true face
            lo, hi = self.own
            return max(lo - 4, 0), min(hi + 4, nzl)
        v0, v1 = self._valid
        z0 = v0 if self.gz0 + v0 == 0 else v0 + rz              # a true face reflects, no shrink
        z1 = v1 if self.gz0 + v1 == self.shape[0] else v1 - rz
        z0 = max(z0, 0)
        z1 = min(z1, nzl)
        self._valid = (z0, z1)
        return z0, z1

    def _after_cascade_step(self, k):
        """steps scheme: the r_z planes step k+1 needs beyond owned +- 4 leave now and travel beside scale k's work."""
        if self.halo_mode == "steps" and self.world > 1 and k + 1 < len(self._rz) and self._rz[k + 1] > 0:
            self.comm.exchange_halo(self.ctx, FIELD_GAUSS, self._rz[k + 1], 4, True)

    def _chain_ahead(self, n_voxels):
        return False             # a slab's next cascade step needs ghost planes that are still travelling (_after_cascade_step)

    def _vess_range(self):
        lo, hi = self.own
        return max(lo - 2, 0), min(hi + 2, self.lshape[0])

    # ---- exact global scalars -----------------------------------------------------------------------------
    def _reduce_minmax(self, mn, mx, npos):
        """Global (min, max, any positive sample?) in ONE collective: max over (max, -min, has-samples).  Callers only
        ask whether the count is zero, so the third value is 0 / 1."""
        if self._fused_reduce:
            return mn, mx, npos
        big = np.float32(np.inf)
        r = self.comm.allreduce(np.array([mx if npos else -big, -mn if npos else -big, 1.0 if npos else 0.0], np.float32), "max")
        if r[2] <= 0:
            return mn, mx, 0
        return np.float32(-r[1]), np.float32(r[0]), 1

    def _reduce_counts(self, counts):
        if self._fused_reduce:
            return counts
        return self.comm.allreduce(np.ascontiguousarray(counts, dtype=np.int64), "sum")

    def _reduce_stats(self, max_abs, max_fsq, any_inf, overflow=0, fused_call=False):
        if self._fused_reduce and fused_call:
            return max_abs, max_fsq, bool(any_inf), bool(overflow)
        r = self.comm.allreduce(np.array([max_abs, max_fsq, 1.0 if any_inf else 0.0, 1.0 if overflow else 0.0], np.float32), "max")
        return np.float32(r[0]), np.float32(r[1]), bool(r[2] > 0), bool(r[3] > 0)

    def _reduce_sum(self, n):
        return int(self.comm.allreduce(np.array([n], np.int64), "sum")[0])

    def _tail_reductions_on_device(self) -> bool:
        return bool(self._fused_reduce)      # the percentile's histograms are all-reduced between the kernels (nl_tail_enqueue)

    def _reduce_fused_count(self, n):
        return int(n) if self._fused_reduce else self._reduce_sum(n)     # nl_mask_volume_fused reduces on the device when fused

    # The device-resident threshold chain needs every reduction of a scale on the device, between the kernels (a fused
    # communicator).  Exact on slabs (tests/test_hip_sharded.py runs it over the loopback transport) and ON by default since round 4:
    # one rank's 128 x 2048 x 2048 step measures 25.3 ms with it against 26.0 ms on the synchronous path (tools/prof_slab.py;
    # the bench's harness: 26.5 vs 26.7), and what it removes -- ~20 host-synchronous reductions per frame -- is what would each
    # become a collective plus a wait on 8 ranks.  (Round 3 measured it slower in the bench harness, 38.5 vs 29.8 ms: that was the
    # cost of communicators destroyed earlier in the process, gone since they are pooled.)  NELLIE_DEVICE_CHAIN_SLABS=0: off.
    def _chain_reductions_on_device(self) -> bool:
        return bool(self._fused_reduce) and os.environ.get("NELLIE_DEVICE_CHAIN_SLABS", "1") == "1"

    def _all_ranks_agree(self, ok: bool) -> bool:
        """Every rank decides from the same global histograms and statistics; the reduction only guards the fallback (a
        collective sequence of its own) against a rank that disagrees."""
        if self.world == 1:
            return ok
        return bool(self.comm.allreduce(np.array([1 if ok else 0], np.int64), "min")[0])

    def _reduce_mask_count(self, n):
        return n            # this rank's share; _settle_mask_counts turns all of a frame's counts global in one collective

    def _settle_mask_counts(self):
        hit = [sc for sc in self.trace.scales if sc.one_pass]
        if hit:
            tot = self.comm.allreduce(np.array([sc.mask_count for sc in hit], np.int64), "sum")
            for sc, t in zip(hit, tot):
                sc.mask_count = int(t)

    def _gather(self, samples):
        return self.comm.allgather(np.ascontiguousarray(samples, dtype=np.float32))

    # Over RCCL the positive samples of a threshold never visit the host on their own: compaction, all-gather in fixed blocks
    # and ONE download happen inside one library call (nl_positive_samples_world).  The block is a bound on any rank's sample
    # points, derived from the global shape so that every rank passes the same number.
    def _positive_lattice_samples(self, fld, strides):
        f = getattr(self.comm, "positive_samples_world", None)
        if f is None:
            return super()._positive_lattice_samples(fld, strides)
        sz, sy, sx = (int(v) for v in strides)
        gnz, ny, nx = self.shape
        per_plane = -(-ny // sy) * -(-nx // sx)
        most = max(-(-o1 // sz) - -(-o0 // sz) for o0, o1 in (slab_range(gnz, self.world, r) for r in range(self.world)))
        return f(self.ctx, fld, 0, sz, sy, sx, max(1, most * per_plane))

    def _positive_flat_samples(self, fld, offset, step):
        f = getattr(self.comm, "positive_samples_world", None)
        if f is None:
            return super()._positive_flat_samples(fld, offset, step)
        gnz, ny, nx = self.shape
        plane = ny * nx
        most = max(-(-(o1 - o0) * plane // int(step)) + 1 for o0, o1 in (slab_range(gnz, self.world, r) for r in range(self.world)))
        return f(self.ctx, fld, 1, int(offset), int(step), 0, max(1, most))

    # ---- outputs ------------------------------------------------------------------------------------------
    def download_frangi(self, out=None):
        """This rank's OWN planes of the Filter output."""
        lo, hi = self.own
        return self.ctx.filter_store(z0=lo, z1=hi, out=out)

    def upload_frangi(self, frangi):
        """`frangi` = this rank's OWN planes (Label run stand-alone on slabs)."""
        lo, hi = self.own
        self.ctx.label_load_frangi(np.asarray(frangi, dtype=np.float32), z0=lo, z1=hi)

    def intensity_mask(self, original, thresh):
        """labelling.py:550-552 on this rank's OWN planes (`original` = those planes of the original image)."""
        lo, hi = self.own
        self.ctx.label_intensity_mask(np.asarray(original), thresh, z0=lo, z1=hi)

    def _gather_list(self, arr):
        f = getattr(self.comm, "allgather_list", None)
        if f is not None:
            return f(arr)
        # communicators that only concatenate: gather the sizes first
        sizes = self.comm.allgather(np.array([arr.size], np.int64))
        flat = self.comm.allgather(arr)
        cuts = np.cumsum(np.concatenate([[0], sizes]))
        return [flat[cuts[r]:cuts[r + 1]] for r in range(self.world)]

    def _slab_phase(self, phase):
        """Components of this slab for one phase + the joined view of the trees that continue on other ranks.  Over RCCL the
        tables are all-gathered on the device inside the context's own call (one wait per phase); other communicators gather
        the rank's blob through the host."""
        fused = getattr(self.comm, "slab_phase_gather", None)
        blobs = fused(self.ctx, phase) if fused is not None else self._gather_list(self.ctx.slab_phase(phase)[0])
        return join_slab_blobs(blobs)

    def label(self, frangi_thresh, min_area, fill_holes=True):
        """labelling.py:467-509 across slabs (see the module docstring); returns the GLOBAL label count."""
        ctx, comm, me = self.ctx, self.comm, self.rank
        self.trace.label_thr = None if frangi_thresh is None else float(frangi_thresh)
        ctx.slab_label_pack(frangi_thresh)
        if fill_holes:
            comm.exchange_bits(ctx, 0)
            j = self._slab_phase(SLAB_FILL)
            if j.ncomp:                                            # (nothing crosses an interface: nothing to learn from the others)
                mine = j.rank == me
                outside = np.zeros(j.ncomp, bool)
                np.logical_or.at(outside, j.comp, j.val != 0)
                fix = mine & (j.val == 0) & outside[j.comp]
                ctx.slab_patch(j.root[fix], np.ones(int(fix.sum()), np.int32))
            ctx.slab_apply()
        comm.exchange_bits(ctx, 0)
        j = self._slab_phase(SLAB_AREA)
        if j.ncomp:
            mine = j.rank == me
            area = np.zeros(j.ncomp, np.int64)
            np.add.at(area, j.comp, j.val.astype(np.int64))
            ctx.slab_patch(j.root[mine], np.minimum(area[j.comp[mine]], 2 ** 31 - 1).astype(np.int32))
        ctx.slab_apply(int(min_area))
        comm.exchange_bits(ctx, 1)
        ctx.slab_majority()
        comm.exchange_bits(ctx, 0)
        j = self._slab_phase(SLAB_NUMBER)
        empty = np.zeros(0, np.int32)
        if j.ncomp == 0:                                        # no tree crosses an interface: every rank numbers its own, ids offset
            k_local, _ = ctx.slab_number(empty, empty)
            my_comps = local_id = np.zeros(0, np.int64)
            mine = np.zeros(0, bool)
        else:
            mine = j.rank == me
            none = np.int64(2 ** 31 - 1)
            # the owner of a component: the lowest rank holding voxels of it; its defining run there: the first one
            owner = np.full(j.ncomp, self.world, np.int64)
            has = j.val != none
            np.minimum.at(owner, j.comp[has], j.rank[has])
            first = np.full(j.ncomp, none, np.int64)
            own_nodes = mine & has & (owner[j.comp] == me)
            np.minimum.at(first, j.comp[own_nodes], j.val[own_nodes].astype(np.int64))
            my_comps = np.flatnonzero(first != none)
            k_local, local_id = ctx.slab_number(j.root[mine], first[my_comps].astype(np.int32))
        mine_part = np.concatenate([np.array([k_local], np.int64), my_comps.astype(np.int64), local_id.astype(np.int64)])
        parts = [mine_part] if self.world == 1 else self._gather_list(mine_part)
        counts = np.array([int(p[0]) for p in parts], np.int64)
        base = np.concatenate([[0], np.cumsum(counts)])
        if j.ncomp == 0:
            ctx.slab_paint(int(base[me]), empty, empty)
        else:
            label_of = np.zeros(j.ncomp, np.int64)
            for r, p in enumerate(parts):
                m = (p.size - 1) // 2
                label_of[p[1:1 + m]] = base[r] + p[1 + m:]
            ctx.slab_paint(int(base[me]), j.root[mine], label_of[j.comp[mine]].astype(np.int32))
        self.trace.n_labels = int(base[-1])
        return self.trace.n_labels

    def label_replicated(self, frangi_thresh, min_area, fill_holes=True):
        """The first implementation (kept for A/B and as a cross-check in tests): all-gather the global bit mask and run
        the run-level labelling redundantly on every rank."""
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
