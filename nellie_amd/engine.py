"""
Frame engines behind the stage classes (`Filter`, `Label`, `run`): how ONE (Z, Y, X) frame is laid out over contexts.

  SingleContext   one context on one GPU (nellie_amd/pipeline.py): the common case.
  LocalSlabs      the frame as W Z-slab contexts driven from THIS process, one host thread per slab
                  (nellie_amd/sharded.py) -- on the GPUs named by `devices=[...]`, or on one GPU when a frame is too large
                  for one context (a context indexes < 2^31 voxels).  Ghost planes, bit planes, reductions and tables travel
                  through the library's in-process transport (include/nellie_amd.h: nl_comm_loopback_id): device-to-device
                  copies on the contexts' own streams, same call sites as RCCL.
  RankSlab        one rank of a multi-process job (`torchrun`, `mpirun`: WORLD_SIZE / RANK / LOCAL_RANK): this process
                  owns one slab on its GPU and exchanges over RCCL; the id of the communicator is handed over through a
                  file next to the outputs.  Every rank writes its own planes of the shared output files.

All three give the same bits: the sharded frame equals the single-context frame (tests/test_hip_sharded.py,
tests/test_sharded_cpu.py).  This replaces the reference's memory ladder (nellie/utils/adaptive_run.py:88-113,
filtering.py:1047-1076), whose chunked rungs change the result.
"""
from __future__ import annotations

import os
import threading
import time
from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np

from nellie_amd.pipeline import FilterParams, FramePipeline

MAX_LOCAL_SLABS = 16                        # csrc/loopback.inc: ranks of one in-process communicator
MAX_CONTEXT_VOXELS = (1 << 31) - 1          # nl_ctx_create: int32 voxel indices inside a context


@dataclass
class ShardSpec:
    """One rank of a multi-process Z-slab job.  comm_factory(ctx) -> communicator (default: RCCL, the id travelling through
    `rendezvous_dir`); ctx_factory as ShardedFramePipeline takes it (tests put a CPU double there)."""
    rank: int
    world: int
    device: int = 0
    comm_factory: Optional[Callable] = None
    ctx_factory: Optional[Callable] = None
    rendezvous_dir: Optional[str] = None
    tag: str = ""

    @staticmethod
    def from_env(rendezvous_dir=None):
        """WORLD_SIZE / RANK / LOCAL_RANK as torchrun and mpirun export them; None for a single process."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world <= 1:
            return None
        return ShardSpec(rank=int(os.environ.get("RANK", "0")), world=world, device=int(os.environ.get("LOCAL_RANK", "0")),
                         rendezvous_dir=rendezvous_dir, tag=os.environ.get("MASTER_PORT", ""))


def context_bytes(local_shape) -> Optional[int]:
    """HBM a context of this local shape takes: nl_ctx_bytes (35 B/voxel: four float32 volumes, the mask bit planes, the eigen queue --
    DESIGN.md section 3) plus the resident input (4 B/voxel at most).  None where the library cannot be asked."""
    try:
        from nellie_amd import hipnative
        nz, ny, nx = (int(s) for s in local_shape)
        return int(hipnative.load().cdll.nl_ctx_bytes(nz, ny, nx)) + 4 * nz * ny * nx
    except Exception:  # noqa: BLE001
        return None


def _free_bytes(device):
    from nellie_amd.utils import adaptive_run
    return adaptive_run.get_gpu_free_bytes(int(device))


HBM_HEADROOM = 0.95                         # of the free bytes a plan may claim (adaptive_run.frame_fits_on_device uses the same figure)


def memory_plan(shape_zyx, halo: int, w: int, devices, free_of=None, bytes_of=None):
    """Where the w slabs of a frame go (contiguous blocks of slabs per device, as LocalSlabs places them) and whether they fit:
    -> (fits, {device: bytes needed}, {device: bytes free}).  A device whose free memory cannot be asked counts as fitting."""
    free_of = free_of or _free_bytes
    bytes_of = bytes_of or context_bytes
    nz, ny, nx = (int(s) for s in shape_zyx)
    devs = [int(d) for d in (devices or [0])]
    owned = -(-nz // w)
    need = {}
    for r in range(w):
        lo, hi = r * owned, min(nz, (r + 1) * owned)
        if hi <= lo:
            continue
        local = (hi - lo) + ((halo if r > 0 else 0) + (halo if hi < nz else 0) if w > 1 else 0)
        b = bytes_of((local, ny, nx))
        if b is None:
            return True, {}, {}
        d = devs[r * len(devs) // w]
        need[d] = need.get(d, 0) + b
    free = {d: free_of(d) for d in need}
    fits = all(free[d] is None or need[d] <= free[d] * HBM_HEADROOM for d in need)
    return fits, need, free


def slabs_needed(shape_zyx, halo: int, n_devices: int = 1, devices=None, free_of=None, bytes_of=None) -> int:
    """Smallest slab count (a multiple of the device count) whose slabs, ghost planes included, fit a context's index range AND the
    free HBM of the devices they land on (round 6; the reference's ladder asks the same question with a 6 x frame heuristic,
    nellie/utils/adaptive_run.py:88-113).  The slabs of a frame are all resident at once (they exchange ghost planes every cascade step),
    so more slabs only help where there are more devices: when the smallest layout the index range allows does not fit the memory,
    this raises MemoryError with the figures -- the message carries "out of memory", which adaptive_run.is_oom_error and the
    reference's callers parse."""
    nz, ny, nx = (int(s) for s in shape_zyx)
    plane = ny * nx
    nd = max(1, int(n_devices))
    devs = [int(d) for d in devices] if devices else list(range(nd))
    w = nd
    while True:
        owned = -(-nz // w)
        local = owned + (2 * halo if w > 1 else 0)
        if local * plane <= MAX_CONTEXT_VOXELS:
            break
        if owned <= halo:
            raise MemoryError(f"a {nz} x {ny} x {nx} frame cannot be cut into Z slabs that fit a context "
                              f"({plane} voxels per plane, {halo} ghost planes per side)")
        w += nd
    fits, need, free = memory_plan((nz, ny, nx), halo, w, devs, free_of, bytes_of)
    if not fits:
        gb = lambda b: "?" if b is None else f"{b / 2**30:.1f}"            # noqa: E731
        detail = ", ".join(f"GPU {d}: needs {gb(need[d])} GiB, {gb(free.get(d))} GiB free" for d in sorted(need))
        raise MemoryError(f"out of memory: a {nz} x {ny} x {nx} frame as {w} Z slab(s) on device(s) {sorted(set(devs))} does not fit the free HBM "
                          f"({detail}; a plan may claim {HBM_HEADROOM:.0%} of the free bytes). Name more GPUs (devices=[...]) or run one rank "
                          f"per GPU (shard='env'): the slabs of a frame are resident together, so more slabs on the same GPUs need more memory, not less")
    return w


class SingleContext:
    kind = "single"

    def __init__(self, shape, device=0):
        self.pipe = FramePipeline(shape, device=device)
        self.world = 1

    @property
    def trace(self):
        return self.pipe.trace

    def filter(self, frame, params, mask=True, remove_edges=False):
        return self.pipe.filter(frame, params, mask=mask, remove_edges=remove_edges)

    def download_frangi(self, out=None):
        return self.pipe.download_frangi(out=out)

    def upload_frangi(self, frangi):
        self.pipe.upload_frangi(frangi)

    def intensity_mask(self, original, thresh):
        """frangi *= (original > thresh) on the uploaded Frangi frame (labelling.py:550-552)."""
        self.pipe.ctx.label_intensity_mask(np.asarray(original), thresh)

    def frangi_threshold(self, max_samples=1_000_000, nbins=256):
        return self.pipe.frangi_threshold(max_samples, nbins)

    def label(self, thr, min_area, fill_holes=True):
        return self.pipe.label(thr, min_area, fill_holes=fill_holes)

    def download_labels(self, out=None):
        return self.pipe.download_labels(out=out)

    def barrier(self):
        pass

    def close(self):
        self.pipe.close()


def _planes(frame, z0, z1):
    """Planes [z0, z1) of a host frame / memory map as a contiguous array."""
    return np.ascontiguousarray(frame[z0:z1])


class _SlabBase:
    """What LocalSlabs and RankSlab share: a ShardedFramePipeline takes its planes out of the whole frame (with the raw
    ghost planes the first cascade step reads, so no raw plane is exchanged) and puts its own planes into whole-frame outputs."""

    @staticmethod
    def _filter_one(pipe, frame, params, mask, remove_edges):
        from nellie_amd.sharded import slab_range
        o0, o1 = slab_range(pipe.shape[0], pipe.world, pipe.rank)
        g_lo, g_hi = pipe.raw_ghost_needed()
        return pipe.filter(_planes(frame, o0 - g_lo, o1 + g_hi), params, mask=mask, remove_edges=remove_edges)

    @staticmethod
    def _own(pipe):
        from nellie_amd.sharded import slab_range
        return slab_range(pipe.shape[0], pipe.world, pipe.rank)

    @classmethod
    def _download(cls, pipe, which, out):
        o0, o1 = cls._own(pipe)
        view = out[o0:o1]
        get = pipe.download_frangi if which == "frangi" else pipe.download_labels
        if isinstance(view, np.ndarray) and view.flags.c_contiguous and view.flags.writeable:
            got = get(out=view)
            if got is not view:             # a context that does not fill `out` in place
                view[...] = got
        else:
            view[...] = get()


class LocalSlabs(_SlabBase):
    kind = "local-slabs"

    def __init__(self, shape, params: FilterParams, devices=(0,), n_slabs=None, halo_mode=None, halo=None,
                 comm_factory_of_rank=None, ctx_factory=None):
        from nellie_amd import hipnative
        from nellie_amd.sharded import RcclComm, ShardedFramePipeline, halo_depth, halo_depth_steps
        self.shape = tuple(int(s) for s in shape)
        devices = [int(d) for d in devices] or [0]
        mode = halo_mode or os.environ.get("NELLIE_HALO", "steps")
        need = halo_depth_steps(params) if mode == "steps" else halo_depth(params)
        self.world = int(n_slabs) if n_slabs else slabs_needed(self.shape, need if halo is None else halo, len(devices))
        W = self.world
        if comm_factory_of_rank is None:
            uid, uid2 = hipnative.comm_unique_id(loopback=True), hipnative.comm_unique_id(loopback=True)
            comm_factory_of_rank = lambda rank: (lambda ctx: RcclComm(ctx, W, rank, uid, uid2=uid2))
        self.devices = [devices[r * len(devices) // W] for r in range(W)]       # contiguous blocks of slabs per device
        self.pipes = [None] * W

        def build(r):
            self.pipes[r] = ShardedFramePipeline(self.shape, r, W, comm_factory_of_rank(r), params, device=self.devices[r],
                                                 ctx_factory=ctx_factory, halo=halo, halo_mode=halo_mode)
        self._each(build)

    def _each(self, fn):
        """fn(rank) on one host thread per slab (the library calls release the GIL; the slabs rendezvous inside them)."""
        out, errs = [None] * self.world, []

        def work(r):
            try:
                out[r] = fn(r)
            except BaseException as exc:  # noqa: BLE001
                errs.append(exc)
        ts = [threading.Thread(target=work, args=(r,)) for r in range(self.world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs:
            raise errs[0]
        return out

    @property
    def trace(self):
        return self.pipes[0].trace

    def filter(self, frame, params, mask=True, remove_edges=False):
        return self._each(lambda r: self._filter_one(self.pipes[r], frame, params, mask, remove_edges))[0]

    def download_frangi(self, out=None):
        out = np.empty(self.shape, np.float32) if out is None else out
        self._each(lambda r: self._download(self.pipes[r], "frangi", out))
        return out

    def upload_frangi(self, frangi):
        self._each(lambda r: self.pipes[r].upload_frangi(_planes(frangi, *self._own(self.pipes[r]))))

    def intensity_mask(self, original, thresh):
        self._each(lambda r: self.pipes[r].intensity_mask(_planes(original, *self._own(self.pipes[r])), thresh))

    def frangi_threshold(self, max_samples=1_000_000, nbins=256):
        return self._each(lambda r: self.pipes[r].frangi_threshold(max_samples, nbins))[0]

    def label(self, thr, min_area, fill_holes=True):
        return self._each(lambda r: self.pipes[r].label(thr, min_area, fill_holes=fill_holes))[0]

    def download_labels(self, out=None):
        out = np.empty(self.shape, np.int32) if out is None else out
        self._each(lambda r: self._download(self.pipes[r], "labels", out))
        return out

    def barrier(self):
        pass

    def close(self):
        def shut(r):
            if self.pipes[r] is not None:
                self.pipes[r].close()
        self._each(shut)


def _exchange_ids(spec: ShardSpec, n_ids: int, make_id, timeout_s=300.0):
    """Rank 0 creates `n_ids` communicator ids and publishes them through this launch's file rendezvous (nellie_amd/rendezvous.py:
    the marker carries a nonce the ranks of THIS launch agreed on, so an id left behind by a launch that died is never picked up)."""
    from nellie_amd.rendezvous import rendezvous_for
    rdv = rendezvous_for(spec)
    _exchange_ids.counter = getattr(_exchange_ids, "counter", 0) + 1
    name = f"comm_{_exchange_ids.counter}.id"
    if spec.rank == 0:
        blob = b"".join(make_id() for _ in range(n_ids))
        rdv.publish(name, blob)
    else:
        blob = rdv.wait(name, timeout_s)
    return [blob[i * 128:(i + 1) * 128] for i in range(n_ids)], (rdv, name)


class RankSlab(_SlabBase):
    kind = "rank-slab"

    def __init__(self, shape, params: FilterParams, spec: ShardSpec, halo_mode=None, halo=None):
        from nellie_amd.sharded import RcclComm, ShardedFramePipeline
        self.shape = tuple(int(s) for s in shape)
        self.spec, self.world = spec, spec.world
        self._id_file = None
        comm_factory = spec.comm_factory
        if comm_factory is None:
            from nellie_amd import hipnative
            (uid, uid2), self._id_file = _exchange_ids(spec, 2, hipnative.comm_unique_id)
            comm_factory = lambda ctx: RcclComm(ctx, spec.world, spec.rank, uid, uid2=uid2)
        self.pipe = ShardedFramePipeline(self.shape, spec.rank, spec.world, comm_factory, params, device=spec.device,
                                         ctx_factory=spec.ctx_factory, halo=halo, halo_mode=halo_mode)
        self.barrier()
        if self._id_file and spec.rank == 0:          # every rank holds its communicators (the barrier above ran on them)
            self._id_file[0].remove(self._id_file[1])

    @property
    def trace(self):
        return self.pipe.trace

    def filter(self, frame, params, mask=True, remove_edges=False):
        return self._filter_one(self.pipe, frame, params, mask, remove_edges)

    def download_frangi(self, out=None):
        """This rank's planes into the whole-frame array (the other ranks fill theirs)."""
        out = np.zeros(self.shape, np.float32) if out is None else out
        self._download(self.pipe, "frangi", out)
        return out

    def upload_frangi(self, frangi):
        self.pipe.upload_frangi(_planes(frangi, *self._own(self.pipe)))

    def intensity_mask(self, original, thresh):
        self.pipe.intensity_mask(_planes(original, *self._own(self.pipe)), thresh)

    def frangi_threshold(self, max_samples=1_000_000, nbins=256):
        return self.pipe.frangi_threshold(max_samples, nbins)

    def label(self, thr, min_area, fill_holes=True):
        return self.pipe.label(thr, min_area, fill_holes=fill_holes)

    def download_labels(self, out=None):
        out = np.zeros(self.shape, np.int32) if out is None else out
        self._download(self.pipe, "labels", out)
        return out

    def barrier(self):
        """Every rank has reached this point (a one-element all-reduce on the communicator)."""
        self.pipe.comm.allreduce(np.array([1], np.int64), "sum")

    def close(self):
        self.pipe.close()


def plan_engine(shape_zyx, params: FilterParams, devices=None, shard: Optional[ShardSpec] = None, halo_mode=None, label_only=False,
                free_of=None, bytes_of=None):
    """("single" | "local-slabs" | "rank-slab", slab count) make_engine would build -- without building it.  The plan looks at the index
    range of a context AND at the free HBM of the devices (free_of / bytes_of: test doubles for the two questions)."""
    shape = tuple(int(s) for s in shape_zyx)
    if len(shape) == 2:
        return "single", 1
    if shard is not None and shard.world > 1:
        return "rank-slab", shard.world
    from nellie_amd.sharded import halo_depth, halo_depth_steps
    mode = halo_mode or os.environ.get("NELLIE_HALO", "steps")
    need = 1 if label_only else (halo_depth_steps(params) if mode == "steps" else halo_depth(params))
    w = slabs_needed(shape, need, len(devices) if devices else 1, devices=devices, free_of=free_of, bytes_of=bytes_of)
    forced = int(os.environ.get("NELLIE_FORCE_SLABS", "0"))
    if forced > 1:
        w = max(w, forced)
    if w > MAX_LOCAL_SLABS:
        raise ValueError(f"a {shape[0]} x {shape[1]} x {shape[2]} frame needs {w} Z slabs in this process, more than the {MAX_LOCAL_SLABS} "
                         f"the in-process transport carries: run it as one rank per GPU (shard='env') or on more GPUs")
    return ("single", 1) if w == 1 else ("local-slabs", w)


def make_engine(shape_zyx, params: FilterParams, device_index=0, devices=None, shard: Optional[ShardSpec] = None,
                halo_mode=None, label_only=False):
    """The engine for a frame of this shape.  2-D images and frames that fit one context on one GPU: SingleContext.
    `devices` with more than one entry, or a frame beyond a context's index range: LocalSlabs.  `shard`: RankSlab.
    label_only: the slabs only ever run Label (one ghost plane per side instead of the Filter halo)."""
    shape = tuple(int(s) for s in shape_zyx)
    halo = 1 if label_only else None
    if len(shape) == 2:
        if shard is not None and shard.world > 1:
            raise NotImplementedError("Z-slab sharding needs a Z axis: run 2-D images in one process")
        return SingleContext(shape, device=device_index)
    if shard is not None and shard.world > 1:
        return RankSlab(shape, params, shard, halo_mode=halo_mode, halo=halo)
    devs = [int(d) for d in devices] if devices else [int(device_index)]
    _, w = plan_engine(shape, params, devs, None, halo_mode, label_only)
    try:
        if w == 1:
            return SingleContext(shape, device=devs[0])
        return LocalSlabs(shape, params, devices=devs, n_slabs=w, halo_mode=halo_mode, halo=halo)
    except Exception as exc:  # noqa: BLE001
        # the plan asked for the free HBM a moment ago; another process can still take it before the contexts exist: the library's
        # "[out of memory]" then becomes the MemoryError the reference's ladder (filtering.py:1053-1076) and its callers expect
        from nellie_amd.utils import adaptive_run
        if adaptive_run.is_oom_error(exc) and not isinstance(exc, MemoryError):
            raise MemoryError(f"out of memory while building the engine of a {shape} frame on device(s) {devs}: {exc}") from exc
        raise
