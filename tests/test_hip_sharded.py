"""
GPU test of the Z-slab Filter on the real HIP engine: `world` contexts on ONE device, one Python thread per
rank, ghost planes exchanged through nl_planes_get / nl_planes_put ("fake multi-GPU").
Filter AND Label: the concatenated slabs must equal the single-context result BIT FOR BIT (every cross-slab quantity is an
integer sum, a min or a max, or is recomputed from identical inputs).  The RCCL path differs only in how the
same planes and scalars travel.
"""
import threading

import numpy as np
import pytest

from comms import ThreadComm, ThreadGroup

pytestmark = pytest.mark.gpu


def _run_sharded(gshape, dr, seed, world):
    from nellie_amd.pipeline import FilterParams, min_area_pixels_of
    from nellie_amd.sharded import ShardedFramePipeline, slab_range
    from nellie_amd.synthetic import make_volume
    group = ThreadGroup(world)
    out, errs = [None] * world, []

    def worker(rank):
        try:
            p = FilterParams(dim_res=dr)
            o0, o1 = slab_range(gshape[0], world, rank)
            own = make_volume((o1 - o0,) + tuple(gshape[1:]), seed, z_offset=o0, global_nz=gshape[0])
            pipe = ShardedFramePipeline(gshape, rank, world, lambda ctx: ThreadComm(group, rank), p)
            pipe.filter(own, p)
            thr = pipe.frangi_threshold()
            n = pipe.label(thr, min_area_pixels_of(dr))
            out[rank] = (pipe.download_frangi(), thr, [s.mask_count for s in pipe.trace.scales],
                         pipe.download_labels(), n)
            pipe.close()
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)
            group.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errs:
        raise errs[0]
    return out


@pytest.mark.parametrize("gshape,aniso,world", [((96, 64, 80), False, 2), ((100, 48, 70), False, 3),
                                               ((60, 64, 64), True, 4)])
def test_zslab_filter_equals_single_gpu(hip, gshape, aniso, world):
    from nellie_amd import pipeline as pl
    from nellie_amd.synthetic import ANISO_03, ISO_01, make_volume
    dr = ANISO_03 if aniso else ISO_01
    vol = make_volume(gshape, 91)
    single = pl.FramePipeline(gshape)
    p = pl.FilterParams(dim_res=dr)
    single.filter(vol, p)
    ref = single.download_frangi()
    ref_thr = single.frangi_threshold()
    ref_counts = [s.mask_count for s in single.trace.scales]
    ref_n = single.label(ref_thr, pl.min_area_pixels_of(dr))
    ref_lab = single.download_labels()
    single.close()
    parts = _run_sharded(gshape, dr, 91, world)
    got = np.concatenate([p_[0] for p_ in parts])
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} voxels differ"
    for _, thr, counts, _, n in parts:
        assert thr == ref_thr and counts == ref_counts and n == ref_n
    assert (ref > 0).any()
    lab = np.concatenate([p_[3] for p_ in parts])
    assert np.array_equal(lab, ref_lab), f"{int((lab != ref_lab).sum())} label voxels differ"
    assert ref_n >= 1


def test_rccl_communicator_world1(hip):
    """RCCL plumbing on a single device: unique id, communicator, all-reduce, and a 1-rank sharded run."""
    from nellie_amd import hipnative
    from nellie_amd import pipeline as pl
    from nellie_amd.sharded import RcclComm, ShardedFramePipeline
    from nellie_amd.synthetic import ISO_01, make_volume
    gshape = (40, 48, 56)
    vol = make_volume(gshape, 5)
    p = pl.FilterParams(dim_res=ISO_01)
    uid = hipnative.comm_unique_id()
    assert len(uid) == 128
    pipe = ShardedFramePipeline(gshape, 0, 1, lambda ctx: RcclComm(ctx, 1, 0, uid, lambda a: a), p)
    assert np.array_equal(pipe.comm.allreduce(np.array([3, 4], np.int64), "sum"), [3, 4])
    assert np.array_equal(pipe.comm.allreduce(np.array([1.5], np.float32), "max"), np.array([1.5], np.float32))
    pipe.filter(vol, p)
    got = pipe.download_frangi()
    pipe.close()
    single = pl.FramePipeline(gshape)
    single.filter(vol, p)
    assert np.array_equal(got, single.download_frangi())
    single.close()
