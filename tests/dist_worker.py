"""One rank of the world_size-2 gloo test (CPU): Z-slab Filter AND Label with the oracle-backed context."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    import torch.distributed as dist
    out_dir, aniso = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo", init_method="env://")
    rank, world = dist.get_rank(), dist.get_world_size()
    from comms import GlooComm
    from fake_ctx import OracleCtx
    from nellie_amd.pipeline import FilterParams, min_area_pixels_of
    from nellie_amd.sharded import ShardedFramePipeline, slab_range
    from nellie_amd.synthetic import ANISO_03, ISO_01, make_volume
    dr = ANISO_03 if aniso else ISO_01
    gshape = (40, 36, 44) if aniso else (64, 30, 34)
    p = FilterParams(dim_res=dr)
    o0, o1 = slab_range(gshape[0], world, rank)
    pipe = ShardedFramePipeline(gshape, rank, world, lambda ctx: GlooComm(dist, rank, world), p,
                                ctx_factory=lambda shp, dev, g0, gn, ow: OracleCtx(shp, dev, g0, gn, ow))
    # NELLIE_TEST_RAW_GHOSTS=1: the raw ghost planes the first cascade step reads come with the frame (no exchange of raw planes)
    g_lo, g_hi = pipe.raw_ghost_needed() if os.environ.get("NELLIE_TEST_RAW_GHOSTS") == "1" else (0, 0)
    own = make_volume((o1 - o0 + g_lo + g_hi,) + gshape[1:], 77, z_offset=o0 - g_lo, global_nz=gshape[0])
    calls = []
    inner = pipe.comm.exchange_halo
    pipe.comm.exchange_halo = lambda ctx, field, depth, offset=0, run_async=False: (calls.append((depth, offset)), inner(ctx, field, depth, offset, run_async))[1]
    pipe.filter(own, p)
    if g_lo or g_hi:
        assert all(off > 0 for _, off in calls), f"raw planes were exchanged although they came with the frame: {calls}"
    fr = pipe.download_frangi()
    thr = pipe.frangi_threshold()
    n_labels = pipe.label(thr, min_area_pixels_of(dr))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), frangi=fr, o0=o0, o1=o1, labels=pipe.download_labels(), n_labels=n_labels,
             thr=np.float64(np.nan if thr is None else thr), gamma=[s.gamma for s in pipe.trace.scales],
             mask_count=[s.mask_count for s in pipe.trace.scales], halo=pipe.halo)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
