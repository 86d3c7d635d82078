"""One rank of the world_size-2 gloo test of the STAGE API: `Filter(im_info, shard=...).run()` and `Label(im_info, shard=...).run()`
with the gloo communicator (host-staged exchanges, tests/comms.py) -- every rank writes its own planes of the shared output files
(nellie_amd/engine.py: RankSlab).  On CPU the contexts are oracle-backed doubles; with `--hip` as the last argument they are the
real HIP contexts, both ranks on device 0 (tests/test_hip_sharded.py: the multi-process path with the real library on a one-GPU box;
RCCL itself refuses two ranks on one device, so the communicator stays gloo there)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    import torch.distributed as dist
    hip = sys.argv[-1] == "--hip"
    if hip:
        sys.argv.pop()
    src, out_dir = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo", init_method="env://")
    rank, world = dist.get_rank(), dist.get_world_size()
    from comms import GlooComm
    from fake_ctx import OracleCtx
    from nellie_amd.utils import adaptive_run
    if not hip:
        adaptive_run.gpu_available = lambda: True       # no GPU here: the contexts below are oracle-backed
    from nellie_amd.engine import ShardSpec
    from nellie_amd.im_info.verifier import ImInfo
    from nellie_amd.segmentation.filtering import Filter
    from nellie_amd.segmentation.labelling import Label
    im_info = ImInfo(src, output_dir=out_dir)            # the canonical copy was made by the parent: reused, not rewritten
    spec = ShardSpec(rank=rank, world=world, comm_factory=lambda ctx: GlooComm(dist, rank, world),
                     ctx_factory=None if hip else (lambda shp, dev, g0, gn, ow: OracleCtx(shp, dev, g0, gn, ow)))
    Filter(im_info, shard=spec).run()
    Label(im_info, shard=spec).run()
    dist.barrier()
    if len(sys.argv) > 3:                                # then once more with a fixed intensity threshold (labelling.py:513-520, 550-552)
        import numpy as np
        if rank == 0:
            np.save(os.path.join(out_dir, "labels_plain.npy"), np.asarray(im_info.get_memmap(im_info.pipeline_paths["im_instance_label"], read_mode="r")))
        dist.barrier()
        Label(im_info, threshold=float(sys.argv[3]), shard=spec).run()
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
