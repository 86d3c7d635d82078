"""One rank of the world_size-2 gloo test of the STAGE API on CPU: `Filter(im_info, shard=...).run()` and
`Label(im_info, shard=...).run()` with the oracle-backed context and the gloo communicator -- every rank writes its own
planes of the shared output files (nellie_amd/engine.py: RankSlab)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    import torch.distributed as dist
    src, out_dir = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo", init_method="env://")
    rank, world = dist.get_rank(), dist.get_world_size()
    from comms import GlooComm
    from fake_ctx import OracleCtx
    from nellie_amd.utils import adaptive_run
    adaptive_run.gpu_available = lambda: True           # no GPU here: the contexts below are oracle-backed
    from nellie_amd.engine import ShardSpec
    from nellie_amd.im_info.verifier import ImInfo
    from nellie_amd.segmentation.filtering import Filter
    from nellie_amd.segmentation.labelling import Label
    im_info = ImInfo(src, output_dir=out_dir)            # the canonical copy was made by the parent: reused, not rewritten
    spec = ShardSpec(rank=rank, world=world, comm_factory=lambda ctx: GlooComm(dist, rank, world),
                     ctx_factory=lambda shp, dev, g0, gn, ow: OracleCtx(shp, dev, g0, gn, ow))
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
