"""HIP against the oracle on volumes scaled until the squared Frobenius norm of the Hessian overflows float32 (the trace test of the walk\nmust stay conservative there): tools/probe_overflow.py"""
import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import oracle.nellie_oracle as o
from nellie_amd import pipeline as pl
from nellie_amd.synthetic import ISO_01, make_volume
shape=(40,96,96)
for scale in [float(a) for a in sys.argv[1:]] or (1.0, 1e10, 1e15, 1e16, 3e16, 1e17, 3e17):
    vol=(make_volume(shape,5).astype(np.float64)*scale).astype(np.float32)
    with np.errstate(all="ignore"):
        ref=o.run_frame(vol, ISO_01)
    pipe=pl.FramePipeline(shape)
    pipe.compute_vesselness(vol, pl.FilterParams(dim_res=ISO_01))
    out=pipe.download_frangi()
    tr=[(s.mask_count, s.one_pass, s.skipped) for s in pipe.trace.scales]
    pipe.close()
    sup_eq=np.array_equal(ref>0, out>0)
    d=np.abs(ref.astype(np.float64)-out)
    print("scale",scale,"ref>0",int((ref>0).sum()),"hip>0",int((out>0).sum()),"support equal",sup_eq,"max diff/max",float(d.max()/max(ref.max(),1e-30)), "nan in ref",int(np.isnan(ref).sum()), tr)
