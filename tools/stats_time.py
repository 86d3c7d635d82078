import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from nellie_amd import pipeline as pl
from nellie_amd.synthetic import ISO_01, make_volume
vol = make_volume((512,1024,1024), 1234)
pipe = pl.FramePipeline(vol.shape); pipe.load_input(vol)
ctx = pipe.ctx
ctx.filter_begin() if hasattr(ctx, "filter_begin") else None
sp = pl.spacing_of(ISO_01)
for dpp in ("0", "1"):
    os.environ["NELLIE_HV_DPP"] = dpp
    r = ctx.hessian_stats(sp)
    ctx.sync(); ctx.prof_reset(); ctx.prof_enable(True)
    for _ in range(10): r = ctx.hessian_stats(sp)
    ctx.sync(); ctx.prof_enable(False)
    ms, k = ctx.prof_get("hessian_stats")
    print("MODE 0 in the library, NELLIE_HV_DPP=%s: %.4f ms per launch  stats %s" % (dpp, ms / max(k, 1), [float(x) for x in r[:2]] + [int(r[2])]))
