import numpy as np, os, sys
sys.path.insert(0, os.getcwd())
from nellie_amd.pipeline import FilterParams, FramePipeline
from nellie_amd.synthetic import ISO_01
shape = (64, 64, 128)
z, y, x = np.mgrid[:shape[0], :shape[1], :shape[2]]
vol = np.random.default_rng(3).normal(100, 0.02, shape).astype(np.float32)
vol += (50.0 * (np.abs(np.sin(0.15 * x)) + np.abs(np.sin(0.15 * y)) + np.abs(np.sin(0.15 * z)))).astype(np.float32)
res = {}
for dpp in ("0", "1"):
    os.environ["NELLIE_HV_DPP"] = dpp
    pipe = FramePipeline(shape)
    pipe._device_chain = False
    log = []
    spec = pipe.ctx.vesselness_spec
    def hooked(*a, **k):
        r = spec(*a, **k)
        log.append((bool(r[3]), pipe.ctx.info("queue_entries"), float(r[0]), float(r[1])))
        return r
    pipe.ctx.vesselness_spec = hooked
    pipe.compute_vesselness(vol, FilterParams(dim_res=ISO_01))
    res[dpp] = (pipe.download_frangi(), [s.mask_count for s in pipe.trace.scales], [s.one_pass for s in pipe.trace.scales])
    print("dpp", dpp, log, res[dpp][1], res[dpp][2], flush=True)
    pipe.close()
print("equal", np.array_equal(res["0"][0], res["1"][0]), res["0"][1] == res["1"][1])
