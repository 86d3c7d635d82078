#!/usr/bin/env python3
"""Find an input whose one-pass walk overflows a queue region (dense masks), and check the fallback."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import pipeline as pl
from nellie_amd.synthetic import ISO_01
shape = (64, 64, 128)
rng = np.random.default_rng(3)
z, y, x = np.mgrid[:shape[0], :shape[1], :shape[2]]
for amp in (50.0, 400.0):
    vol = rng.normal(100, 1, shape).astype(np.float32)
    vol += (amp * (np.sin(x * 0.9) * np.sin(y * 0.9) * np.sin(z * 0.9))).astype(np.float32)      # dense texture
    res = {}
    for mode in (False, True):
        pipe = pl.FramePipeline(shape); pipe.one_pass = mode
        ovf = []
        orig = pipe.ctx.vesselness_spec
        def wrapped(*a, orig=orig, pipe=pipe, ovf=ovf, **k):
            r = orig(*a, **k); ovf.append(bool(r[3])); return r
        pipe.ctx.vesselness_spec = wrapped
        pipe.compute_vesselness(vol, pl.FilterParams(dim_res=ISO_01))
        res[mode] = pipe.download_frangi()
        print("amp", amp, "one_pass" if mode else "two_pass", "mask frac", [round(s.mask_count / vol.size, 3) for s in pipe.trace.scales],
              "hits", [int(s.one_pass) for s in pipe.trace.scales], "overflow", ovf)
        pipe.close()
    print("   equal:", np.array_equal(res[False], res[True]))
