#!/usr/bin/env python3
"""Device-resident threshold chain on / off, alternating step by step inside ONE process (so clock and thermal drift hit both):
wall time per step and per-group kernel times.   tools/ab_chain.py Z Y X pairs"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import pipeline as pl
from nellie_amd.synthetic import ISO_01, make_volume

shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1024, 1024, 1024)
pairs = int(sys.argv[4]) if len(sys.argv) >= 5 else 6
pipe = pl.FramePipeline(shape)
pipe.load_input(make_volume(shape, 2345))
p = pl.FilterParams(dim_res=ISO_01)
ma = pl.min_area_pixels_of(ISO_01)
GROUPS = ("gauss_zyx<4,4>", "gauss_zyx<3,3>", "gauss_zyx<5,5>", "gauss_zyx<1,4>", "gauss_zyx<1,3>", "gauss_zyx<2,5>", "gauss_z", "gauss_yx", "sample", "vesselness", "vesselness_resolve", "mask_volume", "label")


def step():
    pipe.filter(None, p)
    return pipe.label(pipe.frangi_threshold(), ma)


for mode in (True, False):
    pipe._device_chain = mode
    step()
acc = {True: [], False: []}
grp = {True: {g: 0.0 for g in GROUPS}, False: {g: 0.0 for g in GROUPS}}
for k in range(pairs):
    for mode in ((True, False) if k % 2 == 0 else (False, True)):
        pipe._device_chain = mode
        pipe.ctx.prof_reset(); pipe.ctx.prof_enable(True)
        pipe.ctx.sync(); t0 = time.perf_counter()
        step()
        pipe.ctx.sync(); acc[mode].append((time.perf_counter() - t0) * 1e3)
        pipe.ctx.prof_enable(False)
        for g in GROUPS:
            grp[mode][g] += pipe.ctx.prof_get(g)[0]
out = {"shape": list(shape), "pairs": pairs, "fallbacks": pipe.chain_fallbacks}
for mode, name in ((True, "chain"), (False, "sync")):
    out[name] = {"ms_per_step": round(float(np.mean(acc[mode])), 3), "min": round(float(np.min(acc[mode])), 3),
                 "groups": {g: round(v / pairs, 3) for g, v in grp[mode].items()}, "kernel_sum": round(sum(grp[mode].values()) / pairs, 3)}
print(json.dumps(out))
pipe.close()
