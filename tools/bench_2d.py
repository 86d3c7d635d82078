#!/usr/bin/env python3
"""Filter + Label on 2-D (no_z) frames: ms per frame and Mpixel/s, per-group kernel times.   tools/bench_2d.py [Y X] [N]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import pipeline as pl
from nellie_amd.synthetic import make_volume

yx = tuple(int(a) for a in sys.argv[1:3]) if len(sys.argv) >= 3 else (2048, 2048)
n = int(sys.argv[3]) if len(sys.argv) >= 4 else 50
dr = {"X": 0.1, "Y": 0.1, "Z": None, "T": 1.0}
img = make_volume((8,) + yx, 99)[4].copy()
pipe = pl.FramePipeline(img.shape)
assert pipe.two_d
p = pl.FilterParams(dim_res=dr)
ma = pl.min_area_pixels_of(dr, no_z=True)


def step():
    pipe.filter(img, p)
    return pipe.label(pipe.frangi_threshold(), ma, fill_holes=False)


for _ in range(3):
    nl = step()
pipe.ctx.sync(); t0 = time.perf_counter()
for _ in range(n):
    step()
pipe.ctx.sync(); dt = (time.perf_counter() - t0) / n
pipe.ctx.prof_reset(); pipe.ctx.prof_enable(True)
step(); pipe.ctx.sync(); pipe.ctx.prof_enable(False)
groups = {}
for g in ("gauss_zyx<4,4>", "gauss_zyx<3,3>", "gauss_zyx<5,5>", "gauss_zyx<1,4>", "gauss_zyx<1,3>", "gauss_zyx<2,5>", "gauss_z", "gauss_yx", "sample", "hessian_stats", "vesselness", "vesselness_resolve", "finish", "log2d", "mask_volume", "label", "load"):
    try:
        ms, k = pipe.ctx.prof_get(g)
        if k:
            groups[g] = round(ms, 3)
    except Exception:
        pass
print(json.dumps({"shape": list(yx), "ms_per_frame": round(dt * 1e3, 3), "mpixel_s": round(yx[0] * yx[1] / dt / 1e6, 1), "labels": int(nl), "groups_ms": groups}))
