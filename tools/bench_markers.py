#!/usr/bin/env python3
"""Markers stage timing on a synthetic volume (device-resident labels and input, Filter -> Label first).
    python tools/bench_markers.py [Z Y X]          default 1024^3"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import pipeline as pl
from nellie_amd.synthetic import ISO_01, make_volume

shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1024, 1024, 1024)
vol = make_volume(shape, 2345)
pipe = pl.FramePipeline(shape)
pipe.load_input(vol)
p = pl.FilterParams(dim_res=ISO_01)
times = []
for rep in range(3):
    pipe.filter(None, p)
    n_lab = pipe.label(pipe.frangi_threshold(), pl.min_area_pixels_of(ISO_01))
    pipe.ctx.prof_reset(); pipe.ctx.prof_enable(True)
    pipe.ctx.sync(); t0 = time.perf_counter()
    n = pipe.markers(ISO_01)                          # labels and input are on the device
    pipe.ctx.sync(); times.append(time.perf_counter() - t0)
    pipe.ctx.prof_enable(False)
groups = {k: round(pipe.ctx.prof_get(k)[0], 3) for k in ("markers_begin", "markers_distance", "markers_log", "markers_peaks", "markers_nms")}
print(json.dumps({"shape": list(shape), "labels": n_lab, "markers": n, "ms_best": round(min(times) * 1e3, 2),
                  "mvoxel_s": round(np.prod(shape) / min(times) / 1e6, 1), "groups_ms_last": groups,
                  "note": "labels and input resident in HBM (Filter -> Label -> Markers on the device)"}))
pipe.close()
