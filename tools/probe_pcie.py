#!/usr/bin/env python3
"""PCIe probe through the C-ABI: H2D / D2H rates of one 128x512x512 frame from page-locked (hipHostMalloc),
registered (hipHostRegister) and pageable host memory."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import hipnative, pipeline as pl
from nellie_amd.synthetic import ISO_01, make_volume

shape = (128, 512, 512)
vol = make_volume(shape, 1)
pipe = pl.FramePipeline(shape)
ctx = pipe.ctx
p = pl.FilterParams(dim_res=ISO_01)
pipe.filter(vol, p); pipe.label(pipe.frangi_threshold(), pl.min_area_pixels_of(ISO_01))
res = {}
mb = vol.nbytes / 1e6
pin_in = hipnative.PinnedArray(shape, np.float32); pin_in.array[...] = vol
pin_fr = hipnative.PinnedArray(shape, np.float32); pin_lab = hipnative.PinnedArray(shape, np.int32)
def timeit(f, n=8):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n
def h2d_pinned():
    ctx.input_load_async(0, pin_in); ctx.input_select(0)
res["h2d_pinned_GBs"] = round(mb / 1e3 / timeit(h2d_pinned), 1)
def d2h_pinned():
    ctx.outputs_stage(True); ctx.outputs_fetch_async(pin_fr, pin_lab); ctx.outputs_wait()
res["d2h_pinned_GBs"] = round(2 * mb / 1e3 / timeit(d2h_pinned), 1)
def stage_only():
    ctx.outputs_stage(True); ctx.sync()
res["stage_ms"] = round(timeit(stage_only) * 1e3, 3)
fr = np.empty(shape, np.float32); lab = np.empty(shape, np.int32)
t0 = time.perf_counter(); r1 = hipnative.RegisteredArray(fr); r2 = hipnative.RegisteredArray(lab); res["register_ms_per_268MB"] = round((time.perf_counter() - t0) * 1e3, 2)
def d2h_reg():
    ctx.outputs_stage(True); ctx.outputs_fetch_async(fr, lab); ctx.outputs_wait()
res["d2h_registered_GBs"] = round(2 * mb / 1e3 / timeit(d2h_reg), 1)
t0 = time.perf_counter(); r1.release(); r2.release(); res["unregister_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
def d2h_pageable():
    pipe.download_frangi(out=fr); pipe.download_labels(out=lab)
res["d2h_pageable_GBs"] = round(2 * mb / 1e3 / timeit(d2h_pageable, 4), 1)
def h2d_pageable():
    pipe.load_input(vol)
res["h2d_pageable_GBs"] = round(mb / 1e3 / timeit(h2d_pageable, 4), 1)
def compute():
    pipe.filter(None, p); pipe.label(pipe.frangi_threshold(), pl.min_area_pixels_of(ISO_01))
res["compute_ms"] = round(timeit(compute, 6) * 1e3, 2)
print(json.dumps(res))
pipe.close()
