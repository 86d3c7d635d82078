#!/usr/bin/env python3
"""Where one frame's wall time goes on the host side of the single-GPU path: every library call of the last of `reps` steps with
its start / end, the time inside the calls, the time between them, the longest of both.  tools/prof_calls.py Z Y X [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import pipeline as pl
from nellie_amd.synthetic import ISO_01, make_volume

shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 512, 512)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
pipe = pl.FramePipeline(shape)
pipe.load_input(make_volume(shape, 1234))
p = pl.FilterParams(dim_res=ISO_01)
ma = pl.min_area_pixels_of(ISO_01)
log = []
lib_call = pipe.ctx.lib.call
def logged(name, *a):
    t0 = time.perf_counter(); r = lib_call(name, *a); log.append((name, t0, time.perf_counter())); return r
pipe.ctx.lib.call = logged
for _ in range(reps):
    del log[:]
    t0 = time.perf_counter()
    pipe.filter(None, p)
    n = pipe.label(pipe.frangi_threshold(), ma)
    pipe.ctx.sync()
    wall = time.perf_counter() - t0
inside = sum(e - s for _, s, e in log)
print(f"wall {wall * 1e3:.3f} ms, {len(log)} calls, inside the library {inside * 1e3:.3f} ms, between calls {(wall - inside) * 1e3:.3f} ms")
agg = {}
for name, s, e in log:
    a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += e - s
for name, (k, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {name:34s} {k:3d} calls {t * 1e3:7.3f} ms")
gaps = sorted(((log[i + 1][1] - log[i][2]) * 1e3, log[i][0], log[i + 1][0]) for i in range(len(log) - 1))[::-1]
for g in gaps[:10]:
    print("  %.3f ms between %s and %s" % g)
pipe.close()
