#!/usr/bin/env python3
"""Where the Z-slab Label spends its time on one rank (world 1, RCCL communicator of one rank or a null communicator):
wall time per host-level call of ShardedFramePipeline.label on a slab of BASELINE config 4's per-GPU size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import pipeline as pl
from nellie_amd import sharded
from nellie_amd.synthetic import ISO_01, make_volume

shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 2048, 2048)


class Null:
    world, rank = 1, 0
    def exchange_halo(self, *a): pass
    def exchange_bits(self, *a): pass
    def allreduce(self, arr, op): return arr
    def allgather(self, arr): return arr
    def allgather_list(self, arr): return [arr]


p = pl.FilterParams(dim_res=ISO_01)
ma = pl.min_area_pixels_of(ISO_01)
pipe = sharded.ShardedFramePipeline(shape, 0, 1, lambda ctx: Null(), p)
pipe.load_input(make_volume(shape, 3456))
pipe.filter(None, p)
thr = pipe.frangi_threshold()
ctx = pipe.ctx
acc = {}
for name in [n for n in dir(ctx) if n.startswith("slab_")]:
    f = getattr(ctx, name)
    def wrap(*a, _f=f, _n=name, **k):
        key = _n + (str(a[0]) if _n == "slab_components" else "")
        t0 = time.perf_counter(); r = _f(*a, **k); ctx.sync(); acc[key] = acc.get(key, 0.0) + time.perf_counter() - t0; return r
    setattr(ctx, name, wrap)
orig_join = sharded.join_slab_tables
def join(t):
    t0 = time.perf_counter(); r = orig_join(t); acc["host join"] = acc.get("host join", 0.0) + time.perf_counter() - t0; return r
sharded.join_slab_tables = join
for rep in range(3):
    acc.clear(); ctx.sync(); t0 = time.perf_counter()
    n = pipe.label(thr, ma); ctx.sync()
    tot = time.perf_counter() - t0
print("labels", n, "total ms %.2f" % (tot * 1e3), {k: round(v * 1e3, 2) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])})
t0 = time.perf_counter(); pipe.filter(None, p); ctx.sync(); print("filter ms %.2f" % ((time.perf_counter() - t0) * 1e3))
pipe.close()
single = pl.FramePipeline(shape); single.load_input(make_volume(shape, 3456)); single.filter(None, p); thr = single.frangi_threshold()
for rep in range(2):
    single.filter(None, p); single.ctx.sync(); t0 = time.perf_counter(); single.label(thr, ma); single.ctx.sync(); t1 = time.perf_counter() - t0
print("single-context label ms %.2f" % (t1 * 1e3))
single.close()
