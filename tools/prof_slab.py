#!/usr/bin/env python3
"""One rank's share of a Z-slab run (world 1 over a real one-rank RCCL communicator): wall time per step, HIP-event time per
kernel group and the wall time of every host-level context call, for a slab of BASELINE config 4's per-GPU size.
tools/prof_slab.py [Z Y X] [reps]
  NELLIE_PROF_CALLS=0  no per-call timers (for rocprofv3 runs)      NELLIE_PROF_PY=1  cProfile of the timed steps
  NELLIE_PROF_LOG=1    (start, end) of every library call of step NELLIE_PROF_STEP (default: the last): gaps and longest calls"""
import os, sys, time
if os.environ.get("NELLIE_PROF_TORCH_FIRST") == "1":      # what bench.py's multi-process path does: torch (and its HIP runtime) before the library
    import torch  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import hipnative, pipeline as pl, sharded
from nellie_amd.synthetic import ISO_01, make_volume

shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 2048, 2048)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
p = pl.FilterParams(dim_res=ISO_01)
ma = pl.min_area_pixels_of(ISO_01)
_pre = os.environ.get("NELLIE_PROF_PRELUDE", "")     # what came before in the process matters: 1 = what bench.py's
if _pre:                                              # slab run used to do first; a / b / c / d = its parts
    small = (48, 192, 256)
    if _pre in ("1", "a", "a1", "a3"):
        u1, u2 = hipnative.comm_unique_id(), hipnative.comm_unique_id()
        q = sharded.ShardedFramePipeline(small, 0, 1, lambda ctx: sharded.RcclComm(ctx, 1, 0, u1, uid2=u2), p)
        if _pre != "a3":                                  # a3: communicators created and destroyed, never used
            q.load_input(make_volume(small, 4242)); q.filter(None, p); q.label(q.frangi_threshold(), ma); q.download_frangi()
        if _pre != "a1":                                  # a1: used, kept alive
            q.close()
        else:
            _keep = q
    if _pre in ("1", "b"):
        q = pl.FramePipeline(small); q.filter(make_volume(small, 4242), p); q.label(q.frangi_threshold(), ma); q.close()
    if _pre == "c":
        q = pl.FramePipeline(small); q.close()
    if _pre == "d":
        q = pl.FramePipeline(shape); q.close()
uid, uid2 = hipnative.comm_unique_id(), hipnative.comm_unique_id()
pipe = sharded.ShardedFramePipeline(shape, 0, 1, lambda ctx: sharded.RcclComm(ctx, 1, 0, uid, uid2=uid2), p)
pipe.load_input(make_volume(shape, 3456))
ctx = pipe.ctx

def step():
    pipe.filter(None, p)
    return pipe.label(pipe.frangi_threshold(), ma)

ctx.prof_enable(True)
step()
acc, cnt = {}, {}
log = []
if os.environ.get("NELLIE_PROF_LOG") == "1":
    lib_call = ctx.lib.call
    def logged(name, *a):
        t0 = time.perf_counter(); r = lib_call(name, *a); log.append((name, t0, time.perf_counter())); return r
    ctx.lib.call = logged
if os.environ.get("NELLIE_PROF_CALLS", "1") == "1":
    orig = ctx._call
    def timed(name, *a):
        t0 = time.perf_counter(); r = orig(name, *a); dt = time.perf_counter() - t0
        acc[name] = acc.get(name, 0.0) + dt; cnt[name] = cnt.get(name, 0) + 1
        return r
    ctx._call = timed
ctx.prof_reset(); ctx.prof_enable(True); ctx.sync()
prof = None
if os.environ.get("NELLIE_PROF_PY") == "1":
    import cProfile
    prof = cProfile.Profile()
    prof.enable()
t0 = time.perf_counter()
per_step = []
for _ in range(reps):
    t1 = time.perf_counter()
    n = step()
    per_step.append(round((time.perf_counter() - t1) * 1e3, 2))
ctx.sync()
wall = (time.perf_counter() - t0) / reps * 1e3
print("per step (host return, ms):", per_step)
if prof is not None:
    import pstats
    prof.disable()
    pstats.Stats(prof).sort_stats("tottime").print_stats(25)
ctx.prof_enable(False)
groups = {}
for g in ("gauss_zyx<4,4>", "gauss_zyx<3,3>", "gauss_zyx<5,5>", "gauss_zyx<1,4>", "gauss_zyx<1,3>", "gauss_zyx<2,5>", "gauss_z", "gauss_yx", "sample", "hessian_stats", "vesselness", "vesselness_resolve", "mask_volume", "label", "halo"):
    ms, k = ctx.prof_get(g)
    if k:
        groups[g] = round(ms / reps, 3)
print(f"labels {n}  wall {wall:.2f} ms/step  kernels {sum(groups.values()):.2f} ms  {groups}")
tot = sum(acc.values()) / reps * 1e3
print(f"host-level calls: {tot:.2f} ms/step inside the library, {wall - tot:.2f} ms/step in Python between them")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:40]:
    print(f"  {k:32s} {v / reps * 1e3:7.3f} ms/step  {cnt[k] / reps:5.1f} calls")
if log:
    begins = [i for i, e in enumerate(log) if e[0] == "nl_filter_begin"] + [len(log)]
    which = int(os.environ.get("NELLIE_PROF_STEP", "-1")) % (len(begins) - 1)
    ev = log[begins[which]:begins[which + 1]]
    print("step", which, ": calls", len(ev), "span %.2f ms" % ((ev[-1][2] - ev[0][1]) * 1e3))
    gaps = sorted(((ev[i + 1][1] - ev[i][2]) * 1e3, ev[i][0], ev[i + 1][0]) for i in range(len(ev) - 1))[::-1]
    print("time between calls: %.2f ms" % sum(g[0] for g in gaps))
    for g in gaps[:25]:
        print("  %.3f ms between %s and %s" % g)
    worst = max(range(len(ev)), key=lambda i: ev[i][2] - ev[i][1])
    print("longest call is #%d; calls around it:" % worst)
    for i in range(max(0, worst - 12), min(len(ev), worst + 4)):
        print("   #%d %-28s %.3f ms" % (i, ev[i][0], (ev[i][2] - ev[i][1]) * 1e3))
    longest = sorted(((e[2] - e[1]) * 1e3, e[0]) for e in ev)[::-1]
    for d, n in longest[:15]:
        print("  call %.3f ms %s" % (d, n))
pipe.close()
