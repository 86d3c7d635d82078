#!/usr/bin/env python3
"""Debug helper: Filter on a volume as one context and as two Z slabs; report where the Frangi volumes differ."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from comms import ThreadComm, ThreadGroup
from nellie_amd import pipeline as pl
from nellie_amd.sharded import ShardedFramePipeline, slab_range
from nellie_amd.synthetic import ISO_01, make_volume

SHAPE = tuple(int(a) for a in sys.argv[1:4])
vol = make_volume(SHAPE, 2345)
p = pl.FilterParams(dim_res=ISO_01)
pipe = pl.FramePipeline(SHAPE)
pipe.filter(vol, p)
fr = pipe.download_frangi()
pipe.filter(vol, p)
fr2 = pipe.download_frangi()
print("repeat equal:", np.array_equal(fr, fr2), "ndiff", int((fr != fr2).sum()))
pipe.close()
world = 2
group = ThreadGroup(world)
out = [None] * world
def worker(rank):
    o0, o1 = slab_range(SHAPE[0], world, rank)
    sp = ShardedFramePipeline(SHAPE, rank, world, lambda ctx: ThreadComm(group, rank), p)
    sp.filter(vol[o0:o1], p)
    out[rank] = (o0, o1, sp.download_frangi())
    sp.close()
ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
[t.start() for t in ts]; [t.join() for t in ts]
for o0, o1, f in out:
    d = f != fr[o0:o1]
    print("slab", o0, o1, "ndiff", int(d.sum()))
    if d.any():
        z, y, x = np.nonzero(d)
        print(" z hist", np.bincount(z // 16)[:80])
        print(" first", [(int(a) + o0, int(b), int(c), float(f[a, b, c]), float(fr[o0 + a, b, c])) for a, b, c in list(zip(z, y, x))[:8]])
