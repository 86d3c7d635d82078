#!/usr/bin/env python3
"""cProfile of the host side of N Filter + Label passes over a resident frame (where does Python spend a frame's host time?).
    tools/prof_py_frame.py [Z Y X] [N]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nellie_amd import pipeline as pl
from nellie_amd.synthetic import ISO_01, make_volume

shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (128, 512, 512)
n = int(sys.argv[4]) if len(sys.argv) >= 5 else 200
pipe = pl.FramePipeline(shape)
pipe.load_input(make_volume(shape, 2345))
p = pl.FilterParams(dim_res=ISO_01)
ma = pl.min_area_pixels_of(ISO_01)


def step():
    pipe.filter(None, p)
    return pipe.label(pipe.frangi_threshold(), ma)


for _ in range(5):
    step()
pipe.ctx.sync(); t0 = time.perf_counter()
for _ in range(n):
    step()
pipe.ctx.sync(); wall = (time.perf_counter() - t0) / n * 1e3
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
pr.disable()
print(f"{shape}: {wall:.3f} ms/frame without the profiler")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
