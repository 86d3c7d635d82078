import sys, os, cProfile, pstats, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from nellie_amd import pipeline as pl
from nellie_amd.synthetic import ISO_01, make_volume
shape = (128, 512, 512)
pipe = pl.FramePipeline(shape); pipe.load_input(make_volume(shape, 1234))
p = pl.FilterParams(dim_res=ISO_01); ma = pl.min_area_pixels_of(ISO_01)
def step():
    pipe.filter(None, p); return pipe.label(pipe.frangi_threshold(), ma)
for _ in range(3): step()
pipe.ctx.sync(); t0 = time.perf_counter()
for _ in range(20): step()
pipe.ctx.sync(); print("ms/step", (time.perf_counter() - t0) / 20 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
