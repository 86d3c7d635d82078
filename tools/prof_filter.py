#!/usr/bin/env python3
"""Small driver for rocprofv3 runs: one Filter(+Label) pass on a synthetic volume (no CPU baseline, no bench extras)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import pipeline as pl
from nellie_amd.synthetic import ISO_01, make_volume

shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (256, 512, 512)
reps = int(sys.argv[4]) if len(sys.argv) >= 5 else 1
vol = make_volume(shape, 1234)
pipe = pl.FramePipeline(shape)
pipe.load_input(vol)
p = pl.FilterParams(dim_res=ISO_01)
for _ in range(reps):
    pipe.filter(None, p)
    n = pipe.label(pipe.frangi_threshold(), pl.min_area_pixels_of(ISO_01))
pipe.ctx.sync()
print("labels", n, "positives", pipe.trace.n_positive)
pipe.close()
