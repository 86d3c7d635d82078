#!/usr/bin/env python3
"""Debug helper: per-scale bracket / exact threshold / hit of the one-pass vesselness, one-pass vs two-pass results."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import pipeline as pl
from nellie_amd.synthetic import ISO_01, make_volume

shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 128, 128)
vol = make_volume(shape, 1234)
p = pl.FilterParams(dim_res=ISO_01)
res = {}
for mode in (False, True):
    pipe = pl.FramePipeline(shape)
    pipe.one_pass = mode
    orig = pipe._fsq_bracket
    def wrapped(strides, division, orig=orig, pipe=pipe):
        b = orig(strides, division)
        pipe._last_bracket = b
        return b
    pipe._fsq_bracket = wrapped
    orig_res = pipe.ctx.vesselness_resolve
    def wres(*a, pipe=pipe, orig_res=orig_res):
        r = orig_res(*a)
        print("   bracket", pipe._last_bracket, "exact", pipe.ctx.info("last_fsq_min"), "hit", r)
        return r
    pipe.ctx.vesselness_resolve = wres
    pipe.filter(vol, p)
    fr = pipe.download_frangi()
    print("one_pass" if mode else "two_pass", [(s.mask_count, s.one_pass) for s in pipe.trace.scales], pipe.trace.n_positive)
    res[mode] = fr
    pipe.close()
d = res[False] != res[True]
print("differing voxels:", int(d.sum()))
