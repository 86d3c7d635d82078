#!/usr/bin/env python3
"""One-pass vesselness hit rate over the golden Filter cases (diagnostics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import FILTER_CASES, load_golden
from nellie_amd import pipeline as pl
for name in FILTER_CASES:
    g = load_golden(name)
    if "error_type" in g:
        continue
    vol = g["input"]
    p = pl.FilterParams(dim_res=g["dim_res_dict"], **g["kwargs"])
    for margin in (1e-3, 1e-2):
        pipe = pl.FramePipeline(vol.shape)
        pipe.one_pass_margin = margin
        pipe.compute_vesselness(vol, p)
        print(f"{name:28s} margin {margin:g}: hits {[int(s.one_pass) for s in pipe.trace.scales]} skipped {[int(s.skipped) for s in pipe.trace.scales]}")
        pipe.close()
