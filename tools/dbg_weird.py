#!/usr/bin/env python3
"""HIP path vs oracle on awkward shapes and parameters (diagnostics)."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import pipeline as pl
from nellie_amd.synthetic import make_volume
from oracle import nellie_oracle as orc
cases = [((3, 5, 7), {"X": 0.1, "Y": 0.1, "Z": 0.1, "T": 1.0}, {}),
         ((2, 2, 2), {"X": 0.1, "Y": 0.1, "Z": 0.1, "T": 1.0}, {}),
         ((5, 40, 3), {"X": 0.1, "Y": 0.1, "Z": 0.1, "T": 1.0}, {}),
         ((9, 9, 200), {"X": 0.1, "Y": 0.1, "Z": 0.5, "T": 1.0}, {}),
         ((30, 60, 60), {"X": 0.05, "Y": 0.05, "Z": 0.05, "T": 1.0}, {}),                      # sigmas x2: radii > 8
         ((20, 50, 50), {"X": 0.1, "Y": 0.1, "Z": 0.1, "T": 1.0}, {"max_radius_um": 3.0}),     # big radii, generic kernels
         ((16, 33, 65), {"X": 0.2, "Y": 0.1, "Z": 0.3, "T": 1.0}, {})]                            # X != Y spacing
for shape, dr, kw in cases:
    vol = make_volume(shape, 5)
    try:
        ref = orc.filter_frame(vol, dr, **({"sigmas": orc.default_sigmas(dr, **kw)} if kw else {}))
        ref_err = None
    except Exception as e:
        ref, ref_err = None, f"{type(e).__name__}: {e}"
    try:
        pipe = pl.FramePipeline(shape)
        pipe.filter(vol, pl.FilterParams(dim_res=dr, **kw))
        got = pipe.download_frangi(); pipe.close(); got_err = None
    except Exception as e:
        got, got_err = None, f"{type(e).__name__}: {e}"
    if ref is None or got is None:
        print(shape, kw, "oracle:", ref_err, "| hip:", got_err)
        continue
    tol = 1e-4 * np.abs(ref) + 1e-6 * np.abs(ref).max()
    bad = int((np.abs(got - ref) > tol).sum())
    print(shape, kw, "sigmas", [round(s, 2) for s in pl.FilterParams(dim_res=dr, **kw).resolved_sigmas()], "bad", bad, "nonzero", int((ref > 0).sum()), int((got > 0).sum()))
