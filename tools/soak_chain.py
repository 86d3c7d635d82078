#!/usr/bin/env python3
"""Soak: N different frames through the default path (device chain, percentile threshold selected on the device) and through the
synchronous path with the percentile on the host (round 3's) on a second context; Frangi frames, labels and traces must agree frame
by frame; counts the frames the chain handed back.
    tools/soak_chain.py [Z Y X] [N]"""
import json, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import pipeline as pl
from nellie_amd.synthetic import ANISO_03, ISO_01, make_volume

shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 160, 200)
n = int(sys.argv[4]) if len(sys.argv) >= 5 else 300
a, b = pl.FramePipeline(shape), pl.FramePipeline(shape)
b._device_chain = False
b._device_tail = False
bad, t0 = [], time.time()
rng = np.random.default_rng(1)
for k in range(n):
    dr = ISO_01 if k % 3 else ANISO_03
    vol = make_volume(shape, 10_000 + k)
    if k % 7 == 0:
        vol = (vol * np.float32(rng.uniform(0.01, 50.0))).astype(np.float32)          # other intensity ranges
    if k % 11 == 0:
        vol[: shape[0] // 2] = vol[shape[0] // 2:].mean()                            # a flat half
    p = pl.FilterParams(dim_res=dr)
    ma = pl.min_area_pixels_of(dr)
    out = []
    for pipe in (a, b):
        try:
            pipe.filter(vol, p)
            thr = pipe.frangi_threshold()
            nl = pipe.label(thr, ma)
            tr = [(s.gamma, s.max_abs, s.frob_thr, s.mask_count, s.skipped) for s in pipe.trace.scales]
            out.append((zlib.crc32(pipe.download_frangi().tobytes()), zlib.crc32(pipe.download_labels().tobytes()), nl, thr, tr,
                        pipe.trace.percentile_thr, pipe.trace.n_positive))
        except ValueError as exc:
            out.append(("raised", str(exc)[:60]))
    if out[0] != out[1]:
        bad.append((k, out[0][:4], out[1][:4]))
print(json.dumps({"shape": list(shape), "frames": n, "mismatches": len(bad), "first": [str(x) for x in bad[:3]],
                  "chain_fallbacks": a.chain_fallbacks, "seconds": round(time.time() - t0, 1)}))
