#!/usr/bin/env python3
"""CRCs of the Frangi frame and the labels of N different frames, one line per frame: run it under two builds of the library
(NELLIE_HIP_LIB=...) and diff the outputs.  Used for the trace test of the walk (a build with -DHM_TRACE_ALL queues and solves
every masked voxel): the frames must be identical.
    tools/soak_crc.py [Z Y X] [N]"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import pipeline as pl
from nellie_amd.synthetic import ANISO_03, ISO_01, make_volume

shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 160, 200)
n = int(sys.argv[4]) if len(sys.argv) >= 5 else 200
pipe = pl.FramePipeline(shape)
rng = np.random.default_rng(7)
for k in range(n):
    dr = ISO_01 if k % 3 else ANISO_03
    vol = make_volume(shape, 20_000 + k)
    if k % 5 == 0:
        vol = (vol * np.float32(rng.uniform(0.01, 50.0))).astype(np.float32)
    if k % 4 == 1:                                   # pure noise and textures: Hessians of every signature
        vol = rng.normal(100, rng.uniform(0.5, 20.0), shape).astype(np.float32)
    if k % 4 == 3:
        z, y, x = np.mgrid[:shape[0], :shape[1], :shape[2]]
        f = rng.uniform(0.1, 0.9, 3)
        vol = (vol + 30.0 * np.sin(f[0] * z) * np.sin(f[1] * y) * np.sin(f[2] * x)).astype(np.float32)
    pipe.filter(vol, pl.FilterParams(dim_res=dr))
    thr = pipe.frangi_threshold()
    nl = pipe.label(thr, pl.min_area_pixels_of(dr))
    fr = pipe.download_frangi()
    print(k, zlib.crc32(fr.tobytes()), zlib.crc32(pipe.download_labels().tobytes()), nl, int((fr > 0).sum()), pipe.chain_fallbacks, flush=True)
