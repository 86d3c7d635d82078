#!/usr/bin/env python3
"""Config 5 streamed (64 frames of 128 x 512 x 512), lanes 1 .. 4, float32 and uint16 stacks, every configuration REPS times in alternation on one box."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import pipeline as pl
from nellie_amd.streaming import StreamedSegmenter
from nellie_amd.synthetic import ISO_01, make_volume
T, REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 3
fs = (128, 512, 512)
f32 = np.stack([make_volume(fs, 4567 + t) for t in range(T)])
u16 = np.clip(f32 * 64.0, 0, 65535).astype(np.uint16)
p = pl.FilterParams(dim_res=ISO_01)
res = {}
for rep in range(REPS):
    for dt, frames in (("f32", f32), ("u16", u16)):
        for L in (1, 2, 3, 4):
            fr, lab = np.zeros(frames.shape, np.float32), np.zeros(frames.shape, np.int32)
            seg = StreamedSegmenter(fs, frames.dtype, p, lanes=L)
            seg.run(frames[:8], fr[:8], lab[:8], flush=False)
            t0 = time.perf_counter()
            seg.run(frames, fr, lab, flush=False)
            res.setdefault(f"{dt}_lanes{L}", []).append(round((time.perf_counter() - t0) / T * 1e3, 3))
            seg.close()
print(json.dumps(res))
