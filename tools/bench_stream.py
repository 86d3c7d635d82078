#!/usr/bin/env python3
"""BASELINE config 5: a 3-D+T stack streamed frame by frame (PCIe-inclusive: host arrays in, host arrays out).
    python tools/bench_stream.py [T Z Y X]        default 16 x 128 x 512 x 512 float32
Prints Mvoxel/s for (a) blocking per-frame transfers and (b) the overlapped streamer (nellie_amd/streaming.py)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from nellie_amd import pipeline as pl
from nellie_amd.streaming import StreamedSegmenter
from nellie_amd.synthetic import ISO_01, make_volume

T, Z, Y, X = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (16, 128, 512, 512)
frames = np.stack([make_volume((Z, Y, X), 4567 + t) for t in range(T)])
p = pl.FilterParams(dim_res=ISO_01)
min_area = pl.min_area_pixels_of(ISO_01)
fr = np.empty(frames.shape, np.float32)
lab = np.empty(frames.shape, np.int32)

pipe = pl.FramePipeline((Z, Y, X))
pipe.filter(frames[0], p)                      # warm-up
t0 = time.perf_counter()
for t in range(T):
    pipe.filter(frames[t], p)
    pipe.label(pipe.frangi_threshold(), min_area)
    pipe.download_frangi(out=fr[t])
    pipe.download_labels(out=lab[t])
serial = time.perf_counter() - t0
pipe.load_input(frames[0])
t0 = time.perf_counter()
for t in range(4):
    pipe.filter(None, p)
    pipe.label(pipe.frangi_threshold(), min_area)
compute_only = (time.perf_counter() - t0) / 4
pipe.close()

fr2 = np.empty_like(fr)
lab2 = np.empty_like(lab)
seg = StreamedSegmenter((Z, Y, X), frames.dtype, p)
seg.run(frames[:2], fr2[:2], lab2[:2], flush=False)        # warm-up
t0 = time.perf_counter()
seg.run(frames, fr2, lab2, flush=False)                    # output pages are touched for the first time here
streamed_cold = time.perf_counter() - t0
for k in seg.timing:
    seg.timing[k] = 0
t0 = time.perf_counter()
seg.run(frames, fr2, lab2, flush=False)                    # steady state: every host page resident
streamed = time.perf_counter() - t0
tm = {k: round(v / T * 1e3, 2) for k, v in seg.timing.items() if k != "frames"}
seg.close()
n = float(frames.size)
print(json.dumps({"stack": [T, Z, Y, X], "blocking_mvoxel_s": round(n / serial / 1e6, 1),
                  "streamed_mvoxel_s": round(n / streamed / 1e6, 1), "ms_per_frame_blocking": round(serial / T * 1e3, 2),
                  "ms_per_frame_streamed": round(streamed / T * 1e3, 2), "ms_per_frame_streamed_first_touch": round(streamed_cold / T * 1e3, 2), "ms_per_frame_compute_only": round(compute_only * 1e3, 2),
                  "main_thread_ms_per_frame": tm, "identical": bool(np.array_equal(fr, fr2) and np.array_equal(lab, lab2))}))
