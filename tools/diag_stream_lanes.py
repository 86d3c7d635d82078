#!/usr/bin/env python3
"""Where does a second streaming lane lose its time?  Config 5 (frames of 128 x 512 x 512) through L StreamedSegmenters on L host threads, with the host-side
costs switched off one at a time: contiguous per-lane stacks (page-locked in place, no staging copy), outputs known to be zero (no zero fill).
    python tools/diag_stream_lanes.py [T]"""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import pipeline as pl
from nellie_amd.streaming import StreamedSegmenter
from nellie_amd.synthetic import ISO_01, make_volume

T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
fs = (128, 512, 512)
frames = np.stack([make_volume(fs, 4567 + t) for t in range(T)])
p = pl.FilterParams(dim_res=ISO_01)
CONFIGS = ((1, True, False), (1, True, True), (2, False, False), (2, True, False), (2, True, True), (3, True, True))
if os.environ.get("NELLIE_DIAG_ONLY"):          # "L contiguous zeroed"
    a = [int(x) for x in os.environ["NELLIE_DIAG_ONLY"].split()]
    CONFIGS = ((a[0], bool(a[1]), bool(a[2])),)
for L, contiguous, zeroed in CONFIGS:
    ins = [np.ascontiguousarray(frames[k::L]) if contiguous else frames[k::L] for k in range(L)]
    frs = [np.zeros(a.shape, np.float32) for a in ins]
    labs = [np.zeros(a.shape, np.int32) for a in ins]
    segs = [StreamedSegmenter(fs, frames.dtype, p) for _ in range(L)]

    def run_all():
        ts = [threading.Thread(target=lambda k=k: segs[k].run(ins[k], frs[k], labs[k], flush=False, outputs_zeroed=zeroed)) for k in range(L)]
        for t in ts: t.start()
        for t in ts: t.join()
    run_all()
    for s in segs:
        s.timing = {"wait_upload": 0.0, "compute": 0.0, "wait_download": 0.0, "frames": 0}
    t0 = time.perf_counter()
    run_all()
    dt = time.perf_counter() - t0
    tm = [{k: round(v / max(1, s.timing["frames"]) * 1e3, 3) if k != "frames" else v for k, v in s.timing.items()} for s in segs]
    print(json.dumps({"lanes": L, "contiguous_in": contiguous, "outputs_zeroed": zeroed, "ms_per_frame": round(dt / T * 1e3, 3), "per_lane_ms": tm}), flush=True)
    for s in segs:
        s.close()
