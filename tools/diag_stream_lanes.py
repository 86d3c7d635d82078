#!/usr/bin/env python3
"""Config 5 (frames of 128 x 512 x 512) through ONE StreamedSegmenter with 1 / 2 / 3 lanes (contexts of the GPU fed by one upload thread), float32 and
uint16 stacks, outputs to be zero-filled or known to be zero.  NELLIE_DIAG_ONLY="L zeroed dtype" runs one configuration (tools/trace_lanes.sh).
    python tools/diag_stream_lanes.py [T]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import pipeline as pl
from nellie_amd.streaming import StreamedSegmenter
from nellie_amd.synthetic import ISO_01, make_volume

T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
fs = (128, 512, 512)
frames32 = np.stack([make_volume(fs, 4567 + t) for t in range(T)])
frames16 = np.clip(frames32 * 64.0, 0, 65535).astype(np.uint16)
p = pl.FilterParams(dim_res=ISO_01)
CONFIGS = ((1, False, "f32"), (2, False, "f32"), (3, False, "f32"), (2, True, "f32"), (1, False, "u16"), (2, False, "u16"), (3, False, "u16"))
if os.environ.get("NELLIE_DIAG_ONLY"):
    a = os.environ["NELLIE_DIAG_ONLY"].split()
    CONFIGS = ((int(a[0]), bool(int(a[1])), a[2]),)
ref = {}
for L, zeroed, dt in CONFIGS:
    frames = frames32 if dt == "f32" else frames16
    fr, lab = np.zeros(frames.shape, np.float32), np.zeros(frames.shape, np.int32)
    seg = StreamedSegmenter(fs, frames.dtype, p, lanes=L)
    seg.run(frames, fr, lab, flush=False, outputs_zeroed=zeroed)
    seg.timing = {"wait_upload": 0.0, "compute": 0.0, "wait_download": 0.0, "frames": 0}
    t0 = time.perf_counter()
    seg.run(frames, fr, lab, flush=False, outputs_zeroed=zeroed)
    dt_s = time.perf_counter() - t0
    tm = {k: round(v / max(1, seg.timing["frames"]) * 1e3, 3) if k != "frames" else v for k, v in seg.timing.items()}
    seg.close()
    key = dt
    if key not in ref:
        ref[key] = (fr.copy(), lab.copy())
    same = bool(np.array_equal(fr, ref[key][0]) and np.array_equal(lab, ref[key][1]))
    print(json.dumps({"lanes": L, "dtype": dt, "outputs_zeroed": zeroed, "ms_per_frame": round(dt_s / T * 1e3, 3), "per_frame_ms_in_compute_threads": tm,
                      "identical_to_first_of_dtype": same}), flush=True)
