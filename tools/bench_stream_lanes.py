#!/usr/bin/env python3
"""BASELINE config 5 streamed by L lanes sharing ONE GPU (each lane = a StreamedSegmenter with its own context and host thread,
lane k takes frames k, k + L, ...): does a second lane fill the GPU's idle time between one lane's waits?
    python tools/bench_stream_lanes.py [T] [lanes ...]        default 64 frames of 128 x 512 x 512, lanes 1 2 3"""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import pipeline as pl
from nellie_amd.streaming import StreamedSegmenter
from nellie_amd.synthetic import ISO_01, make_volume

# lanes are Python threads: a lane coming back from a C call waits for the GIL up to one switch interval (5 ms by default -- longer than a frame)
if os.environ.get("NELLIE_SWITCH_INTERVAL"):
    sys.setswitchinterval(float(os.environ["NELLIE_SWITCH_INTERVAL"]))
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lanes_list = [int(a) for a in sys.argv[2:]] or [1, 2, 3]
fs = (128, 512, 512)
frames = np.stack([make_volume(fs, 4567 + t) for t in range(T)])
p = pl.FilterParams(dim_res=ISO_01)
ref = None
out = {"stack": [T] + list(fs)}
for L in lanes_list:
    fr, lab = np.empty(frames.shape, np.float32), np.empty(frames.shape, np.int32)
    segs = [StreamedSegmenter(fs, frames.dtype, p, lanes=1) for _ in range(L)]     # (round 6: the streamer has lanes of its own -- tools/diag_stream_lanes.py)

    def run_all():
        errs = []
        def work(k):
            try:
                segs[k].run(frames[k::L], fr[k::L], lab[k::L], flush=False)
            except BaseException as exc:  # noqa: BLE001
                errs.append(exc)
        ts = [threading.Thread(target=work, args=(k,)) for k in range(L)]
        for t in ts: t.start()
        for t in ts: t.join()
        if errs:
            raise errs[0]
    run_all()                                   # first touch of the output pages
    t0 = time.perf_counter()
    run_all()
    dt = time.perf_counter() - t0
    for s in segs:
        s.close()
    if ref is None:
        ref = (fr.copy(), lab.copy())
    out[f"lanes_{L}"] = {"ms_per_frame": round(dt / T * 1e3, 3), "mvoxel_s": round(frames.size / dt / 1e6, 1),
                         "identical_to_one_lane": bool(np.array_equal(fr, ref[0]) and np.array_equal(lab, ref[1]))}
print(json.dumps(out))
