#!/usr/bin/env python3
"""The frame streamer with lanes, opened, run over a stack (run_streamed's pattern: one StreamedSegmenter per file) and closed N times over -- and one streamer
reused for N stacks: host RSS, threads and device memory must level off.    tools/leak_check_stream.py ROUNDS [LANES]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import hipnative, pipeline as pl
from nellie_amd.streaming import StreamedSegmenter
from nellie_amd.synthetic import ISO_01, make_volume


def rss_mb():
    with open("/proc/self/status") as f:
        for line in f:
            if line.startswith("VmRSS"):
                return int(line.split()[1]) / 1024.0


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lib = hipnative.load()
fs = (48, 128, 160)
frames = np.stack([make_volume(fs, 300 + t) for t in range(7)])
p = pl.FilterParams(dim_res=ISO_01)
fr, lab = np.zeros(frames.shape, np.float32), np.zeros(frames.shape, np.int32)
ref = None
rows = []
t0 = time.time()
keep = StreamedSegmenter(fs, frames.dtype, p, lanes=lanes)
for r in range(rounds):
    seg = StreamedSegmenter(fs, frames.dtype, p, lanes=lanes)
    seg.run(frames, fr, lab, flush=False)
    seg.close()
    if ref is None:
        ref = (fr.copy(), lab.copy())
    assert np.array_equal(fr, ref[0]) and np.array_equal(lab, ref[1])
    keep.run(frames, fr, lab, flush=False)
    assert np.array_equal(fr, ref[0]) and np.array_equal(lab, ref[1])
    if r in (0, 1, 2) or (r + 1) % max(1, rounds // 10) == 0:
        free, total = lib.device_mem_info(0)
        rows.append((r + 1, round(rss_mb(), 1), round((total - free) / 2 ** 20, 1), threading.active_count()))
        print(f"round {r + 1:5d}  host RSS {rows[-1][1]:9.1f} MiB   device memory in use {rows[-1][2]:9.1f} MiB   threads {rows[-1][3]}   ({time.time() - t0:.0f} s)", flush=True)
keep.close()
mid, last = rows[len(rows) // 2], rows[-1]
print(f"second half of the run: host RSS {last[1] - mid[1]:+.1f} MiB, device memory {last[2] - mid[2]:+.1f} MiB, threads {last[3] - mid[3]:+d}; "
      f"{rounds} rounds x 2 stacks of {len(frames)} frames, {lanes} lanes, identical outputs every time")
