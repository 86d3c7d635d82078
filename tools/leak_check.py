#!/usr/bin/env python3
"""Contexts created, used for a whole frame (Filter, Label, Markers; one context and 3 Z slabs through the stage classes) and
closed, N times over: host RSS and free device memory must level off (a long-lived service opens contexts per file).
  tools/leak_check.py ROUNDS"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from fakes import ArrayImInfo
from nellie_amd import hipnative
from nellie_amd.segmentation.filtering import Filter
from nellie_amd.segmentation.labelling import Label
from nellie_amd.segmentation.mocap_marking import Markers
from nellie_amd.synthetic import ISO_01, make_volume


def rss_mb():
    with open("/proc/self/status") as f:
        for line in f:
            if line.startswith("VmRSS"):
                return int(line.split()[1]) / 1024.0


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
lib = hipnative.load()
vols = np.stack([make_volume((90, 64, 96), 5 + t) for t in range(2)])
fine = {"X": 0.065, "Y": 0.065, "Z": 0.2, "T": 1.0}            # Markers' any-radius path and its scratch volume
rows = []
t0 = time.time()
for r in range(rounds):
    for devices, dr in ((None, ISO_01), ([0, 0, 0], ISO_01), (None, fine)):
        im = ArrayImInfo(vols, dr)
        Filter(im, devices=devices).run(); Label(im, devices=devices).run(); Markers(im, devices=devices).run()
    if r in (0, 1, 2) or (r + 1) % max(1, rounds // 10) == 0:
        free, total = lib.device_mem_info(0)
        rows.append((r + 1, round(rss_mb(), 1), round((total - free) / 2 ** 20, 1)))
        print(f"round {r + 1:5d}  host RSS {rows[-1][1]:9.1f} MiB   device memory in use {rows[-1][2]:9.1f} MiB   ({time.time() - t0:.0f} s)", flush=True)
mid = rows[len(rows) // 2]
last = rows[-1]
grow_host, grow_dev = last[1] - mid[1], last[2] - mid[2]
print(f"second half of the run: host RSS {grow_host:+.1f} MiB, device {grow_dev:+.1f} MiB over {last[0] - mid[0]} rounds ({3 * (last[0] - mid[0])} stage runs of 3 stages, 2 frames each)")
sys.exit(0 if grow_host < 64 and grow_dev < 64 else 1)
