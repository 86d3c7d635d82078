#!/usr/bin/env python3
"""Host-side probe for the streamer: page-locking cost of touched / untouched arrays, threaded copy rates."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from nellie_amd import hipnative
hipnative.load()
res = {}
n = 1 << 30
a = np.empty(n, np.uint8)
t0 = time.perf_counter(); r = hipnative.RegisteredArray(a); res["register_untouched_ms_per_GiB"] = round((time.perf_counter() - t0) * 1e3, 1); r.release()
t0 = time.perf_counter(); r = hipnative.RegisteredArray(a); res["register_touched_ms_per_GiB"] = round((time.perf_counter() - t0) * 1e3, 1); r.release()
b = np.empty(n, np.uint8)
t0 = time.perf_counter(); b[...] = 1; res["first_touch_ms_per_GiB"] = round((time.perf_counter() - t0) * 1e3, 1)
pin = hipnative.PinnedArray((n,), np.uint8)
pin.array[...] = 2
t0 = time.perf_counter(); np.copyto(b, pin.array); res["copy_1thread_GBs"] = round(n / 1e9 / (time.perf_counter() - t0), 1)
for k in (4, 8, 16, 32):
    ex = ThreadPoolExecutor(k)
    step = n // k
    def cp(i): np.copyto(b[i * step:(i + 1) * step], pin.array[i * step:(i + 1) * step])
    list(ex.map(cp, range(k)))
    t0 = time.perf_counter(); list(ex.map(cp, range(k))); res[f"copy_{k}threads_GBs"] = round(n / 1e9 / (time.perf_counter() - t0), 1)
    ex.shutdown()
print(json.dumps(res))
