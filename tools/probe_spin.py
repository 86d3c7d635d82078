#!/usr/bin/env python3
"""Does an open RCCL communicator cost the host thread anything?  Times a tiny numpy function before / after comm_init."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import hipnative, pipeline as pl

def probe(tag):
    ts = []
    for _ in range(200):
        t = time.perf_counter(); pl.gaussian_weights(1.5); ts.append(time.perf_counter() - t)
    ts = np.array(ts) * 1e6
    print(f"{tag:28s} median {np.median(ts):8.1f} us  mean {ts.mean():8.1f} us  max {ts.max():8.1f} us", flush=True)

print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "threads", len(os.listdir("/proc/self/task")))
probe("fresh")
ctx = hipnative.Context((16, 64, 64))
probe("context")
print("threads", len(os.listdir("/proc/self/task")))
ctx.comm_init(1, 0, hipnative.comm_unique_id())
probe("one communicator")
print("threads", len(os.listdir("/proc/self/task")))
ctx.comm_init2(1, 0, hipnative.comm_unique_id())
probe("two communicators")
print("threads", len(os.listdir("/proc/self/task")))
a = ctx.allreduce(np.array([1], np.int64), "sum")
probe("after an allreduce")
time.sleep(0.5)
probe("after 0.5 s")
for t in sorted(os.listdir("/proc/self/task")):
    try:
        st = open(f"/proc/self/task/{t}/stat").read().split()
        print(t, open(f"/proc/self/task/{t}/comm").read().strip(), "state", st[2], "utime", st[13], "stime", st[14])
    except Exception:
        pass
ctx.close()
