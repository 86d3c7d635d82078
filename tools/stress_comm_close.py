#!/usr/bin/env python3
"""Open / use / close loopback communicators from `world` threads at once, many times: the ranks of a job close their contexts
simultaneously, which is where lb::comm_destroy once touched its Group after dropping its reference (csrc/loopback.inc).
  tools/stress_comm_close.py SECONDS [WORLD]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import hipnative
from nellie_amd.sharded import RcclComm

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
world = int(sys.argv[2]) if len(sys.argv) > 2 else 6
bar = threading.Barrier(world)
uids = [None, None]
rounds, errs = [0], []


def worker(rank):
    t0 = time.time()
    try:
        while True:
            if rank == 0:
                uids[0], uids[1] = hipnative.comm_unique_id(loopback=True), hipnative.comm_unique_id(loopback=True)
                stop[0] = time.time() - t0 > budget
            bar.wait()
            if stop[0]:
                return
            ctx = hipnative.Context((4, 8, 64), gz0=4 * rank, gnz=4 * world)
            comm = RcclComm(ctx, world, rank, uids[0], uid2=uids[1])
            s = comm.allreduce(np.array([rank + 1], np.int64), "sum")
            assert int(s[0]) == world * (world + 1) // 2
            bar.wait()
            ctx.close()                      # all ranks at once
            if rank == 0:
                rounds[0] += 1
    except Exception as exc:  # noqa: BLE001
        errs.append(repr(exc))
        bar.abort()


stop = [False]
ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
[t.start() for t in ts]
[t.join() for t in ts]
print({"rounds": rounds[0], "world": world, "closes": rounds[0] * world * 2, "errors": errs[:2]})
sys.exit(1 if errs else 0)
