#!/usr/bin/env python3
"""Network stage's two dense steps (pixel classes, branch labels) on a synthetic skeleton image.
    python tools/bench_network.py [Z Y X]          default 1024^3 (a 128^3 random-walk skeleton tiled)
Times include the H2D of the int32 skeleton and the D2H of both products (the C-ABI hands over host arrays);
`kernel_ms` is the HIP-event time of the device work alone."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import hipnative
from nellie_amd.synthetic import make_skeleton

shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1024, 1024, 1024)
tile = make_skeleton((128, 128, 128), 7)
reps = [-(-s // 128) for s in shape]
skel = np.ascontiguousarray(np.tile(tile, reps)[:shape[0], :shape[1], :shape[2]])
ctx = hipnative.Context(shape)
wall, kern = [], []
for rep in range(3):
    ctx.prof_reset(); ctx.prof_enable(True)
    t0 = time.perf_counter()
    pc, n_skel = ctx.skel_pixel_class(skel)
    labels, n_branches = ctx.skel_branch_labels(None)
    wall.append(time.perf_counter() - t0)
    ctx.prof_enable(False)
    kern.append(ctx.prof_get("network")[0])
print(json.dumps({"shape": list(shape), "skeleton_voxels": n_skel, "branches": n_branches,
                  "classes": np.bincount(pc.ravel(), minlength=5).tolist(),
                  "wall_ms_best": round(min(wall) * 1e3, 2), "device_ms_best": round(min(kern), 3),
                  "note": "wall includes H2D of the int32 skeleton and D2H of uint8 classes + int32 labels from pageable memory"}))
ctx.close()
