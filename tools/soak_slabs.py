#!/usr/bin/env python3
"""Soak of the Z-slab engine on one GPU (loopback transport): a T stack through Filter + Label as ONE context and as 2, 3, 5 slab
contexts (NELLIE_FORCE_SLABS), with and without random transfer delays; the products must be identical.
    tools/soak_slabs.py [T] [Z Y X]"""
import json, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import numpy as np
from fakes import ArrayImInfo
from nellie_amd.segmentation.filtering import Filter
from nellie_amd.segmentation.labelling import Label
from nellie_amd.synthetic import ANISO_03, ISO_01, make_volume

T = int(sys.argv[1]) if len(sys.argv) > 1 else 12
shape = tuple(int(a) for a in sys.argv[2:5]) if len(sys.argv) >= 5 else (90, 112, 136)
out = {"frames": T, "shape": list(shape), "runs": []}
for dr, name in ((ISO_01, "iso"), (ANISO_03, "aniso")):
    vols = np.stack([make_volume(shape, 500 + 17 * t) for t in range(T)])
    os.environ.pop("NELLIE_FORCE_SLABS", None)
    ref = ArrayImInfo(vols, dr)
    Filter(ref, device="gpu").run(); Label(ref, device="gpu").run()
    for slabs in (2, 3, 5):
        os.environ["NELLIE_FORCE_SLABS"] = str(slabs)
        t0 = time.time()
        got = ArrayImInfo(vols, dr)
        Filter(got, device="gpu").run(); Label(got, device="gpu").run()
        ok = bool(np.array_equal(ref.store["frangi"], got.store["frangi"]) and np.array_equal(ref.store["labels"], got.store["labels"]))
        out["runs"].append({"spacing": name, "slabs": slabs, "identical": ok, "labels": int(got.store["labels"].max()), "s": round(time.time() - t0, 1)})
print(json.dumps(out))
