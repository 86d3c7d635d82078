"""What the device path does with NaN / +-Inf voxels in the input, beside the oracle (= the reference: filtering.py:421-426 replaces
infinite Frobenius norms, :764-766 cleans the response; numpy's histogram raises on an infinite range).
    python tools/probe_nonfinite.py            (needs a GPU)"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nellie_amd import pipeline as pl                      # noqa: E402
from nellie_amd.synthetic import ISO_01, make_volume       # noqa: E402
from oracle import nellie_oracle as orc                    # noqa: E402

warnings.simplefilter("ignore")
for shape in ((20, 40, 40), (40, 96, 80)):
    for tag, val, where in (("nan", np.nan, "one"), ("-inf", -np.inf, "one"), ("+inf", np.inf, "one"), ("nan", np.nan, "plane"), ("nan", np.nan, "all")):
        vol = make_volume(shape, 9)
        if where == "one":
            vol[shape[0] // 2, shape[1] // 2, shape[2] // 2] = val
        elif where == "plane":
            vol[shape[0] // 2] = val
        else:
            vol[:] = val
        for mask in (True, False):
            try:
                ref = orc.filter_frame(vol.copy(), ISO_01, mask=mask)
                rs = f"oracle nnz {int((ref > 0).sum())}"
            except Exception as e:  # noqa: BLE001
                ref, rs = None, f"oracle raised {type(e).__name__}: {str(e)[:60]}"
            pipe = pl.FramePipeline(shape)
            t0 = time.time()
            try:
                pipe.filter(vol.copy(), pl.FilterParams(dim_res=ISO_01), mask=mask)
                out = pipe.download_frangi()
                ds = f"device nnz {int((out > 0).sum())} nan {int(np.isnan(out).sum())} inf {int(np.isinf(out).sum())} chain_fallbacks {pipe.chain_fallbacks}"
                if ref is not None:
                    tol = 1e-4 * np.abs(ref) + 1e-6 * np.abs(ref).max()
                    ds += f" | differing voxels {int((np.abs(out - ref) > tol).sum())} support differs {int(((out > 0) != (ref > 0)).sum())}"
            except Exception as e:  # noqa: BLE001
                ds = f"device raised {type(e).__name__}: {str(e)[:80]}"
            pipe.close()
            print(f"{shape} {tag:5s} {where:5s} mask={mask}: {rs} || {ds} ({time.time() - t0:.2f} s)", flush=True)
