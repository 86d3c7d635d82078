#!/usr/bin/env python3
"""A/B driver for kernel experiments: runs the hot path under several environments (one subprocess each, the library reads
its knobs at load time) and prints per-group HIP-event times plus checksums of both outputs, so a variant that is faster
but not bit-identical shows at once.   tools/ab_hess.py Z Y X reps -- 'NAME=VAL ...' 'NAME=VAL ...'"""
import json, os, subprocess, sys, zlib

def child(shape, reps):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import time
    import numpy as np
    from nellie_amd import pipeline as pl
    from nellie_amd.synthetic import ISO_01, make_volume
    vol = make_volume(shape, 1234)
    pipe = pl.FramePipeline(shape)
    pipe.load_input(vol)
    p = pl.FilterParams(dim_res=ISO_01)
    ma = pl.min_area_pixels_of(ISO_01)
    def step():
        pipe.filter(None, p)
        return pipe.label(pipe.frangi_threshold(), ma)
    step()
    pipe.ctx.prof_reset(); pipe.ctx.prof_enable(True)
    pipe.ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps):
        n = step()
    pipe.ctx.sync(); dt = (time.perf_counter() - t0) / reps * 1e3
    pipe.ctx.prof_enable(False)
    out = {"ms_per_step": round(dt, 3), "labels": int(n), "npos": int(pipe.trace.n_positive),
           "mask_counts": [int(sc.mask_count) for sc in pipe.trace.scales], "one_pass": [bool(sc.one_pass) for sc in pipe.trace.scales]}
    for g in ("gauss_zyx<4,4>", "gauss_zyx<3,3>", "gauss_zyx<5,5>", "gauss_zyx<1,4>", "gauss_zyx<1,3>", "gauss_zyx<2,5>", "gauss_z", "gauss_yx", "sample", "hessian_stats", "vesselness", "vesselness_resolve", "mask_volume", "label"):
        ms, k = pipe.ctx.prof_get(g)
        if k:
            out[g] = round(ms / reps, 3)
    fr = pipe.download_frangi(); lab = pipe.download_labels()
    out["crc_frangi"] = zlib.crc32(fr.tobytes()); out["crc_labels"] = zlib.crc32(lab.tobytes())
    print("AB " + json.dumps(out), flush=True)
    pipe.close()

if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(tuple(int(a) for a in sys.argv[2:5]), int(sys.argv[5]))
        sys.exit(0)
    i = sys.argv.index("--")
    shape, reps = sys.argv[1:4], sys.argv[4]
    for cfg in sys.argv[i + 1:]:
        env = dict(os.environ)
        for kv in cfg.split():
            k, v = kv.split("=", 1)
            env[k] = v
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", *shape, reps], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("AB ")]
        print(f"[{cfg}]", line[0][3:] if line else f"FAILED rc={r.returncode} {r.stderr[-800:]}", flush=True)
