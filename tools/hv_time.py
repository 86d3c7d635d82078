#!/usr/bin/env python3
"""In-situ timing of the one-pass Hessian walk (nl_vesselness_spec) for A/B builds of the library -- including builds whose
results are deliberately wrong (timing experiments): the frame runs as usual, and every time the pipeline calls the walk,
the same call is repeated N times on the very Gaussian volume and bracket of that scale; the "vesselness" HIP-event group
gives the average launch.  What happens after the walk does not matter to the figure.

  tools/hv_time.py Z Y X reps -- 'NELLIE_HIP_LIB=... [ENV=...]' ...         (one subprocess per configuration)"""
import json, os, subprocess, sys, zlib


def child(shape, reps):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from nellie_amd import pipeline as pl
    from nellie_amd.synthetic import ISO_01, make_volume
    vol = make_volume(shape, 1234)
    pipe = pl.FramePipeline(shape)
    pipe._device_chain = False            # the synchronous path calls nl_vesselness_spec, which is what gets repeated
    pipe.load_input(vol)
    p = pl.FilterParams(dim_res=ISO_01)
    ctx = pipe.ctx
    real = ctx.vesselness_spec
    per_scale, stats = [], []

    def timed(spacing, lo, hi, z0=-1, z1=-1):
        real(spacing, lo, hi, z0=z0, z1=z1)                   # warm
        ctx.sync(); ctx.prof_reset(); ctx.prof_enable(True)
        for _ in range(reps):
            r = real(spacing, lo, hi, z0=z0, z1=z1)
        ctx.sync(); ctx.prof_enable(False)
        ms, k = ctx.prof_get("vesselness")
        per_scale.append(round(ms / max(k, 1), 4))
        stats.append([float(r[0]), float(r[1]), int(r[2]), int(r[3])])
        return r

    ctx.vesselness_spec = timed
    try:
        pipe.filter(None, p)
        ok = True
    except Exception as exc:  # noqa: BLE001  (a deliberately wrong build may derail the rest of the frame)
        ok = repr(exc)[:200]
    out = {"walk_ms": per_scale, "mean": round(sum(per_scale) / max(len(per_scale), 1), 4), "stats": stats, "frame_ok": ok}
    if ok is True:
        out["crc_frangi"] = zlib.crc32(pipe.download_frangi().tobytes())
    print("HV " + json.dumps(out), flush=True)
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
        line = [l for l in r.stdout.splitlines() if l.startswith("HV ")]
        print(f"[{cfg}]", line[0][3:] if line else f"FAILED rc={r.returncode} {r.stderr[-600:]}", flush=True)
