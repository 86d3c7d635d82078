#!/usr/bin/env python3
"""Randomised run of the Z-slab engine (the N >= 2 code: ghost-plane send / recv, fused all-reduces, table gathers, the Label
join) against the single-context HIP run of the same volume, on ONE GPU over the library's loopback transport: random shapes,
slab counts 2..6, halo schemes, spacings, textures and dtypes of tools/fuzz_parity.py.  Bar: Filter output, every threshold,
mask counts, label volume and label count IDENTICAL (the single context is what tools/fuzz_parity.py holds against the oracle).

  tools/fuzz_slabs.py SECONDS [SEED] [OUT] [big]
"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_parity as F  # noqa: E402


def run_slabs(vol, dr, world, halo_mode, raw_ghosts, kw):
    from nellie_amd import hipnative
    from nellie_amd.pipeline import FilterParams, min_area_pixels_of
    from nellie_amd.sharded import RcclComm, ShardedFramePipeline, slab_range
    gshape = vol.shape
    uid, uid2 = hipnative.comm_unique_id(loopback=True), hipnative.comm_unique_id(loopback=True)
    out, errs = [None] * world, []

    def worker(rank):
        try:
            p = FilterParams(dim_res=dr, **kw)
            o0, o1 = slab_range(gshape[0], world, rank)
            pipe = ShardedFramePipeline(gshape, rank, world, lambda ctx: RcclComm(ctx, world, rank, uid, uid2=uid2), p, halo_mode=halo_mode)
            g_lo, g_hi = pipe.raw_ghost_needed() if raw_ghosts else (0, 0)
            pipe.filter(np.ascontiguousarray(vol[o0 - g_lo:o1 + g_hi]), p)
            thr = pipe.frangi_threshold()
            n = pipe.label(thr, min_area_pixels_of(dr))
            out[rank] = (pipe.download_frangi(), thr, [s.mask_count for s in pipe.trace.scales], pipe.download_labels(), n)
            pipe.close()
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)

    threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    deadline = time.time() + 120
    for t in threads:
        t.join(max(0.1, deadline - time.time()))
    if any(t.is_alive() for t in threads):
        raise TimeoutError("slab ranks hung" + (f" after {errs[0]!r}" if errs else ""))
    if errs:
        raise errs[0]
    return out


BIG = False      # argv[4] == "big": 20 - 250 Mvoxel volumes of the synthetic generator (several Z chunks of the walk, thousands of tiles,
                 # slab tables of several blocks) -- too large for the oracle, so the bar is slabs == one context, bit for bit


def one_case(rng, idx):
    from nellie_amd import pipeline as pl
    world = int(rng.integers(2, 7))
    if BIG:
        from nellie_amd.synthetic import make_volume
        per = int(rng.integers(30, 90))
        extra = int(rng.integers(0, world))
        gshape = (per * world + extra, int(rng.integers(250, 800)), int(rng.integers(250, 900)))
        dr = F.SPACINGS[int(rng.integers(0, 2))]
        vol = make_volume(gshape, int(rng.integers(0, 1 << 30)), dtype=[np.float32, np.uint16][int(rng.integers(0, 2))])
    else:
        per = int(rng.integers(9, 40))
        extra = int(rng.integers(0, world))                  # uneven split
        gshape = (per * world + extra, int(rng.choice(F.PRIMES[8:28])), int(rng.choice(F.PRIMES[8:30])))
        dr = F.SPACINGS[int(rng.integers(0, len(F.SPACINGS)))]
        vol = F.draw_volume(rng, gshape)
    halo_mode = ["steps", "fat"][int(rng.integers(0, 2))]
    raw = bool(rng.integers(0, 2))
    kw = {}
    if rng.integers(0, 4) == 0:
        kw["frob_thresh_division"] = int(rng.choice([2, 3, 4, 8]))
    info = {"case": idx, "shape": list(gshape), "world": world, "halo": halo_mode + ("+raw" if raw else ""), "dtype": str(vol.dtype), "z_um": dr["Z"], "kw": kw}
    if idx < int(os.environ.get("FUZZ_SKIP", "0")):      # replay the draws of the earlier cases without running them
        info.update(ok=True, result="skipped")
        return info
    single = pl.FramePipeline(gshape)
    try:
        p = pl.FilterParams(dim_res=dr, **kw)
        try:
            single.filter(vol, p)
        except ValueError as exc:
            info.update(ok=True, result="single context raises " + str(exc)[:60])
            return info
        ref = single.download_frangi()
        ref_thr = single.frangi_threshold()
        ref_counts = [s.mask_count for s in single.trace.scales]
        ref_n = single.label(ref_thr, pl.min_area_pixels_of(dr))
        ref_lab = single.download_labels()
    finally:
        single.close()
    from nellie_amd.sharded import halo_depth, halo_depth_steps, slab_range
    need = halo_depth_steps(p) if halo_mode == "steps" else halo_depth(p)
    thinnest = min(b - a for a, b in (slab_range(gshape[0], world, r) for r in range(world)))
    if thinnest < need:                                  # the engine refuses such a split (ValueError on the thin rank)
        info.update(ok=True, result=f"refused: slab of {thinnest} planes under the {need}-plane halo")
        return info
    try:
        parts = run_slabs(vol, dr, world, halo_mode, raw, kw)
    except ValueError as exc:
        msg = str(exc)
        thin = "halo" in msg or "thin" in msg or "planes" in msg
        info.update(ok=thin, result=("refused: " if thin else "ERROR: ") + msg[:120])
        return info
    except TimeoutError as exc:
        info.update(ok=False, result="HANG: " + str(exc)[:200])
        print(json.dumps(info), flush=True)
        os._exit(3)
    except Exception as exc:  # noqa: BLE001
        info.update(ok=False, result="ERROR: " + repr(exc)[:300])
        return info
    got = np.concatenate([q[0] for q in parts])
    lab = np.concatenate([q[3] for q in parts])
    problems = []
    if not np.array_equal(got, ref):
        problems.append(f"{int((got != ref).sum())} Frangi voxels differ")
    for r, (_, thr, counts, _, n) in enumerate(parts):
        if thr != ref_thr or counts != ref_counts or n != ref_n:
            problems.append(f"rank {r}: thr {thr} vs {ref_thr}, counts {counts} vs {ref_counts}, n {n} vs {ref_n}")
            break
    if not np.array_equal(lab, ref_lab):
        problems.append(f"{int((lab != ref_lab).sum())} label voxels differ")
    info.update(ok=not problems, result="identical" if not problems else "MISMATCH: " + "; ".join(problems)[:300], labels=int(ref_n), nnz=int(np.count_nonzero(ref)))
    return info


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    out = sys.argv[3] if len(sys.argv) > 3 else None
    global BIG
    BIG = len(sys.argv) > 4 and sys.argv[4] == "big"
    rng = np.random.default_rng(seed)
    t0 = time.time()
    lines, bad, idx, res = [], 0, 0, {}
    while time.time() - t0 < budget:
        info = one_case(rng, idx)
        idx += 1
        bad += 0 if info["ok"] else 1
        key = info["result"].split(":")[0][:40]
        res[key] = res.get(key, 0) + 1
        lines.append(json.dumps(info))
        print(lines[-1], flush=True)
    summary = {"summary": True, "cases": idx, "failed": bad, "results": res, "seed": seed, "seconds": round(time.time() - t0, 1)}
    lines.append(json.dumps(summary))
    print(lines[-1], flush=True)
    if out:
        with open(out, "w") as f:
            f.write("# tools/fuzz_slabs.py: Z slabs over the loopback transport vs one context, random shapes / slab counts / halo schemes (bit-identical)\n")
            f.write("\n".join(lines) + "\n")
    os._exit(1 if bad else 0)


if __name__ == "__main__":
    main()
