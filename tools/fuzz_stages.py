#!/usr/bin/env python3
"""Randomised differential run of the "next" rows against the oracle (GPU box): Markers (distance / border / marker images
bit-exact, use_im = distance and frangi, 3-D and 2-D), Network's two dense steps (pixel classes, branch labels: bit-exact) and
the 2-D Filter + Label path -- random shapes / spacings / object layouts, as tools/fuzz_parity.py does for the 3-D hot path.

  tools/fuzz_stages.py SECONDS [SEED] [OUT]
"""
import json
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_parity as F  # noqa: E402


def random_labels(rng, shape, vol):
    """Objects as a threshold of a smoothed random field plus the volume's own bright structures; ids by a coarse hash (the
    Markers stage only reads labels > 0)."""
    thr = np.percentile(vol.astype(np.float32), float(rng.uniform(90, 99.5)))
    m = vol.astype(np.float32) > thr
    if rng.integers(0, 3) == 0:                    # a few solid boxes touching faces
        for _ in range(int(rng.integers(1, 4))):
            lo = [int(rng.integers(0, max(1, s - 2))) for s in shape]
            hi = [min(s, l + int(rng.integers(2, 9))) for s, l in zip(shape, lo)]
            m[tuple(slice(a, b) for a, b in zip(lo, hi))] = True
    return np.where(m, 1 + (np.arange(m.size).reshape(shape) % 7), 0).astype(np.int32)


def case_markers(rng, idx):
    from nellie_amd import pipeline as pl
    from oracle import nellie_oracle as orc
    two_d = rng.integers(0, 4) == 0
    if two_d:
        shape = (int(rng.choice(F.PRIMES[6:])), int(rng.choice(F.PRIMES[6:])))
        dr = {"X": 0.1, "Y": 0.1, "Z": None, "T": 1.0}
        vol3 = F.draw_volume(rng, (5,) + shape)
        vol = np.ascontiguousarray(vol3[2])
    else:
        shape = F.draw_shape(rng)
        dr = F.SPACINGS[int(rng.integers(0, len(F.SPACINGS)))]
        vol = F.draw_volume(rng, shape)
    lab = random_labels(rng, shape, vol)
    use_fr = rng.integers(0, 3) == 0
    fr = (np.abs(vol.astype(np.float32)) * np.float32(1e-3) * (lab > 0)).astype(np.float32) if use_fr else None
    kw = {}
    if rng.integers(0, 3) == 0:
        kw["peak_min_distance"] = int(rng.integers(1, 4))
    if rng.integers(0, 4) == 0:
        kw["num_sigma"] = int(rng.integers(2, 7))
    info = {"stage": "markers", "case": idx, "shape": list(shape), "dtype": str(vol.dtype), "frangi": bool(use_fr), "kw": kw, "mask": int((lab > 0).sum())}
    ref = orc.markers_frame(vol, lab, dr, frangi=fr, **kw)
    pipe = pl.FramePipeline(shape)
    try:
        n = pipe.markers(dr, labels=lab, intensity=vol, use_image=fr, **kw)
        marker, dist, border = (a.reshape(shape) for a in pipe.download_markers())
        assert np.array_equal(dist, ref[1]), f"distance differs on {int((dist != ref[1]).sum())} voxels"
        assert np.array_equal(border, ref[2]), f"border differs on {int((border != ref[2]).sum())} voxels"
        assert np.array_equal(marker, ref[0]), f"markers differ on {int((marker != ref[0]).sum())} voxels"
        assert n == int(ref[0].sum())
        info.update(ok=True, result="equal", markers=int(n))
    finally:
        pipe.close()
    return info


def case_network(rng, idx):
    from nellie_amd.segmentation.networking import HipNetworkKernels
    from nellie_amd.synthetic import make_skeleton
    from oracle import nellie_oracle as orc
    shape = F.draw_shape(rng)
    skel = make_skeleton(shape, int(rng.integers(0, 1 << 30)), n_walks=int(rng.integers(1, 120)))
    if rng.integers(0, 3) == 0:                    # clumps: junction-rich neighbourhoods
        pts = rng.integers(0, 1 << 30, size=(int(rng.integers(5, 60)), 3)) % np.array(shape)
        for z, y, x in pts:
            skel[z, y, x] = skel.max() + 1 if skel.max() > 0 else 1
    info = {"stage": "network", "case": idx, "shape": list(shape), "voxels": int((skel > 0).sum())}
    k = HipNetworkKernels()
    try:
        pc = k._get_pixel_class(skel)
        ref = orc.network_pixel_class(skel)
        assert np.array_equal(pc, ref), f"pixel classes differ on {int((pc != ref).sum())} voxels"
        bl = k._get_branch_skel_labels(pc)
        rb = orc.network_branch_skel_labels(ref)
        assert np.array_equal(bl, rb), f"branch labels differ on {int((bl != rb).sum())} voxels"
        info.update(ok=True, result="equal", branches=int(rb.max()))
    finally:
        k.close()
    return info


def case_2d(rng, idx):
    from nellie_amd import pipeline as pl
    from oracle import nellie_oracle as orc
    shape = (int(rng.choice(F.PRIMES[8:])), int(rng.choice(F.PRIMES[8:])))
    dr = {"X": float(rng.choice([0.065, 0.1, 0.2])), "Y": None, "Z": None, "T": 1.0}
    dr["Y"] = dr["X"]
    vol = np.ascontiguousarray(F.draw_volume(rng, (5,) + shape)[2])
    info = {"stage": "2d", "case": idx, "shape": list(shape), "dtype": str(vol.dtype), "x_um": dr["X"]}
    ref_err = None
    try:
        ref_run = orc.run_frame_2d(vol, dr)
    except ValueError as exc:                        # numpy raises on degenerate histograms (filtering.py:432-441 via gpu_functions.py:64-94)
        ref_err = str(exc)
    pipe = pl.FramePipeline(shape)
    try:
        p = pl.FilterParams(dim_res=dr)
        if ref_err is not None:
            try:
                pipe.compute_vesselness(vol, p)
            except ValueError as exc:
                assert str(exc)[:30] == ref_err[:30], f"messages differ: {exc} / {ref_err}"
                info.update(ok=True, result="both raise")
                return info
            raise AssertionError(f"oracle raised ({ref_err}), device did not")
        pipe.compute_vesselness(vol, p)
        run = pipe.download_frangi()
        level = None
        for floor, name in ((0.0, "equal"), (F.FLOOR, "equal_at_exp_floor")):
            try:
                F.frangi_close(run.reshape(shape), ref_run, floor, "run_frame_2d")
                level = name
                break
            except AssertionError as exc:
                last = exc
        if level is None:
            raise last
        # Label on the oracle's masked image: bit-exact
        ref_fr = orc.mask_volume_2d(ref_run) if float(np.sum(ref_run)) > 0 else ref_run
        try:
            ref_lab, ref_thr = orc.label_frame_2d(ref_fr, dr, return_thr=True)
        except ValueError as exc:                    # one non-empty bin in the log-domain histogram of Label's threshold (labelling.py:385-455): numpy raises,
            pipe.upload_frangi(ref_fr)               # and so must the product (seed 33, cases 2950 and 6174: the first two such 2-D frames any seed drew)
            try:
                pipe.frangi_threshold()
            except ValueError as exc2:
                assert str(exc2)[:30] == str(exc)[:30], f"messages differ: {exc2} / {exc}"
                info.update(ok=True, result=level + ", Label: both raise")
                return info
            raise AssertionError(f"the oracle's Label threshold raised ({exc}), the device's did not")
        pipe.upload_frangi(ref_fr)
        thr = pipe.frangi_threshold()
        assert (thr is None and ref_thr is None) or float(thr) == float(ref_thr), f"label threshold {thr} vs {ref_thr}"
        pipe.label(thr, pl.min_area_pixels_of(dr, no_z=True), fill_holes=False)
        lab = pipe.download_labels().reshape(shape)
        assert np.array_equal(lab, ref_lab), f"labels differ on {int((lab != ref_lab).sum())} pixels"
        info.update(ok=True, result=level, labels=int(ref_lab.max()))
    finally:
        pipe.close()
    return info


def case_label(rng, idx):
    """Label on dense random structure (fill 2 - 80 %, blobs with cavities, shells, single voxels): hole filling, both
    26-connected labellings, area filter, majority smoothing and numbering, bit for bit (labelling.py:467-509)."""
    from nellie_amd import pipeline as pl
    from oracle import nellie_oracle as orc
    shape = F.draw_shape(rng)
    dr = F.SPACINGS[int(rng.integers(0, len(F.SPACINGS)))]
    nz, ny, nx = shape
    f = rng.random(shape).astype(np.float32)
    for ax in range(3):                                  # a few box-filter passes: structure at a random scale
        for _ in range(int(rng.integers(0, 3))):
            f = (f + np.roll(f, 1, ax) + np.roll(f, -1, ax)) / np.float32(3)
    fill = float(rng.uniform(0.02, 0.8))
    cut = np.quantile(f, 1.0 - fill)
    fr = np.where(f > cut, f - np.float32(cut) + np.float32(1e-3), 0).astype(np.float32)
    for _ in range(int(rng.integers(0, 4))):             # hollow boxes: cavities for the hole filling, shells for the area filter
        lo = [int(rng.integers(0, max(1, s_ - 3))) for s_ in shape]
        hi = [min(s_, l + int(rng.integers(3, 14))) for s_, l in zip(shape, lo)]
        sl = tuple(slice(a, b) for a, b in zip(lo, hi))
        inner = tuple(slice(a + 1, max(a + 1, b - 1)) for a, b in zip(lo, hi))
        fr[sl] = np.float32(rng.uniform(0.5, 2.0))
        if rng.integers(0, 2):
            fr[inner] = 0
    if rng.integers(0, 3) == 0:
        fr[rng.random(shape) < 0.01] = np.float32(0.7)   # isolated voxels
    kw = {}
    if rng.integers(0, 3) == 0:
        kw["min_radius_um"] = float(rng.choice([0.05, 0.1, 0.25, 0.5]))
    info = {"stage": "label", "case": idx, "shape": list(shape), "fill": round(float((fr > 0).mean()), 3), "kw": kw}
    ref_lab, ref_thr = orc.label_frame(fr, dr, return_thr=True, **kw)
    pipe = pl.FramePipeline(shape)
    try:
        pipe.upload_frangi(fr)
        thr = pipe.frangi_threshold()
        assert (thr is None and ref_thr is None) or float(thr) == float(ref_thr), f"label threshold {thr} vs {ref_thr}"
        n = pipe.label(thr, pl.min_area_pixels_of(dr, **({"min_radius_um": kw["min_radius_um"]} if kw else {})))
        lab = pipe.download_labels()
        assert np.array_equal(lab, ref_lab), f"labels differ on {int((lab != ref_lab).sum())} voxels"
        assert n == int(ref_lab.max())
        info.update(ok=True, result="equal", labels=int(n))
    finally:
        pipe.close()
    return info


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    out = sys.argv[3] if len(sys.argv) > 3 else None
    rng = np.random.default_rng(seed)
    t0 = time.time()
    lines, bad, idx, res = [], 0, 0, {}
    cases = [case_markers, case_network, case_2d, case_label]
    while time.time() - t0 < budget:
        fn = cases[idx % len(cases)]
        try:
            info = fn(rng, idx)
        except AssertionError as exc:
            info = {"stage": fn.__name__[5:], "case": idx, "ok": False, "result": "MISMATCH: " + str(exc)[:300]}
        except Exception as exc:  # noqa: BLE001
            info = {"stage": fn.__name__[5:], "case": idx, "ok": False, "result": "ERROR: " + repr(exc)[:200] + " | " + " / ".join(traceback.format_exc().splitlines()[-4:])[:400]}
        idx += 1
        bad += 0 if info["ok"] else 1
        key = info["stage"] + " " + info["result"].split(":")[0][:30]
        res[key] = res.get(key, 0) + 1
        lines.append(json.dumps(info))
        print(lines[-1], flush=True)
    summary = {"summary": True, "cases": idx, "failed": bad, "results": res, "seed": seed, "seconds": round(time.time() - t0, 1)}
    lines.append(json.dumps(summary))
    print(lines[-1], flush=True)
    if out:
        with open(out, "w") as f:
            f.write("# tools/fuzz_stages.py: Markers, Network steps and the 2-D path vs the oracle on random inputs\n")
            f.write("\n".join(lines) + "\n")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
