#!/usr/bin/env python3
"""Randomised runs THROUGH FILES: an OME-TIFF written by nellie_amd/im_info/ome_tiff.py (random shape, dtype, T stack or
single frame, 3-D or 2-D, spacing) -> FileInfo / ImInfo -> run() (Filter, Label, Markers stage by stage) and run_streamed()
(the overlapped streamer, one or two lanes / Z slabs) into separate output directories; every product is reopened from its
file and held against the oracle (Filter: bars of tools/fuzz_parity.py; labels and Markers products, computed by the oracle
from the files' own Filter output / labels: bit for bit), the two runs against each other (identical files), the input file
against what was written (never modified).

  tools/fuzz_files.py SECONDS [SEED] [OUT]
"""
import json
import os
import shutil
import sys
import tempfile
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_parity as F  # noqa: E402


def one_case(rng, idx, tmp):
    from nellie_amd.im_info import ome_tiff
    from nellie_amd.im_info.verifier import FileInfo, ImInfo
    from nellie_amd.run import run, run_streamed
    from oracle import nellie_oracle as orc
    nt = int(rng.integers(1, 5))
    shape = F.draw_shape(rng)
    dr = F.SPACINGS[int(rng.integers(0, len(F.SPACINGS)))]
    vols = [F.draw_volume(rng, shape) for _ in range(nt)]
    dt = np.dtype(vols[0].dtype if vols[0].dtype != np.float64 else np.float32)         # the file layer stores what microscopes write
    vols = np.stack([v.astype(dt) for v in vols])
    streamed_kw = [{}, {"devices": [0, 0]}][int(rng.integers(0, 2))]
    info = {"case": idx, "shape": [nt] + list(shape), "dtype": str(dt), "z_um": dr["Z"], "x_um": dr["X"], "streamed": "two lanes (devices=[0,0])" if streamed_kw else "one lane"}
    d = os.path.join(tmp, f"c{idx}")
    os.makedirs(d)
    src = os.path.join(d, "img.ome.tif")
    ome_tiff.create(src, vols.shape, dt, dr, "raw", data=vols)
    level = "equal"
    try:
        ref_err = None
        refs = []
        try:
            for t in range(nt):
                run_ = orc.run_frame(vols[t], dr)
                fr, thr = orc.mask_volume(run_, return_thr=True) if float(np.sum(run_)) > 0 else (run_, None)
                refs.append((run_, fr, thr))
        except ValueError as exc:
            ref_err = str(exc)
        fa = FileInfo(src, output_dir=os.path.join(d, "a"))
        if ref_err is not None:
            try:
                run(fa, device="gpu")
            except ValueError:
                info.update(ok=True, result="both raise")
                return info
            raise AssertionError(f"oracle raised ({ref_err}), run() did not")
        a = run(fa, device="gpu", markers=True)
        assert a.shape[-3:] == shape and (nt == 1 or a.shape[0] == nt)
        fb = FileInfo(src, output_dir=os.path.join(d, "b"))
        fb.find_metadata(); fb.load_metadata()
        b = ImInfo(fb)
        run_streamed(b, **streamed_kw)
        get = lambda im, key: np.asarray(im.get_memmap(im.pipeline_paths[key], read_mode="r")).reshape((nt,) + shape)
        fr_a, lab_a = get(a, "im_preprocessed"), get(a, "im_instance_label")
        assert fr_a.dtype == np.float32 and lab_a.dtype == np.int32
        assert np.array_equal(np.asarray(a.get_memmap(a.im_path, read_mode="r")).reshape(vols.shape), vols), "the input file was modified"
        for key in ("im_preprocessed", "im_instance_label"):
            assert np.array_equal(get(a, key), get(b, key)), f"run() and run_streamed() wrote different {key}"
        mk, di, bo = get(a, "im_marker"), get(a, "im_distance"), get(a, "im_border")
        for t in range(nt):
            run_, fr, thr = refs[t]
            last = None
            for floor, name in ((0.0, "equal"), (F.FLOOR, "equal_at_exp_floor"), (F.FLOOR, "equal_labels_only")):
                try:
                    if name == "equal_labels_only":      # the percentile moved by a rank (see fuzz_parity's third level): the files'
                        pass                             # own Filter output is what Label and Markers are held to below
                    elif thr is None:
                        F.frangi_close(fr_a[t], fr, floor, f"frangi[{t}]")
                    else:
                        F.masked_close(orc, fr_a[t], fr, run_, thr, floor, f"frangi[{t}]")
                    last = None
                    if name != "equal" and level != "equal_labels_only":
                        level = name
                    break
                except AssertionError as exc:
                    last = exc
            if last is not None:
                raise last
            try:
                ref_lab = orc.label_frame(fr_a[t], dr)
            except ValueError:
                continue                                 # numpy raises on a degenerate histogram: run() would have raised too
            assert np.array_equal(lab_a[t], ref_lab), f"labels[{t}] differ on {int((lab_a[t] != ref_lab).sum())} voxels"
            m, dd, bb = orc.markers_frame(vols[t], lab_a[t], dr)
            assert np.array_equal(di[t], dd) and np.array_equal(bo[t], bb), f"distance / border [{t}] differ"
            assert np.array_equal(mk[t], m), f"markers[{t}] differ on {int((mk[t] != m).sum())} voxels"
        info.update(ok=True, result=level, labels=[int(l.max()) for l in lab_a])
    except AssertionError as exc:
        info.update(ok=False, result="MISMATCH: " + str(exc)[:300])
    except Exception as exc:  # noqa: BLE001
        info.update(ok=False, result="ERROR: " + repr(exc)[:200] + " | " + " / ".join(traceback.format_exc().splitlines()[-4:])[:500])
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return info


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    out = sys.argv[3] if len(sys.argv) > 3 else None
    rng = np.random.default_rng(seed)
    tmp = tempfile.mkdtemp(prefix="nellie_fuzz_files_")
    t0 = time.time()
    lines, bad, idx, res = [], 0, 0, {}
    while time.time() - t0 < budget:
        info = one_case(rng, idx, tmp)
        idx += 1
        bad += 0 if info["ok"] else 1
        key = info["result"] if info["ok"] else info["result"].split(":")[0]
        res[key] = res.get(key, 0) + 1
        lines.append(json.dumps(info))
        print(lines[-1], flush=True)
    shutil.rmtree(tmp, ignore_errors=True)
    summary = {"summary": True, "cases": idx, "failed": bad, "results": res, "seed": seed, "seconds": round(time.time() - t0, 1)}
    lines.append(json.dumps(summary))
    print(lines[-1], flush=True)
    if out:
        with open(out, "w") as f:
            f.write("# tools/fuzz_files.py: OME-TIFF -> FileInfo / ImInfo -> run() and run_streamed() -> files reopened, vs the oracle and vs each other\n")
            f.write("\n".join(lines) + "\n")
    os._exit(1 if bad else 0)


if __name__ == "__main__":
    main()
