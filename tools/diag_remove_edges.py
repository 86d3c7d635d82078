#!/usr/bin/env python3
"""A frame on which Filter(remove_edges=True) and the oracle disagree on whole rows: is it the edge removal, or the input it is
given?  filtering.py:969-1000 zeroes 15 rows at both ends of the bounding box of ANY non-zero value of a plane -- a response of
1e-8 that is 0 on the other side (the float32 exp, see tools/fuzz_parity.py) moves the box by a row, and whole rows of large
responses with it.  Checks: (1) where the supports of the two run_frame products differ; (2) the device's edge removal against the
oracle's remove_edges applied to the DEVICE's own run_frame (must be bit-identical).
  tools/diag_remove_edges.py VOLS.npy T Z_UM X_UM"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import pipeline as pl
from oracle import nellie_oracle as orc

vols = np.load(sys.argv[1]); t = int(sys.argv[2])
dr = {"X": float(sys.argv[4]), "Y": float(sys.argv[4]), "Z": float(sys.argv[3]), "T": 1.0}
vol = vols[t]
ref_run = orc.run_frame(vol, dr)
pipe = pl.FramePipeline(vol.shape)
p = pl.FilterParams(dim_res=dr)
pipe.compute_vesselness(vol, p)
run = pipe.download_frangi()
d = np.abs(run.astype(np.float64) - ref_run)
sup = (run > 0) != (ref_run > 0)
print(f"run_frame: max|d| {d.max():.3e}, support differs on {int(sup.sum())} voxels; the values there: device {np.sort(run[sup])[-5:]}, oracle {np.sort(ref_run[sup])[-5:]}")
for q in np.argwhere(sup)[:12]:
    print("   ", tuple(int(v) for v in q), "device", float(run[tuple(q)]), "oracle", float(ref_run[tuple(q)]))
n = pipe.ctx.remove_edges(15)
got = pipe.download_frangi()
own = orc.remove_edges(run)
print(f"device remove_edges vs the oracle's remove_edges of the device's own run_frame: {int((got != own).sum())} voxels differ (count {n} vs {int((own > 0).sum())})")
ref = orc.remove_edges(ref_run)
bad = np.argwhere((got > 0) != (ref > 0))
print(f"device vs oracle after remove_edges: support differs on {len(bad)} voxels, planes {sorted(set(int(b[0]) for b in bad))}")
for z in sorted(set(int(b[0]) for b in bad))[:4]:
    rd, ro = np.where(np.any(run[z], axis=1))[0], np.where(np.any(ref_run[z], axis=1))[0]
    print(f"   plane {z}: rows with any non-zero value: device {rd[0]}..{rd[-1]}, oracle {ro[0]}..{ro[-1]}")
thr = pipe.mask_volume(p)
fr = pipe.download_frangi()
own_fr, own_thr = orc.mask_volume(got, return_thr=True)
print(f"_mask_volume after the edge removal: device threshold {thr!r}, oracle's on the device's frame {own_thr!r}; masked frames differ on {int((fr != own_fr).sum())} voxels")
w = np.argwhere(fr != own_fr)
if len(w):
    print("   box", w.min(0).tolist(), "..", w.max(0).tolist())
    for q in w[:8]:
        print("   ", tuple(int(v) for v in q), "device", float(fr[tuple(q)]), "oracle", float(own_fr[tuple(q)]), "frame before", float(got[tuple(q)]))
    z = int(w[0][0])
    rows = np.where(np.any(got[z], axis=1))[0]
    print(f"   plane {z}: rows with any non-zero value after the edge removal: {rows[0]}..{rows[-1]}; threshold-positive rows: {np.where(np.any(got[z] > own_thr, axis=1))[0][[0, -1]].tolist()}")
# and through pipe.filter(remove_edges=True) in one go, as the stage class does
pipe.filter(vol, p, remove_edges=True)
fr2 = pipe.download_frangi()
print(f"pipe.filter(remove_edges=True): differs from the three separate steps on {int((fr2 != fr).sum())} voxels, from the oracle's chain on the device's run_frame on {int((fr2 != own_fr).sum())}")
pipe.close()
# the same on a FRESH pipeline, pipe.filter(remove_edges=True) as its first call (what the stage class does), twice
pipe2 = pl.FramePipeline(vol.shape)
for k in range(2):
    pipe2.filter(vol, p, remove_edges=True)
    fr3 = pipe2.download_frangi()
    w = np.argwhere(fr3 != own_fr)
    print(f"fresh pipeline, call {k + 1}: differs from the oracle's chain on the device's run_frame on {len(w)} voxels" + (f", box {w.min(0).tolist()}..{w.max(0).tolist()}" if len(w) else ""))
pipe2.close()
# and through the stage class on the whole stack
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from fakes import ArrayImInfo
from nellie_amd.segmentation.filtering import Filter
im = ArrayImInfo(vols, dr)
Filter(im, remove_edges=True).run()
for tt in range(vols.shape[0]):
    g = np.asarray(im.store["frangi"][tt])
    if tt == t:
        w = np.argwhere(g != own_fr)
        print(f"Filter(remove_edges=True).run(), frame {tt}: differs on {len(w)} voxels" + (f", box {w.min(0).tolist()}..{w.max(0).tolist()}" if len(w) else ""))
# the oracle's chain on ITS OWN run_frame against the oracle's chain on the device's run_frame: what an ulp in run_frame is worth
ref_fr, ref_thr = orc.mask_volume(ref, return_thr=True)
w = np.argwhere(ref_fr != own_fr)
big = np.argwhere(np.abs(ref_fr.astype(np.float64) - own_fr) > 1e-4 * np.abs(ref_fr) + 2.4e-7)
print(f"oracle chain on the oracle's vs on the device's run_frame: thresholds {ref_thr!r} / {own_thr!r}; {len(w)} voxels differ, {len(big)} beyond the bars" + (f", box {big.min(0).tolist()}..{big.max(0).tolist()}" if len(big) else ""))
if len(big):
    lo, hi = np.maximum(big.min(0) - 2, 0), big.max(0) + 3
    sub_r, sub_d = ref[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]], got[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
    flips = np.argwhere((sub_r > ref_thr) != (sub_d > own_thr))
    print(f"   voxels around the box on different sides of their thresholds: {len(flips)}")
    for q in flips[:10]:
        print("      ", (q + lo).tolist(), "oracle", float(sub_r[tuple(q)]), "device", float(sub_d[tuple(q)]))
