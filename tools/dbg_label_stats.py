import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nellie_amd import pipeline as pl
from nellie_amd.synthetic import ISO_01, make_volume
shape = (512, 1024, 1024)
vol = make_volume(shape, 2345)
pipe = pl.FramePipeline(shape)
pipe.filter(vol, pl.FilterParams(dim_res=ISO_01))
thr = pipe.frangi_threshold()
fr = pipe.download_frangi()
n = pipe.label(thr, pl.min_area_pixels_of(ISO_01))
lab = pipe.download_labels()
m = fr > thr
runs = int((m[:, :, 1:] & ~m[:, :, :-1]).sum() + m[:, :, 0].sum())
l = lab > 0
runs2 = int((l[:, :, 1:] & ~l[:, :, :-1]).sum() + l[:, :, 0].sum())
print("frangi>0", float((fr > 0).mean()), "thr", thr, "mask frac", float(m.mean()), "fg runs", runs, "labels", n, "labelled frac", float(l.mean()), "runs after", runs2)
pipe.close()
