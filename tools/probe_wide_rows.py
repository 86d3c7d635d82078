"""Filter + Label against the oracle on rows wider than 64 mask words (nx > 4096: the generic paths of the pack kernels) and other odd
shapes: tools/probe_wide_rows.py"""
import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import oracle.nellie_oracle as o
from nellie_amd import pipeline as pl
from nellie_amd.synthetic import ISO_01, make_volume
for shape in ((6,12,4200),(4,70,4097),(3,6,8200),(9,33,130),(2,2,5000)):
    vol=make_volume(shape, 11)
    ref=o.filter_frame(vol, ISO_01); ref_lab=o.label_frame(ref, ISO_01)
    pipe=pl.FramePipeline(shape)
    pipe.filter(vol, pl.FilterParams(dim_res=ISO_01))
    got=pipe.download_frangi()
    thr=pipe.frangi_threshold(); n=pipe.label(thr, pl.min_area_pixels_of(ISO_01)); lab=pipe.download_labels()
    pipe.upload_frangi(ref); thr2=pipe.frangi_threshold(); n2=pipe.label(thr2, pl.min_area_pixels_of(ISO_01)); lab2=pipe.download_labels()
    pipe.close()
    d=np.abs(got.astype(np.float64)-ref)
    print(shape,"support equal",np.array_equal(got>0,ref>0),"max rel",float(d.max()/max(ref.max(),1e-30)),"labels given oracle frangi equal",np.array_equal(lab2,ref_lab), n2, int(ref_lab.max()))
