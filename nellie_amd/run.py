"""
`run()`: the hot-path half of nellie.run.run (reference nellie/run.py:18-130) -- Filter then Label on the
MI355X engine, same keyword names, same `timeit` prints.  The later stages (Network, Markers, tracking,
Hierarchy) are Nellie's own and consume the two files this writes.
"""
from __future__ import annotations

import time

from nellie_amd.segmentation.filtering import Filter
from nellie_amd.segmentation.labelling import Label


def run(im_info, remove_edges=False, otsu_thresh_intensity=False, threshold=None, timeit=False, device="auto",
        low_memory=False):
    t0 = time.perf_counter() if timeit else None
    preprocessing = Filter(im_info, remove_edges=remove_edges, device=device, low_memory=low_memory)
    preprocessing.run()
    if timeit:
        t1 = time.perf_counter()
        print(f"[timeit] Filter: {t1 - t0:.3f}s")
    segmenting = Label(im_info, otsu_thresh_intensity=otsu_thresh_intensity, threshold=threshold, device=device,
                       low_memory=low_memory)
    segmenting.run()
    if timeit:
        t2 = time.perf_counter()
        print(f"[timeit] Label: {t2 - t1:.3f}s")
        print(f"[timeit] Total: {t2 - t0:.3f}s")
    return im_info
