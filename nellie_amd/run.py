"""
`run()`: the hot-path half of nellie.run.run (reference nellie/run.py:18-130) -- Filter then Label on the
MI355X engine, same keyword names, same `timeit` prints.  The later stages (Network, Markers, tracking,
Hierarchy) are Nellie's own and consume the two files this writes; `markers=True` also runs this package's
Markers stage (reference run.py:88-89; it does not depend on Network) and writes im_marker / im_distance /
im_border.
"""
from __future__ import annotations

import time

from nellie_amd.segmentation.filtering import Filter
from nellie_amd.segmentation.labelling import Label


def run_streamed(im_info, viewer=None, device_index=0):
    """
    Filter + Label of every frame with the three legs overlapped (nellie_amd/streaming.py): same two files as
    `run()`, default parameters only (no remove_edges / intensity thresholds), 3-D frames.
    """
    from nellie_amd.pipeline import FilterParams
    from nellie_amd.streaming import StreamedSegmenter
    if im_info.no_z:
        raise NotImplementedError("streaming covers 3-D frames")
    im = im_info.get_memmap(im_info.im_path)
    fr = im_info.allocate_memory(im_info.pipeline_paths["im_preprocessed"], dtype="float32",
                                 description="frangi filtered im", return_memmap=True)
    lab = im_info.allocate_memory(im_info.pipeline_paths["im_instance_label"], dtype="int32",
                                  description="instance segmentation", return_memmap=True)
    seg = StreamedSegmenter(im.shape[1:], im.dtype, FilterParams(dim_res=im_info.dim_res), device=device_index)
    try:
        def status(t, n):
            if viewer is not None:
                viewer.status = f"Preprocessing + extracting organelles. Frame: {t + 1} of {n}."
        seg.run(im, fr, lab, status=status, outputs_zeroed=True)      # both files were created (zero-filled) a few lines up
    finally:
        seg.close()
    return im_info


def run(file_info, remove_edges=False, otsu_thresh_intensity=False, threshold=None, timeit=False, device="auto",
        low_memory=False, markers=False):
    """nellie.run.run's signature (run.py:18-26): `file_info` is a FileInfo (this package's or any object with the same
    fields) from which the ImInfo is built as run.py:49 does; an ImInfo (anything with `pipeline_paths`) or a path / array
    is accepted too.  Returns the ImInfo."""
    from nellie_amd.im_info.verifier import ImInfo
    im_info = file_info if hasattr(file_info, "pipeline_paths") else ImInfo(file_info)
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
    if markers:
        from nellie_amd.segmentation.mocap_marking import Markers
        Markers(im_info, device=device, low_memory=low_memory).run()
        if timeit:
            print(f"[timeit] Markers: {time.perf_counter() - t2:.3f}s")
    if timeit:
        print(f"[timeit] Total: {time.perf_counter() - t0:.3f}s")
    return im_info
