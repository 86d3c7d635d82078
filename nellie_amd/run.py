"""
`run()`: the hot-path half of nellie.run.run (reference nellie/run.py:18-130) -- Filter then Label on the
MI355X engine, same keyword names, same `timeit` prints.  The later stages (Network, Markers, tracking,
Hierarchy) are Nellie's own and consume the two files this writes; `markers=True` also runs this package's
Markers stage (reference run.py:88-89; it does not depend on Network) and writes im_marker / im_distance /
im_border.
"""
from __future__ import annotations

import os
import time

from nellie_amd.segmentation.filtering import Filter
from nellie_amd.segmentation.labelling import Label


def _wait_for(path, timeout_s=600.0):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout_s:
            raise TimeoutError(f"no {path} after {timeout_s} s")
        time.sleep(0.02)


def _build_im_info(file_info, shard):
    """ImInfo(file_info) as run.py:49 -- in a multi-process launch (shard="env") rank 0 first, so that the canonical copy of the
    input (verifier.py:620-695) is written once; the other ranks then find it complete (ome_tiff.create moves files into place
    atomically) and reuse it instead of re-creating it under readers."""
    from nellie_amd.engine import ShardSpec
    from nellie_amd.im_info.verifier import ImInfo
    spec = ShardSpec.from_env() if shard == "env" or (shard is None and os.environ.get("NELLIE_SHARD") == "env") else None
    if spec is None or spec.world <= 1:
        return ImInfo(file_info)
    import tempfile
    from nellie_amd.rendezvous import rendezvous_for
    src = getattr(file_info, "filepath", None) or (file_info if isinstance(file_info, (str, os.PathLike)) else None)
    d = getattr(file_info, "output_dir", None) or (os.path.dirname(os.path.abspath(os.fspath(src))) if src else tempfile.gettempdir())
    rdv = rendezvous_for(spec, d)
    if spec.rank == 0:
        im_info = ImInfo(file_info)
        rdv.publish("im_info_ready")
    else:
        rdv.wait("im_info_ready")
        im_info = ImInfo(file_info)
    rdv.barrier("im_info_built")
    if spec.rank == 0:
        rdv.remove("im_info_ready")
    return im_info


def run_streamed(im_info, viewer=None, device_index=0, devices=None, shard=None):
    """
    Filter + Label of every frame with the three legs overlapped (nellie_amd/streaming.py): same two files as
    `run()`, default parameters only (no remove_edges / intensity thresholds), 3-D frames.

    Frames of a T stack are independent ("for C5 prefer frame-parallel"):
      devices=[...]  one streamer per GPU in this process (one host thread each), GPU k takes frames k, k + N, ...;
      shard="env"    this process is one rank of a multi-process job (WORLD_SIZE / RANK / LOCAL_RANK): it takes frames
                     RANK, RANK + WORLD_SIZE, ... on GPU LOCAL_RANK; rank 0 creates the two files, the others map them.
    No data crosses between the GPUs; the files are the ones a single GPU writes.
    """
    from nellie_amd.engine import ShardSpec
    from nellie_amd.pipeline import FilterParams
    from nellie_amd.streaming import StreamedSegmenter
    if im_info.no_z:
        raise NotImplementedError("streaming covers 3-D frames")
    spec = ShardSpec.from_env() if shard == "env" else shard
    rank, world = (spec.rank, spec.world) if spec is not None else (0, 1)
    fr_path, lab_path = im_info.pipeline_paths["im_preprocessed"], im_info.pipeline_paths["im_instance_label"]
    # a multi-process run meets through files that carry this launch's nonce (nellie_amd/rendezvous.py): the markers of a launch
    # that died on the same MASTER_PORT are never mistaken for this one's
    rdv = None
    if spec is not None and world > 1:
        from nellie_amd.rendezvous import rendezvous_for
        rdv = rendezvous_for(spec, os.path.dirname(lab_path))
    im = im_info.get_memmap(im_info.im_path)
    if rank == 0:
        fr = im_info.allocate_memory(fr_path, dtype="float32", description="frangi filtered im", return_memmap=True)
        lab = im_info.allocate_memory(lab_path, dtype="int32", description="instance segmentation", return_memmap=True)
        if rdv:
            rdv.publish("streamed_files_ready")
    else:
        rdv.wait("streamed_files_ready")
        fr, lab = im_info.get_memmap(fr_path), im_info.get_memmap(lab_path)
    devs = [spec.device] if spec is not None else ([int(d) for d in devices] if devices else [int(device_index)])
    params = FilterParams(dim_res=im_info.dim_res)
    num_t = im.shape[0]

    def lane(k, n_lanes, first, device):
        """frames first + k, first + k + n_lanes, ... on one GPU"""
        sl = slice(first + k, None, n_lanes)
        if len(range(num_t)[sl]) == 0:
            return
        seg = StreamedSegmenter(im.shape[1:], im.dtype, params, device=device)
        try:
            def status(t, n):
                if viewer is not None and k == 0:
                    viewer.status = f"Preprocessing + extracting organelles. Frame: {t * n_lanes + 1} of {num_t}."
            seg.run(im[sl], fr[sl], lab[sl], status=status, outputs_zeroed=True)   # both files were created (zero-filled) above
        finally:
            seg.close()

    if len(devs) == 1:
        lane(0, world, rank, devs[0])
    else:
        import threading
        errs = []

        def work(k):
            try:
                lane(k, len(devs), 0, devs[k])
            except BaseException as exc:  # noqa: BLE001
                errs.append(exc)
        ts = [threading.Thread(target=work, args=(k,)) for k in range(len(devs))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs:
            raise errs[0]
    fr.flush(); lab.flush()
    if rdv:                                                    # EVERY rank returns only when every rank has flushed its frames
        rdv.barrier("streamed_done")
        if rank == 0:
            rdv.remove("streamed_files_ready")
    return im_info


def run(file_info, remove_edges=False, otsu_thresh_intensity=False, threshold=None, timeit=False, device="auto",
        low_memory=False, markers=False, devices=None, shard=None):
    """nellie.run.run's signature (run.py:18-26): `file_info` is a FileInfo (this package's or any object with the same
    fields) from which the ImInfo is built as run.py:49 does; an ImInfo (anything with `pipeline_paths`) or a path / array
    is accepted too.  Returns the ImInfo.
    devices=[...]: every 3-D frame runs as Z slabs over these GPUs; shard="env": this process is one rank (WORLD_SIZE / RANK /
    LOCAL_RANK) of a multi-process Z-slab job over RCCL -- see nellie_amd/engine.py.  Frames beyond one context's size are
    cut into slabs without being asked."""
    from nellie_amd.im_info.verifier import ImInfo
    if hasattr(file_info, "pipeline_paths"):
        im_info = file_info
    else:
        im_info = _build_im_info(file_info, shard)
    t0 = time.perf_counter() if timeit else None
    preprocessing = Filter(im_info, remove_edges=remove_edges, device=device, low_memory=low_memory, devices=devices, shard=shard)
    preprocessing.run()
    if timeit:
        t1 = time.perf_counter()
        print(f"[timeit] Filter: {t1 - t0:.3f}s")
    segmenting = Label(im_info, otsu_thresh_intensity=otsu_thresh_intensity, threshold=threshold, device=device,
                       low_memory=low_memory, devices=devices, shard=shard)
    segmenting.run()
    if timeit:
        t2 = time.perf_counter()
        print(f"[timeit] Label: {t2 - t1:.3f}s")
    if markers:
        from nellie_amd.segmentation.mocap_marking import Markers
        Markers(im_info, device=device, low_memory=low_memory, devices=devices, shard=shard).run()
        if timeit:
            print(f"[timeit] Markers: {time.perf_counter() - t2:.3f}s")
    if timeit:
        print(f"[timeit] Total: {time.perf_counter() - t0:.3f}s")
    return im_info
