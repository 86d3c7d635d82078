"""
`Markers`: drop-in for nellie.segmentation.mocap_marking.Markers (reference mocap_marking.py:21-836) on the MI355X
HIP engine -- the stage after Label: distance transform of the labelled objects, their border shell, and the
motion-capture markers (multi-scale LoG peaks of the distance image, intensity-based non-maximum suppression).

Same constructor keywords, same `.run()`, same on-disk products (`im_marker` uint8, `im_distance` float32,
`im_border` uint8 in `im_info.pipeline_paths`).  All three are bit-identical to the reference's numpy path.

3-D volumes and 2-D (`no_z`) images, `use_im='distance'` (the default) and `use_im='frangi'`.
Differences (documented in DESIGN.md): `low_memory` / `max_chunk_voxels` are accepted and ignored (the reference's
chunked path gives the same results as its full-frame path); there is no CPU engine behind this class
(`device="cpu"` raises).
"""
from __future__ import annotations

import numpy as np

from nellie_amd.pipeline import FramePipeline, marker_sigmas
from nellie_amd.utils import adaptive_run
from nellie_amd.utils.base_logger import logger


class Markers:
    def __init__(self, im_info, num_t=None, min_radius_um=0.20, max_radius_um=1, use_im="distance", num_sigma=5,
                 viewer=None, prefer_gpu=True, peak_min_distance=2, device="auto", low_memory=False,
                 max_chunk_voxels=int(1e6), device_index: int = 0, devices=None, shard=None):
        """The reference's keywords, plus where a frame runs: `device_index`, `devices=[...]` (the frame as Z slabs over these
        GPUs) and `shard="env"` (this process is one rank of a multi-process run), as for Filter and Label.  Every step of this
        stage has bounded support (the distance transform is clamped at 2 max_radius_px, the widest LoG kernel reaches
        truncate * sigma_z planes, peak tests +- 1, suppression +- peak_min_distance), so a slab computed together with that many
        planes of labels and intensities on each side -- read from the files, nothing is exchanged -- gives the whole-volume
        result on its own planes; frames beyond a context's 2^31 voxels are cut that way without being asked."""
        self.im_info = im_info
        self.devices = list(devices) if devices else None
        self.shard = shard
        self.num_t = num_t
        if self.im_info.no_t:
            self.num_t = 1
        elif num_t is None:
            self.num_t = im_info.shape[im_info.axes.index("T")]
        x_res = self.im_info.dim_res.get("X") or 1.0
        z_res = self.im_info.dim_res.get("Z") or x_res
        self.z_ratio = float(z_res) / float(x_res) if not self.im_info.no_z else 1.0
        self.min_radius_um = max(min_radius_um, float(x_res))        # mocap_marking.py:128-132
        self.max_radius_um = max_radius_um
        self.min_radius_px = self.min_radius_um / float(x_res)
        self.max_radius_px = self.max_radius_um / float(x_res)
        self.use_im = use_im
        self.num_sigma = num_sigma
        self.sigmas = []
        self.shape = ()
        self.im_memmap = self.im_frangi_memmap = self.label_memmap = None
        self.im_marker_memmap = self.im_distance_memmap = self.im_border_memmap = None
        self.debug = None
        self.viewer = viewer
        dev = str(device or "auto").lower()
        if dev not in ("auto", "cpu", "gpu", "cuda", "hip"):
            raise ValueError(f"Unsupported device '{device}'. Use 'auto', 'cpu', or 'gpu'.")
        if dev == "cpu" or (dev == "auto" and not prefer_gpu):
            raise RuntimeError("nellie_amd provides the MI355X HIP backend only: device='cpu' is not available "
                               "(no CPU fallback exists in this package; use the reference implementation on CPU)")
        if not adaptive_run.gpu_available():
            raise RuntimeError("GPU backend requested but no HIP device / libnellie_hip.so is available.")
        self.device = device or "auto"
        self.device_type = "hip"
        self.device_index = int(device_index)
        self.use_gpu = True
        self.peak_min_distance = peak_min_distance
        self.low_memory = bool(low_memory)
        self.max_chunk_voxels = int(max_chunk_voxels)
        self.truncate = 4.0
        self._pipeline = None
        self._pipeline_key = None

    # ------------------------------------------------------------------ setup (mocap_marking.py:329-417)
    def _set_default_sigmas(self):
        logger.debug("Setting sigma values.")
        self.sigmas, _ = marker_sigmas(self.im_info.dim_res, self.min_radius_um, self.max_radius_um, self.num_sigma)

    def _get_t(self):
        if self.num_t is None:
            self.num_t = 1 if self.im_info.no_t else self.im_info.shape[self.im_info.axes.index("T")]

    def _allocate_memory(self):
        logger.debug("Allocating memory for mocap marking.")
        self.label_memmap = self.im_info.get_memmap(self.im_info.pipeline_paths["im_instance_label"])
        self.im_memmap = self.im_info.get_memmap(self.im_info.im_path)
        if self.use_im == "frangi":
            self.im_frangi_memmap = self.im_info.get_memmap(self.im_info.pipeline_paths["im_preprocessed"])
        self.shape = self.label_memmap.shape
        paths = self.im_info.pipeline_paths
        _, spec = self._slab_plan(self.shape[1:]) if (len(self.shape) == 4 and self.sigmas) else (1, None)
        self._spec = spec
        self._rdv = None
        if spec is not None and spec.world > 1:                # a multi-process run meets through this launch's file rendezvous
            from nellie_amd.rendezvous import rendezvous_for
            self._rdv = rendezvous_for(spec, os.path.dirname(paths["im_border"]))
        if spec is not None and spec.rank != 0:                # rank 0 creates the files, the others map them
            self._rdv.wait("markers_files_ready")
            self.im_marker_memmap = self.im_info.get_memmap(paths["im_marker"])
            self.im_distance_memmap = self.im_info.get_memmap(paths["im_distance"])
            self.im_border_memmap = self.im_info.get_memmap(paths["im_border"])
            return
        alloc = self.im_info.allocate_memory
        self.im_marker_memmap = alloc(paths["im_marker"], dtype="uint8", description="mocap marker image", return_memmap=True)
        self.im_distance_memmap = alloc(paths["im_distance"], dtype="float32", description="distance transform image", return_memmap=True)
        self.im_border_memmap = alloc(paths["im_border"], dtype="uint8", description="border image", return_memmap=True)
        if self._rdv is not None:
            self._rdv.publish("markers_files_ready")

    def _get_pipeline(self, shape) -> FramePipeline:
        key = tuple(int(s) for s in shape)
        if self._pipeline is None or self._pipeline_key != key:
            self.close()
            self._pipeline = FramePipeline(key, device=self.device_index)
            self._pipeline_key = key
        return self._pipeline

    def close(self):
        if self._pipeline is not None:
            self._pipeline.close()
            self._pipeline = None

    # ------------------------------------------------------------------ frames (mocap_marking.py:648-703)
    def _run_frame_impl(self, t, low_memory=False, chunk_voxels=None):
        """(marker uint8, distance float32, border uint8) of frame t.  `low_memory` / `chunk_voxels` select the
        reference's chunked CPU path, whose results equal the full-frame ones (tests/test_mocap_marking.py:34-58);
        here the frame is always processed whole on the device."""
        logger.info(f"Running motion capture marking, volume {t}/{self.num_t - 1}")
        intensity = np.asarray(self.im_memmap[t])
        labels = np.asarray(self.label_memmap[t])
        use_image = None
        if self.use_im == "frangi":                                   # mocap_marking.py:675-679
            if self.im_frangi_memmap is None:
                raise RuntimeError("Frangi image requested for peak detection but not available.")
            use_image = np.asarray(self.im_frangi_memmap[t], dtype=np.float32)
        elif self.use_im != "distance":
            raise ValueError(f"Unknown use_im value: {self.use_im}")
        pipe = self._get_pipeline(labels.shape)
        pipe.markers(self.im_info.dim_res, labels=labels, intensity=intensity, min_radius_um=self.min_radius_um,
                     max_radius_um=self.max_radius_um, num_sigma=self.num_sigma, peak_min_distance=self.peak_min_distance,
                     use_image=use_image)
        return tuple(a.reshape(labels.shape) for a in pipe.download_markers())

    def _run_frame(self, t):
        return self._run_frame_impl(t)

    # ------------------------------------------------------------------ Z slabs
    def _slab_halo(self) -> int:
        """Planes of context a slab needs on each side for its own planes to be exact (see the constructor)."""
        rz = max([int(self.truncate * (float(s) / self.z_ratio) + 0.5) for s in self.sigmas] or [0])
        return int(np.ceil(2.0 * self.max_radius_px)) + rz + 1 + int(self.peak_min_distance) + 1

    def _slab_plan(self, shape3):
        """(slab count, this process's (rank, world) or None) for a frame of this shape."""
        import os
        from nellie_amd.engine import ShardSpec, slabs_needed
        shard = self.shard if self.shard is not None else (os.environ.get("NELLIE_SHARD") or None)
        spec = ShardSpec.from_env() if shard == "env" else shard
        if self.im_info.no_z:
            return 1, None
        if spec is not None and spec.world > 1:
            return spec.world, spec
        w = slabs_needed(shape3, self._slab_halo(), len(self.devices) if self.devices else 1)
        return max(w, int(os.environ.get("NELLIE_FORCE_SLABS", "0") or 0), 1), None

    def _run_slab(self, t, o0, o1, device):
        """Planes [o0, o1) of frame t: the stage on the slab extended by the halo (clipped at the volume's faces, where the
        real boundary rules then apply), owned planes written straight into the three output maps."""
        nz = self.label_memmap.shape[1]
        h = self._slab_halo()
        e0, e1 = max(0, o0 - h), min(nz, o1 + h)
        labels = np.ascontiguousarray(self.label_memmap[t, e0:e1])
        intensity = np.ascontiguousarray(self.im_memmap[t, e0:e1])
        use_image = None
        if self.use_im == "frangi":
            use_image = np.ascontiguousarray(self.im_frangi_memmap[t, e0:e1], dtype=np.float32)
        elif self.use_im != "distance":
            raise ValueError(f"Unknown use_im value: {self.use_im}")
        pipe = FramePipeline(labels.shape, device=device)
        try:
            pipe.markers(self.im_info.dim_res, labels=labels, intensity=intensity, min_radius_um=self.min_radius_um,
                         max_radius_um=self.max_radius_um, num_sigma=self.num_sigma, peak_min_distance=self.peak_min_distance,
                         use_image=use_image)
            marker, distance, border = (a.reshape(labels.shape) for a in pipe.download_markers())
        finally:
            pipe.close()
        self.im_marker_memmap[t, o0:o1] = marker[o0 - e0:o1 - e0]
        self.im_distance_memmap[t, o0:o1] = distance[o0 - e0:o1 - e0]
        self.im_border_memmap[t, o0:o1] = border[o0 - e0:o1 - e0]

    def _run_frame_as_slabs(self, t, n_slabs, spec):
        from nellie_amd.sharded import slab_range
        nz = self.label_memmap.shape[1]
        logger.info(f"Running motion capture marking, volume {t}/{self.num_t - 1}, as {n_slabs} Z slabs")
        if spec is not None:                                   # one rank of a multi-process run: its own slab
            self._run_slab(t, *slab_range(nz, n_slabs, spec.rank), spec.device)
            return
        devs = self.devices or [self.device_index]
        if len(devs) == 1:
            for r in range(n_slabs):
                self._run_slab(t, *slab_range(nz, n_slabs, r), devs[0])
            return
        import threading
        errs = []

        def work(k):                                           # GPU k takes a contiguous block of slabs, one after the other
            try:
                for r in range(n_slabs):
                    if r * len(devs) // n_slabs == k:
                        self._run_slab(t, *slab_range(nz, n_slabs, r), devs[k])
            except BaseException as exc:  # noqa: BLE001
                errs.append(exc)
        ts = [threading.Thread(target=work, args=(k,)) for k in range(len(devs))]
        for th in ts:
            th.start()
        for th in ts:
            th.join()
        if errs:
            raise errs[0]

    def _distance_im(self, mask):
        """mocap_marking.py:419-450: (float32 distance to the background clamped at 2 * max_radius_px, bool border shell)."""
        mask = np.asarray(mask).astype(bool)
        pipe = self._get_pipeline(mask.shape)
        ctx = pipe.ctx
        ctx.markers_begin(pipe._as_frame(mask.astype(np.int32)), pipe._as_frame(np.zeros(mask.shape, np.float32)))
        ctx.markers_distance(np.float32(self.max_radius_px * 2.0))
        ctx.markers_finish(0)
        _, dist, border = ctx.markers_store(marker=False)
        return dist.reshape(mask.shape), border.reshape(mask.shape).astype(bool)

    def _run_mocap_marking(self):
        for t in range(self.num_t):
            if self.viewer is not None:
                self.viewer.status = f"Mocap marking. Frame: {t + 1} of {self.num_t}."
            n_slabs, spec = self._slab_plan(self.label_memmap.shape[1:]) if self.label_memmap.ndim == 4 else (1, None)
            if n_slabs > 1:
                self._run_frame_as_slabs(t, n_slabs, spec)
                for mm in (self.im_marker_memmap, self.im_distance_memmap, self.im_border_memmap):
                    if hasattr(mm, "flush"):
                        mm.flush()
                continue
            marker, distance, border = self._run_frame(t)
            if self.im_info.no_t or self.num_t == 1 and self.im_marker_memmap.ndim == marker.ndim:
                self.im_marker_memmap[...] = marker
                self.im_distance_memmap[...] = distance
                self.im_border_memmap[...] = border
            else:
                self.im_marker_memmap[t] = marker
                self.im_distance_memmap[t] = distance
                self.im_border_memmap[t] = border
            for mm in (self.im_marker_memmap, self.im_distance_memmap, self.im_border_memmap):
                if hasattr(mm, "flush"):
                    mm.flush()

    def run(self):
        logger.info("Running Markers (HIP).")
        self._get_t()
        self._set_default_sigmas()
        self._allocate_memory()
        try:
            self._run_mocap_marking()
            rdv = getattr(self, "_rdv", None)
            if rdv is not None:                                # EVERY rank returns only when every rank has flushed its planes
                rdv.barrier("markers_done")
                if rdv.rank == 0:
                    rdv.remove("markers_files_ready")
        finally:
            self.close()
