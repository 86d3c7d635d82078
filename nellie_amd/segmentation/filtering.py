"""
`Filter`: drop-in for nellie.segmentation.filtering.Filter (reference filtering.py:17-1076)
on the MI355X HIP engine.

Same constructor keywords, same `.run(mask=True)`, same on-disk product
(`im_info.pipeline_paths['im_preprocessed']`, float32, written frame by frame with a flush).
Every full-volume operation runs in libnellie_hip.so; the thresholds between passes are
decided on the host from device histograms (nellie_amd/pipeline.py).

Deliberate differences from the reference (documented in DESIGN.md):
  * the input memmap is never written (the reference's in-place blur clobbers float32 inputs);
  * `device="cpu"` raises: this package has no CPU engine and no CPU fallback;
  * `low_memory` / `max_chunk_voxels` are accepted and ignored: the chunked mode of the reference
    changes the result (per-chunk thresholds); large volumes shard over Z across GPUs instead;
"""
from __future__ import annotations

import os

import numpy as np

from nellie_amd.pipeline import FilterParams, FramePipeline, default_sigmas, sample_strides
from nellie_amd.utils import adaptive_run
from nellie_amd.utils.base_logger import logger


class Filter:
    def __init__(
        self,
        im_info,
        num_t=None,
        remove_edges: bool = False,
        min_radius_um: float = 0.25,
        max_radius_um: float = 1.0,
        alpha_sq: float = 0.5,
        beta_sq: float = 0.5,
        frob_thresh=None,
        frob_thresh_division=2,
        viewer=None,
        device: str = "auto",
        low_memory: bool = False,
        max_chunk_voxels: int = int(1e6),
        max_threshold_samples: int = int(1e6),
        device_index: int = 0,
        devices=None,
        shard=None,
    ):
        """The reference's keywords, plus where a frame runs (nellie_amd/engine.py):
        device_index  the GPU of a single-context run;
        devices       list of GPUs: every 3-D frame is cut into Z slabs over them (one process, one host thread per slab);
        shard         "env" (read WORLD_SIZE / RANK / LOCAL_RANK: this process is one rank of a multi-process Z-slab job, RCCL
                      between the ranks) or an engine.ShardSpec.
        A frame too large for one context (>= 2^31 voxels) is cut into slabs on its own; results never depend on the layout."""
        self.im_info = im_info
        self.device = device
        self.device_type = self._resolve_backend(device)
        self.device_index = int(device_index)
        self.devices = list(devices) if devices else None
        self.shard = shard
        self._engine = None
        self.truncate = 3.0
        if not self.im_info.no_z:
            z_res = self.im_info.dim_res.get("Z") or self.im_info.dim_res.get("X") or 1.0
            x_res = self.im_info.dim_res.get("X") or 1.0
            self.z_ratio = float(z_res) / float(x_res)
        self.num_t = num_t
        if num_t is None and not self.im_info.no_t:
            self.num_t = im_info.shape[im_info.axes.index("T")]
        self.remove_edges = remove_edges
        self.min_radius_um = min_radius_um
        self.max_radius_um = max_radius_um
        self.min_radius_px = self.min_radius_um / self.im_info.dim_res["X"]
        self.max_radius_px = self.max_radius_um / self.im_info.dim_res["X"]
        self.im_memmap = None
        self.frangi_memmap = None
        self.sigma_vec = None
        self.sigmas = None
        self.alpha_sq = float(alpha_sq)
        self.beta_sq = float(beta_sq)
        self.frob_thresh = frob_thresh
        self.frob_thresh_division = frob_thresh_division
        self.viewer = viewer
        self.low_memory = low_memory
        if low_memory:
            logger.info('Filter: low_memory is accepted and ignored by the HIP backend (a frame stays resident in HBM; large '
                        'volumes shard over Z instead of being chunked, which would change the result).')
        self.max_chunk_voxels = int(max_chunk_voxels)
        self.max_threshold_samples = int(max_threshold_samples)
        self.work_dtype = "float32"
        self.out_dtype = "float32"
        self.halo = None
        self._pipeline = None

    # ------------------------------------------------------------------ backend
    def _resolve_backend(self, device):
        """filtering.py:117-159 with HIP in the role of CuPy."""
        device = (device or "auto").lower()
        if device not in ("auto", "cpu", "gpu", "cuda", "hip"):          # "hip": what INTEGRATION.md's dispatch forwards; same engine as "gpu"
            raise ValueError(f"Unsupported device '{device}'. Use 'auto', 'cpu', or 'gpu'.")
        if device == "cpu":
            raise RuntimeError(
                "nellie_amd provides the MI355X HIP backend only: device='cpu' is not available "
                "(no CPU fallback exists in this package; use the reference implementation on CPU)")
        if not adaptive_run.gpu_available():
            raise RuntimeError("GPU backend requested but no HIP device / libnellie_hip.so is available.")
        return "hip"

    def _params(self) -> FilterParams:
        return FilterParams(
            dim_res=self.im_info.dim_res, min_radius_um=self.min_radius_um, max_radius_um=self.max_radius_um,
            alpha_sq=self.alpha_sq, beta_sq=self.beta_sq, frob_thresh=self.frob_thresh,
            frob_thresh_division=self.frob_thresh_division,
            max_threshold_samples=self.max_threshold_samples, sigmas=self.sigmas)

    def _get_pipeline(self, shape) -> FramePipeline:
        key = tuple(int(s) for s in shape)
        if self._pipeline is None or self._pipeline_key != key:
            if self._pipeline is not None:
                self._pipeline.close()
            self._pipeline = FramePipeline(key, device=self.device_index)     # (Z, Y, X), or (Y, X) for no_z images
            self._pipeline_key = key
        return self._pipeline

    def _shard_spec(self):
        from nellie_amd.engine import ShardSpec
        shard = self.shard if self.shard is not None else (os.environ.get("NELLIE_SHARD") or None)
        if isinstance(shard, str):
            if shard != "env":
                raise ValueError("shard must be 'env' or an engine.ShardSpec")
            shard = ShardSpec.from_env(rendezvous_dir=os.path.dirname(self.im_info.pipeline_paths["im_preprocessed"]))
        return shard

    def _get_engine(self, shape):
        """The engine of a frame of this shape (nellie_amd/engine.py): one context, Z slabs in this process, or this rank's slab."""
        from nellie_amd.engine import make_engine
        key = tuple(int(s) for s in shape)
        if self._engine is None or self._engine_key != key:
            if self._engine is not None:
                self._engine.close()
            self._engine = make_engine(key, self._params(), device_index=self.device_index, devices=self.devices, shard=self._shard_spec())
            self._engine_key = key
        return self._engine

    def close(self):
        if self._pipeline is not None:
            self._pipeline.close()
            self._pipeline = None
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    # ------------------------------------------------------------------ setup (filtering.py:201-323)
    def _get_t(self):
        if self.num_t is None:
            if self.im_info.no_t:
                self.num_t = 1
            else:
                self.num_t = self.im_info.shape[self.im_info.axes.index("T")]

    def _allocate_memory(self, engine=None):
        """filtering.py:201-216.  In a multi-process run rank 0 creates the file, the other ranks map it once it exists
        (every rank writes its own planes of every frame)."""
        logger.debug("Allocating memory for frangi filter.")
        if self.im_memmap is None:
            self.im_memmap = self.im_info.get_memmap(self.im_info.im_path)
        self.shape = self.im_memmap.shape
        im_frangi_path = self.im_info.pipeline_paths["im_preprocessed"]
        multi = engine is not None and engine.kind == "rank-slab"
        if not multi or engine.spec.rank == 0:
            self.frangi_memmap = self.im_info.allocate_memory(
                im_frangi_path, dtype=self.out_dtype, description="frangi filtered im", return_memmap=True)
        if multi:
            engine.barrier()
            if engine.spec.rank != 0:
                self.frangi_memmap = self.im_info.get_memmap(im_frangi_path)

    def _get_sigma_vec(self, sigma: float):
        if self.im_info.no_z:
            self.sigma_vec = (float(sigma), float(sigma))
        else:
            self.sigma_vec = (float(sigma) / self.z_ratio, float(sigma), float(sigma))
        return self.sigma_vec

    def _set_default_sigmas(self):
        logger.debug("Setting Frangi sigma values.")
        self.sigmas = default_sigmas(self.im_info.dim_res, self.min_radius_um, self.max_radius_um)
        self.sigma_min, self.sigma_max = None, None
        self.halo = self._compute_halo()

    def _compute_halo(self):
        if not self.sigmas:
            return None
        sigma_vec = self._get_sigma_vec(max(self.sigmas))
        return tuple(int(np.ceil(self.truncate * float(s))) for s in sigma_vec)

    def _sample_strides(self, shape, max_samples):
        return sample_strides(shape, max_samples)

    # ------------------------------------------------------------------ frames
    def _run_frame(self, t, mask=True):
        """filtering.py:910-933: vesselness * masks of frame t (2-D: maximum with the LoG blob response; remove_edges
        if requested) as a host float32 array."""
        logger.info(f"Running Frangi filter on t={t}.")
        frame_cpu = self.im_memmap[t, ...]
        pipe = self._get_pipeline(frame_cpu.shape)
        pipe.compute_vesselness(frame_cpu, self._params(), mask=mask)
        if self.remove_edges:
            pipe.ctx.remove_edges(self.edge_margin)
        out = pipe.download_frangi()
        return out[0] if self.im_info.no_z else out

    def _mask_volume(self, frangi_frame):
        """filtering.py:952-967 for a host frame (device does the work)."""
        frangi_frame = np.asarray(frangi_frame, dtype=np.float32)
        pipe = self._get_pipeline(frangi_frame.shape)
        pipe.upload_frangi(frangi_frame)
        if pipe.mask_volume(self._params()) is None:
            return frangi_frame
        out = pipe.download_frangi()
        return out[0] if frangi_frame.ndim == 2 else out

    edge_margin = 15                                            # filtering.py:978, 988

    def _bbox(self, im):
        """filtering.py:227-250: inclusive index range of the non-zero values along every axis, axis by axis
        (rows, columns[, planes] in the reference's order); all zeros when the image is empty."""
        im = np.asarray(im)
        if im.ndim not in (2, 3):
            logger.warning("Image not 2D or 3D... Cannot get bounding box.")
            return None
        nz = im != 0
        out = []
        for axis in ((0, 1) if im.ndim == 2 else (0, 1, 2)):
            flags = nz.any(axis=tuple(a for a in range(im.ndim) if a != axis))
            if not flags.any():
                return (0,) * (2 * im.ndim)
            out += [int(flags.argmax()), int(flags.size - 1 - flags[::-1].argmax())]
        return tuple(out)

    def _remove_edges(self, frangi_frame):
        """filtering.py:969-1000 for a host frame: the device zeroes min(15, height) rows at both ends of the row span
        of every Z plane (of the image, in 2-D); like the reference, the frame is changed in place when it can be."""
        frame = np.asarray(frangi_frame)
        if frame.size == 0:
            return frangi_frame
        pipe = self._get_pipeline(frame.shape)
        pipe.upload_frangi(frame.astype(np.float32, copy=False))
        pipe.ctx.remove_edges(self.edge_margin)
        out = pipe.download_frangi()
        out = out[0] if frame.ndim == 2 else out
        if isinstance(frangi_frame, np.ndarray) and frangi_frame.flags.writeable and frangi_frame.dtype == np.float32:
            frangi_frame[...] = out
            return frangi_frame
        return out

    def _filter_frame(self, t, mask=True):
        """One frame end to end on the device (filtering.py:1012-1020), one download."""
        frame_cpu = self.im_memmap[t, ...]
        pipe = self._get_pipeline(frame_cpu.shape)
        pipe.filter(frame_cpu, self._params(), mask=mask, remove_edges=bool(self.remove_edges))
        out = pipe.download_frangi()
        return out[0] if self.im_info.no_z else out

    def _run_filter(self, mask=True):
        """filtering.py:1005-1031.  A frame that runs as Z slabs (engine.py) goes through the same loop: every slab takes
        its planes of the input map and puts its planes into the output map."""
        for t in range(self.num_t):
            if self.viewer is not None:
                self.viewer.status = f"Preprocessing. Frame: {t + 1} of {self.num_t}."
            logger.info(f"Running Frangi filter on t={t}.")
            frame_view = self.im_memmap[t, ...]
            from nellie_amd.engine import plan_engine
            if plan_engine(frame_view.shape, self._params(), self.devices, self._shard_spec())[0] == "single":
                filtered_im = self._filter_frame(t, mask=mask)
                if self.im_info.no_t or self.num_t == 1:
                    self.frangi_memmap[:] = filtered_im[:]
                else:
                    self.frangi_memmap[t, ...] = filtered_im
            else:
                engine = self._get_engine(frame_view.shape)
                engine.filter(frame_view, self._params(), mask=mask, remove_edges=bool(self.remove_edges))
                engine.download_frangi(out=self.frangi_memmap[t, ...])
            self.frangi_memmap.flush()
        if self._engine is not None:
            self._engine.barrier()

    def run(self, mask=True):
        """filtering.py:1033-1076.  The ladder has GPU rungs only; OOM re-raises as MemoryError."""
        logger.info("Running Frangi filter.")
        adaptive_run.normalize_device(self.device)
        try:
            self._get_t()
            self._set_default_sigmas()
            self.im_memmap = self.im_info.get_memmap(self.im_info.im_path)
            engine = None
            if self._shard_spec() is not None:       # the communicator has to exist before the ranks can agree on the files
                engine = self._get_engine(self.im_memmap.shape[1:])
            self._allocate_memory(engine)
            self._run_filter(mask=mask)
        finally:
            self.close()
