"""
`Label`: drop-in for nellie.segmentation.labelling.Label (reference labelling.py:17-778)
on the MI355X HIP engine.

Same constructor keywords, same `.run()`, same product
(`im_info.pipeline_paths['im_instance_label']`, int32, ids 1..K per frame in raster order of
each object's first voxel, exactly scipy.ndimage.label's numbering).

Each Frangi frame is uploaded once; the log-domain threshold is taken from a strided sample
gathered on the device (labelling.py:385-455), and thresholding, hole filling, both labelling
passes, the small-object filter and the majority smoothing run there too (nl_label_run).  `chunk_z` / `low_memory` are accepted and ignored: the reference's
Z-chunked mode is not equivalent to its full-volume mode (per-chunk hole filling and area
filter); this backend always produces the full-volume result.
"""
from __future__ import annotations

import os

import numpy as np

from nellie_amd.pipeline import FramePipeline, min_area_pixels_of
from nellie_amd.utils import adaptive_run
from nellie_amd.utils.base_logger import logger
from nellie_amd.utils.gpu_functions import otsu_threshold

_UNSET = object()


class Label:
    def __init__(self, im_info,
                 num_t=None,
                 threshold=None,
                 otsu_thresh_intensity=False,
                 viewer=None,
                 chunk_z=None,
                 flush_interval=1,
                 min_radius_um=0.25,
                 threshold_sampling_pixels=1_000_000,
                 histogram_nbins=256,
                 device="auto",
                 low_memory: bool = False,
                 max_chunk_voxels: int = int(1e6),
                 device_index: int = 0,
                 devices=None,
                 shard=None):
        """The reference's keywords, plus where a frame runs -- `device_index`, `devices`, `shard`: as for Filter
        (nellie_amd/segmentation/filtering.py, nellie_amd/engine.py)."""
        self.im_info = im_info
        self.device = device
        self.device_type = self._resolve_backend(device)
        self.device_index = int(device_index)
        self.devices = list(devices) if devices else None
        self.shard = shard
        self._engine = None
        self.num_t = num_t
        if num_t is None and not self.im_info.no_t:
            self.num_t = im_info.shape[im_info.axes.index('T')]
        self.threshold = threshold
        self.otsu_thresh_intensity = otsu_thresh_intensity
        self.im_memmap = None
        self.frangi_memmap = None
        self.semantic_mask_memmap = None
        self.instance_label_memmap = None
        self.shape = ()
        self.debug = {}
        self.viewer = viewer
        self.chunk_z = None           # accepted, ignored (see module docstring)
        self._user_chunk_z = chunk_z
        if chunk_z is not None or low_memory:
            logger.info('Label: chunk_z / low_memory are accepted and ignored by the HIP backend: it always produces the '
                        'full-volume result (the reference\'s chunked mode fills holes and filters areas per chunk).')
        self.flush_interval = max(1, int(flush_interval))
        min_radius_um = float(min_radius_um)
        x_res = self.im_info.dim_res.get("X") or 1.0
        self.min_radius_um = max(min_radius_um, float(x_res))
        self.threshold_sampling_pixels = int(threshold_sampling_pixels)
        self.histogram_nbins = int(histogram_nbins)
        self.eps = 1e-8
        self.low_memory = bool(low_memory)
        self.max_chunk_voxels = int(max_chunk_voxels)
        self.ndim = 2 if self.im_info.no_z else 3
        self.min_area_pixels = self._compute_min_area_pixels()
        self._pipeline = None

    def _resolve_backend(self, device):
        """labelling.py:115-154 with HIP in the role of CuPy."""
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

    def _compute_min_area_pixels(self):
        """labelling.py:209-219 (min_radius_um already clamped to >= X resolution, :95-97)."""
        return min_area_pixels_of(self.im_info.dim_res, self.min_radius_um, no_z=self.im_info.no_z)

    def _get_pipeline(self, shape3) -> FramePipeline:
        if self._pipeline is None or self._pipeline.shape != tuple(shape3):
            if self._pipeline is not None:
                self._pipeline.close()
            self._pipeline = FramePipeline(shape3, device=self.device_index)
        return self._pipeline

    def _shard_spec(self):
        from nellie_amd.engine import ShardSpec
        shard = self.shard if self.shard is not None else (os.environ.get("NELLIE_SHARD") or None)
        if isinstance(shard, str):
            if shard != "env":
                raise ValueError("shard must be 'env' or an engine.ShardSpec")
            shard = ShardSpec.from_env(rendezvous_dir=os.path.dirname(self.im_info.pipeline_paths["im_instance_label"]))
        return shard

    def _engine_params(self):
        from nellie_amd.pipeline import FilterParams
        return FilterParams(dim_res=self.im_info.dim_res)

    def _get_engine(self, shape3):
        """Z slabs of a frame of this shape (nellie_amd/engine.py; one ghost plane per side is all Label needs)."""
        from nellie_amd.engine import make_engine
        key = tuple(int(s) for s in shape3)
        if self._engine is None or self._engine_key != key:
            if self._engine is not None:
                self._engine.close()
            self._engine = make_engine(key, self._engine_params(), device_index=self.device_index, devices=self.devices,
                                       shard=self._shard_spec(), label_only=True)
            self._engine_key = key
        return self._engine

    def close(self):
        if self._pipeline is not None:
            self._pipeline.close()
            self._pipeline = None
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    def _get_t(self):
        if self.num_t is None:
            if self.im_info.no_t:
                self.num_t = 1
            else:
                self.num_t = self.im_info.shape[self.im_info.axes.index('T')]

    def _allocate_memory(self, engine=None):
        """labelling.py:337-353.  Multi-process runs: rank 0 creates the label file, the others map it once it exists."""
        logger.debug('Allocating memory for semantic segmentation.')
        self.im_memmap = self.im_info.get_memmap(self.im_info.im_path)
        self.frangi_memmap = self.im_info.get_memmap(self.im_info.pipeline_paths['im_preprocessed'])
        self.shape = self.frangi_memmap.shape
        path = self.im_info.pipeline_paths['im_instance_label']
        multi = engine is not None and engine.kind == "rank-slab"
        if not multi or engine.spec.rank == 0:
            self.instance_label_memmap = self.im_info.allocate_memory(
                path, dtype='int32', description='instance segmentation', return_memmap=True)
        if multi:
            engine.barrier()
            if engine.spec.rank != 0:
                self.instance_label_memmap = self.im_info.get_memmap(path)

    def _get_frame_views(self, t):
        return self.im_memmap[t, ...], self.frangi_memmap[t, ...]

    def _write_labels_for_frame(self, t, labels):
        self.instance_label_memmap[t, ...] = labels

    # ------------------------------------------------------------------ thresholds
    # The Frangi frame is uploaded once per frame and everything that looks at it does so on the device: the optional
    # intensity mask is applied first (labelling.py:550-552), and the strided positive sample the log-domain threshold
    # is taken from (labelling.py:385-455) is then gathered from the masked, resident frame -- a voxel the mask removed
    # is 0 and therefore not a positive sample, which is exactly the reference's `(sample > 0) & (mask > thresh)`.
    # Only the intensity Otsu threshold (off by default) samples the ORIGINAL image, in its own dtype, on the host.
    def _positive_stride_sample(self, image):
        """labelling.py:385-438 without mask arguments, for a host image: the positive values among every
        (size // threshold_sampling_pixels)-th element; if there are none, among the elements half a stride further;
        if there are none either, all positive values."""
        flat = np.asarray(image).reshape(-1)
        if flat.size == 0:
            return flat
        stride = max(int(flat.size) // max(1, self.threshold_sampling_pixels), 1)
        for start in ([0] if stride == 1 else [0, stride // 2]):
            picked = flat[start::stride]
            picked = picked[picked > 0]
            if picked.size > 0 or stride == 1:
                return picked
        return flat[flat > 0] if float(flat.max()) > 0 else flat[:0]

    def _compute_intensity_otsu_threshold(self, frame):
        """labelling.py:457-465."""
        values = self._positive_stride_sample(frame)
        if values.size == 0:
            return None
        return otsu_threshold(np.asarray(values), nbins=self.histogram_nbins)[0]

    def _intensity_threshold(self, original_view):
        """labelling.py:513-520: Otsu of the original image, the user's fixed threshold, or None."""
        if self.otsu_thresh_intensity:
            found = self._compute_intensity_otsu_threshold(original_view)
            return 0 if found is None else found
        return self.threshold

    def _upload_frame(self, original_view, frangi_view, intensity_thresh):
        """Frangi frame of one time point -> HBM, intensity-masked when an intensity threshold is in force."""
        frangi3 = self._as3d(frangi_view)
        pipe = self._get_pipeline(frangi3.shape)
        pipe.upload_frangi(frangi3)
        if intensity_thresh is not None:
            orig3 = self._as3d(original_view)
            pipe.ctx.label_intensity_mask(orig3, self._effective_threshold(orig3, intensity_thresh))
        return pipe

    def _compute_frangi_threshold(self, frame, mask_frame=None, mask_thresh=None):
        """labelling.py:440-455 for a host Frangi frame (optionally masked by `mask_frame > mask_thresh`)."""
        pipe = self._upload_frame(mask_frame, frame, mask_thresh if mask_frame is not None else None)
        return pipe.frangi_threshold(self.threshold_sampling_pixels, self.histogram_nbins)

    def _compute_frame_thresholds(self, original_view, frangi_view):
        """labelling.py:511-532: (intensity threshold or None, Frangi threshold or None)."""
        intensity_thresh = self._intensity_threshold(original_view)
        return intensity_thresh, self._compute_frangi_threshold(frangi_view, original_view if intensity_thresh is not None else None,
                                                                intensity_thresh)

    # ------------------------------------------------------------------ frames
    @staticmethod
    def _as3d(a):
        a = np.asarray(a)
        return a[None, ...] if a.ndim == 2 else a

    @staticmethod
    def _effective_threshold(original: np.ndarray, thresh) -> float:
        """The value `original > thresh` really compares against under numpy's promotion rules:
        python scalars are weak (cast to the array's float dtype); numpy scalars promote with the array."""
        if isinstance(thresh, (np.generic, np.ndarray)):
            common = np.result_type(original.dtype, np.asarray(thresh).dtype)
            return float(np.asarray(thresh).astype(common))
        if original.dtype.kind == "f":
            return float(original.dtype.type(thresh))
        return float(thresh)

    def _label_resident(self, pipe, frangi_view, frangi_thresh):
        pipe.label(frangi_thresh, self.min_area_pixels, fill_holes=not self.im_info.no_z)
        labels = pipe.download_labels()
        return labels[0] if np.asarray(frangi_view).ndim == 2 else labels

    def _run_frame_full_volume(self, t, original_view, frangi_view, intensity_thresh, frangi_thresh):
        """labelling.py:538-556: int32 labels of frame t (inputs are never modified)."""
        logger.info(f'Running semantic segmentation, volume {t}/{(self.num_t or 1) - 1}')
        pipe = self._upload_frame(original_view, frangi_view, intensity_thresh)
        return self._label_resident(pipe, frangi_view, frangi_thresh)

    def _get_labels(self, frame, frangi_thresh=_UNSET):
        """labelling.py:467-509: (mask, labels) for a host Frangi frame."""
        if frangi_thresh is _UNSET:
            frangi_thresh = self._compute_frangi_threshold(frame)
        labels = self._run_frame_full_volume(0, None, frame, None, frangi_thresh)
        return labels > 0, labels

    def _run_segmentation(self):
        """labelling.py:697-734; the frame is uploaded once and both the threshold and the labels come from that copy."""
        for t in range(self.num_t):
            if self.viewer is not None:
                self.viewer.status = f'Extracting organelles. Frame: {t + 1} of {self.num_t}.'
            original_view, frangi_view = self._get_frame_views(t)
            logger.info(f'Running semantic segmentation, volume {t}/{(self.num_t or 1) - 1}')
            from nellie_amd.engine import plan_engine
            shape3 = self._as3d(frangi_view).shape if np.asarray(frangi_view).ndim == 2 else tuple(frangi_view.shape)
            kind = "single" if self.im_info.no_z else plan_engine(shape3, self._engine_params(), self.devices, self._shard_spec(), label_only=True)[0]
            if kind != "single":
                # the intensity threshold comes from a strided sample of the ORIGINAL image on the host (every rank reads the same
                # file and finds the same value); each slab then masks the planes it owns (labelling.py:513-520, 550-552)
                intensity_thresh = self._intensity_threshold(original_view)
                engine = self._get_engine(shape3)
                engine.upload_frangi(frangi_view)
                if intensity_thresh is not None:
                    engine.intensity_mask(original_view, self._effective_threshold(np.asarray(original_view[:1]), intensity_thresh))
                frangi_thresh = engine.frangi_threshold(self.threshold_sampling_pixels, self.histogram_nbins)
                engine.label(frangi_thresh, self.min_area_pixels, fill_holes=True)
                engine.download_labels(out=self.instance_label_memmap[t, ...])
                if (t + 1) % self.flush_interval == 0:
                    self.instance_label_memmap.flush()
                continue
            pipe = self._upload_frame(original_view, frangi_view, self._intensity_threshold(original_view))
            frangi_thresh = pipe.frangi_threshold(self.threshold_sampling_pixels, self.histogram_nbins)
            self._write_labels_for_frame(t, self._label_resident(pipe, frangi_view, frangi_thresh))
            if (t + 1) % self.flush_interval == 0:
                self.instance_label_memmap.flush()
        self.instance_label_memmap.flush()
        if self._engine is not None:
            self._engine.barrier()

    def run(self):
        """labelling.py:736-778."""
        logger.info('Running semantic segmentation.')
        adaptive_run.normalize_device(self.device)
        try:
            self._get_t()
            engine = None
            if self._shard_spec() is not None and not self.im_info.no_z:   # the communicator first: the ranks agree on the files through it
                frangi = self.im_info.get_memmap(self.im_info.pipeline_paths['im_preprocessed'])
                engine = self._get_engine(frangi.shape[1:])
            self._allocate_memory(engine)
            self._run_segmentation()
        finally:
            self.close()
