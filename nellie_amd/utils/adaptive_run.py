"""
Device / memory ladder helpers (reference: nellie/utils/adaptive_run.py:14-141) for the
HIP backend.  Same function names and meaning; "gpu" means the MI355X HIP engine and there
is no CPU engine in this package, so the ladder only ever contains GPU rungs.
"""
from __future__ import annotations

import math

from nellie_amd import hipnative

_ESTIMATED_PEAK_MULTIPLIER = 6.0
_MEMORY_HEADROOM = 0.7


def normalize_device(device):
    """adaptive_run.py:14-20."""
    device = (device or "auto").lower()
    if device in ("cuda", "hip"):         # "hip": the name INTEGRATION.md's dispatch uses for this backend
        device = "gpu"
    if device not in ("auto", "cpu", "gpu"):
        raise ValueError(f"Unsupported device '{device}'. Use 'auto', 'cpu', or 'gpu'.")
    return device


def gpu_available() -> bool:
    """adaptive_run.py:23-31."""
    return hipnative.gpu_available()


def get_gpu_free_bytes(device: int = 0):
    """adaptive_run.py:34-43."""
    try:
        return hipnative.load().device_mem_info(device)[0]
    except Exception:
        return None


def estimate_frame_bytes(im_info):
    """adaptive_run.py:73-85."""
    if im_info is None or getattr(im_info, "axes", None) is None or getattr(im_info, "shape", None) is None:
        return None
    frame_shape = tuple(dim for axis, dim in zip(im_info.axes, im_info.shape) if axis != "T")
    if not frame_shape:
        return None
    try:
        itemsize = im_info.im.dtype.itemsize
    except Exception:
        return None
    return int(math.prod(frame_shape) * itemsize)


def frame_fits_on_device(shape_zyx, device: int = 0, contexts: int = 1) -> bool:
    """The engine needs nl_ctx_bytes(shape) (35 B/voxel: four float32 volumes, three byte volumes and the eigen queue) of HBM
    per context, plus the resident input; the reference's 6x-frame heuristic (adaptive_run.py:88-100) is replaced by the exact figure.
    contexts: how many contexts of that shape are wanted at once (the frame streamer's lanes).  The engine plan asks the same question per
    device for slab layouts (nellie_amd/engine.py: memory_plan)."""
    from nellie_amd import engine
    need = engine.context_bytes(shape_zyx)
    free = get_gpu_free_bytes(device)
    return need is None or free is None or need * max(1, int(contexts)) <= free * engine.HBM_HEADROOM


def is_oom_error(exc: Exception) -> bool:
    """adaptive_run.py:116-127."""
    if isinstance(exc, MemoryError):
        return True
    msg = repr(exc).lower()
    return "out of memory" in msg or "outofmemory" in msg


def is_gpu_unavailable_error(exc: Exception) -> bool:
    """adaptive_run.py:130-141."""
    msg = repr(exc).lower()
    return ("gpu backend requested" in msg or "no hip device" in msg
            or "libnellie_hip.so is not built" in msg)
