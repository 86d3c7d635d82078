"""
ctypes binding of libnellie_hip.so (C-ABI: include/nellie_amd.h).

There is no CPU fallback: if the library is missing or no MI355X is present, every
entry point raises.  Status codes map to the exception classes the reference's retry
ladder recognises (nellie/utils/adaptive_run.py:116-141):
    NL_ENODEV -> RuntimeError("GPU backend requested but ...")
    NL_ENOMEM -> MemoryError("... out of memory ...")
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NELLIE_HIP_LIB") or os.path.join(_HERE, "libnellie_hip.so")   # the override is for A/B builds (tools/build_variant.sh)

NL_OK, NL_EINVAL, NL_ENODEV, NL_ENOMEM, NL_EHIP, NL_ESTATE, NL_ECOMM = range(7)
FIELD_GAUSS, FIELD_FROB, FIELD_FRANGI, FIELD_VESSELNESS = 0, 1, 2, 3

DTYPE_CODES = {
    np.dtype(np.uint8): 0, np.dtype(np.int8): 1, np.dtype(np.uint16): 2, np.dtype(np.int16): 3,
    np.dtype(np.uint32): 4, np.dtype(np.int32): 5, np.dtype(np.float32): 6, np.dtype(np.float64): 7,
    np.dtype(np.uint64): 8, np.dtype(np.int64): 9,
}

_ERRLEN = 512
_i64, _int, _f32, _f64 = C.c_int64, C.c_int, C.c_float, C.c_double
_p = C.c_void_p
_ERR = (C.c_char_p, C.c_size_t)

# name -> argtypes WITHOUT the trailing (err, errlen) pair; every listed function returns int
_PROTOS = {
    "nl_device_count": [C.POINTER(_int)],
    "nl_device_mem_info": [_int, C.POINTER(_i64), C.POINTER(_i64)],
    "nl_device_name": [_int, C.c_char_p, C.c_size_t],
    "nl_ctx_create": [C.POINTER(_p), _int, _i64, _i64, _i64, _i64, _i64, _i64, _i64],
    "nl_sync": [_p],
    "nl_filter_load": [_p, _p, _int, _i64, _i64],
    "nl_input_load": [_p, _p, _int, _i64, _i64],
    "nl_filter_begin": [_p],
    "nl_gauss_step": [_p, _p, _int, _p, _int, _p, _int, _i64, _i64],
    "nl_gauss_step_ahead": [_p, _p, _int, _p, _int, _p, _int, _i64, _i64],
    "nl_gauss_commit": [_p],
    "nl_sample_gather": [_p, _int, _i64, _i64, _i64, _p, _i64, C.POINTER(_i64)],
    "nl_sample_gather_positive": [_p, _int, _i64, _i64, _i64, _p, _i64, C.POINTER(_i64)],
    "nl_sample_gather_positive_begin": [_p, _int, _i64, _i64, _i64, C.POINTER(_i64)],
    "nl_sample_gather_positive_end": [_p, _p, _i64, C.POINTER(_i64)],
    "nl_sample_minmax": [_p, _int, _i64, _i64, _i64, C.POINTER(_f32), C.POINTER(_f32), C.POINTER(_i64)],
    "nl_sample_hist": [_p, _int, _i64, _i64, _i64, _p, _int, _p],
    "nl_sample_range_hist": [_p, _int, _i64, _i64, _i64, _int, C.POINTER(_f32), C.POINTER(_f32), C.POINTER(_i64), _p, _p, C.POINTER(_int)],
    "nl_sample_range_hist2": [_p, _int, _int, _i64, _i64, _i64, _int, _p, _p, _p, _p, _p, _p],
    "nl_hist_thresholds": [_p, _p, _int, C.POINTER(_f64), C.POINTER(_f64), C.POINTER(_int)],
    "nl_hist_thresholds_ex": [_p, _p, _int, _int, C.POINTER(_f64), C.POINTER(_f64), C.POINTER(_f64), C.POINTER(_int)],
    "nl_host_hist_thresholds_f32": [_p, _i64, _int, C.POINTER(_f64), C.POINTER(_f64), C.POINTER(_int), _p, _p],
    "nl_tail_enqueue": [_p, _i64, _i64, _i64, _f64],
    "nl_tail_finish": [_p, _int, C.POINTER(_i64), C.POINTER(_f32), C.POINTER(_f32), C.POINTER(_f32), C.POINTER(_f32), C.POINTER(_i64)],
    "nl_mask_volume_dev": [_p, _i64, _i64, _i64, _f64, C.POINTER(_i64), C.POINTER(_f32), C.POINTER(_f32), C.POINTER(_f32), C.POINTER(_f32)],
    "nl_debug_percentile": [_p, _p, _i64, _f64, C.POINTER(_f32), C.POINTER(_f32), C.POINTER(_f32)],
    "nl_positive_samples_world": [_p, _int, _int, _i64, _i64, _i64, _i64, _p, _i64, _p],
    "nl_host_slab_join": [_int, _p, _i64, _i64, C.POINTER(_i64), C.POINTER(_i64), _p, _p, _p, _p],
    "nl_outputs_pack": [_p, _int, _p],
    "nl_outputs_pack_with_label": [_p, _int],
    "nl_outputs_fetch_packed_async": [_p, _p, _i64],
    "nl_outputs_unpack": [_p, _i64, _p, _p, _i64, _int, _int],
    "nl_host_zero": [_p, _i64, _int],
    "nl_hessian_stats": [_p, C.POINTER(_f64), C.POINTER(_f32), C.POINTER(_f32), C.POINTER(_int)],
    "nl_set_frob_norm": [_p, _f32, _f32],
    "nl_vesselness_step": [_p, _f32, _f32, _f32, _int, _f32, _i64, _i64, C.POINTER(_i64)],
    "nl_set_spacing": [_p, C.POINTER(_f64)],
    "nl_vesselness_spec": [_p, C.POINTER(_f64), _f32, _f32, _i64, _i64, C.POINTER(_f32), C.POINTER(_f32),
                           C.POINTER(_int), C.POINTER(_int)],
    "nl_vesselness_resolve": [_p, _f32, _f32, _f32, _int, _f32, C.POINTER(_int), C.POINTER(_i64)],
    "nl_vesselness_count": [_p, C.POINTER(_i64)],
    "nl_set_ndim": [_p, _int],
    "nl_markers_begin": [_p, _p, _p, _int],
    "nl_markers_distance": [_p, _f32, C.POINTER(_i64)],
    "nl_markers_log_step": [_p, C.POINTER(_f64), C.POINTER(_f64), _int, C.POINTER(_f64), C.POINTER(_f64), C.POINTER(_f64),
                            C.POINTER(_f64), _int, _f32],
    "nl_markers_finish": [_p, _int, C.POINTER(_i64)],
    "nl_markers_use_image": [_p, _p],
    "nl_markers_store": [_p, _p, _p, _p],
    "nl_skel_pixel_class": [_p, _p, _p, C.POINTER(_i64)],
    "nl_skel_branch_labels": [_p, _p, _p, C.POINTER(_i64)],
    "nl_mask_volume_fused": [_p, _f32, C.POINTER(_i64)],
    "nl_log2d_step": [_p, C.POINTER(_f64), C.POINTER(_f64), C.POINTER(_f64), C.POINTER(_f64), _int, _f32, _int, _int],
    "nl_log2d_finish": [_p, C.POINTER(_i64)],
    "nl_filter_finish": [_p, _i64, _i64, C.POINTER(_i64)],
    "nl_remove_edges": [_p, _int, C.POINTER(_i64)],
    "nl_planes_get": [_p, _int, _i64, _i64, _p],
    "nl_planes_put": [_p, _int, _i64, _i64, _p],
    "nl_comm_unique_id": [C.c_char_p],
    "nl_comm_loopback_id": [C.c_char_p],
    "nl_comm_init": [_p, _int, _int, C.c_char_p],
    "nl_halo_exchange": [_p, _int, _i64],
    "nl_halo_exchange_at": [_p, _int, _i64, _i64, _int],
    "nl_comm_init2": [_p, _int, _int, C.c_char_p],
    "nl_allreduce": [_p, _p, _i64, _int, _int],
    "nl_mask_volume": [_p, _f32],
    "nl_filter_store": [_p, _p, _i64, _i64],
    "nl_gauss_store": [_p, _p, _i64, _i64],
    "nl_label_load_frangi": [_p, _p, _i64, _i64],
    "nl_label_intensity_mask": [_p, _p, _int, _f64],
    "nl_label_intensity_mask_planes": [_p, _p, _int, _f64, _i64, _i64],
    "nl_flat_sample_gather": [_p, _int, _i64, _i64, _p, _i64, C.POINTER(_i64)],
    "nl_flat_sample_gather_positive": [_p, _int, _i64, _i64, _p, _i64, C.POINTER(_i64)],
    "nl_label_run": [_p, _int, _f32, _i64, _int, C.POINTER(_i64)],
    "nl_label_store": [_p, _p, _i64, _i64],
    "nl_label_pack": [_p, _int, _f32],
    "nl_label_bits_get": [_p, _i64, _i64, _p],
    "nl_label_bits_put": [_p, _i64, _i64, _p],
    "nl_label_bits_allgather": [_p, _p],
    "nl_label_run_global": [_p, _i64, _int, C.POINTER(_i64)],
    "nl_slab_label_pack": [_p, _int, _f32],
    "nl_slab_bits_get": [_p, _int, _i64, _p],
    "nl_slab_bits_put": [_p, _int, _i64, _p],
    "nl_slab_bits_exchange": [_p, _int],
    "nl_slab_phase": [_p, _int, _int, _i64, _p, C.POINTER(_i64), C.POINTER(_i64)],
    "nl_slab_patch": [_p, _i64, _p, _p],
    "nl_slab_apply": [_p, _i64],
    "nl_slab_majority": [_p],
    "nl_slab_number": [_p, _i64, _p, _i64, _p, C.POINTER(_i64), _p],
    "nl_slab_paint": [_p, _i64, _i64, _p, _p],
    "nl_allgather_bytes": [_p, _p, _i64, _p, _i64, _p],
    "nl_allgather_var": [_p, _p, _i64, _p, _p, _p],
    "nl_comm_fuse": [_p, _int],
    "nl_chain_begin": [_p, _int],
    "nl_chain_scale": [_p, _p, _i64, _i64, _i64, _f64, _f64, _f64, _f64, _f64, _i64, _i64],
    "nl_chain_flush": [_p],
    "nl_chain_finish": [_p, _p, _p, _p, _p, _p],
    "nl_chain_log": [_p, _int, _int, _p, _p, _p, _p],
    "nl_pinned_alloc": [C.POINTER(_p), _i64],
    "nl_host_register": [_p, _i64],
    "nl_input_load_async": [_p, _int, _p, _int],
    "nl_input_select": [_p, _int],
    "nl_input_wait": [_p, _int],
    "nl_outputs_stage": [_p, _int],
    "nl_outputs_fetch_async": [_p, _p, _p],
    "nl_outputs_wait": [_p],
    "nl_debug_eig_frangi": [_p, _p, _i64, _int, _f32, _f32, _f32, _p],
    "nl_timer_begin": [_p],
    "nl_timer_end_ms": [_p, C.POINTER(_f32)],
}
# functions without the (err, errlen) tail
_PLAIN = {
    "nl_version": (C.c_char_p, []),
    "nl_ctx_destroy": (_int, [_p]),
    "nl_ctx_bytes": (_i64, [_i64, _i64, _i64]),
    "nl_prof_enable": (_int, [_p, _int]),
    "nl_prof_get": (_int, [_p, C.c_char_p, C.POINTER(_f64), C.POINTER(_i64)]),
    "nl_prof_reset": (_int, [_p]),
    "nl_pinned_free": (_int, [_p]),
    "nl_host_unregister": (_int, [_p]),
    "nl_ctx_info": (_int, [_p, C.c_char_p, C.POINTER(_f64)]),
}
ALL_SYMBOLS = sorted(list(_PROTOS) + list(_PLAIN))


class NellieHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class LibraryUnavailable(RuntimeError):
    """libnellie_hip.so is not built or cannot be loaded (as opposed to an error the loaded library reports)."""


def _raise(code: int, msg: str):
    if code == NL_ENODEV:
        if "GPU backend requested" not in msg:
            msg = "GPU backend requested but " + msg
        raise RuntimeError(msg)
    if code == NL_ENOMEM:
        raise MemoryError(f"HIP out of memory: {msg}")
    if code == NL_EINVAL:
        raise ValueError(msg)
    raise NellieHipError(code, f"libnellie_hip error {code}: {msg}")


class _Lib:
    def __init__(self, path: str):
        self.path = path
        self.cdll = C.CDLL(path)
        for name, argtypes in _PROTOS.items():
            fn = getattr(self.cdll, name)
            fn.restype = _int
            fn.argtypes = list(argtypes) + list(_ERR)
        for name, (res, argtypes) in _PLAIN.items():
            fn = getattr(self.cdll, name)
            fn.restype = res
            fn.argtypes = argtypes

    def call(self, name: str, *args):
        buf = C.create_string_buffer(_ERRLEN)
        rc = getattr(self.cdll, name)(*args, buf, _ERRLEN)
        if rc != NL_OK:
            _raise(rc, buf.value.decode("utf-8", "replace"))

    def version(self) -> str:
        return self.cdll.nl_version().decode()

    def device_count(self) -> int:
        n = _int(0)
        buf = C.create_string_buffer(_ERRLEN)
        rc = self.cdll.nl_device_count(C.byref(n), buf, _ERRLEN)
        return int(n.value) if rc == NL_OK else 0

    def device_mem_info(self, device=0):
        f, t = _i64(0), _i64(0)
        self.call("nl_device_mem_info", device, C.byref(f), C.byref(t))
        return int(f.value), int(t.value)

    def device_name(self, device=0) -> str:
        b = C.create_string_buffer(256)
        self.call("nl_device_name", device, b, 256)
        return b.value.decode()


_LIB = None


def load() -> _Lib:
    """Load libnellie_hip.so or raise.  Never falls back to a CPU implementation."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise LibraryUnavailable(
                f"GPU backend requested but {LIB_PATH} is not built "
                "(run `python -m nellie_amd.build`); nellie_amd has no CPU fallback")
        try:
            _LIB = _Lib(LIB_PATH)
        except OSError as e:
            raise LibraryUnavailable(f"GPU backend requested but {LIB_PATH} cannot be loaded: {e}") from e
    return _LIB


def hist_thresholds(counts, edges, with_variance=False):
    """(triangle, otsu) bin-centre thresholds of a finished histogram (nl_hist_thresholds_ex; gpu_functions.py:36-50, 64-94),
    in the dtype of the edges: float32 edges (float32 data) give float32 centres, anything else is numpy's float64 path
    (integer / float64 data).  with_variance: (triangle, otsu, between-class variance at otsu).
    Raises numpy's ValueError where the reference's triangle construction does (one non-empty bin)."""
    counts = np.ascontiguousarray(counts, dtype=np.int64)
    edges = np.asarray(edges)
    f64 = edges.dtype != np.float32
    edges = np.ascontiguousarray(edges, dtype=np.float64 if f64 else np.float32)
    nbins = int(counts.size)
    assert edges.size == nbins + 1
    tri, otsu, var, status = _f64(0.0), _f64(0.0), _f64(0.0), _int(0)
    load().call("nl_hist_thresholds_ex", _ptr(counts), _ptr(edges), 1 if f64 else 0, nbins, C.byref(tri), C.byref(otsu), C.byref(var),
                C.byref(status))
    if status.value == 1:
        raise ValueError("attempt to get argmax of an empty sequence")
    out_t = np.float64 if f64 else np.float32
    if with_variance:
        return out_t(tri.value), out_t(otsu.value), np.float64(var.value)
    return out_t(tri.value), out_t(otsu.value)


def host_hist_thresholds(values, nbins=256, with_histogram=False):
    """(triangle, otsu) of np.histogram(values, bins=nbins) for float32 host data, computed by the library in one call
    (nl_host_hist_thresholds_f32); with_histogram: (triangle, otsu, counts, edges).  Raises ValueError where numpy would."""
    v = np.ascontiguousarray(values, dtype=np.float32).reshape(-1)
    tri, otsu, status = _f64(0.0), _f64(0.0), _int(0)
    counts = np.zeros(nbins, np.int64) if with_histogram else None
    edges = np.zeros(nbins + 1, np.float32) if with_histogram else None
    load().call("nl_host_hist_thresholds_f32", _ptr(v), int(v.size), int(nbins), C.byref(tri), C.byref(otsu), C.byref(status),
                None if counts is None else _ptr(counts), None if edges is None else _ptr(edges))
    if status.value == 2:
        raise ValueError("autodetected range of [%s, %s] is not finite" % (v.min(), v.max()))
    if status.value == 1:
        raise ValueError("attempt to get argmax of an empty sequence")
    out = (np.float32(tri.value), np.float32(otsu.value))
    return out + (counts, edges) if with_histogram else out


def host_slab_join(blobs):
    """The joined view of the slab tables of all ranks (nl_host_slab_join; host code, no device): blobs = one int32 array per
    rank as Context.slab_phase returns them.  -> (rank, root, val, comp, ncomp): one node per (rank, tree)."""
    world = len(blobs)
    block = max(8, max(int(b.size) for b in blobs))
    flat = np.zeros(world * block, np.int32)
    for r, b in enumerate(blobs):
        flat[r * block:r * block + b.size] = b
    cap = max(1, sum(int(b[:4].sum()) for b in blobs))            # nodes <= entries
    n, nc = _i64(0), _i64(0)
    rank, root, val, comp = np.empty(cap, np.int64), np.empty(cap, np.int32), np.empty(cap, np.int64), np.empty(cap, np.int64)
    load().call("nl_host_slab_join", world, _ptr(flat), block, cap, C.byref(n), C.byref(nc), _ptr(rank), _ptr(root), _ptr(val), _ptr(comp))
    k = int(n.value)
    if k > cap:          # (the library reports this as an error: the arrays above are sized from the entry counts, which bound the nodes)
        raise NellieHipError(NL_EINVAL, f"nl_host_slab_join: {k} nodes for arrays of {cap}")
    return rank[:k], root[:k], val[:k], comp[:k], int(nc.value)


def outputs_unpack(blob, nbytes, frangi, labels=None, zero_fill=True, threads=8):
    """Expand a packed-output blob (Context.outputs_pack) into dense C-contiguous arrays; host code, `threads` host threads."""
    assert frangi.dtype == np.float32 and frangi.flags.c_contiguous and frangi.flags.writeable
    if labels is not None:
        assert labels.dtype == np.int32 and labels.flags.c_contiguous and labels.shape == frangi.shape
    src = blob._p if hasattr(blob, "_p") else _ptr(blob)
    load().call("nl_outputs_unpack", src, int(nbytes), _ptr(frangi), None if labels is None else _ptr(labels),
                int(frangi.size), 1 if zero_fill else 0, int(threads))


def host_zero(array, threads=8):
    """Zero a C-contiguous host array with `threads` host threads (nl_host_zero; the call releases the GIL, so it can run on a
    Python thread beside the GPU calls of the frame whose outputs the array will receive)."""
    assert array.flags.c_contiguous and array.flags.writeable
    load().call("nl_host_zero", _ptr(array), int(array.nbytes), int(threads))


def comm_unique_id(loopback: bool = False) -> bytes:
    """128-byte communicator id (rank 0 creates it and hands it to the other ranks out of band).  loopback=True: an id of
    the in-process loopback transport (nl_comm_loopback_id) -- the ranks are then contexts of this process, one host thread
    each, and every exchange runs through the same library code as over RCCL."""
    buf = C.create_string_buffer(128)
    load().call("nl_comm_loopback_id" if loopback else "nl_comm_unique_id", buf)
    return buf.raw


class PinnedArray:
    """A numpy array over page-locked host memory (hipHostMalloc): asynchronous copies need it."""

    def __init__(self, shape, dtype):
        self.lib = load()
        self.dtype = np.dtype(dtype)
        self.shape = tuple(int(s) for s in shape)
        nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = _p()
        self.lib.call("nl_pinned_alloc", C.byref(p), nbytes)
        self._p = p
        buf = (C.c_char * nbytes).from_address(p.value)
        self.array = np.frombuffer(buf, dtype=self.dtype).reshape(self.shape)

    def free(self):
        if getattr(self, "_p", None):
            self.array = None
            self.lib.cdll.nl_pinned_free(self._p)
            self._p = None

    __del__ = free


class RegisteredArray:
    """Page-locks an existing C-contiguous numpy array for the lifetime of this object (hipHostRegister).
    `ok` is False when the runtime refuses (e.g. some file-backed maps): callers then stage through PinnedArray."""

    def __init__(self, array: np.ndarray):
        self.array = array
        self.ok = False
        self._ptr = None
        if isinstance(array, np.ndarray) and array.flags.c_contiguous and array.flags.writeable and array.nbytes > 0:
            try:
                load().call("nl_host_register", _ptr(array), array.nbytes)
                self.ok = True
                self._ptr = _ptr(array)
            except Exception:
                self.ok = False

    def release(self):
        if self._ptr is not None:
            load().cdll.nl_host_unregister(self._ptr)
            self._ptr = None
            self.ok = False

    __del__ = release


def gpu_available() -> bool:
    """adaptive_run.gpu_available (nellie/utils/adaptive_run.py:23-31) for the HIP backend."""
    try:
        return load().device_count() > 0
    except Exception:
        return False


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_p)


class Context:
    """One device context = one (device, local slab shape) pair (include/nellie_amd.h nl_ctx)."""

    def __init__(self, shape, device=0, gz0=0, gnz=None, own=None):
        self.lib = load()
        nz, ny, nx = (int(s) for s in shape)
        self.shape = (nz, ny, nx)
        self.gz0 = int(gz0)
        self.gnz = int(gnz) if gnz is not None else nz
        self.own = (0, nz) if own is None else (int(own[0]), int(own[1]))
        self.device = int(device)
        h = _p()
        self.lib.call("nl_ctx_create", C.byref(h), self.device, nz, ny, nx, self.gz0, self.gnz,
                      self.own[0], self.own[1])
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.lib.cdll.nl_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _call(self, name, *args):
        if not self._h:
            raise NellieHipError(NL_ESTATE, "context is closed")
        self.lib.call(name, self._h, *args)

    def sync(self):
        self._call("nl_sync")

    # ---------------------------------------------------------------- Filter
    def filter_load(self, frame: np.ndarray, z0=0, z1=None):
        z1 = self.shape[0] if z1 is None else z1
        a = np.ascontiguousarray(frame)
        if a.dtype not in DTYPE_CODES:
            a = a.astype(np.float32)
        assert a.shape == (z1 - z0, self.shape[1], self.shape[2]), (a.shape, self.shape, z0, z1)
        self._call("nl_filter_load", _ptr(a), DTYPE_CODES[a.dtype], z0, z1)

    def input_load(self, frame: np.ndarray, z0=0, z1=None):
        z1 = self.shape[0] if z1 is None else z1
        a = np.ascontiguousarray(frame)
        if a.dtype not in DTYPE_CODES:
            a = a.astype(np.float32)
        assert a.shape == (z1 - z0, self.shape[1], self.shape[2]), (a.shape, self.shape, z0, z1)
        self._call("nl_input_load", _ptr(a), DTYPE_CODES[a.dtype], z0, z1)

    def filter_begin(self):
        self._call("nl_filter_begin")

    def gauss_commit(self):
        self._call("nl_gauss_commit")

    def gauss_step(self, wz, wy, wx, z0=0, z1=None, ahead=False):
        """ahead=True: enqueue the step on the side stream; it becomes current at gauss_commit()."""
        z1 = self.shape[0] if z1 is None else z1
        args = []
        keep = []
        for w in (wz, wy, wx):
            if w is None:
                args += [None, 0]
            else:
                w = np.ascontiguousarray(w, dtype=np.float64)
                keep.append(w)
                args += [_ptr(w), (len(w) - 1) // 2]
        self._call("nl_gauss_step_ahead" if ahead else "nl_gauss_step", *args, z0, z1)

    def sample_gather(self, field, strides):
        sz, sy, sx = (int(s) for s in strides)
        n = _i64(0)
        self._call("nl_sample_gather", field, sz, sy, sx, None, 0, C.byref(n))
        out = np.empty(int(n.value), dtype=np.float32)
        if out.size:
            self._call("nl_sample_gather", field, sz, sy, sx, _ptr(out), out.size, C.byref(n))
        return out

    def sample_gather_positive(self, field, strides):
        """The positive lattice samples (device-side compaction, unspecified order)."""
        sz, sy, sx = (int(s) for s in strides)
        n = _i64(0)
        self._call("nl_sample_gather", field, sz, sy, sx, None, 0, C.byref(n))        # lattice size
        out = np.empty(int(n.value), dtype=np.float32)
        if out.size:
            self._call("nl_sample_gather_positive", field, sz, sy, sx, _ptr(out), out.size, C.byref(n))
        return out[:int(n.value)]

    def sample_gather_positive_begin(self, field, strides):
        """Enqueue the compaction; sample_gather_positive_end() fetches.  Only chain_finish / chain_log in between."""
        sz, sy, sx = (int(s) for s in strides)
        n = _i64(0)
        self._call("nl_sample_gather_positive_begin", field, sz, sy, sx, C.byref(n))
        self._gp_cap = int(n.value)

    def sample_gather_positive_end(self):
        out = np.empty(self._gp_cap, dtype=np.float32)
        n = _i64(0)
        self._call("nl_sample_gather_positive_end", _ptr(out), out.size, C.byref(n))
        return out[:int(n.value)]

    def sample_minmax(self, field, strides):
        sz, sy, sx = (int(s) for s in strides)
        mn, mx, n = _f32(0), _f32(0), _i64(0)
        self._call("nl_sample_minmax", field, sz, sy, sx, C.byref(mn), C.byref(mx), C.byref(n))
        return np.float32(mn.value), np.float32(mx.value), int(n.value)

    def sample_hist(self, field, strides, edges: np.ndarray):
        sz, sy, sx = (int(s) for s in strides)
        e = np.ascontiguousarray(edges, dtype=np.float32)
        counts = np.zeros(e.size - 1, dtype=np.int64)
        self._call("nl_sample_hist", field, sz, sy, sx, _ptr(e), e.size - 1, _ptr(counts))
        return counts

    def sample_range_hist(self, field, strides, nbins=256):
        """sample_minmax + sample_hist in one call (the device forms numpy's float32 bin edges itself).
        Returns (min, max, n_positive, counts, edges, valid): valid 0 = no positive sample, 1 = ok, 2 = range not finite."""
        mn, mx, n, valid = _f32(0), _f32(0), _i64(0), _int(0)
        counts = np.zeros(nbins, np.int64)
        edges = np.zeros(nbins + 1, np.float32)
        sz, sy, sx = (int(s) for s in strides)
        self._call("nl_sample_range_hist", int(field), sz, sy, sx, int(nbins), C.byref(mn), C.byref(mx), C.byref(n),
                   _ptr(counts), _ptr(edges), C.byref(valid))
        return np.float32(mn.value), np.float32(mx.value), int(n.value), counts, edges, int(valid.value)

    def sample_range_hist2(self, field_a, field_b, strides, nbins=256):
        """sample_range_hist of two independent fields in one device round trip: two (min, max, n_positive, counts, edges, valid)."""
        mn, mx, n, valid = np.zeros(2, np.float32), np.zeros(2, np.float32), np.zeros(2, np.int64), np.zeros(2, np.int32)
        counts = np.zeros((2, nbins), np.int64)
        edges = np.zeros((2, nbins + 1), np.float32)
        sz, sy, sx = (int(s) for s in strides)
        self._call("nl_sample_range_hist2", int(field_a), int(field_b), sz, sy, sx, int(nbins), _ptr(mn), _ptr(mx), _ptr(n),
                   _ptr(counts), _ptr(edges), _ptr(valid))
        return tuple((np.float32(mn[k]), np.float32(mx[k]), int(n[k]), counts[k], edges[k], int(valid[k])) for k in range(2))

    def hessian_stats(self, spacing):
        sp = (_f64 * 3)(*[float(s) for s in spacing])
        ma, mf, inf = _f32(0), _f32(0), _int(0)
        self._call("nl_hessian_stats", sp, C.byref(ma), C.byref(mf), C.byref(inf))
        return np.float32(ma.value), np.float32(mf.value), bool(inf.value)

    def set_frob_norm(self, max_abs, max_finite):
        self._call("nl_set_frob_norm", float(max_abs), float(max_finite))

    def vesselness_step(self, gamma_sq, alpha_sq, beta_sq, thr, want_count=True, z0=-1, z1=-1):
        n = _i64(0)
        use = 0 if thr is None else 1
        self._call("nl_vesselness_step", float(np.float32(gamma_sq)), float(np.float32(alpha_sq)),
                   float(np.float32(beta_sq)), use, float(np.float32(0.0 if thr is None else thr)),
                   int(z0), int(z1), C.byref(n) if want_count else None)
        return int(n.value)

    def mask_volume_fused(self, thr) -> int:
        n = _i64(0)
        self._call("nl_mask_volume_fused", float(np.float32(thr)), C.byref(n))
        return int(n.value)

    # ---------------------------------------------------------------- Markers stage
    def tail_enqueue(self, strides, q=1.0):
        """The frame's epilogue (percentile threshold selected on the device, mask, opening, product), enqueued without a wait."""
        self._call("nl_tail_enqueue", int(strides[0]), int(strides[1]), int(strides[2]), float(q))

    def tail_finish(self, commit=True):
        """-> dict(n_samples, a, b, gamma, thr, n_positive) of the epilogue nl_tail_enqueue started; commit: it becomes the frame."""
        n, npos = _i64(0), _i64(0)
        a, b, g, t = _f32(0), _f32(0), _f32(0), _f32(0)
        self._call("nl_tail_finish", 1 if commit else 0, C.byref(n), C.byref(a), C.byref(b), C.byref(g), C.byref(t), C.byref(npos))
        return dict(n_samples=int(n.value), a=np.float32(a.value), b=np.float32(b.value), gamma=np.float32(g.value),
                    thr=np.float32(t.value), n_positive=int(npos.value))

    def mask_volume_dev(self, strides, q=1.0):
        """_mask_volume with the percentile selected on the device -> dict(n_samples, a, b, gamma, thr); n_samples = 0: frame unchanged."""
        n = _i64(0)
        a, b, g, t = _f32(0), _f32(0), _f32(0), _f32(0)
        self._call("nl_mask_volume_dev", int(strides[0]), int(strides[1]), int(strides[2]), float(q), C.byref(n), C.byref(a), C.byref(b),
                   C.byref(g), C.byref(t))
        return dict(n_samples=int(n.value), a=np.float32(a.value), b=np.float32(b.value), gamma=np.float32(g.value), thr=np.float32(t.value))

    def debug_percentile(self, values, q=1.0):
        v = np.ascontiguousarray(values, dtype=np.float32).reshape(-1)
        a, b, t = _f32(0), _f32(0), _f32(0)
        self._call("nl_debug_percentile", _ptr(v), v.size, float(q), C.byref(t), C.byref(a), C.byref(b))
        return np.float32(t.value), np.float32(a.value), np.float32(b.value)

    def markers_begin(self, labels=None, intensity=None):
        """labels: int32 (Z, Y, X) host array or None (device labels of label_run); intensity: host array of any
        supported dtype or None (the resident input)."""
        lab_p, int_p, code = None, None, 0
        keep = []
        if labels is not None:
            a = np.ascontiguousarray(labels, dtype=np.int32)
            assert a.shape == self.shape
            keep.append(a); lab_p = _ptr(a)
        if intensity is not None:
            b = np.ascontiguousarray(intensity)
            if b.dtype not in DTYPE_CODES:          # float16, bool, big-endian maps ...: as filter_load / input_load do
                b = b.astype(np.float32)
            assert b.shape == self.shape
            keep.append(b); int_p = _ptr(b); code = DTYPE_CODES[b.dtype]
        self._call("nl_markers_begin", lab_p, int_p, code)

    def markers_distance(self, clamp) -> int:
        n = _i64(0)
        self._call("nl_markers_distance", float(np.float32(clamp)), C.byref(n))
        return int(n.value)

    def markers_log_step(self, wz2, wz0, wy2, wy0, wx2, wx0, s2):
        """One sigma of mocap_marking.py:488-508.  wz2 = wz0 = None: 2-D image (no Z terms)."""
        flat = wz2 is None and wz0 is None
        arrs = [np.ascontiguousarray(w, dtype=np.float64) for w in (wy2, wy0, wx2, wx0)]
        ryx = (arrs[0].size - 1) // 2
        assert all(a.size == arrs[0].size for a in arrs)
        ptr = [a.ctypes.data_as(C.POINTER(_f64)) for a in arrs]
        if flat:
            self._call("nl_markers_log_step", None, None, 0, ptr[0], ptr[1], ptr[2], ptr[3], ryx, float(np.float32(s2)))
            return
        z2, z0 = np.ascontiguousarray(wz2, dtype=np.float64), np.ascontiguousarray(wz0, dtype=np.float64)
        assert z2.size == z0.size
        self._call("nl_markers_log_step", z2.ctypes.data_as(C.POINTER(_f64)), z0.ctypes.data_as(C.POINTER(_f64)), (z2.size - 1) // 2,
                   ptr[0], ptr[1], ptr[2], ptr[3], ryx, float(np.float32(s2)))

    def markers_use_image(self, image=None):
        """use_im='frangi' (mocap_marking.py:675-679): the float32 image the LoG runs on; None = the distance image."""
        if image is None:
            self._call("nl_markers_use_image", None)
            return
        im = np.ascontiguousarray(image, dtype=np.float32)
        if im.size != int(np.prod(self.shape)):
            raise ValueError(f"image shape {im.shape} does not match the context shape {tuple(self.shape)}")
        self._call("nl_markers_use_image", _ptr(im))

    def markers_finish(self, peak_min_distance) -> int:
        n = _i64(0)
        self._call("nl_markers_finish", int(peak_min_distance), C.byref(n))
        return int(n.value)

    def markers_store(self, marker=True, distance=True, border=True):
        m = np.empty(self.shape, np.uint8) if marker else None
        d = np.empty(self.shape, np.float32) if distance else None
        b = np.empty(self.shape, np.uint8) if border else None
        self._call("nl_markers_store", None if m is None else _ptr(m), None if d is None else _ptr(d), None if b is None else _ptr(b))
        return m, d, b

    def skel_pixel_class(self, skel, download=True):
        """networking.py:672-683.  Returns (uint8 pixel classes or None, number of skeleton voxels)."""
        skel = np.ascontiguousarray(skel, dtype=np.int32)
        if skel.size != int(np.prod(self.shape)):
            raise ValueError(f"skeleton shape {skel.shape} does not match the context shape {tuple(self.shape)}")
        out = np.empty(skel.shape, np.uint8) if download else None
        n = _i64(0)
        self._call("nl_skel_pixel_class", _ptr(skel), None if out is None else _ptr(out), C.byref(n))
        return out, int(n.value)

    def skel_branch_labels(self, pixel_class=None):
        """networking.py:758-800.  pixel_class=None: the classes of the previous skel_pixel_class on this context.
        Returns (int32 branch labels, number of branches)."""
        pc = None
        if pixel_class is not None:
            pc = np.ascontiguousarray(pixel_class, dtype=np.uint8)
            if pc.size != int(np.prod(self.shape)):
                raise ValueError(f"pixel_class shape {pc.shape} does not match the context shape {tuple(self.shape)}")
        out = np.empty(self.shape if pc is None else pc.shape, np.int32)
        n = _i64(0)
        self._call("nl_skel_branch_labels", None if pc is None else _ptr(pc), _ptr(out), C.byref(n))
        return out, int(n.value)

    def set_ndim(self, ndim: int):
        self._call("nl_set_ndim", int(ndim))

    def log2d_step(self, wy2, wy0, wx2, wx0, s2, first, use_mask=True):
        arrs = [np.ascontiguousarray(w, dtype=np.float64) for w in (wy2, wy0, wx2, wx0)]
        r = (arrs[0].size - 1) // 2
        assert all(a.size == 2 * r + 1 for a in arrs)
        self._call("nl_log2d_step", *[a.ctypes.data_as(C.POINTER(_f64)) for a in arrs], r, float(np.float32(s2)),
                   1 if first else 0, 1 if use_mask else 0)

    def log2d_finish(self) -> int:
        n = _i64(0)
        self._call("nl_log2d_finish", C.byref(n))
        return int(n.value)

    def one_pass_available(self) -> bool:
        return bool(self.info("vesselness_one_pass"))

    def chain_available(self) -> bool:
        """nl_chain_begin's precondition: 3-D, the one-pass walk, and the pair kernel (off for planes of >= 2^30 voxels and with
        NELLIE_HV_RS=0: the synchronous path then runs the one-voxel walk)."""
        return bool(self.info("chain_available"))

    def set_spacing(self, spacing):
        sp = (_f64 * 3)(*[float(s) for s in spacing])
        self._call("nl_set_spacing", sp)

    def vesselness_spec(self, spacing, fsq_lo, fsq_hi, z0=-1, z1=-1):
        """One walk over the Hessian: statistics + vesselness candidates for fsq_min in [fsq_lo, fsq_hi].
        Returns (max_abs, max_frob_sq, any_inf, overflow)."""
        sp = (_f64 * 3)(*[float(s) for s in spacing])
        ma, mf, inf, ovf = _f32(0), _f32(0), _int(0), _int(0)
        self._call("nl_vesselness_spec", sp, float(np.float32(fsq_lo)), float(np.float32(fsq_hi)), int(z0), int(z1),
                   C.byref(ma), C.byref(mf), C.byref(inf), C.byref(ovf))
        return np.float32(ma.value), np.float32(mf.value), bool(inf.value), bool(ovf.value)

    # ---- device-resident threshold chain (include/nellie_amd.h: nl_chain_*)
    def chain_begin(self, n_scales):
        self._call("nl_chain_begin", int(n_scales))
        self._chain_n = int(n_scales)

    def chain_scale(self, spacing, strides, alpha_sq, beta_sq, division, margin, test_scale=1.0, z0=-1, z1=-1):
        sp = (_f64 * 3)(*[float(s) for s in spacing])
        sz, sy, sx = (int(s) for s in strides)
        self._call("nl_chain_scale", sp, sz, sy, sx, float(alpha_sq), float(beta_sq), float(division), float(margin), float(test_scale),
                   int(z0), int(z1))

    def chain_flush(self):
        self._call("nl_chain_flush")

    def chain_finish(self):
        """-> (flags, gamma, max_abs, thr, mask_count) arrays over the scales; flags all zero: the chain's result stands."""
        n = self._chain_n
        flags = np.zeros(n, np.int32)
        gamma, max_abs, thr = np.zeros(n), np.zeros(n), np.zeros(n)
        counts = np.zeros(n, np.int64)
        self._call("nl_chain_finish", _ptr(flags), _ptr(gamma), _ptr(max_abs), _ptr(thr), _ptr(counts))
        return flags, gamma, max_abs, thr, counts

    def chain_log(self, k, which):
        counts, edges = np.zeros(256, np.int64), np.zeros(257, np.float32)
        rng, sc = np.zeros(2, np.float32), np.zeros(8)
        self._call("nl_chain_log", int(k), int(which), _ptr(counts), _ptr(edges), _ptr(rng), _ptr(sc))
        return counts, edges, rng, dict(zip(("fsq_lo", "fsq_hi", "gamma_sq", "fsq_min", "thr_cmp", "max_frob", "tri", "otsu"), sc))

    def vesselness_resolve(self, gamma_sq, alpha_sq, beta_sq, thr) -> bool:
        """hit: False means the bracket missed and vesselness_step must run.  Asynchronous on a hit: the kernel runs
        on the context's side stream, vesselness_count() waits for it and returns the h_mask count."""
        hit = _int(0)
        use = 0 if thr is None else 1
        self._call("nl_vesselness_resolve", float(np.float32(gamma_sq)), float(np.float32(alpha_sq)),
                   float(np.float32(beta_sq)), use, float(np.float32(0.0 if thr is None else thr)),
                   C.byref(hit), None)
        return bool(hit.value)

    def vesselness_count(self) -> int:
        n = _i64(0)
        self._call("nl_vesselness_count", C.byref(n))
        return int(n.value)

    def filter_finish(self, z0=-1, z1=-1) -> int:
        n = _i64(0)
        self._call("nl_filter_finish", int(z0), int(z1), C.byref(n))
        return int(n.value)

    # ---------------------------------------------------------------- Z-slabs
    def remove_edges(self, margin=15) -> int:
        """filtering.py:969-1000 on the resident frame; returns the number of values > 0 left."""
        n = _i64(0)
        self._call("nl_remove_edges", int(margin), C.byref(n))
        return int(n.value)

    def planes_get(self, field, z0, z1):
        out = np.empty((z1 - z0, self.shape[1], self.shape[2]), dtype=np.float32)
        self._call("nl_planes_get", int(field), int(z0), int(z1), _ptr(out))
        return out

    def planes_put(self, field, z0, z1, planes):
        a = np.ascontiguousarray(planes, dtype=np.float32)
        assert a.shape == (z1 - z0, self.shape[1], self.shape[2])
        self._call("nl_planes_put", int(field), int(z0), int(z1), _ptr(a))

    def comm_init(self, world, rank, uid: bytes):
        assert len(uid) == 128
        self._call("nl_comm_init", int(world), int(rank), uid)

    def halo_exchange(self, field, depth):
        self._call("nl_halo_exchange", int(field), int(depth))

    def halo_exchange_at(self, field, offset, depth, run_async=False):
        self._call("nl_halo_exchange_at", int(field), int(offset), int(depth), 1 if run_async else 0)

    def comm_init2(self, world, rank, uid: bytes):
        self._call("nl_comm_init2", int(world), int(rank), uid)

    def allreduce(self, arr: np.ndarray, op: str):
        a = np.ascontiguousarray(arr)
        assert a.dtype in (np.int64, np.float32)
        self._call("nl_allreduce", _ptr(a), a.size, 0 if a.dtype == np.int64 else 1, {"sum": 0, "min": 1, "max": 2}[op])
        return a

    def mask_volume(self, thr):
        self._call("nl_mask_volume", float(np.float32(thr)))

    def filter_store(self, z0=0, z1=None, out=None):
        z1 = self.shape[0] if z1 is None else z1
        if out is None:
            out = np.empty((z1 - z0, self.shape[1], self.shape[2]), dtype=np.float32)
        assert out.dtype == np.float32 and out.flags.c_contiguous
        self._call("nl_filter_store", _ptr(out), z0, z1)
        return out

    def gauss_store(self, z0=0, z1=None):
        z1 = self.shape[0] if z1 is None else z1
        out = np.empty((z1 - z0, self.shape[1], self.shape[2]), dtype=np.float32)
        self._call("nl_gauss_store", _ptr(out), z0, z1)
        return out

    # ---------------------------------------------------------------- frame streaming
    def input_load_async(self, slot, pinned):
        """`pinned`: a PinnedArray, or a numpy view into registered (page-locked) memory."""
        if isinstance(pinned, np.ndarray):
            self._call("nl_input_load_async", int(slot), _ptr(pinned), DTYPE_CODES[pinned.dtype])
        else:
            self._call("nl_input_load_async", int(slot), pinned._p, DTYPE_CODES[pinned.dtype])

    def input_select(self, slot):
        self._call("nl_input_select", int(slot))

    def input_wait(self, slot):
        """Block until the upload into that slot has arrived (copy-thread call)."""
        self._call("nl_input_wait", int(slot))

    def outputs_stage(self, with_labels=True):
        self._call("nl_outputs_stage", 1 if with_labels else 0)

    def outputs_fetch_async(self, frangi, labels=None):
        def ptr(a):
            if a is None:
                return None
            return _ptr(a) if isinstance(a, np.ndarray) else a._p
        self._call("nl_outputs_fetch_async", ptr(frangi), ptr(labels))

    def outputs_wait(self):
        self._call("nl_outputs_wait")

    def outputs_pack(self, with_labels=True) -> int:
        """Pack the frame's products on the device (see include/nellie_amd.h); bytes of the blob, 0 if it does not pack."""
        n = _i64(0)
        self._call("nl_outputs_pack", 1 if with_labels else 0, C.byref(n))
        return int(n.value)

    def outputs_pack_with_label(self, on=True):
        """label_run() then enqueues the frame's outputs_pack(True) under its own wait; the outputs_pack(True) that follows returns at once."""
        self._call("nl_outputs_pack_with_label", 1 if on else 0)

    def outputs_fetch_packed_async(self, pinned, nbytes):
        self._call("nl_outputs_fetch_packed_async", pinned._p if hasattr(pinned, "_p") else _ptr(pinned), int(nbytes))

    # ---------------------------------------------------------------- Label
    def label_load_frangi(self, frangi: np.ndarray, z0=0, z1=None):
        z1 = self.shape[0] if z1 is None else z1
        a = np.ascontiguousarray(frangi, dtype=np.float32)
        assert a.shape == (z1 - z0, self.shape[1], self.shape[2])
        self._call("nl_label_load_frangi", _ptr(a), z0, z1)

    def label_intensity_mask(self, original: np.ndarray, thresh: float, z0=None, z1=None):
        """frangi *= (original > thresh); z0, z1: only those planes, `original` holding exactly them."""
        a = np.ascontiguousarray(original)
        if a.dtype not in DTYPE_CODES:
            a = a.astype(np.float64)
        if z0 is None and z1 is None:
            assert a.shape == self.shape
            self._call("nl_label_intensity_mask", _ptr(a), DTYPE_CODES[a.dtype], float(thresh))
            return
        z0 = 0 if z0 is None else int(z0)
        z1 = self.shape[0] if z1 is None else int(z1)
        assert a.shape == (z1 - z0,) + tuple(self.shape[1:])
        self._call("nl_label_intensity_mask_planes", _ptr(a), DTYPE_CODES[a.dtype], float(thresh), z0, z1)

    def flat_sample_gather(self, field, offset, step):
        n = _i64(0)
        self._call("nl_flat_sample_gather", field, int(offset), int(step), None, 0, C.byref(n))
        out = np.empty(int(n.value), dtype=np.float32)
        if out.size:
            self._call("nl_flat_sample_gather", field, int(offset), int(step), _ptr(out), out.size, C.byref(n))
        return out

    def flat_sample_gather_positive(self, field, offset, step):
        n = _i64(0)
        self._call("nl_flat_sample_gather", field, int(offset), int(step), None, 0, C.byref(n))
        out = np.empty(int(n.value), dtype=np.float32)
        if out.size:
            self._call("nl_flat_sample_gather_positive", field, int(offset), int(step), _ptr(out), out.size, C.byref(n))
        return out[:int(n.value)]

    def label_run(self, thr, min_area, fill_holes=True) -> int:
        n = _i64(0)
        has = 0 if thr is None else 1
        self._call("nl_label_run", has, float(np.float32(0.0 if thr is None else thr)), int(min_area),
                   1 if fill_holes else 0, C.byref(n))
        return int(n.value)

    def label_pack(self, thr):
        has = 0 if thr is None else 1
        self._call("nl_label_pack", has, float(np.float32(0.0 if thr is None else thr)))

    def label_bits_get(self, row0, nrows):
        out = np.empty((nrows, (self.shape[2] + 63) // 64), dtype=np.uint64)
        self._call("nl_label_bits_get", int(row0), int(nrows), _ptr(out))
        return out

    def label_bits_put(self, row0, words):
        a = np.ascontiguousarray(words, dtype=np.uint64)
        self._call("nl_label_bits_put", int(row0), a.shape[0], _ptr(a))

    def label_bits_allgather(self, slab_plane0):
        a = np.ascontiguousarray(slab_plane0, dtype=np.int64)
        self._call("nl_label_bits_allgather", _ptr(a))

    def label_run_global(self, min_area, fill_holes=True) -> int:
        n = _i64(0)
        self._call("nl_label_run_global", int(min_area), 1 if fill_holes else 0, C.byref(n))
        return int(n.value)

    # ---- Z-slab Label without replication (include/nellie_amd.h "Label on Z-slabs WITHOUT replication") ----
    def slab_label_pack(self, thr):
        has = thr is not None
        self._call("nl_slab_label_pack", 1 if has else 0, float(np.float32(thr)) if has else 0.0)

    def slab_bits_get(self, which, plane):
        words = self.shape[1] * ((self.shape[2] + 63) // 64)
        out = np.empty(words, np.uint64)
        self._call("nl_slab_bits_get", int(which), int(plane), _ptr(out))
        return out

    def slab_bits_put(self, which, plane, words):
        a = np.ascontiguousarray(words, dtype=np.uint64)
        self._call("nl_slab_bits_put", int(which), int(plane), _ptr(a))

    def slab_bits_exchange(self, which):
        self._call("nl_slab_bits_exchange", int(which))

    SLAB_BLOCK_INTS = 16384

    def slab_phase(self, phase, gather_world=0):
        """One phase of the slab protocol up to its tables (nl_slab_phase): -> list of int32 blobs, this rank's only
        (gather_world = 0) or every rank's, all-gathered on the device (gather_world = the communicator's size)."""
        nb = int(gather_world) if gather_world else 1
        block = self.SLAB_BLOCK_INTS
        need, nruns = _i64(0), _i64(0)
        out = np.empty(nb * block, np.int32)
        self._call("nl_slab_phase", int(phase), 1 if gather_world else 0, block, _ptr(out), C.byref(need), C.byref(nruns))
        if need.value > block:                       # rare: tables beyond 64 KiB -- fetch again in larger blocks, nothing is recomputed
            block = int(need.value) + 1024
            out = np.empty(nb * block, np.int32)
            self._call("nl_slab_phase", -1, 1 if gather_world else 0, block, _ptr(out), C.byref(need), C.byref(nruns))
        self.slab_nruns = int(nruns.value)
        return [out[r * block:r * block + int(out[r * block + 4])] for r in range(nb)]

    def slab_patch(self, roots, values):
        r = np.ascontiguousarray(roots, dtype=np.int32)
        v = np.ascontiguousarray(values, dtype=np.int32)
        assert r.size == v.size
        if r.size:
            self._call("nl_slab_patch", r.size, _ptr(r), _ptr(v))

    def slab_apply(self, min_area=0):
        self._call("nl_slab_apply", int(min_area))

    def slab_majority(self):
        self._call("nl_slab_majority")

    def slab_number(self, clear, select):
        """-> (trees this rank numbers, 1-based local rank of every tree of `select`)"""
        c = np.ascontiguousarray(clear, dtype=np.int32)
        s_ = np.ascontiguousarray(select, dtype=np.int32)
        n = _i64(0)
        ids = np.empty(s_.size, np.int32)
        self._call("nl_slab_number", c.size, _ptr(c) if c.size else None, s_.size, _ptr(s_) if s_.size else None, C.byref(n),
                   _ptr(ids) if s_.size else None)
        return int(n.value), ids

    def slab_paint(self, base, roots, labels):
        r = np.ascontiguousarray(roots, dtype=np.int32)
        v = np.ascontiguousarray(labels, dtype=np.int32)
        assert r.size == v.size
        self._call("nl_slab_paint", int(base), r.size, _ptr(r) if r.size else None, _ptr(v) if v.size else None)

    def allgather_bytes(self, data: bytes, max_bytes: int, world: int):
        """Variable-size all-gather over RCCL: the list of every rank's bytes."""
        send = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(0, np.uint8)
        recv = np.empty(int(max_bytes) * int(world), np.uint8)
        sizes = np.zeros(int(world), np.int64)
        self._call("nl_allgather_bytes", _ptr(send) if send.size else None, send.size, _ptr(recv), int(max_bytes), _ptr(sizes))
        return [recv[r * max_bytes:r * max_bytes + int(sizes[r])].tobytes() for r in range(int(world))]

    def positive_samples_world(self, field, mode, a, b, c, block_items, world):
        """Every rank's positive samples (mode 0: lattice strides a, b, c; mode 1: flat offset a, step b), gathered on the device."""
        cap = int(block_items) * int(world)
        out = np.empty(max(cap, 1), np.float32)
        counts = np.zeros(int(world), np.int64)
        self._call("nl_positive_samples_world", int(field), int(mode), int(a), int(b), int(c), int(block_items), _ptr(out), cap, _ptr(counts))
        return out[:int(counts.sum())]

    def allgather_var(self, arr: np.ndarray, world: int):
        """Variable-size all-gather over RCCL of one array per rank (same dtype everywhere): the list of every rank's array."""
        a = np.ascontiguousarray(arr)
        recv, stride = C.c_void_p(), C.c_int64(0)
        sizes = np.zeros(int(world), np.int64)
        self._call("nl_allgather_var", _ptr(a) if a.size else None, a.nbytes, C.byref(recv), C.byref(stride), _ptr(sizes))
        blob = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(int(stride.value) * int(world),))
        st = int(stride.value)
        return [blob[r * st:r * st + int(sizes[r])].view(a.dtype).copy() for r in range(int(world))]

    def comm_fuse(self, on=True):
        self._call("nl_comm_fuse", 1 if on else 0)

    def label_store(self, z0=0, z1=None, out=None):
        z1 = self.shape[0] if z1 is None else z1
        if out is None:
            out = np.empty((z1 - z0, self.shape[1], self.shape[2]), dtype=np.int32)
        assert out.dtype == np.int32 and out.flags.c_contiguous
        self._call("nl_label_store", _ptr(out), z0, z1)
        return out

    def debug_eig_frangi(self, h6, alpha_sq=0.5, beta_sq=0.5, gamma_sq=1.0, impl=0):
        h = np.ascontiguousarray(h6, dtype=np.float32)
        out = np.empty((h.shape[0], 4), dtype=np.float32)
        self._call("nl_debug_eig_frangi", _ptr(h), h.shape[0], int(impl), float(np.float32(alpha_sq)),
                   float(np.float32(beta_sq)), float(np.float32(gamma_sq)), _ptr(out))
        return out

    # ---------------------------------------------------------------- timing
    def timer_begin(self):
        self._call("nl_timer_begin")

    def timer_end_ms(self) -> float:
        ms = _f32(0)
        self._call("nl_timer_end_ms", C.byref(ms))
        return float(ms.value)

    def info(self, key: str) -> float:
        v = _f64(0)
        if self.lib.cdll.nl_ctx_info(self._h, key.encode(), C.byref(v)) != NL_OK:
            raise KeyError(key)
        return float(v.value)

    def prof_enable(self, on=True):
        self.lib.cdll.nl_prof_enable(self._h, 1 if on else 0)

    def prof_reset(self):
        self.lib.cdll.nl_prof_reset(self._h)

    def prof_get(self, name: str):
        ms, k = _f64(0), _i64(0)
        self.lib.cdll.nl_prof_get(self._h, name.encode(), C.byref(ms), C.byref(k))
        return float(ms.value), int(k.value)
