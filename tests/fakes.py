"""Duck-typed ImInfo over in-memory arrays, in the spirit of the reference's test fakes
(tests/test_labelling.py:7-22), plus the get_memmap / allocate_memory pair the stages use
(nellie/im_info/verifier.py:967-1070)."""
import numpy as np


class _Mem(np.ndarray):
    def flush(self):
        pass


def _mem(a):
    return np.asarray(a).view(_Mem)


class ArrayImInfo:
    def __init__(self, volume_tzyx, dim_res, no_z=False):
        v = np.asarray(volume_tzyx)
        self.no_t = False
        self.no_z = no_z
        self.axes = "TYX" if no_z else "TZYX"
        self.shape = v.shape
        self.dim_res = dict(dim_res)
        self.im_path = "im"
        self.pipeline_paths = {"im_preprocessed": "frangi", "im_instance_label": "labels", "im_marker": "marker",
                               "im_distance": "distance", "im_border": "border"}
        self.store = {"im": _mem(v)}
        self.im = self.store["im"]

    def get_memmap(self, path):
        return self.store[path]

    def allocate_memory(self, path, dtype="float", description="", return_memmap=False, **kw):
        self.store[path] = _mem(np.zeros(self.shape, dtype=dtype))
        return self.store[path] if return_memmap else None
