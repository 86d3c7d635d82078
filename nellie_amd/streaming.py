"""
Streaming a 3-D+T stack through the engine (BASELINE config 5): frame t+1 is read from the (memory-mapped) stack
into pinned memory and copied host -> HBM while frame t computes, and the two outputs of frame t-1 are copied
HBM -> host and written to their memory maps, each leg on its own HIP stream / host thread.

The reference loops `for t in range(num_t)` with blocking transfers inside each stage (filtering.py:1007-1031,
labelling.py:701-734) and round-trips the Frangi frame through the file between the two stages; here one frame
stays resident in HBM from upload to label download.  Results are the same arrays the per-stage path writes.
"""
from __future__ import annotations

import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from nellie_amd import hipnative
from nellie_amd.pipeline import FilterParams, FramePipeline, min_area_pixels_of


class StreamedSegmenter:
    def __init__(self, frame_shape, dtype, params: FilterParams, min_area=None, min_radius_um=0.25, device: int = 0,
                 threshold_sampling_pixels=1_000_000, histogram_nbins=256):
        self.shape = tuple(int(s) for s in frame_shape)
        self.params = params
        self.min_area = min_area if min_area is not None else min_area_pixels_of(params.dim_res, min_radius_um)
        self.sampling = (threshold_sampling_pixels, histogram_nbins)
        self.pipe = FramePipeline(self.shape, device=device)
        dtype = np.dtype(dtype)
        if dtype not in hipnative.DTYPE_CODES:
            dtype = np.dtype(np.float32)
        self.in_buf = [hipnative.PinnedArray(self.shape, dtype) for _ in range(2)]
        self.fr_buf = [hipnative.PinnedArray(self.shape, np.float32) for _ in range(2)]
        self.lab_buf = [hipnative.PinnedArray(self.shape, np.int32) for _ in range(2)]
        self.io = ThreadPoolExecutor(max_workers=2)
        self.stats = []
        self._reg_in = None

    def close(self):
        self.io.shutdown(wait=True)
        self.pipe.close()
        for b in self.in_buf + self.fr_buf + self.lab_buf:
            b.free()

    def _stage_in(self, frames, t, slot):
        """host thread: the asynchronous H2D of that slot -- straight from the stack when it is page-locked,
        else through a pinned staging buffer"""
        if self._reg_in is not None and self._reg_in.ok:
            self.pipe.ctx.input_load_async(slot, frames[t])
            return
        np.copyto(self.in_buf[slot].array, frames[t], casting="unsafe")
        self.pipe.ctx.input_load_async(slot, self.in_buf[slot])

    def run(self, frames, out_frangi, out_labels, status=None, flush=True):
        """frames: (T, Z, Y, X) array / memmap; out_*: writable (T, Z, Y, X) float32 / int32 arrays (memmaps).
        In-memory arrays are page-locked in place (hipHostRegister) so that no host-side staging copy is needed."""
        ctx = self.pipe.ctx
        num_t = len(frames)
        plain = lambda a: isinstance(a, np.ndarray) and not isinstance(a, np.memmap)   # noqa: E731
        self._reg_in = hipnative.RegisteredArray(frames) if plain(frames) and frames.dtype == self.in_buf[0].dtype else None
        reg_fr = hipnative.RegisteredArray(out_frangi) if plain(out_frangi) and out_frangi.dtype == np.float32 else None
        reg_lab = hipnative.RegisteredArray(out_labels) if plain(out_labels) and out_labels.dtype == np.int32 else None
        direct_out = bool(reg_fr and reg_fr.ok and reg_lab and reg_lab.ok)
        try:
            return self._run(ctx, frames, out_frangi, out_labels, num_t, status, flush, direct_out)
        finally:
            for r in (self._reg_in, reg_fr, reg_lab):
                if r is not None:
                    r.release()
            self._reg_in = None

    def _run(self, ctx, frames, out_frangi, out_labels, num_t, status, flush, direct_out):
        load = self.io.submit(self._stage_in, frames, 0, 0)
        pending = None                                        # (t, slot, future that waits + writes)
        for t in range(num_t):
            slot = t & 1
            if status is not None:
                status(t, num_t)
            load.result()
            ctx.input_select(slot)
            # while frame t computes, frame t+1 is read and uploaded into the other slot (its previous user,
            # frame t-1, has finished computing: every frame ends with a synchronising label count)
            if t + 1 < num_t:
                load = self.io.submit(self._stage_in, frames, t + 1, slot ^ 1)
            self.pipe.filter(None, self.params)
            thr = self.pipe.frangi_threshold(*self.sampling)
            n = self.pipe.label(thr, self.min_area)
            self.stats.append((self.pipe.trace.n_positive, n))
            if pending is not None:
                pending.result()                              # frame t-1 landed on the host and in the memmaps
            ctx.outputs_stage(True)
            if direct_out:
                ctx.outputs_fetch_async(out_frangi[t], out_labels[t])
            else:
                ctx.outputs_fetch_async(self.fr_buf[slot], self.lab_buf[slot])

            def drain(tt=t, ss=slot):
                ctx.outputs_wait()
                if direct_out:
                    return
                out_frangi[tt] = self.fr_buf[ss].array
                out_labels[tt] = self.lab_buf[ss].array
                if flush and hasattr(out_frangi, "flush"):
                    out_frangi.flush()
                    out_labels.flush()
            pending = self.io.submit(drain)
        if pending is not None:
            pending.result()
        return self.stats
