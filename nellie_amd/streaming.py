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

    def close(self):
        self.io.shutdown(wait=True)
        self.pipe.close()
        for b in self.in_buf + self.fr_buf + self.lab_buf:
            b.free()

    def _stage_in(self, frames, t, slot):
        """host thread: stack -> pinned buffer, then the asynchronous H2D of that slot"""
        np.copyto(self.in_buf[slot].array, frames[t], casting="unsafe")
        self.pipe.ctx.input_load_async(slot, self.in_buf[slot])

    def run(self, frames, out_frangi, out_labels, status=None, flush=True):
        """frames: (T, Z, Y, X) array / memmap; out_*: writable (T, Z, Y, X) float32 / int32 arrays (memmaps)."""
        ctx = self.pipe.ctx
        num_t = len(frames)
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
            ctx.outputs_fetch_async(self.fr_buf[slot], self.lab_buf[slot])

            def drain(tt=t, ss=slot):
                ctx.outputs_wait()
                out_frangi[tt] = self.fr_buf[ss].array
                out_labels[tt] = self.lab_buf[ss].array
                if flush and hasattr(out_frangi, "flush"):
                    out_frangi.flush()
                    out_labels.flush()
            pending = self.io.submit(drain)
        if pending is not None:
            pending.result()
        return self.stats
