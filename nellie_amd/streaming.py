"""
Streaming a 3-D+T stack through the engine (BASELINE config 5): frame t+1 is read from the (memory-mapped) stack
into pinned memory and copied host -> HBM while frame t computes, and the two outputs of frame t-1 are copied
HBM -> host and written to their memory maps, each leg on its own HIP stream / host thread.

The reference loops `for t in range(num_t)` with blocking transfers inside each stage (filtering.py:1007-1031,
labelling.py:701-734) and round-trips the Frangi frame through the file between the two stages; here one frame
stays resident in HBM from upload to label download.  Results are the same arrays the per-stage path writes.
"""
from __future__ import annotations

import os
import threading
import time
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
        # Packed download (nl_outputs_pack): both products are ~98 % zeros, and the dense 8 B/voxel over PCIe is what bounds the
        # stream.  Per slot one page-locked landing buffer for the blob; a frame that does not pack goes the dense way.
        self.packed = os.environ.get("NELLIE_STREAM_PACKED", "1") == "1"
        self.blob_buf = [hipnative.PinnedArray((2 * int(np.prod(self.shape)) + 4096,), np.uint8) for _ in range(2)] if self.packed else []
        self.packed_frames = 0
        # the pack of a frame rides under Label's own wait (two host waits per frame less; NELLIE_STREAM_PACK_WITH_LABEL=0: as before)
        if self.packed and os.environ.get("NELLIE_STREAM_PACK_WITH_LABEL", "1") == "1" and hasattr(self.pipe.ctx, "outputs_pack_with_label"):
            self.pipe.ctx.outputs_pack_with_label(True)
        self._zero_fill = True
        self.io = ThreadPoolExecutor(max_workers=3)
        # host threads of the copies / the packed-output expansion (NELLIE_STREAM_COPY_THREADS: A/B)
        self.copy_threads = int(os.environ.get("NELLIE_STREAM_COPY_THREADS", "0")) or max(1, min(8, (os.cpu_count() or 2) // 2))
        self.copiers = ThreadPoolExecutor(max_workers=self.copy_threads)
        self.stats = []
        self.timing = {"wait_upload": 0.0, "compute": 0.0, "wait_download": 0.0, "frames": 0}   # main-thread seconds
        self._reg_in = None

    def _copy(self, dst, src):
        """dst[...] = src, split over the copy threads along the first axis (numpy releases the GIL inside the
        copy; one thread moves ~20 GB/s, eight > 100 GB/s, and first-touch page faults of a fresh destination --
        a new memmap, an untouched array -- are taken in parallel too)."""
        n = dst.shape[0]
        k = min(self.copy_threads, n)
        if k <= 1 or dst.nbytes < (8 << 20):
            np.copyto(dst, src, casting="unsafe")
            return
        bounds = [n * i // k for i in range(k + 1)]
        list(self.copiers.map(lambda i: np.copyto(dst[bounds[i]:bounds[i + 1]], src[bounds[i]:bounds[i + 1]], casting="unsafe"),
                              range(k)))

    def close(self):
        self.io.shutdown(wait=True)
        self.copiers.shutdown(wait=True)
        self.pipe.close()
        for b in self.in_buf + self.fr_buf + self.lab_buf + self.blob_buf:
            b.free()

    def _stage_in(self, frames, t, slot):
        """host thread: the asynchronous H2D of that slot -- straight from the stack when it is page-locked,
        else through a pinned staging buffer"""
        if self._reg_in is not None and self._reg_in.ok:
            self.pipe.ctx.input_load_async(slot, frames[t])
            return
        self._copy(self.in_buf[slot].array, frames[t])
        self.pipe.ctx.input_load_async(slot, self.in_buf[slot])

    def run(self, frames, out_frangi, out_labels, status=None, flush=True, outputs_zeroed=False):
        """frames: (T, Z, Y, X) array / memmap; out_*: writable (T, Z, Y, X) float32 / int32 arrays (memmaps).
        An in-memory input stack is page-locked in place (hipHostRegister: its pages hold data, so this costs
        ~0.1 ms/GiB) and uploaded without a staging copy.  Outputs always land in page-locked staging buffers and are
        copied out by the copy threads: locking a fresh, never-touched output array in place costs 170 ms/GiB on
        this host (single-threaded page population), three times the parallel copy including its page faults.
        outputs_zeroed: the output arrays are known to hold zeros (files just created by allocate_memory, calloc'ed arrays):
        the packed download then touches only the rows that have content -- the untouched pages of a new file stay holes."""
        ctx = self.pipe.ctx
        num_t = len(frames)
        plain = lambda a: isinstance(a, np.ndarray) and not isinstance(a, np.memmap)   # noqa: E731
        self._reg_in = hipnative.RegisteredArray(frames) if plain(frames) and frames.dtype == self.in_buf[0].dtype else None
        try:
            self._zero_fill = not outputs_zeroed
            return self._run(ctx, frames, out_frangi, out_labels, num_t, status, flush, False)
        finally:
            if self._reg_in is not None:
                self._reg_in.release()
            self._reg_in = None

    def _run(self, ctx, frames, out_frangi, out_labels, num_t, status, flush, direct_out):
        load = self.io.submit(self._stage_in, frames, 0, 0)
        landed = None                                         # event: the D2H of the previous frame has finished
        copied = [None, None]                                 # per staging slot: future of the copy into the caller's arrays
        for t in range(num_t):
            slot = t & 1
            if status is not None:
                status(t, num_t)
            t_a = time.perf_counter()
            load.result()
            ctx.input_select(slot)
            t_b = time.perf_counter()
            # while frame t computes, frame t+1 is read and uploaded into the other slot (its previous user,
            # frame t-1, has finished computing: every frame ends with a synchronising label count)
            if t + 1 < num_t:
                load = self.io.submit(self._stage_in, frames, t + 1, slot ^ 1)
            self.pipe.filter(None, self.params)
            thr = self.pipe.frangi_threshold(*self.sampling)
            n = self.pipe.label(thr, self.min_area)
            self.stats.append((self.pipe.trace.n_positive, n))
            t_c = time.perf_counter()
            if landed is not None:
                landed.wait()                                 # the device-side staging copy of frame t-1 has been read out
            if copied[slot] is not None:
                copied[slot].result()                         # frame t-2 left this slot's page-locked buffers
            t_d = time.perf_counter()
            tm = self.timing
            tm["wait_upload"] += t_b - t_a; tm["compute"] += t_c - t_b; tm["wait_download"] += t_d - t_c; tm["frames"] += 1
            nbytes = ctx.outputs_pack(True) if self.packed else 0
            if nbytes and nbytes <= self.blob_buf[slot].array.nbytes:
                ctx.outputs_fetch_packed_async(self.blob_buf[slot], nbytes)
                self.packed_frames += 1
            else:
                nbytes = 0
                ctx.outputs_stage(True)
                ctx.outputs_fetch_async(self.fr_buf[slot], self.lab_buf[slot])
            landed = threading.Event()

            def drain(tt=t, ss=slot, ev=landed, nb=nbytes):
                try:
                    ctx.outputs_wait()
                finally:
                    ev.set()
                if nb:
                    # expand straight into the caller's arrays (host threads; rows without content are only zero-filled)
                    fr_t, lab_t = out_frangi[tt], out_labels[tt]
                    if fr_t.flags.c_contiguous and lab_t.flags.c_contiguous and fr_t.dtype == np.float32 and lab_t.dtype == np.int32:
                        hipnative.outputs_unpack(self.blob_buf[ss], nb, fr_t, lab_t, zero_fill=self._zero_fill, threads=self.copy_threads)
                    else:
                        hipnative.outputs_unpack(self.blob_buf[ss], nb, self.fr_buf[ss].array, self.lab_buf[ss].array, True, self.copy_threads)
                        self._copy(fr_t, self.fr_buf[ss].array)
                        self._copy(lab_t, self.lab_buf[ss].array)
                else:
                    self._copy(out_frangi[tt], self.fr_buf[ss].array)
                    self._copy(out_labels[tt], self.lab_buf[ss].array)
                if flush and hasattr(out_frangi, "flush"):
                    out_frangi.flush()
                    out_labels.flush()
            copied[slot] = self.io.submit(drain)
        for f in copied:
            if f is not None:
                f.result()
        return self.stats

