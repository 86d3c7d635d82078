"""
Streaming a 3-D+T stack through the engine (BASELINE config 5): frame t+1 is read from the (memory-mapped) stack
into pinned memory and copied host -> HBM while frame t computes, and the two outputs of frame t-1 are copied
HBM -> host and written to their memory maps, each leg on its own HIP stream / host thread.

The reference loops `for t in range(num_t)` with blocking transfers inside each stage (filtering.py:1007-1031,
labelling.py:701-734) and round-trips the Frangi frame through the file between the two stages; here one frame
stays resident in HBM from upload to label download.  Results are the same arrays the per-stage path writes.

Round 6 -- LANES.  A config-5 frame (33.5 Mvoxel) is too small for this GPU: its 132 launches are 2.6-2.7 ms of kernel time of
which 0.6 ms are one-wave threshold kernels, 44 % of the per-voxel efficiency of a 1024^3 frame.  The frames of a stack are
independent (labelling.py:697-734: ids restart per frame), so small frames run on `lanes` CONTEXTS of the same GPU, one compute
thread each, frame t on lane t % lanes: the GPU runs two frames' kernel chains side by side (resident frames: 2.87 -> 2.17 ms per
frame with two lanes, profiles/r06_resident_lanes.json).  The lanes share ONE upload thread that keeps exactly one host -> HBM copy
in flight, frame after frame in stack order: the upload is what bounds a float32 stack (134 MB per frame at 54 GB/s = 2.49 ms), and
two concurrent copies share the link badly (3.57 ms each as a pair, profiles/r06_stream_lanes_trace.txt -- what made two independent
streamers SLOWER than one in round 5).  Every frame goes through one context exactly as before, so the results do not depend on the
number of lanes (tests/test_hip_full_size.py::test_c5_streamed_equals_per_frame_at_its_frame_size).
"""
from __future__ import annotations

import os
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from nellie_amd import hipnative
from nellie_amd.pipeline import FilterParams, FramePipeline, min_area_pixels_of

# Lanes by default (profiles/r06_stream_lanes.txt: 64 frames of 128 x 512 x 512, three alternating repetitions on one box --
# float32 3.2-4.3 / 3.5-3.8 / 2.67-2.71 / 2.8-3.1 ms per frame with 1 / 2 / 3 / 4 lanes, uint16 4.1-4.2 / 2.7 / 2.3-2.6 / 2.6-2.8):
# three below 2^26 voxels, two below 2^27 (a resident 256 x 512 x 512 frame gains 17 % from the second), one from there on -- a frame
# of that size fills the GPU by itself and another context would only cost HBM.
LANES_BELOW_VOXELS = 1 << 27


def default_lanes(frame_shape, device: int = 0) -> int:
    env = os.environ.get("NELLIE_STREAM_LANES")
    if env:
        return max(1, int(env))
    n = int(np.prod(frame_shape))
    want = 3 if n < (1 << 26) else (2 if n < LANES_BELOW_VOXELS else 1)
    from nellie_amd.utils import adaptive_run
    while want > 1 and not adaptive_run.frame_fits_on_device(frame_shape, device, contexts=want):      # a lane is a context's HBM
        want -= 1
    return want


class _Lane:
    """One context of the GPU with its page-locked landing buffers (two slots each: frame j of the lane uses slot j & 1)."""

    def __init__(self, shape, dtype, device, packed):
        self.pipe = FramePipeline(shape, device=device)
        self.in_buf = [hipnative.PinnedArray(shape, dtype) for _ in range(2)]
        self.fr_buf = [hipnative.PinnedArray(shape, np.float32) for _ in range(2)]
        self.lab_buf = [hipnative.PinnedArray(shape, np.int32) for _ in range(2)]
        # Packed download (nl_outputs_pack): both products are ~98 % zeros, and the dense 8 B/voxel over PCIe is what would bound
        # the stream.  Per slot one page-locked landing buffer for the blob; a frame that does not pack goes the dense way.
        self.blob_buf = [hipnative.PinnedArray((2 * int(np.prod(shape)) + 4096,), np.uint8) for _ in range(2)] if packed else []
        self.uploaded = {}           # frame index -> threading.Event: its upload has been ENQUEUED into the lane's slot
        self.computed = {}           # frame index -> threading.Event: its compute has finished (the input slot is free again)

    def free(self):
        self.pipe.close()
        for b in self.in_buf + self.fr_buf + self.lab_buf + self.blob_buf:
            b.free()


class StreamedSegmenter:
    def __init__(self, frame_shape, dtype, params: FilterParams, min_area=None, min_radius_um=0.25, device: int = 0,
                 threshold_sampling_pixels=1_000_000, histogram_nbins=256, lanes=None):
        self.shape = tuple(int(s) for s in frame_shape)
        self.params = params
        self.min_area = min_area if min_area is not None else min_area_pixels_of(params.dim_res, min_radius_um)
        self.sampling = (threshold_sampling_pixels, histogram_nbins)
        dtype = np.dtype(dtype)
        if dtype not in hipnative.DTYPE_CODES:
            dtype = np.dtype(np.float32)
        self.in_dtype = dtype
        self.packed = os.environ.get("NELLIE_STREAM_PACKED", "1") == "1"
        self.n_lanes = int(lanes) if lanes else default_lanes(self.shape, device)
        self.lanes = [_Lane(self.shape, dtype, device, self.packed) for _ in range(self.n_lanes)]
        self.pipe = self.lanes[0].pipe                      # (the single-lane names of round 5: tools and tests read them)
        self.in_buf, self.fr_buf, self.lab_buf, self.blob_buf = (self.lanes[0].in_buf, self.lanes[0].fr_buf, self.lanes[0].lab_buf,
                                                                 self.lanes[0].blob_buf)
        self.packed_frames = 0
        # the pack of a frame rides under Label's own wait (two host waits per frame less; NELLIE_STREAM_PACK_WITH_LABEL=0: as before)
        if self.packed and os.environ.get("NELLIE_STREAM_PACK_WITH_LABEL", "1") == "1":
            for ln in self.lanes:
                if hasattr(ln.pipe.ctx, "outputs_pack_with_label"):
                    ln.pipe.ctx.outputs_pack_with_label(True)
        self._zero_fill = True
        self.io = ThreadPoolExecutor(max_workers=1 + 2 * self.n_lanes)
        # host threads of the copies / the packed-output expansion (NELLIE_STREAM_COPY_THREADS: A/B)
        self.copy_threads = int(os.environ.get("NELLIE_STREAM_COPY_THREADS", "0")) or max(1, min(8, (os.cpu_count() or 2) // 2))
        self.copiers = ThreadPoolExecutor(max_workers=self.copy_threads * self.n_lanes)
        self.stats = []
        self.timing = {"wait_upload": 0.0, "compute": 0.0, "wait_download": 0.0, "frames": 0}   # compute-thread seconds, all lanes
        self._tlock = threading.Lock()
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
        for ln in self.lanes:
            ln.free()

    def _stage_in(self, ln, frames, t, slot, before_issue=None):
        """upload thread: the asynchronous H2D into that slot of the lane -- straight from the stack when it is page-locked,
        else through a pinned staging buffer (filled BEFORE `before_issue` waits for the copy in flight: the host copy of frame
        t + 1 runs beside the H2D of frame t)"""
        src = frames[t]
        if not (self._reg_in is not None and self._reg_in.ok):
            self._copy(ln.in_buf[slot].array, src)
            src = ln.in_buf[slot]
        if before_issue is not None:
            before_issue()
        ln.pipe.ctx.input_load_async(slot, src)

    def run(self, frames, out_frangi, out_labels, status=None, flush=True, outputs_zeroed=False):
        """frames: (T, Z, Y, X) array / memmap; out_*: writable (T, Z, Y, X) float32 / int32 arrays (memmaps).
        An in-memory input stack is page-locked in place (hipHostRegister: its pages hold data, so this costs
        ~0.1 ms/GiB) and uploaded without a staging copy.  Outputs always land in page-locked staging buffers and are
        copied out by the copy threads: locking a fresh, never-touched output array in place costs 170 ms/GiB on
        this host (single-threaded page population), three times the parallel copy including its page faults.
        outputs_zeroed: the output arrays are known to hold zeros (files just created by allocate_memory, calloc'ed arrays):
        the packed download then touches only the rows that have content -- the untouched pages of a new file stay holes."""
        num_t = len(frames)
        plain = lambda a: isinstance(a, np.ndarray) and not isinstance(a, np.memmap)   # noqa: E731
        self._reg_in = hipnative.RegisteredArray(frames) if plain(frames) and frames.dtype == self.in_dtype else None
        try:
            self._zero_fill = not outputs_zeroed
            self.stats = [None] * num_t
            return self._run(frames, out_frangi, out_labels, num_t, status, flush)
        finally:
            if self._reg_in is not None:
                self._reg_in.release()
            self._reg_in = None

    # ------------------------------------------------------------------------------------------------------------------
    def _uploader(self, frames, num_t, stop):
        """ONE thread feeds every lane, in stack order, one copy in flight: frame t goes into slot (t // L) & 1 of lane t % L as soon
        as the frame that used that slot before (t - 2 L) has been computed."""
        L = self.n_lanes
        in_flight = None
        try:
            for t in range(num_t):
                ln, j = self.lanes[t % L], t // L
                prev = ln.computed.get(t - 2 * L)
                while prev is not None and not prev.wait(0.05):
                    if stop.is_set():
                        return
                if stop.is_set():
                    return
                self._stage_in(ln, frames, t, j & 1, in_flight)
                ln.uploaded[t].set()
                # the next copy starts when this one has arrived (module header); one lane: as in round 5, no wait
                in_flight = (lambda c=ln.pipe.ctx, sl=j & 1: c.input_wait(sl)) if L > 1 else None
        except BaseException:
            stop.set()
            raise
        finally:
            if stop.is_set():                               # nobody may keep waiting for a frame that will never arrive
                for lane in self.lanes:
                    for ev in list(lane.uploaded.values()):
                        ev.set()

    def _lane_loop(self, k, frames, out_frangi, out_labels, num_t, status, flush, stop):
        """compute thread of lane k: frames k, k + L, ..."""
        L, ln = self.n_lanes, self.lanes[k]
        pipe, ctx = ln.pipe, ln.pipe.ctx
        landed = None                                         # event: the D2H of the lane's previous frame has finished
        copied = [None, None]                                 # per staging slot: future of the copy into the caller's arrays
        try:
            for j, t in enumerate(range(k, num_t, L)):
                slot = j & 1
                if status is not None and k == 0:
                    status(t, num_t)
                t_a = time.perf_counter()
                ln.uploaded[t].wait()
                if stop.is_set():
                    raise RuntimeError("the frame stream was aborted (another lane or the upload thread failed)")
                ctx.input_select(slot)
                t_b = time.perf_counter()
                pipe.filter(None, self.params)
                thr = pipe.frangi_threshold(*self.sampling)
                n = pipe.label(thr, self.min_area)
                self.stats[t] = (pipe.trace.n_positive, n)
                t_c = time.perf_counter()
                if landed is not None:
                    landed.wait()                             # the device-side staging copy of the lane's previous frame has been read out
                if copied[slot] is not None:
                    copied[slot].result()                     # the frame before that left this slot's page-locked buffers
                t_d = time.perf_counter()
                nbytes = ctx.outputs_pack(True) if self.packed else 0
                if nbytes and nbytes <= ln.blob_buf[slot].array.nbytes:
                    ctx.outputs_fetch_packed_async(ln.blob_buf[slot], nbytes)
                    packed = 1
                else:
                    nbytes, packed = 0, 0
                    ctx.outputs_stage(True)
                    ctx.outputs_fetch_async(ln.fr_buf[slot], ln.lab_buf[slot])
                # every kernel that reads the input slot has been enqueued AND the frame's last wait is behind us: the slot is free
                ln.computed[t].set()
                with self._tlock:
                    tm = self.timing
                    tm["wait_upload"] += t_b - t_a; tm["compute"] += t_c - t_b; tm["wait_download"] += t_d - t_c; tm["frames"] += 1
                    self.packed_frames += packed
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
                            hipnative.outputs_unpack(ln.blob_buf[ss], nb, fr_t, lab_t, zero_fill=self._zero_fill, threads=self.copy_threads)
                        else:
                            hipnative.outputs_unpack(ln.blob_buf[ss], nb, ln.fr_buf[ss].array, ln.lab_buf[ss].array, True, self.copy_threads)
                            self._copy(fr_t, ln.fr_buf[ss].array)
                            self._copy(lab_t, ln.lab_buf[ss].array)
                    else:
                        self._copy(out_frangi[tt], ln.fr_buf[ss].array)
                        self._copy(out_labels[tt], ln.lab_buf[ss].array)
                    if flush and hasattr(out_frangi, "flush"):
                        out_frangi.flush()
                        out_labels.flush()
                copied[slot] = self.io.submit(drain)
        finally:
            for t in range(k, num_t, L):                      # an exception must not leave the upload thread waiting for this lane
                ln.computed[t].set()
            for f in copied:
                if f is not None:
                    f.result()

    def _run(self, frames, out_frangi, out_labels, num_t, status, flush):
        L = self.n_lanes
        for t in range(num_t):
            ln = self.lanes[t % L]
            ln.uploaded[t], ln.computed[t] = threading.Event(), threading.Event()
        stop = threading.Event()
        up = self.io.submit(self._uploader, frames, num_t, stop)
        errs = []

        def work(k):
            try:
                self._lane_loop(k, frames, out_frangi, out_labels, num_t, status, flush, stop)
            except BaseException as exc:  # noqa: BLE001
                errs.append(exc)
                stop.set()
        try:
            if L == 1:
                work(0)
            else:
                ts = [threading.Thread(target=work, args=(k,)) for k in range(L)]
                for th in ts:
                    th.start()
                for th in ts:
                    th.join()
        finally:
            try:
                up.result()
            except BaseException as exc:  # noqa: BLE001
                errs.insert(0, exc)
            for ln in self.lanes:
                ln.uploaded.clear(); ln.computed.clear()
        if errs:
            raise errs[0]
        return self.stats
