#!/usr/bin/env python3
"""Config-5 frames (128 x 512 x 512) RESIDENT in HBM, L contexts on L host threads sharing one GPU: does the GPU run two frames' kernel chains side
by side?  (tools/bench_stream_lanes.py measures the same with the PCIe traffic and the host copies in; this isolates the GPU.)
    python tools/bench_resident_lanes.py [frames per lane] [lanes ...]"""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nellie_amd import pipeline as pl
from nellie_amd.synthetic import ISO_01, make_volume

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
lanes_list = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4]
fs = tuple(int(a) for a in os.environ.get("NELLIE_LANE_SHAPE", "128,512,512").split(","))
p = pl.FilterParams(dim_res=ISO_01)
minarea = pl.min_area_pixels_of(ISO_01)
out = {"frame": list(fs), "frames_per_lane": N}
for L in lanes_list:
    pipes = [pl.FramePipeline(fs) for _ in range(L)]
    for k, pipe in enumerate(pipes):
        pipe.load_input(make_volume(fs, 4567 + k))
    counts = [None] * L
    PACK = os.environ.get("NELLIE_BG_PACK") == "1"
    if PACK:
        import numpy as np
        from nellie_amd import hipnative
        blobs = [hipnative.PinnedArray((2 * fs[0] * fs[1] * fs[2] + 4096,), np.uint8) for _ in range(L)]

    def work(k, n):
        pipe = pipes[k]
        for _ in range(n):
            pipe.filter(None, p)
            counts[k] = pipe.label(pipe.frangi_threshold(), minarea)
            if PACK:
                nb = pipe.ctx.outputs_pack(True)
                pipe.ctx.outputs_fetch_packed_async(blobs[k], nb)
                pipe.ctx.outputs_wait()

    def run_all(n):
        ts = [threading.Thread(target=work, args=(k, n)) for k in range(L)]
        for t in ts: t.start()
        for t in ts: t.join()
    run_all(3)
    # NELLIE_BG_H2D=1: a thread keeps one 134 MB host -> HBM copy in flight all the time (what the streamer's upload thread does): does
    # the DMA traffic alone slow the kernels of the lanes down?   NELLIE_BG_PACK=1: every frame also packs its outputs and fetches the blob.
    bg_stop, bg_n = threading.Event(), [0]
    bg = None
    if os.environ.get("NELLIE_BG_H2D") == "1":
        import numpy as np
        from nellie_amd import hipnative
        up = pl.FramePipeline(fs)
        src = hipnative.PinnedArray(fs, np.float32)
        src.array[:] = 1.0

        def pump():
            while not bg_stop.is_set():
                up.ctx.input_load_async(bg_n[0] & 1, src)
                up.ctx.input_wait(bg_n[0] & 1)
                bg_n[0] += 1
        bg = threading.Thread(target=pump)
        bg.start()
    t0 = time.perf_counter()
    run_all(N)
    dt = time.perf_counter() - t0
    if bg is not None:
        bg_stop.set(); bg.join(); up.close(); src.free()
    out[f"lanes_{L}"] = {"ms_per_frame": round(dt / (N * L) * 1e3, 3), "labels": counts, "background_uploads": bg_n[0],
                         "background_upload_gbs": round(bg_n[0] * 4 * fs[0] * fs[1] * fs[2] / dt / 1e9, 1)}
    for pipe in pipes:
        pipe.close()
print(json.dumps(out))
