#!/usr/bin/env python3
"""
bench.py -- Mvoxel/s of Nellie's segmentation hot path (5-scale Frangi Filter + Label) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--shape Z Y X] [--no-cpu-baseline] [--no-io]

One "step" = one pass of the whole hot path (float32 conversion, 5-scale Gaussian cascade, Hessian, eigenvalues, Frangi,
scale-max, masks, percentile mask + opening, Label thresholds, hole filling, two 26-connected labellings, area filter,
majority filter, raster renumbering) over synthetic float32 data that is ALREADY RESIDENT IN HBM when the timed region
starts; outputs stay in HBM.  `value` is that figure.  The host-to-host rates SURVEY.md 8(d) also asks for (pinned host
frame in -> both outputs back in pinned host memory, and the streamed 3-D+T stack of BASELINE config 5) are reported in
the same line under `io`, never as `value`.

N = 1: BASELINE.json configs[2], one 1024 x 1024 x 1024 float32 frame (headline).
N > 1: ONE volume cut into Z slabs over the N GPUs (nellie_amd/sharded.py): every rank owns 128 planes of 2048 x 2048
       voxels -- the per-GPU share of BASELINE config 4 -- plus ghost planes, so N = 8 is exactly config 4
       (1024 x 2048 x 2048, seed 3456) and N = 2, 4 are its first 256 / 512 planes' worth of the same generator (weak
       scaling: one context holds < 2^31 voxels, so config 4 itself does not fit two GPUs).  Ghost planes, bit planes,
       scalar reductions and the sample / table gathers all travel over RCCL (xGMI).  `value` = global voxels per wall
       second of that run.  The slab run lives in a child process per rank (own rendezvous, timeout): if the
       communication path fails the line is still printed, with `zslab.error`, `value` = null and `zslab_failed` = true; the
       frame-replica figure (one 1024^3 frame per GPU, no data-path collective) is a different workload and is only
       ever reported as `replicas`.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` and `cpu_baseline`.
The oracle is used here only for the `cpu_baseline` leg and the accuracy check that rides with it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s measured float4 copy
HBM_COPY_GBS = 6290.0
MEASURED_COPY_GBS = 5000.0     # what a read + write stream reaches on these boxes (profiles/r02_stream_patterns_*.txt: 4.7-5.65 TB/s)
B_ALG_TOTAL = 301.0            # Filter 257 + Label 44 bytes/voxel
# SURVEY.md 8(d)'s pass-structured bytes per voxel, split over the kernel groups of this build (one launch each):
#   per scale: the cascade step 24 (the model's three axis passes; one fused kernel since round 5 -- gauss_zyx -- or the Z pass 8 +
#              the fused Y+X pass 16 where the radii have no fused instantiation);
#              Hessian/eigen/Frangi pass 16 + mask pass 6 = 22, of which the walk (Gaussian read 4, frob_sq 4, mask pass 6)
#              does 14 and the resolve kernel (running maximum 4 r + 4 w) 8;
#   frame: product 9 + _mask_volume 18 = 27 (one fused epilogue here); Label 44.        5 (24 + 22) + 27 + 44 = 301
B_ALG_KERNEL = {
    "load": 8.0, "gauss_z": 8.0, "gauss_yx": 16.0, "gauss_y": 8.0, "gauss_x": 8.0,
    "hessian_stats": 4.0, "vesselness": 14.0, "vesselness_resolve": 8.0,
    "finish": 9.0, "mask_volume": 27.0, "label": 44.0,
}
GAUSS_ZYX = tuple(f"gauss_zyx<{rz},{r}>" for rz in range(1, 6) for r in range(3, 6))     # the fused cascade step, one group per instantiation
B_ALG_KERNEL.update({g: 24.0 for g in GAUSS_ZYX})
B_ALG_PASS = 22.0              # walk + resolve: what SURVEY 8(d) calls the Hessian/eigen/Frangi + mask passes of a scale
# Round 6 (VERDICT r05 "next 4"): the build has outgrown that pass model -- its fused kernels never make the passes the model counts, and
# five groups "achieved" more than the HBM peak against it.  Beside it now: the FUSED-DESIGN bytes, i.e. the HBM streams each kernel must
# make AS IT IS STRUCTURED (reads of halo columns / rows that neighbouring workgroups share are meant to hit the L2 / Infinity Cache and
# are not counted; sparse reads are priced with the fractions this very run measured).  bytes per voxel and launch:
#   cascade step (gauss_zyx, or gauss_z / gauss_yx each)   4 r + 4 w; the frame's first step also zeroes the scale maximum: + 4 w
#   walk (vesselness)         4 r (Gaussian) + 2/8 (cumulative mask bit read + written) + 28 q   (q = queue entries / voxels of the scale)
#   resolve                   28 q (entries read) + 8 q (running maximum read + written at the queued voxels)
#   mask_volume               4 w (dense output) + 1 (eight bit-plane passes of 1/8: pack, opening x 2, apply) + 4 m (values read through the
#                             mask bits, m = smallest per-scale mask fraction, an upper bound of the cumulative mask) + 4 s (s = survivors)
#   label                     4 w (dense int32 labels) + 1 (bit planes: threshold, fill, majority, paint) + 4 s (Frangi values read through the bits)
# `design_frac` of a group = these bytes / its time / the HBM peak: how close the kernel runs to the floor of ITS OWN design -- never above
# `counter_frac`, which is what the PMC counters saw moving.
PMC_KERNEL_OF_GROUP = {"vesselness": ("hessian_v_kernel<2", "hessian_g_kernel<2"), "vesselness_resolve": ("vesselness_queue_kernel<true",),
                       "hessian_stats": ("hessian_v_kernel<0", "hessian_g_kernel<0"), "gauss_yx": ("gauss_yx_tile_kernel<4",),
                       "gauss_zyx<4,4>": ("gauss_zyx_kernel<4, 4>",), "gauss_zyx<3,3>": ("gauss_zyx_kernel<3, 3>",),
                       "gauss_zyx<5,5>": ("gauss_zyx_kernel<5, 5>",),
                       "gauss_z": ("gauss_march_z2_kernel<4", "gauss_march_kernel<0, 4"),
                       # groups of several kernels: every kernel whose name starts with one of these prefixes, summed per step
                       "mask_volume": ("pct_", "pack_masked_kernel", "void bits_morph6_kernel", "apply_bits_pos_kernel", "tail_"),
                       "label": ("rl_", "void rl_", "majority_bits_kernel", "root_", "ccl_flatten_kernel", "flat_gather_pos_kernel", "blk_scan_kernel",
                                 "chunk_scan_kernel", "chunk_sum_kernel"),
                       "sample": ("sample_", "chain_")}
PMC_SUMMED_GROUPS = ("mask_volume", "label", "sample")
GROUPS = ("load",) + GAUSS_ZYX + ("gauss_z", "gauss_yx", "gauss_y", "gauss_x", "sample", "hessian_stats", "vesselness", "vesselness_resolve",
          "finish", "mask_volume", "label", "halo", "halo_wait")
SLAB_PLANES = 128              # owned planes per GPU of the Z-slab run: BASELINE config 4 / 8
SLAB_YX = (2048, 2048)


def pmc_profile_file():
    """The committed PMC profile the `traffic` / `counter_*` fields fall back to when this run could not measure them, repo-relative."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_hbm_bytes_1024cube.json")))
    return os.path.relpath(files[-1], REPO) if files else None


def sq_bound_of(kernel_group):
    """What the committed SQ-counter pass says bounds the dominant kernel (profiles/r*_pmc_sq_*.json: {"kernel_group": ..., "bound":
    "hbm" | "issue", ...}); the roofline object keeps the contract's vocabulary in `bound` and carries this verdict beside it."""
    import glob
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_sq_*.json")), reverse=True):
        try:
            rec = json.load(open(f))
        except ValueError:
            continue
        if rec.get("kernel_group") == kernel_group:
            return {"limited_by": rec.get("bound"), "source": os.path.relpath(f, REPO), "evidence": rec.get("evidence")}
    return None


class PmcTable:
    """HBM bytes per kernel launch: rows {"kernel", "launches", "fetch_size_kb_per_launch", "write_size_kb_per_launch"} from separate
    rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (KB units; FETCH_SIZE x 2 on gfx950: MI355X_MICROARCH.md, HBM section).
    `source`: "this run" (pmc_this_run: one step of this very command re-executed under the counters, outside the timed region) or the
    committed file's path."""

    def __init__(self, rows, source, shape):
        self.rows, self.source, self.shape = rows, source, tuple(shape)

    @staticmethod
    def committed(shape):
        f = pmc_profile_file()
        if f is None or tuple(shape) != (1024, 1024, 1024):
            return PmcTable([], None, shape)
        return PmcTable(json.load(open(os.path.join(REPO, f))), f, shape)

    @staticmethod
    def _bytes(rec):
        return (2.0 * rec["fetch_size_kb_per_launch"] + rec["write_size_kb_per_launch"]) * 1024.0

    def traffic(self, group):
        """bytes per launch of a one-kernel group; for the groups of many small kernels (PMC_SUMMED_GROUPS) bytes per STEP."""
        pats = PMC_KERNEL_OF_GROUP.get(group)
        if not pats or not self.rows:
            return None
        hit = [r for r in self.rows if any(r["kernel"].startswith(pat) or (" " + pat) in r["kernel"] for pat in pats)]
        if not hit:
            return None
        if group in PMC_SUMMED_GROUPS:
            return round(sum(self._bytes(r) * r.get("launches", 0) for r in hit) / max(1, self.steps()))
        return round(self._bytes(hit[0]))

    def steps(self):
        """passes over the volume the profiled command made (the committed files and pmc_this_run both profile ONE step)"""
        return 1

    def bytes_per_step(self):
        if not self.rows:
            return None
        return sum(self._bytes(r) * r.get("launches", 0) for r in self.rows) / max(1, self.steps())


def pmc_this_run(shape, seed, vol, timeout_s=240.0):
    """One step of this very workload re-executed under `rocprofv3 --pmc FETCH_SIZE` and, separately, `--pmc WRITE_SIZE` (counters only,
    no tracing; the two do not fit one pass: MI355X_MICROARCH.md, PMC slots), OUTSIDE the timed region, in child processes that read the
    volume from /dev/shm.  Returns a PmcTable with source "this run", or None (no rocprofv3 on PATH, a failed pass, a timeout)."""
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3")
    if rocprof is None:
        return None
    tmp = tempfile.mkdtemp(prefix="nellie_pmc_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        vpath = os.path.join(tmp, "vol.npy")
        np.save(vpath, vol)
        acc = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [rocprof, "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", vpath, "--shape"] + [str(v) for v in shape]
            env = dict(os.environ, TMPDIR=tmp)
            r = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0:
                return None
            files = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith("counter_collection.csv")]
            if not files:
                return None
            import csv
            for row in csv.DictReader(open(files[0])):
                if row.get("Counter_Name") != counter:
                    continue
                k = row["Kernel_Name"].split("(")[0]
                a = acc.setdefault(k, {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
                a[counter][0] += float(row["Counter_Value"]); a[counter][1] += 1
        rows = []
        for k, a in sorted(acc.items(), key=lambda kv: -kv[1]["FETCH_SIZE"][0]):
            L = max(a["FETCH_SIZE"][1], a["WRITE_SIZE"][1])
            rows.append({"kernel": k, "launches": L, "fetch_size_kb_per_launch": a["FETCH_SIZE"][0] / max(1, a["FETCH_SIZE"][1]),
                         "write_size_kb_per_launch": a["WRITE_SIZE"][0] / max(1, a["WRITE_SIZE"][1])})
        return PmcTable(rows, "this run", shape) if rows else None
    except (subprocess.TimeoutExpired, OSError, ValueError, KeyError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_child_main(args):
    """internal (pmc_this_run): ONE step of the hot path on the volume the parent saved -- what the counters of this process are about."""
    from nellie_amd import pipeline as pl
    from nellie_amd.synthetic import ISO_01
    vol = np.load(args.pmc_child)
    pipe = pl.FramePipeline(vol.shape)
    pipe.load_input(vol)
    p = pl.FilterParams(dim_res=ISO_01)
    pipe.filter(None, p)
    n = pipe.label(pipe.frangi_threshold(), pl.min_area_pixels_of(ISO_01))
    pipe.ctx.sync()
    print("pmc child: labels", n, flush=True)
    pipe.close()


def queue_fractions(pl, pipe, p):
    """Queue entries per voxel of every scale (the `q` of the fused-design byte model): one more frame on the synchronous path -- same
    bits as the chain -- with the entry count read back after each scale's walk (nl_ctx_info "queue_entries").  Outside the timed region."""
    ctx = pipe.ctx
    real = ctx.vesselness_spec
    n = float(np.prod(pipe.shape))
    q = []

    def counted(*a, **kw):
        r = real(*a, **kw)
        q.append(ctx.info("queue_entries") / n)
        return r
    chain = pipe._device_chain
    try:
        pipe._device_chain = False
        ctx.vesselness_spec = counted
        pipe.filter(None, p)
    except Exception:  # noqa: BLE001  (a figure of the model, not of the measurement: never costs the line)
        return None
    finally:
        ctx.vesselness_spec = real
        pipe._device_chain = chain
    return [round(v, 5) for v in q] or None


class Control:
    """The control plane of an N > 1 run -- a barrier, the maximum over the ranks, a value from rank 0 -- over the PRODUCT's own file
    rendezvous (nellie_amd/rendezvous.py, the one `shard="env"` launches use): the bench needs torch for nothing (round 5; the
    driver may still start the ranks with torch.distributed.run, only RANK / WORLD_SIZE / MASTER_PORT of its environment are read).
    Barriers that bracket a timed region poll every 0.2 ms; every rank stops ITS clock after its own device sync, the maximum over
    the ranks is taken afterwards -- what the barrier-bracketed region of the contract measures, without the polling interval in it."""

    def __init__(self, rank, world, tag, timeout_s=600.0):
        import tempfile
        from nellie_amd.rendezvous import FileRendezvous
        d = os.environ.get("NELLIE_RENDEZVOUS_DIR") or os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir(), "nellie_bench_rdv")
        self.rank, self.world = rank, world
        self.rdv = FileRendezvous(rank, world, d, tag=str(tag), timeout_s=timeout_s, poll_s=0.0002)

    def barrier(self, name="b"):
        self.rdv.barrier(name)

    def allgather(self, name, obj):
        return [json.loads(b.decode()) for b in self.rdv.allgather(name, json.dumps(obj).encode())]

    def max(self, name, value):
        return max(float(v) for v in self.allgather(name, float(value)))

    def min_ints(self, name, values):
        rows = self.allgather(name, [int(v) for v in values])
        return [min(col) for col in zip(*rows)]

    def from_rank0(self, name, make):
        """rank 0's bytes (e.g. a communicator id), hex-encoded in transit"""
        rows = self.allgather(name, make().hex() if self.rank == 0 else "")
        return bytes.fromhex(rows[0])


def launch_ranks(n, argv):
    """`python bench.py --gpus N` started by hand: become the launcher (one process per GPU with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set, as torch.distributed.run would).  Returns the worst exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    return max(abs(p.wait()) for p in procs)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--shape", type=int, nargs=3, default=None, help="frame Z Y X (default 1024^3)")
    ap.add_argument("--seed", type=int, default=2345)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-shape", type=int, nargs=3, default=[256, 512, 512], help="CPU baseline sample (default: BASELINE config 2)")
    ap.add_argument("--no-io", action="store_true", help="skip the host-to-host and streamed figures")
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="skip the frame-parallel all-cores CPU baseline")
    ap.add_argument("--cpu-all-cores-child", action="store_true", help="internal: the all-cores CPU baseline (a process that never loads HIP)")
    ap.add_argument("--cpu-all-cores-shape", type=int, nargs=3, default=[64, 256, 256],
                    help="frame per worker of the all-cores baseline (default: an eighth of a BASELINE config 5 frame -- 256 workers on full frames "
                         "need ~1 TB of host RAM and took the GPU box down in round 4)")
    ap.add_argument("--no-zslab", action="store_true", help="N > 1: frame replicas only")
    ap.add_argument("--zslab-timeout", type=float, default=420.0)
    ap.add_argument("--zslab-child", action="store_true", help="internal: the Z-slab run (spawned by the bench)")
    ap.add_argument("--zslab-planes", type=int, default=SLAB_PLANES)
    ap.add_argument("--zslab-yx", type=int, nargs=2, default=list(SLAB_YX))
    ap.add_argument("--zslab-on-one-gpu", type=int, default=0, metavar="W",
                    help="only: the W-slab volume (W x --zslab-planes x --zslab-yx) as W slab contexts on ONE GPU over the loopback transport")
    ap.add_argument("--no-pmc", action="store_true", help="do not re-execute one step under rocprofv3 --pmc (counter figures then come from profiles/)")
    ap.add_argument("--pmc-child", default=None, metavar="VOL.npy", help="internal: one step on the saved volume (the process rocprofv3 --pmc wraps)")
    ap.add_argument("--share-device", action="store_true",
                    help="testing only: every rank uses device 0 (exercise the multi-process control flow on a 1-GPU box)")
    return ap.parse_args()


def cpu_baseline(shape, seed):
    """The oracle (CPU restatement of the reference, kind 'port') timed on the host cores."""
    from nellie_amd.synthetic import ISO_01, make_volume
    from oracle import nellie_oracle as orc
    orc.build_c_helper()
    vol = make_volume(shape, seed)
    t0 = time.perf_counter()
    run = orc.run_frame(vol, ISO_01)                      # filter_frame = run_frame + mask_volume, kept apart for the parity check
    fr, thr = orc.mask_volume(run, return_thr=True) if float(np.sum(run)) > 0.0 else (run, None)
    t1 = time.perf_counter()
    lab = orc.label_frame(fr, ISO_01)
    t2 = time.perf_counter()
    n = float(np.prod(shape))
    return {
        "value": round(n / (t2 - t0) / 1e6, 4), "unit": "Mvoxel/s", "cores": 1, "kind": "port",
        # the reference ITSELF (numpy / scipy path, 8 vCPU of the build container) on this sample's shape; the
        # oracle is ~3x that because its eigen-solve is closed-form where the reference calls LAPACK per voxel
        "reference_anchor_mvoxel_s": 0.391,
        "sample": f"oracle Filter+Label on a synthetic {shape[0]}x{shape[1]}x{shape[2]} float32 volume "
                  f"(BASELINE config 2 when 256x512x512; same generator, seed {seed}); Filter {n / (t1 - t0) / 1e6:.3f} Mvoxel/s, "
                  f"Label {n / (t2 - t1) / 1e6:.2f} Mvoxel/s, {float(np.mean(fr > 0)) * 100:.2f}% voxels survive, "
                  f"{int(lab.max())} labels; numpy oracle, 1 thread of {os.cpu_count()} host cores (numpy/scipy kernels are single-threaded here)",
    }, (vol, run, fr, thr, lab)


def _oracle_frame(job):
    """One frame through the oracle in a worker process: (seconds inside the worker, labels)."""
    shape, seed = job
    from nellie_amd.synthetic import ISO_01, make_volume
    from oracle import nellie_oracle as orc
    vol = make_volume(shape, seed)
    t0 = time.perf_counter()
    fr = orc.filter_frame(vol, ISO_01)
    lab = orc.label_frame(fr, ISO_01)
    return time.perf_counter() - t0, int(lab.max())


def cpu_all_cores_child(args):
    """north_star: "the reference's own CPU path timed on the same box's host cores (core count stated)".  The path is single-threaded
    numpy / scipy, so all cores are used the way a user would use them on a 3-D+T stack: FRAME-PARALLEL, one worker process per
    core, one frame of BASELINE config 5's size each (seeds 4567 + i).  Wall time covers the workers' compute only (the volumes
    are generated inside the workers before their timers start; the pool is warm)."""
    import multiprocessing as mp
    from oracle import nellie_oracle as orc
    orc.build_c_helper()
    shape = tuple(args.cpu_all_cores_shape)
    cores = os.cpu_count() or 1
    budget_gb = float(os.environ.get("NELLIE_BENCH_CPU_RAM_GB", "0")) or None
    try:
        import psutil
        avail = psutil.virtual_memory().available / 1e9
    except ImportError:
        avail = 64.0
    try:                                                                # a container's limit may be far below what the host reports
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            avail = min(avail, float(lim) / 1e9)
    except OSError:
        pass
    per_proc_gb = 30.0 * float(np.prod(shape)) * 4 / 1e9 + 0.4        # ~22 float32 volume equivalents at the peak, + margin
    workers = max(1, min(cores, int((budget_gb or min(0.25 * avail, 160.0)) / per_proc_gb)))
    with mp.get_context("fork").Pool(workers) as pool:
        pool.map(_oracle_frame, [((8, 32, 32), 1)] * workers)           # imports done, pool warm
        t0 = time.perf_counter()
        res = pool.map(_oracle_frame, [(shape, 4567 + i) for i in range(workers)], chunksize=1)
        wall = time.perf_counter() - t0
    n = float(np.prod(shape)) * workers
    what = {(128, 512, 512): "a BASELINE config 5 frame", (64, 256, 256): "the generator of BASELINE config 5, an eighth of its frame"}.get(shape, "same generator")
    print(json.dumps({
        "value": round(n / wall / 1e6, 3), "unit": "Mvoxel/s", "cores": workers, "host_cores": cores, "kind": "port",
        "sample": f"oracle Filter+Label, frame-parallel: {workers} worker processes (one per core{'' if workers == cores else ', capped by free RAM'}), "
                  f"one synthetic {shape[0]}x{shape[1]}x{shape[2]} float32 frame each ({what}, seeds 4567+i); "
                  f"wall {wall:.1f} s, slowest worker {max(r[0] for r in res):.1f} s, fastest {min(r[0] for r in res):.1f} s, "
                  f"labels per frame {min(r[1] for r in res)}..{max(r[1] for r in res)}",
        "per_core_mvoxel_s": round(n / wall / 1e6 / workers, 4),
    }), flush=True)


def cpu_baseline_all_cores(args):
    """Runs cpu_all_cores_child in a fresh interpreter (fork-based worker pools and an initialised HIP runtime do not mix)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-all-cores-child", "--cpu-all-cores-shape"] + [str(v) for v in args.cpu_all_cores_shape]
    env = dict(os.environ)
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        env[var] = "1"                                     # one thread per worker process
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=float(os.environ.get("NELLIE_BENCH_CPU_ALL_TIMEOUT", "420")))
        if r.returncode != 0:
            return {"error": (r.stderr or r.stdout)[-400:]}
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as exc:  # noqa: BLE001
        return {"error": repr(exc)[:400]}


def accuracy_check(pl, vol, ref_run, ref_fr, ref_thr, ref_lab):
    """Every timing run also runs the parity check on the CPU-baseline volume: Frangi within
    |a - b| <= 1e-4 |ref| + 1e-6 max|ref| with identical support -- outside the threshold-tie zone: a voxel whose
    unmasked value lies within that tolerance of the percentile threshold of _mask_volume may fall on either side, and
    the opening carries the decision to voxels within L1 distance 2 (counted and reported) --, labels bit-exact given
    the oracle's Frangi frame, and the end-to-end label agreement."""
    from nellie_amd.synthetic import ISO_01
    from oracle import nellie_oracle as orc
    pipe = pl.FramePipeline(vol.shape)
    pipe.filter(vol, pl.FilterParams(dim_res=ISO_01))
    fr = pipe.download_frangi()
    scale = float(ref_fr.max()) if ref_fr.size else 1.0
    err = np.abs(fr.astype(np.float64) - ref_fr)
    bad = err > 1e-4 * np.abs(ref_fr) + 1e-6 * scale
    if ref_thr is None:
        zone = np.zeros(ref_fr.shape, bool)
    else:
        zone = np.abs(ref_run - np.float32(ref_thr)) <= (2e-4 * abs(float(ref_thr)) + 1e-6 * float(ref_run.max()))
    n_tie = int(zone.sum())
    for _ in range(2):
        zone = orc.binary_dilation6(zone)
    used = int((zone & (fr != ref_fr)).sum())
    ok_out = not bool((bad & ~zone).any()) and bool(np.array_equal((fr > 0) & ~zone, (ref_fr > 0) & ~zone))
    pipe.label(pipe.frangi_threshold(), pl.min_area_pixels_of(ISO_01))
    e2e = float(np.mean(pipe.download_labels() == ref_lab))
    pipe.upload_frangi(ref_fr)
    pipe.label(pipe.frangi_threshold(), pl.min_area_pixels_of(ISO_01))
    lab_ok = bool(np.array_equal(pipe.download_labels(), ref_lab))
    pipe.close()
    return {"volume": list(vol.shape), "frangi_within_tol": ok_out, "frangi_max_norm_err_outside_tie_zone": float(err[~zone].max() / scale) if scale else 0.0,
            "threshold_tie_voxels": n_tie, "voxels_differing_inside_tie_zone": used,
            "percentile_threshold_rel_diff": abs(pipe.trace.percentile_thr - float(ref_thr)) / float(ref_thr) if ref_thr else 0.0,
            "labels_bit_exact_given_same_frangi": lab_ok, "labels_end_to_end_match_fraction": round(e2e, 6)}


def io_figures(pl, hipnative, shape, p, min_area, vol):
    """Host-to-host rates (SURVEY 8(d)'s metric definition): a frame in pinned host memory -> both outputs back in pinned host
    memory (blocking transfers around the resident step), and BASELINE config 5 streamed (H2D / D2H overlapped with compute)."""
    out = {}
    n = float(np.prod(shape))
    pin_in = hipnative.PinnedArray(shape, np.float32)
    pin_fr = hipnative.PinnedArray(shape, np.float32)
    pin_lab = hipnative.PinnedArray(shape, np.int32)
    pin_in.array[...] = vol
    pipe = pl.FramePipeline(shape)
    for rep in range(2):                                         # the second pass is the figure (first touches of the pages done)
        pipe.ctx.sync()
        t0 = time.perf_counter()
        pipe.load_input(pin_in.array)
        pipe.filter(None, p)
        pipe.label(pipe.frangi_threshold(), min_area)
        pipe.download_frangi(out=pin_fr.array)
        pipe.download_labels(out=pin_lab.array)
        dt = time.perf_counter() - t0
    out["pinned_host_to_host_mvoxel_s"] = round(n / dt / 1e6, 1)
    out["pinned_host_to_host_ms"] = round(dt * 1e3, 1)
    out["bytes_over_pcie_per_voxel"] = 12
    # the same with the packed download (nl_outputs_pack: bit planes + non-zero values + one label per run; the host expands it
    # into the same dense page-locked arrays, zero-filling them: every byte of both outputs is written)
    blob = hipnative.PinnedArray((2 * int(n) + 4096,), np.uint8)
    threads = max(1, min(16, (os.cpu_count() or 2) // 2))
    import threading
    for rep in range(2):
        pin_fr.array[...] = 1.0
        pin_lab.array[...] = -1
        pipe.ctx.sync()
        t0 = time.perf_counter()
        # the zero fill of both dense outputs (8 B/voxel of host memory traffic) runs on host threads WHILE the frame travels and the
        # GPU works (round 5); the unpack then only scatters the non-zero items.  Everything inside the timed region.
        zero = threading.Thread(target=lambda: (hipnative.host_zero(pin_fr.array, threads), hipnative.host_zero(pin_lab.array, threads)))
        zero.start()
        pipe.load_input(pin_in.array)
        pipe.filter(None, p)
        pipe.label(pipe.frangi_threshold(), min_area)
        nb = pipe.ctx.outputs_pack(True)
        if nb:
            pipe.ctx.outputs_fetch_packed_async(blob, nb)
            pipe.ctx.outputs_wait()
            zero.join()
            hipnative.outputs_unpack(blob, nb, pin_fr.array, pin_lab.array, zero_fill=False, threads=threads)
        zero.join()
        dtp = time.perf_counter() - t0
    if nb:
        out["pinned_host_to_host_packed_mvoxel_s"] = round(n / dtp / 1e6, 1)
        out["pinned_host_to_host_packed_ms"] = round(dtp * 1e3, 1)
        out["packed_bytes_over_pcie_per_voxel"] = round(4.0 + nb / n, 3)
        out["packed_unpack_threads"] = threads
        out["packed_zero_fill"] = "on host threads, concurrent with the upload and the GPU step (inside the timed region)"
    blob.free()
    pipe.close()
    for a in (pin_in, pin_fr, pin_lab):
        a.free()
    # BASELINE config 5: a 3-D+T stack of 64 frames of 128 x 512 x 512, host arrays in, host arrays out
    from nellie_amd.streaming import StreamedSegmenter
    from nellie_amd.synthetic import make_volume
    T, fs = int(os.environ.get("NELLIE_BENCH_C5_FRAMES", "64")), (128, 512, 512)
    frames = np.stack([make_volume(fs, 4567 + t) for t in range(T)])
    fr, lab = np.empty(frames.shape, np.float32), np.empty(frames.shape, np.int32)
    seg = StreamedSegmenter(fs, frames.dtype, p)
    seg.run(frames, fr, lab, flush=False)                        # first touch of the output pages
    t0 = time.perf_counter()
    seg.run(frames, fr, lab, flush=False)
    dt = time.perf_counter() - t0
    seg.close()
    out["streamed_config5_mvoxel_s"] = round(frames.size / dt / 1e6, 1)
    out["streamed_config5_ms_per_frame"] = round(dt / T * 1e3, 2)
    out["streamed_config5_stack"] = [T] + list(fs)
    return out


def design_bytes_per_voxel(group, g, steps, facts):
    """Fused-design bytes per voxel and launch of a group (the table in the header); None where the model has no entry.  facts: what the
    run measured -- q (queue entries per voxel, per scale), mask_fraction (per scale), survival, zeroing cascade launches per step."""
    q = facts.get("queue_fraction_per_scale")
    qm = None if not q else sum(q) / len(q)
    if group in GAUSS_ZYX or group in ("gauss_z", "gauss_yx", "gauss_y", "gauss_x"):
        b = 8.0
        if group == facts.get("zeroing_group"):          # the frame's first cascade step also writes the zeros of the scale maximum
            b += 4.0 * steps / max(1, g["launches"])
        return b
    if group == "vesselness":
        return None if qm is None else 4.0 + 0.25 + 28.0 * qm
    if group == "vesselness_resolve":
        return None if qm is None else 36.0 * qm
    s_ = facts.get("survival")
    m_ = facts.get("mask_fraction_min")
    if group == "mask_volume" and s_ is not None and m_ is not None:
        return 4.0 + 1.0 + 4.0 * m_ + 4.0 * s_
    if group == "label" and s_ is not None:
        return 4.0 + 1.0 + 4.0 * s_
    return None


def roofline_of(groups, shape, steps, ms_per_step, pmc=None, facts=None):
    """The `roofline` object.  Top level: the dominant kernel against SURVEY 8(d)'s algorithmic bytes (`achieved` / `frac`, the contract's
    definition) with the PMC bytes of this run beside it (`traffic`).  `groups`: per kernel group the time, the counter bytes and -- since
    round 6 -- the fused-design bytes; no figure of this object is a rate against the pass model except the dominant kernel's `achieved`."""
    facts = facts or {}
    pmc = pmc or PmcTable.committed(shape)
    n_local = float(np.prod(shape))
    kernel_ms_per_step = sum(g["ms_total"] for g in groups.values()) / max(1, steps)
    dom = max((g for g in groups if g in B_ALG_KERNEL), key=lambda g: groups[g]["ms_total"])
    dom_bytes = B_ALG_KERNEL[dom] * n_local
    dom_gbs = dom_bytes / (groups[dom]["ms_avg"] * 1e-3) / 1e9
    traffic = pmc.traffic(dom)
    step_bytes = pmc.bytes_per_step()
    table = {}
    design_step = 0.0
    design_complete = True
    for name, g in groups.items():
        t = pmc.traffic(name)
        per_step = name in PMC_SUMMED_GROUPS          # groups of many small kernels: bytes and time per step, not per launch
        ms = g["ms_total"] / max(1, steps) if per_step else g["ms_avg"]
        d = design_bytes_per_voxel(name, g, steps, facts)
        d_bytes = None if d is None else d * n_local            # per launch (mask_volume and label are timed once per step)
        row = {"ms_per_step": round(g["ms_total"] / max(1, steps), 3), "launches_per_step": round(g["launches"] / max(1, steps), 2),
               "ms_avg": round(g["ms_avg"], 4),
               "pass_model_bytes_per_voxel": B_ALG_KERNEL.get(name),
               "design_bytes_per_voxel": None if d is None else round(d, 3),
               "design_gbs": None if d is None else round(d_bytes / (g["ms_avg"] * 1e-3) / 1e9, 1),
               "design_frac": None if d is None else round(d_bytes / (g["ms_avg"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "counter_bytes_per_voxel": None if t is None else round(t / n_local, 3),
               "counter_gbs": None if t is None else round(t / (ms * 1e-3) / 1e9, 1),
               "counter_frac": None if t is None else round(t / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "traffic_over_design": None if (t is None or d is None) else round(t / d_bytes, 3)}
        table[name] = row
        if d is not None:
            design_step += d_bytes * g["launches"] / max(1, steps)
        elif name in B_ALG_KERNEL:
            design_complete = False
    sq = sq_bound_of(dom)
    out = {
        # `bound`: the roofline this byte / integer path is priced against (MFMA is not used); `limited_by`: what the SQ counters say
        # actually holds the dominant kernel back (profiles/r*_pmc_sq_*.json) -- "issue" means instruction issue / LDS latency, not HBM
        "bound": "hbm", "limited_by": None if sq is None else sq.get("limited_by"),
        "kernel": dom, "achieved": round(dom_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(dom_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
        # where `traffic` and every counter_* figure below come from: "this run" (one step of this command re-executed under
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE outside the timed region) or, failing that, the committed profile of the same command
        "traffic_source": pmc.source if traffic is not None else None,
        "limited_by_counters": sq,
        "achieved_by_counters": None if traffic is None else round(traffic / (groups[dom]["ms_avg"] * 1e-3) / 1e9, 1),
        "frac_by_counters": None if traffic is None else round(traffic / (groups[dom]["ms_avg"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "frac_of_measured_copy_peak": round(dom_gbs / HBM_COPY_GBS, 4),
        "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": round(groups[dom]["ms_avg"], 4),
        "design_bytes_per_launch": None if table[dom]["design_bytes_per_voxel"] is None else round(table[dom]["design_bytes_per_voxel"] * n_local),
        "design_frac": table[dom]["design_frac"],
        "design_model_facts": facts or None,
        "groups": table,
        # Whole step.  `frac_by_counters` is the utilisation: bytes the PMC passes saw moving / wall time / peak.  `design_*`: the fused
        # design's own floor (sum of the groups' design bytes).  SURVEY 8(d)'s pass model (301 B/voxel) is kept as a constant for
        # reference only: the fused kernels never make the passes it counts, so no rate is quoted against it any more.
        "pipeline": {
            "kernel_ms_per_step": round(kernel_ms_per_step, 3),
            "counter_bytes_per_voxel": None if step_bytes is None else round(step_bytes / n_local, 1),
            "achieved_by_counters": None if step_bytes is None else round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
            "frac_by_counters": None if step_bytes is None else round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "design_bytes_per_voxel": round(design_step / n_local, 1) if design_complete and design_step else None,
            "design_gbs": round(design_step / (ms_per_step * 1e-3) / 1e9, 1) if design_complete and design_step else None,
            "design_frac": round(design_step / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if design_complete and design_step else None,
            "ms_floor_of_the_design_at_hbm_peak": round(design_step / (HBM_PEAK_GBS * 1e9) * 1e3, 2) if design_complete and design_step else None,
            "pass_model_bytes_per_voxel_survey_8d": B_ALG_TOTAL,
            "ms_floor_at_measured_copy_rate": None if step_bytes is None else round(step_bytes / (MEASURED_COPY_GBS * 1e9) * 1e3, 2),
        },
    }
    if "vesselness" in groups and "vesselness_resolve" in groups:
        # the Hessian -> eigen -> Frangi pass of a scale: walk + resolve together, by the counters and against the design
        t_pass = groups["vesselness"]["ms_avg"] + groups["vesselness_resolve"]["ms_avg"]
        tw, tr = pmc.traffic("vesselness"), pmc.traffic("vesselness_resolve")
        dw, dr = table["vesselness"]["design_bytes_per_voxel"], table["vesselness_resolve"]["design_bytes_per_voxel"]
        out["hessian_eigen_pass"] = {"ms_per_scale": round(t_pass, 4),
                                     "design_bytes_per_voxel": None if dw is None or dr is None else round(dw + dr, 3),
                                     "design_frac": None if dw is None or dr is None else round((dw + dr) * n_local / (t_pass * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                     "achieved_by_counters": None if tw is None or tr is None else round((tw + tr) / (t_pass * 1e-3) / 1e9, 1)}
    return out


def main():
    args = parse_args()
    if args.cpu_all_cores_child:
        cpu_all_cores_child(args)
        return
    if args.zslab_child:
        zslab_child_main(args)
        return
    if args.pmc_child:
        pmc_child_main(args)
        return
    if args.zslab_on_one_gpu:
        print(json.dumps(zslab_on_one_gpu(args.zslab_on_one_gpu, args.zslab_planes, args.zslab_yx, 0, args.steps, max(1, args.warmup))), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        # started by hand as `python bench.py --gpus N`: become the launcher the contract describes (one rank per GPU)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    n_gpus = world if world > 1 else args.gpus
    dist = None
    if world > 1:
        dist = Control(rank, world, tag="bench_" + os.environ.get("MASTER_PORT", "29500"))

    from nellie_amd import hipnative
    from nellie_amd import pipeline as pl
    from nellie_amd.synthetic import ISO_01, make_volume

    lib = hipnative.load()
    if args.share_device:
        local_rank = 0
    if lib.device_count() <= local_rank:
        raise RuntimeError(f"GPU backend requested but device {local_rank} is not visible")

    shape = tuple(args.shape) if args.shape else (1024, 1024, 1024)
    p = pl.FilterParams(dim_res=ISO_01)
    min_area = pl.min_area_pixels_of(ISO_01)

    # ---- the resident-frame measurement: rank r owns frame r of an (N, Z, Y, X) stack (N = 1: THE headline figure)
    t_gen = time.perf_counter()
    vol = make_volume(shape, args.seed + rank)
    t_gen = time.perf_counter() - t_gen
    pipe = pl.FramePipeline(shape, device=local_rank)
    t_up = time.perf_counter()
    pipe.load_input(vol)
    t_up = time.perf_counter() - t_up

    def step():
        pipe.filter(None, p)
        thr = pipe.frangi_threshold()
        return pipe.label(thr, min_area)

    def barrier():
        pipe.ctx.sync()
        if dist is not None:
            dist.barrier("step")

    pipe.ctx.prof_enable(True)      # the event pairs of the per-group timers exist (and have been used) before the timed region
    for _ in range(args.warmup):
        step()
    if os.environ.get("NELLIE_BENCH_GC", "freeze") == "freeze":
        # the interpreter's cyclic collector stays out of the timed region, as in timeit: with torch imported (N > 1) a full
        # collection takes 50-100 ms and used to land in one step of a run (round 3's "one-off stall of the queue": found in round 4)
        import gc
        gc.collect(); gc.freeze()
    pipe.ctx.prof_reset()
    barrier()
    t0 = time.perf_counter()
    n_labels = 0
    for _ in range(args.steps):
        n_labels = step()
    pipe.ctx.sync()
    elapsed = time.perf_counter() - t0             # this rank's clock stops after ITS device sync ...
    if dist is not None:
        dist.barrier("step")
        elapsed = dist.max("elapsed", elapsed)     # ... and the figure is the maximum over the ranks (see Control)
    pipe.ctx.prof_enable(False)

    groups = {}
    for name in GROUPS:
        ms, k = pipe.ctx.prof_get(name)
        if k:
            groups[name] = {"ms_total": ms, "launches": k, "ms_avg": ms / k}
    n_local = float(np.prod(shape))
    ms_per_step = elapsed / args.steps * 1e3
    tr = pipe.trace
    # what the fused-design byte model needs from the run (outside the timed region): queue entries per scale, mask / survival fractions
    sig = p.resolved_sigmas()
    d0 = pl.cascade_deltas(sig, pl.z_ratio_of(ISO_01))[0] if len(sig) else (0.0, 0.0, 0.0)
    facts = {"queue_fraction_per_scale": queue_fractions(pl, pipe, p),
             "mask_fraction_per_scale": [round(sc.mask_count / n_local, 4) for sc in tr.scales],
             "mask_fraction_min": round(min([sc.mask_count / n_local for sc in tr.scales if not sc.skipped] or [0.0]), 4),
             "survival": round(tr.n_positive / n_local, 5),
             "zeroing_group": "gauss_zyx<%d,%d>" % (int(3.0 * d0[0] + 0.5), int(3.0 * d0[1] + 0.5))}
    pmc = None
    if rank == 0 and world == 1 and not args.no_pmc:
        t_pmc = time.perf_counter()
        pmc = pmc_this_run(shape, args.seed, vol)
        facts["pmc_this_run_s"] = round(time.perf_counter() - t_pmc, 1)
    roofline = roofline_of(groups, shape, args.steps, ms_per_step, pmc=pmc, facts=facts)
    # Since round 5 the chain runs the resolve kernel of scale s on the side stream beside the threshold kernels ("sample") of scale s+1
    # on volumes of 2^26 voxels and more (nl_chain_scale).  Each group's HIP-event timer then counts the time it shared: the groups sum
    # to more than the step.  Say so in the line instead of leaving a sum that exceeds `ms_per_step` unexplained.
    if os.environ.get("NELLIE_RESOLVE_DEFER", "1") != "0" and float(np.prod(shape)) >= float(1 << 26) and pipe._chain_usable(p, True):
        roofline["pipeline"]["overlapped_groups"] = ["sample", "vesselness_resolve"]
        roofline["pipeline"]["kernel_ms_per_step_note"] = ("sample and vesselness_resolve run side by side on two streams; their timers each count "
                                                          "the shared time, so kernel_ms_per_step (the sum of the groups) exceeds ms_per_step")
    chain_info = {"enabled": bool(pipe._chain_usable(p, True)), "frames_redone_synchronously": int(pipe.chain_fallbacks),
                  "last_flags": getattr(pipe, "last_chain_flags", None)}
    fast_div = int(pipe.ctx.info("fast_div"))
    tile_rows = int(pipe.ctx.info("hessian_tile_rows"))
    pipe.close()

    io = None
    if not args.no_io and world == 1:
        io = io_figures(pl, hipnative, shape, p, min_area, vol)
    del vol

    out = None
    if rank == 0:
        cpu = None
        acc = None
        cpu_all = None
        if not args.no_cpu_baseline:
            cpu, (cvol, crun, cfr, cthr, clab) = cpu_baseline(tuple(args.cpu_shape), 1234)
            acc = accuracy_check(pl, cvol, crun, cfr, cthr, clab)
            del cvol, crun, cfr, clab
            if not args.no_cpu_all_cores:
                cpu_all = cpu_baseline_all_cores(args)
        replica_value = n_local * n_gpus * args.steps / elapsed / 1e6
        out = {
            # `value`: the frame is ALREADY in HBM when the timed region starts and the outputs stay there (the contract's
            # definition); SURVEY 8(d)'s host-to-host definition is `host_to_host_mvoxel_s` below (PCIe-bound, never `value`)
            "metric": "Mvoxel/s multiscale Frangi (5 sigma) + Label, float32, frame resident in HBM", "value": round(replica_value, 1),
            "unit": "Mvoxel/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak" if n_gpus > 1 else None, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "host_to_host_mvoxel_s": None if io is None else io.get("pinned_host_to_host_mvoxel_s"),
            "host_to_host_packed_mvoxel_s": None if io is None else io.get("pinned_host_to_host_packed_mvoxel_s"),
            "config": {
                "workload": f"synthetic {shape[0]}x{shape[1]}x{shape[2]} float32 volume "
                            f"(N(100,5) noise + Gaussian tube segments, seed {args.seed}), 0.1 um isotropic, "
                            f"{len(p.resolved_sigmas())}-scale Frangi + Label, full hot path per step, frame resident in HBM",
                "voxels": int(n_local * n_gpus), "per_gpu_shape": list(shape),
                "survival_fraction": round(tr.n_positive / n_local, 5), "labels": int(n_labels),
                "mask_fraction_per_scale": [round(sc.mask_count / n_local, 4) for sc in tr.scales],
                "one_pass_scales": int(sum(1 for sc in tr.scales if sc.one_pass)),
                "host_gen_s": round(t_gen, 1), "h2d_s": round(t_up, 2), "fast_div_proven": fast_div, "hessian_tile_rows": tile_rows,
                "device_chain": chain_info,
            },
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_all,
        }
        if acc is not None:
            out["accuracy"] = acc
        if io is not None:
            out["io"] = io

    if dist is None:
        print(json.dumps(out), flush=True)
        return

    # ---- N > 1: the measured line is the Z-slab run of ONE volume over RCCL.  It runs in a CHILD process per rank (own
    # rendezvous on another port, own HIP contexts), so that nothing in the communication path -- a hang inside a collective,
    # a crash inside the library -- can cost the line: the parent waits with a timeout, kills its child if need be, and
    # rank 0 prints the line either way.
    zslab = None
    if not args.no_zslab:
        dist.barrier("children")                           # rank 0 comes here late (CPU baseline): start the children together
        zslab = run_zslab_child(args, rank)
    if rank == 0:
        merge_zslab_into_line(out, zslab, n_gpus)
        print(json.dumps(out), flush=True)
    sys.stdout.flush()
    os._exit(0)      # skip collective teardown: nothing after the JSON line may hang the job


def run_zslab_child(args, rank):
    import signal
    import subprocess
    env = dict(os.environ)
    env["MASTER_PORT"] = str((int(env.get("MASTER_PORT", "29500")) - 1024 + 101) % 60000 + 1024)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)          # the children rendezvous among themselves (rank 0 hosts the store)
    # the host side of a slab step is a few hundred calls on tiny arrays: BLAS / OpenMP pools of 64-128 threads per rank only
    # cost wake-ups there (measured: a one-off 60-80 ms stall in the second step of a run, gone with one thread)
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        env.setdefault(var, "1")
    cmd = [sys.executable, os.path.abspath(__file__), "--zslab-child", "--gpus", str(args.gpus), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--zslab-planes", str(args.zslab_planes), "--zslab-yx", str(args.zslab_yx[0]), str(args.zslab_yx[1])]
    if args.share_device:
        cmd.append("--share-device")
    try:
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True, text=True)
    except OSError as exc:
        return {"error": f"could not start the Z-slab run: {exc}"[:300]}
    try:
        so, se = proc.communicate(timeout=args.zslab_timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)             # exactly the process group started above
        except OSError:
            pass
        try:
            proc.communicate(timeout=10)
        except Exception:  # noqa: BLE001
            pass
        return {"error": f"no answer within {args.zslab_timeout} s (child killed)"}
    if rank != 0:
        return None
    for line in reversed(so.strip().splitlines()):
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                break
    return {"error": f"child exit code {proc.returncode}: {se.strip()[-400:]}"}


def zslab_on_one_gpu(world, planes, yx, device, steps, warmup=2):
    """The SAME global volume as `world` Z-slab contexts on ONE GPU (one host thread per rank, exchanges through the library's
    loopback transport: nl_comm_loopback_id): the single-GPU time of the multi-GPU workload, so that a speed-up can be read
    beside the weak-scaling ratio against the 1024^3 line.  The slabs share the GPU, their kernels interleave; the step time
    is the wall time until every slab has finished its step."""
    import threading
    from nellie_amd import hipnative
    from nellie_amd import pipeline as pl
    from nellie_amd.sharded import RcclComm, ShardedFramePipeline, slab_range
    from nellie_amd.synthetic import ISO_01, make_volume
    p = pl.FilterParams(dim_res=ISO_01)
    min_area = pl.min_area_pixels_of(ISO_01)
    gshape = (planes * world, int(yx[0]), int(yx[1]))
    uid, uid_x = hipnative.comm_unique_id(loopback=True), hipnative.comm_unique_id(loopback=True)
    bar = threading.Barrier(world)
    times, labels, errs = [0.0] * world, [0] * world, []

    def worker(rank):
        try:
            pipe = ShardedFramePipeline(gshape, rank, world, lambda ctx: RcclComm(ctx, world, rank, uid, uid2=uid_x), p, device=device)
            o0, o1 = slab_range(gshape[0], world, rank)
            g_lo, g_hi = pipe.raw_ghost_needed()
            pipe.load_input(make_volume((o1 - o0 + g_lo + g_hi,) + gshape[1:], 3456, z_offset=o0 - g_lo, global_nz=gshape[0]))
            def step():
                pipe.filter(None, p)
                return pipe.label(pipe.frangi_threshold(), min_area)
            for _ in range(warmup):
                step()
            pipe.ctx.sync(); bar.wait()
            t0 = time.perf_counter()
            for _ in range(steps):
                labels[rank] = step()
            pipe.ctx.sync(); bar.wait()
            times[rank] = time.perf_counter() - t0
            pipe.close()
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)
            bar.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        return {"error": f"{type(errs[0]).__name__}: {errs[0]}"[:300]}
    ms = max(times) / steps * 1e3
    return {"ms_per_step": round(ms, 3), "value": round(float(np.prod(gshape)) / ms / 1e3, 1), "unit": "Mvoxel/s", "slabs": world,
            "volume": list(gshape), "labels": int(labels[0]), "steps": steps,
            "what": f"the same volume as {world} Z-slab contexts sharing ONE GPU (loopback transport, one host thread per slab)"}


def zslab_run(dist, rank, world, local_rank, args):
    """ONE volume over `world` GPUs (nellie_amd/sharded.py).  First the timed run at 128 owned planes of 2048 x 2048 per GPU (a
    fresh process: see (2) below), then a small volume against a single-GPU run of the same volume (bit-for-bit equality of both
    outputs on every rank), then -- rank 0 -- the timed workload as `world` slab contexts on one GPU."""
    from nellie_amd import hipnative
    from nellie_amd import pipeline as pl
    from nellie_amd.sharded import RcclComm, ShardedFramePipeline, slab_range
    from nellie_amd.synthetic import ISO_01, make_volume
    p = pl.FilterParams(dim_res=ISO_01)
    min_area = pl.min_area_pixels_of(ISO_01)
    def fresh_uid():                    # an RCCL unique id opens exactly one communicator
        return dist.from_rank0("uid", hipnative.comm_unique_id)

    res = {"world": world, "transport": "RCCL: ncclSend/ncclRecv (ghost planes, bit planes), ncclAllReduce (scalars, histograms), "
                                        "ncclAllGather (threshold samples, slab run tables); the per-step ghost-plane exchanges on a second communicator and stream"}

    # ---- (1) the timed run
    planes = int(args.zslab_planes)
    gshape = (planes * world, int(args.zslab_yx[0]), int(args.zslab_yx[1]))
    o0, o1 = slab_range(gshape[0], world, rank)
    uid2, uid2_x = fresh_uid(), fresh_uid()
    pipe = ShardedFramePipeline(gshape, rank, world, lambda ctx: RcclComm(ctx, world, rank, uid2, uid2=uid2_x), p, device=local_rank)
    # The resident input of a rank is its owned planes plus the raw ghost planes the first cascade step reads (8 per interior
    # side here): whoever loads a slab from the source image loads those planes with it, so no raw plane crosses xGMI.  The
    # ghost planes of every COMPUTED volume (one exchange per cascade step) and the bit planes of Label do travel over RCCL,
    # inside the timed region.  NELLIE_BENCH_RAW_EXCHANGE=1: owned planes only, the raw ghosts are exchanged in every step too.
    g_lo, g_hi = (0, 0) if os.environ.get("NELLIE_BENCH_RAW_EXCHANGE") == "1" else pipe.raw_ghost_needed()
    t_gen = time.perf_counter()
    own = make_volume((o1 - o0 + g_lo + g_hi,) + gshape[1:], 3456, z_offset=o0 - g_lo, global_nz=gshape[0])
    t_gen = time.perf_counter() - t_gen
    pipe.load_input(own)
    del own

    def step():
        pipe.filter(None, p)
        return pipe.label(pipe.frangi_threshold(), min_area)

    pipe.ctx.prof_enable(True)      # as in the N = 1 run: the timers' event pairs exist and have been used before the timed region
    # at least two untimed steps: the SECOND pass over a fresh slab context still carries a one-off 40-80 ms stall of the
    # queue (measured with tools/prof_slab.py; not in any kernel -- the single-GPU context does not show it), steady after
    n_warm = max(2, args.warmup)
    for _ in range(n_warm):
        step()
    if os.environ.get("NELLIE_BENCH_GC", "freeze") == "freeze":
        import gc
        gc.collect(); gc.freeze()      # the interpreter's cyclic collector out of the timed region (as timeit does): see DESIGN.md section 5
    pipe.ctx.prof_reset()
    pipe.ctx.sync()
    dist.barrier("zstep")
    t0 = time.perf_counter()
    each = []
    for _ in range(args.steps):
        t1 = time.perf_counter()
        n_labels = step()
        each.append(round((time.perf_counter() - t1) * 1e3, 2))
    pipe.ctx.sync()
    elapsed = time.perf_counter() - t0
    elapsed_own = elapsed
    dist.barrier("zstep")
    elapsed = dist.max("zelapsed", elapsed)
    pipe.ctx.prof_enable(False)
    groups = {}
    for name in GROUPS:
        ms, k = pipe.ctx.prof_get(name)
        if k:
            groups[name] = round(ms / args.steps, 3)
    crc = __import__("zlib").crc32(pipe.download_labels().tobytes())
    n_global = float(np.prod(gshape))
    # every rank's own figures (rank 0 assembles them: zslab_rank_summary)
    mine = {"rank": rank, "ms_per_step": round(elapsed_own / args.steps * 1e3, 3), "ms_of_each_step": each,
            "halo_ms": groups.get("halo"), "halo_wait_ms": groups.get("halo_wait", 0.0),
            "cascade_steps_per_step": len(p.resolved_sigmas()),
            "kernel_groups_ms": {k: v for k, v in groups.items() if k not in ("halo", "halo_wait")}}
    res.update(zslab_rank_summary(dist.allgather("zrows", mine)))
    tr = pipe.trace
    # The event pairs of two kernels that overlap (the cascade step running ahead beside the walk and the threshold kernels)
    # each count the overlap: the per-group table above does not add up on slabs.  Two more steps with nothing running ahead
    # give groups that do (their sum is the kernel time of a step; rocprofv3's kernel sum agrees, profiles/r03_kernel_stats_zslab*).
    pipe._gauss_ahead = False
    pipe.ctx.prof_reset(); pipe.ctx.prof_enable(True)
    for _ in range(2):
        step()
    pipe.ctx.sync(); pipe.ctx.prof_enable(False)
    groups_serial = {}
    for name in GROUPS:
        ms, k = pipe.ctx.prof_get(name)
        if k:
            groups_serial[name] = round(ms / 2, 3)
    res.update({
        "value": round(n_global * args.steps / elapsed / 1e6, 1), "unit": "Mvoxel/s", "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "workload": f"ONE synthetic {gshape[0]}x{gshape[1]}x{gshape[2]} float32 volume (seed 3456"
                    + (", BASELINE config 4" if gshape == (1024, 2048, 2048) else f", the first {gshape[0]} planes' worth of BASELINE config 4's generator")
                    + f") cut into {world} Z slabs of {planes} owned planes + {pipe.halo} ghost planes per interior side "
                      f"({'raw ghost planes of the input resident with it, ' if (g_lo or g_hi or world == 1) else ''}ghost planes of every computed volume and Label's bit planes exchanged over RCCL in the step); "
                      "5-scale Frangi + Label (no replication), full hot path per step, slabs resident in HBM",
        "voxels": int(n_global), "per_gpu_owned_shape": [planes, gshape[1], gshape[2]], "halo_planes": pipe.halo, "halo_scheme": pipe.halo_mode, "untimed_warmup_steps": n_warm, "raw_ghost_planes_resident_with_input": [int(g_lo), int(g_hi)],
        "halo_ms": groups.get("halo"), "groups_ms_per_step_rank0": groups, "ms_of_each_step_rank0": each,
        "groups_ms_per_step_rank0_nothing_ahead": groups_serial, "kernel_sum_ms_per_step_rank0_nothing_ahead": round(sum(groups_serial.values()), 3),
        "labels": int(n_labels),
        "survival_fraction": round(tr.n_positive / n_global, 5),
        "mask_fraction_per_scale": [round(sc.mask_count / n_global, 4) for sc in tr.scales],
        "one_pass_scales": int(sum(1 for sc in tr.scales if sc.one_pass)), "host_gen_s": round(t_gen, 1),
        "labels_crc32_rank0": int(crc),
        "device_chain": {"enabled": bool(pipe._chain_usable(p, True)), "frames_redone_synchronously": int(pipe.chain_fallbacks),
                         "last_flags": getattr(pipe, "last_chain_flags", None)},
    })
    pipe.close()
    # ---- (2) equality on a small volume, against a single-GPU run of the same volume.  AFTER the timed run: a process that has
    # created, used and closed other contexts before runs the same slab step 9 % (synchronous path) to 18 % (device chain) slower
    # (measured, tools/prof_slab.py NELLIE_PROF_PRELUDE=1: 29.9 -> 32.7 and 30.1 -> 35.4 ms), so the measurement goes first
    gshape = (48 * world, 192, 256)
    o0, o1 = slab_range(gshape[0], world, rank)
    vol = make_volume(gshape, 4242)
    uid, uid_x = fresh_uid(), fresh_uid()
    pipe = ShardedFramePipeline(gshape, rank, world, lambda ctx: RcclComm(ctx, world, rank, uid, uid2=uid_x), p, device=local_rank)
    ones = pipe.comm.allreduce(np.array([1], np.int64), "sum")
    res["rccl_ranks"] = int(ones[0])
    pipe.load_input(vol[o0:o1])
    pipe.filter(None, p)
    n_small = pipe.label(pipe.frangi_threshold(), min_area)
    fr, lab = pipe.download_frangi(), pipe.download_labels()
    pipe.close()
    single = pl.FramePipeline(gshape, device=local_rank)
    single.filter(vol, p)
    ok_fr = bool(np.array_equal(single.download_frangi()[o0:o1], fr))
    n_ref = single.label(single.frangi_threshold(), min_area)
    ok_lab = bool(np.array_equal(single.download_labels()[o0:o1], lab)) and n_ref == n_small
    single.close()
    flags = dist.min_ints("equal", [int(ok_fr), int(ok_lab)])
    res["equality_check"] = {"volume": list(gshape), "labels": int(n_small), "frangi_equal": bool(flags[0]), "labels_equal": bool(flags[1])}
    res["frangi_equal"], res["labels_equal"] = bool(flags[0]), bool(flags[1])

    # the same workload on one GPU (rank 0's), after the other ranks are done with theirs
    if rank == 0 and world > 1 and os.environ.get("NELLIE_BENCH_SAME_WORKLOAD", "1") == "1":
        try:
            one = zslab_on_one_gpu(world, planes, args.zslab_yx, local_rank, max(1, min(2, args.steps)))
        except Exception as exc:  # noqa: BLE001
            one = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        res["same_workload_single_gpu"] = one
        if "ms_per_step" in one:
            res["same_workload_single_gpu_ms"] = one["ms_per_step"]
            res["speedup_vs_same_workload_on_one_gpu"] = round(one["ms_per_step"] / res["ms_per_step"], 3)
    return res


def zslab_rank_summary(rows):
    """What the N > 1 line says about the ranks (round 6: the first SCALE record must explain itself): per rank the step time on its own
    clock, the time the ghost-plane exchanges took on their stream (`halo_ms`) and the time the main stream WAITED for them
    (`halo_wait_ms`: the exposed part, also per cascade step), the slowest rank and the skew.  rows: one dict per rank, as every rank's
    zslab_run builds it (tests/test_bench_multi.py feeds eight made-up ones)."""
    rows = sorted(rows, key=lambda r: r["rank"])
    ms = [float(r["ms_per_step"]) for r in rows]
    waits = [float(r.get("halo_wait_ms") or 0.0) for r in rows]
    steps = max(1, int(rows[0].get("cascade_steps_per_step") or 1))
    slow = max(range(len(rows)), key=lambda k: ms[k])
    return {
        "per_rank": [{"rank": r["rank"], "ms_per_step": round(float(r["ms_per_step"]), 3), "halo_ms": r.get("halo_ms"),
                      "halo_wait_ms": round(float(r.get("halo_wait_ms") or 0.0), 3),
                      "kernel_sum_ms": round(sum(float(v) for v in (r.get("kernel_groups_ms") or {}).values()), 3)} for r in rows],
        "slowest_rank": rows[slow]["rank"], "rank_skew_ms": round(max(ms) - min(ms), 3),
        "exchange_exposed_ms_per_step_max_over_ranks": round(max(waits), 3),
        "exchange_exposed_ms_per_cascade_step_max_over_ranks": round(max(waits) / steps, 4),
    }


def n1_reference():
    """The N = 1 figures a multi-GPU line is read against, from the committed profiles of the same commands: the 1024^3 line (what
    `replicas` re-measures in the same run) and ONE rank's share of the slab run at world 1 (a 128 x 2048 x 2048 slab alone on a GPU)."""
    import glob
    out = {}
    for key, pat in (("one_rank_alone_128x2048x2048", "r*_zslab_world1_128x2048x2048.json"), ("frame_1024cube", "r*_bench_n1_1024cube.json")):
        files = sorted(glob.glob(os.path.join(REPO, "profiles", pat)))
        if not files:
            continue
        try:
            rec = json.loads(open(files[-1]).read().strip().splitlines()[-1])
            out[key] = {"ms_per_step": rec.get("ms_per_step"), "mvoxel_s": rec.get("value"), "source": os.path.relpath(files[-1], REPO)}
        except (ValueError, IndexError, OSError):
            pass
    return out or None


def merge_zslab_into_line(out, zslab, n_gpus):
    """rank 0: the Z-slab child's result becomes the line's `value` (or its absence the line's failure); the frame-replica figure moves
    to `replicas`.  Pure function of its arguments (tests/test_bench_multi.py::test_assembled_eight_gpu_line)."""
    out["replicas"] = {"value": out["value"], "unit": "Mvoxel/s", "ms_per_step": out["ms_per_step"],
                       "workload": f"3-D+T stack of {n_gpus} frames, one frame per GPU, no data-path collective"}
    if zslab is None:
        return out
    out["zslab"] = zslab
    if "error" not in zslab and zslab.get("value"):
        out["value"] = zslab["value"]
        out["ms_per_step"] = zslab["ms_per_step"]
        out["config"]["workload"] = zslab["workload"]
        out["config"]["voxels"] = zslab["voxels"]
        out["config"]["per_gpu_shape"] = zslab["per_gpu_owned_shape"]
        out["config"]["parallelism"] = f"zslab{n_gpus}"
        for k in ("survival_fraction", "labels", "mask_fraction_per_scale", "one_pass_scales"):
            out["config"][k] = zslab.get(k)
        ref = n1_reference()
        out["zslab"]["n1_reference"] = ref
        one = (ref or {}).get("one_rank_alone_128x2048x2048", {}).get("ms_per_step")
        if one and zslab.get("per_gpu_owned_shape") == [SLAB_PLANES, SLAB_YX[0], SLAB_YX[1]]:
            # weak scaling read directly: a rank alone takes `one` ms for its slab; N ranks take ms_per_step for N slabs
            out["zslab"]["step_over_one_rank_alone"] = round(zslab["ms_per_step"] / one, 3)
    else:
        # the decomposition this line is about did not run: no number is reported for it (the frame-replica figure
        # stays under `replicas`, it is a different workload)
        out["value"] = None
        out["ms_per_step"] = None
        out["zslab_failed"] = True
        out["config"]["parallelism"] = f"zslab{n_gpus} (failed: see zslab.error; `replicas` = {n_gpus} independent frames, no collective)"
    return out


def zslab_child_main(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_device else int(os.environ.get("LOCAL_RANK", "0"))
    dist = Control(rank, world, tag="zslab_" + os.environ.get("MASTER_PORT", "29500"), timeout_s=180.0)
    fake = os.environ.get("NELLIE_ZSLAB_FAKE", "")      # tests of the isolation: "crash" (the last rank aborts), "hang"
    if fake == "crash" and rank == world - 1:
        os.abort()
    if fake == "hang":
        time.sleep(1e6)
    try:
        res = zslab_run(dist, rank, world, local_rank, args)
    except Exception as exc:  # noqa: BLE001
        res = {"error": f"{type(exc).__name__}: {exc}"[:400]}
    if rank == 0:
        print(json.dumps(res), flush=True)
    sys.stdout.flush()
    if os.environ.get("NELLIE_BENCH_CLEAN_EXIT") == "1":      # under a profiler: let its exit handlers write their files
        return
    os._exit(0)


if __name__ == "__main__":
    main()
