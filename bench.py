#!/usr/bin/env python3
"""
bench.py -- Mvoxel/s of Nellie's segmentation hot path (5-scale Frangi Filter + Label) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--shape Z Y X] [--no-cpu-baseline]

One "step" = one pass of the whole hot path (float32 conversion, 5-scale Gaussian cascade,
Hessian, eigenvalues, Frangi, scale-max, masks, percentile mask + opening, Label thresholds,
hole filling, two 26-connected labellings, area filter, majority filter, raster renumbering)
over one synthetic float32 frame that is ALREADY RESIDENT IN HBM when the timed region starts;
outputs stay in HBM (the PCIe-inclusive rate is a separate, untimed-by-default figure, see
DESIGN.md).  N = 1: BASELINE.json configs[2], 1024 x 1024 x 1024 float32 (headline).
N > 1: a 3-D+T stack of N such frames, one frame per rank / GPU (frames are the path's independent
units, nellie/segmentation/filtering.py:1007, labelling.py:701): no data-path collective, weak scaling.
The Z-slab decomposition of ONE volume (ghost planes over RCCL) is the sharded pipeline of
nellie_amd/sharded.py; see DESIGN.md "Multi-GPU" for what is and is not done yet.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` and `cpu_baseline`.
The oracle is used here only for the `cpu_baseline` leg and the small accuracy check.
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

RESULT_HOLDER = {}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s measured copy
B_ALG_TOTAL = 301.0            # Filter 257 + Label 44 bytes/voxel
# algorithmic bytes per voxel of one launch of each kernel group (DESIGN.md "Kernels")
B_ALG_KERNEL = {
    "load": 8.0,               # 4 r + 4 w
    "gauss_z": 8.0,            # one axis pass: 4 r + 4 w
    "gauss_yx": 16.0,          # two axis passes of the model (4 r + 4 w each), one fused launch here
    "gauss_y": 8.0,
    "gauss_x": 8.0,
    "hessian_stats": 4.0,      # 4 r
    "vesselness": 22.0,        # Hessian/eigen/Frangi pass 16 + mask pass 6, one launch here
    "finish": 9.0,             # 4 r + 1 r + 4 w
    "mask_volume": 8.0,        # 4 r + 4 w (fused threshold + opening + multiply)
    "label": 44.0,             # SURVEY.md 8(d) Label row
}


PMC_KERNEL_OF_GROUP = {"vesselness": "hessian_g_kernel<2", "hessian_stats": "hessian_g_kernel<0",
                       "gauss_yx": "gauss_yx_kernel<4", "gauss_z": "gauss_march_kernel<0, 4"}


def pmc_traffic(group, shape):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r*_pmc_hbm_bytes_1024cube.json: separate FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE x2 on gfx950,
    calibrated on the streaming convert kernel, x1024 for KB).  None when no matching profile is committed."""
    import glob
    if tuple(shape) != (1024, 1024, 1024) or group not in PMC_KERNEL_OF_GROUP:
        return None
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_hbm_bytes_1024cube.json")))
    if not files:
        return None
    for rec in json.load(open(files[-1])):
        if PMC_KERNEL_OF_GROUP[group] in rec["kernel"]:
            return round((2.0 * rec["fetch_size_kb_per_launch"] + rec["write_size_kb_per_launch"]) * 1024.0)
    return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--shape", type=int, nargs=3, default=None, help="per-GPU slab Z Y X (default 1024^3)")
    ap.add_argument("--seed", type=int, default=2345)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-shape", type=int, nargs=3, default=[96, 384, 384])
    ap.add_argument("--with-io", action="store_true", help="also report the PCIe-inclusive rate (untimed otherwise)")
    ap.add_argument("--no-zslab-check", action="store_true", help="N > 1: skip the RCCL Z-slab equality check")
    ap.add_argument("--zslab-timeout", type=float, default=150.0)
    ap.add_argument("--zslab-child", action="store_true", help="internal: run only the Z-slab check (spawned by the bench)")
    ap.add_argument("--zslab-force", action="store_true", help="testing: run the check even with --share-device")
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
    fr = orc.filter_frame(vol, ISO_01)
    t1 = time.perf_counter()
    lab = orc.label_frame(fr, ISO_01)
    t2 = time.perf_counter()
    n = float(np.prod(shape))
    return {
        "value": round(n / (t2 - t0) / 1e6, 4), "unit": "Mvoxel/s", "cores": 1, "kind": "port",
        "sample": f"oracle Filter+Label on a synthetic {shape[0]}x{shape[1]}x{shape[2]} float32 volume "
                  f"(same generator, seed {seed}); Filter {n / (t1 - t0) / 1e6:.3f} Mvoxel/s, "
                  f"Label {n / (t2 - t1) / 1e6:.2f} Mvoxel/s, {float(np.mean(fr > 0)) * 100:.2f}% voxels survive, "
                  f"{int(lab.max())} labels; numpy oracle, 1 thread of {os.cpu_count()} host cores",
    }, (vol, fr, lab)


def accuracy_check(pl, vol, ref_fr, ref_lab):
    """Every timing run also runs the parity check on the CPU-baseline volume."""
    from nellie_amd.synthetic import ISO_01
    pipe = pl.FramePipeline(vol.shape)
    pipe.filter(vol, pl.FilterParams(dim_res=ISO_01))
    fr = pipe.download_frangi()
    scale = float(ref_fr.max()) if ref_fr.size else 1.0
    err = np.abs(fr.astype(np.float64) - ref_fr)
    ok = bool(np.all(err <= 1e-4 * np.abs(ref_fr) + 1e-6 * scale))
    pipe.upload_frangi(ref_fr)
    pipe.label(pipe.frangi_threshold(), pl.min_area_pixels_of(ISO_01))
    lab_ok = bool(np.array_equal(pipe.download_labels(), ref_lab))
    pipe.close()
    return {"frangi_within_tol": ok, "frangi_max_norm_err": float(err.max() / scale) if scale else 0.0,
            "labels_bit_exact_given_same_frangi": lab_ok}


def zslab_check(dist, rank, world, local_rank, gshape=None):
    """
    Z-slab decomposition of ONE volume across the ranks (nellie_amd/sharded.py): ghost planes, all-reduces and the
    mask bit planes travel over RCCL/xGMI.  Every rank checks its own slab against a single-GPU run of the same
    volume, bit for bit.  Small volume: this is a correctness + plumbing check on real hardware, not the metric.
    """
    import torch
    from nellie_amd import hipnative
    from nellie_amd import pipeline as pl
    from nellie_amd.sharded import RcclComm, ShardedFramePipeline, slab_range
    from nellie_amd.synthetic import ISO_01, make_volume
    gshape = gshape or (48 * world, 192, 256)
    p = pl.FilterParams(dim_res=ISO_01)
    min_area = pl.min_area_pixels_of(ISO_01)
    box = [hipnative.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)

    def host_gather(a):
        out = [None] * world
        dist.all_gather_object(out, np.asarray(a))
        return np.concatenate(out)

    o0, o1 = slab_range(gshape[0], world, rank)
    vol = make_volume(gshape, 4242)
    pipe = ShardedFramePipeline(gshape, rank, world, lambda ctx: RcclComm(ctx, world, rank, box[0], host_gather), p,
                                device=local_rank)
    pipe.load_input(vol[o0:o1])
    t0 = time.perf_counter()
    pipe.filter(None, p)
    thr = pipe.frangi_threshold()
    n = pipe.label(thr, min_area)
    pipe.ctx.sync()
    ms = (time.perf_counter() - t0) * 1e3
    fr, lab = pipe.download_frangi(), pipe.download_labels()
    halo_ms, halo_n = pipe.ctx.prof_get("halo")
    pipe.close()
    single = pl.FramePipeline(gshape, device=local_rank)
    single.filter(vol, p)
    ok_fr = bool(np.array_equal(single.download_frangi()[o0:o1], fr))
    single.label(single.frangi_threshold(), min_area)
    ok_lab = bool(np.array_equal(single.download_labels()[o0:o1], lab))
    single.close()
    flags = torch.tensor([int(ok_fr), int(ok_lab)], dtype=torch.int64)
    dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    return {"volume": list(gshape), "world": world, "halo_planes": pipe.halo, "labels": int(n),
            "frangi_equal_to_single_gpu": bool(flags[0]), "labels_equal_to_single_gpu": bool(flags[1]),
            "first_pass_ms": round(ms, 1), "transport": "RCCL ncclSend/ncclRecv + ncclAllReduce + ncclBroadcast"}


def main():
    args = parse_args()
    if args.zslab_child:
        zslab_child_main(args)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        # started by hand as `python bench.py --gpus N`: become the launcher the contract describes (one rank per GPU)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    n_gpus = args.gpus
    if world != n_gpus and world > 1:
        n_gpus = world
    dist = None
    if world > 1:
        import torch.distributed as dist   # rendezvous + barrier + max-over-ranks only (control plane)
        dist.init_process_group(backend="gloo", init_method="env://")

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

    # rank r owns frame r of the (N, Z, Y, X) stack
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
            dist.barrier()

    for _ in range(args.warmup):
        step()
    pipe.ctx.prof_reset()
    pipe.ctx.prof_enable(True)
    barrier()
    t0 = time.perf_counter()
    n_labels = 0
    for _ in range(args.steps):
        n_labels = step()
    barrier()
    elapsed = time.perf_counter() - t0
    pipe.ctx.prof_enable(False)
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel-group HIP-event times over the timed region (this rank)
    groups = {}
    # "vesselness" = the marching Hessian pass of a scale (statistics + masks + queue; one launch per scale),
    # "vesselness_resolve" = the dense eigen/Frangi kernel over its queue; "hessian_stats" only appears when a
    # scale falls back to the two-pass scheme
    for name in ("load", "gauss_z", "gauss_yx", "gauss_y", "gauss_x", "sample", "hessian_stats", "vesselness", "vesselness_resolve", "finish", "mask_volume", "label"):
        ms, k = pipe.ctx.prof_get(name)
        if k:
            groups[name] = {"ms_total": ms, "launches": k, "ms_avg": ms / k}
    n_local = float(np.prod(shape))
    n_global = n_local * n_gpus
    kernel_ms_per_step = sum(g["ms_total"] for g in groups.values()) / max(1, args.steps)
    ms_per_step = elapsed / args.steps * 1e3
    dom = max((g for g in groups if g in B_ALG_KERNEL), key=lambda g: groups[g]["ms_total"])
    dom_bytes = B_ALG_KERNEL[dom] * n_local
    dom_gbs = dom_bytes / (groups[dom]["ms_avg"] * 1e-3) / 1e9
    roofline = {
        "bound": "hbm", "kernel": dom, "achieved": round(dom_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(dom_gbs / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(dom, shape),
        "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": round(groups[dom]["ms_avg"], 4),
        "pipeline": {
            "algorithmic_bytes_per_voxel": B_ALG_TOTAL,
            "kernel_ms_per_step": round(kernel_ms_per_step, 3),
            # over the wall time of a step (host round trips included): the resolve kernel overlaps the Gaussian of
            # the next scale on a side stream, so the per-group times above add up to more than the step takes
            "achieved": round(B_ALG_TOTAL * n_local / (ms_per_step * 1e-3) / 1e9, 1),
            "frac": round(B_ALG_TOTAL * n_local / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        },
        "groups_ms_per_step": {k: round(v["ms_total"] / max(1, args.steps), 3) for k, v in groups.items()},
    }

    io = None
    if args.with_io and world == 1:
        pipe.ctx.sync()
        t0 = time.perf_counter()
        pipe.load_input(vol)
        step()
        fr = pipe.download_frangi()
        lab = pipe.download_labels()
        io = {"pcie_inclusive_mvoxel_s": round(n_local / (time.perf_counter() - t0) / 1e6, 1)}
        del fr, lab

    tr = pipe.trace
    fast_div = int(pipe.ctx.info("fast_div"))
    pipe.close()

    out = None
    if rank == 0:
        cpu = None
        acc = None
        if not args.no_cpu_baseline:
            cpu, (cvol, cfr, clab) = cpu_baseline(tuple(args.cpu_shape), 1234)
            acc = accuracy_check(pl, cvol, cfr, clab)
        value = n_global * args.steps / elapsed / 1e6
        out = {
            "metric": "Mvoxel/s multiscale Frangi (5 sigma) + Label, float32", "value": round(value, 1),
            "unit": "Mvoxel/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"synthetic {shape[0]}x{shape[1]}x{shape[2]} float32 volume "
                            f"(N(100,5) noise + Gaussian tube segments, seed {args.seed}), 0.1 um isotropic, "
                            f"{len(p.resolved_sigmas())}-scale Frangi + Label, full hot path per step"
                            + ("" if n_gpus == 1 else f"; 3-D+T stack of {n_gpus} such frames, one frame per GPU"),
                "voxels": int(n_global), "per_gpu_shape": list(shape),
                "survival_fraction": round(tr.n_positive / n_local, 5), "labels": int(n_labels),
                "mask_fraction_per_scale": [round(sc.mask_count / n_local, 4) for sc in tr.scales],
                "one_pass_scales": int(sum(1 for sc in tr.scales if sc.one_pass)),
                "host_gen_s": round(t_gen, 1), "h2d_s": round(t_up, 2), "fast_div_proven": fast_div,
            },
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if acc is not None:
            out["accuracy"] = acc
        if io is not None:
            out["io"] = io

    if dist is None:
        print(json.dumps(out), flush=True)
        return

    # N > 1: optionally prove the Z-slab decomposition on the real GPUs (RCCL).  It runs in a CHILD process per rank
    # (own rendezvous on another port, own HIP contexts), so that nothing in the communication path -- a hang inside a
    # collective, a crash inside the library -- can cost the measured line: the parent waits with a timeout, kills its
    # child if need be, and rank 0 prints the line either way.
    zslab = None
    if not args.no_zslab_check and (not args.share_device or args.zslab_force):
        dist.barrier()                                     # rank 0 comes here late (CPU baseline): start the children together
        zslab = run_zslab_child(args, rank)
    if rank == 0:
        if zslab is not None:
            out["zslab"] = zslab
        print(json.dumps(out), flush=True)
    sys.stdout.flush()
    os._exit(0)      # skip collective teardown: nothing after the JSON line may hang the job


def run_zslab_child(args, rank):
    import signal
    import subprocess
    env = dict(os.environ)
    env["MASTER_PORT"] = str((int(env.get("MASTER_PORT", "29500")) - 1024 + 101) % 60000 + 1024)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)          # the children rendezvous among themselves (rank 0 hosts the store)
    cmd = [sys.executable, os.path.abspath(__file__), "--zslab-child", "--gpus", str(args.gpus)]
    if args.share_device:
        cmd.append("--share-device")
    try:
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True, text=True)
    except OSError as exc:
        return {"error": f"could not start the check: {exc}"[:300]}
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
    return {"error": f"child exit code {proc.returncode}: {se.strip()[-300:]}"}


def zslab_child_main(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_device else int(os.environ.get("LOCAL_RANK", "0"))
    import datetime
    import torch.distributed as dist
    dist.init_process_group(backend="gloo", init_method="env://", timeout=datetime.timedelta(seconds=120))
    fake = os.environ.get("NELLIE_ZSLAB_FAKE", "")      # tests of the isolation: "crash" (rank 1 aborts), "hang"
    if fake == "crash" and rank == world - 1:
        os.abort()
    if fake == "hang":
        time.sleep(1e6)
    try:
        res = zslab_check(dist, rank, world, local_rank)
    except Exception as exc:  # noqa: BLE001
        res = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    if rank == 0:
        print(json.dumps(res), flush=True)
    sys.stdout.flush()
    os._exit(0)


if __name__ == "__main__":
    main()
