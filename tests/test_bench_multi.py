"""bench.py's N > 1 control flow on a ONE-GPU box: launcher -> ranks -> the Z-slab child per rank -> ONE JSON line
that either carries the slab run over RCCL or says why it has none.  Two ranks share device 0 (`--share-device`, testing only): RCCL
may refuse two ranks on one device -- then `zslab.error` is populated and `value` is null, which is exactly the failure path the
driver's 8-GPU run must be able to take without losing its line.  NELLIE_ZSLAB_FAKE=crash|hang: a rank of the child run aborts / never
answers; the parents kill their children after --zslab-timeout and rank 0 still prints the line."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--share-device", "--steps", "1", "--warmup", "1", "--shape", "64", "64", "64",
         "--no-cpu-baseline", "--no-io", "--zslab-planes", "24", "--zslab-yx", "96", "128"]


def _run(extra_env=None, extra_args=(), timeout=420):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["NELLIE_BENCH_SAME_WORKLOAD"] = "0"
    env.update(extra_env or {})
    proc = subprocess.run(BENCH + list(extra_args), env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line expected, got {len(lines)}\nstdout: {proc.stdout[-600:]}\nstderr: {proc.stderr[-1200:]}"
    return json.loads(lines[0]), proc


@pytest.mark.gpu
def test_two_ranks_sharing_one_device_print_one_line():
    out, proc = _run()
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 1
    assert out["replicas"]["value"] > 0, "the frame-replica figure of the two ranks is measured either way"
    z = out.get("zslab")
    assert z is not None, out
    if "error" in z:                     # RCCL refused two ranks on one device (or the run failed): no number for the decomposition
        assert z["error"] and out["value"] is None and out["ms_per_step"] is None and out.get("zslab_failed") is True
    else:
        assert z["rccl_ranks"] == 2 and z["frangi_equal"] and z["labels_equal"]
        assert out["value"] == z["value"] and out["config"]["parallelism"] == "zslab2"


@pytest.mark.gpu
@pytest.mark.parametrize("fake", ["crash", "hang"])
def test_a_child_that_dies_or_hangs_does_not_cost_the_line(fake):
    out, proc = _run({"NELLIE_ZSLAB_FAKE": fake}, ["--zslab-timeout", "25"])
    assert out["value"] is None and out.get("zslab_failed") is True
    assert "error" in out["zslab"] and out["zslab"]["error"]
    assert out["replicas"]["value"] > 0
    assert "failed" in out["config"]["parallelism"]


def test_control_plane_needs_no_torch():
    """north_star: no PyTorch on the host path -- the bench harness included (round 5: the file rendezvous of the product)."""
    src = open(os.path.join(REPO, "bench.py")).read()
    assert "import torch" not in src and "torch.distributed.run" not in src.replace("as torch.distributed.run would", "").replace("with torch.distributed.run,", "")


def test_control_plane_primitives(tmp_path, monkeypatch):
    """Control over threads standing in for ranks: barrier, maximum, minimum of flags, a value from rank 0 -- twice in a row."""
    import threading
    sys.path.insert(0, REPO)
    import bench
    monkeypatch.setenv("NELLIE_RENDEZVOUS_DIR", str(tmp_path))
    world, res, errs = 3, {}, []

    def rank(r):
        try:
            c = bench.Control(r, world, tag="t", timeout_s=30)
            got = []
            for k in range(2):
                c.barrier("step")
                got.append((c.max("m", 10.0 * r + k), c.min_ints("f", [r, 1, 5 - r]), c.from_rank0("uid", lambda: bytes([7, k, 9]))))
            res[r] = got
        except BaseException as exc:  # noqa: BLE001
            errs.append(repr(exc))

    ts = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    assert not errs, errs
    for r in range(world):
        assert res[r] == [(20.0, [0, 1, 3], bytes([7, 0, 9])), (21.0, [0, 1, 3], bytes([7, 1, 9]))]


def test_assembled_eight_gpu_line():
    """Round 6 (VERDICT r05 "next 8"): the first 8-GPU record must explain itself.  Eight made-up rank rows go through the functions rank 0
    runs -- zslab_rank_summary, merge_zslab_into_line -- and the line says: how many ranks RCCL joined, every rank's own step time, the
    exposed part of the ghost-plane exchanges (per step and per cascade step), the slowest rank and the skew, the equality flags, and the
    N = 1 figures it is to be read against (from the committed profiles).  The failure path: no number, `zslab_failed`."""
    import copy
    sys.path.insert(0, REPO)
    import bench
    rows = [{"rank": r, "ms_per_step": 23.0 + 0.1 * r + (1.4 if r == 5 else 0.0), "ms_of_each_step": [23.0, 23.2], "halo_ms": 4.1,
             "halo_wait_ms": 0.05 * r, "cascade_steps_per_step": 5,
             "kernel_groups_ms": {"gauss_zyx<4,4>": 8.5, "vesselness": 7.4, "vesselness_resolve": 2.0, "label": 2.4}} for r in reversed(range(8))]
    summ = bench.zslab_rank_summary(rows)
    assert [r["rank"] for r in summ["per_rank"]] == list(range(8))
    assert summ["slowest_rank"] == 5 and summ["rank_skew_ms"] == pytest.approx(1.9, abs=1e-6)
    assert summ["exchange_exposed_ms_per_step_max_over_ranks"] == pytest.approx(0.35) and summ["exchange_exposed_ms_per_cascade_step_max_over_ranks"] == pytest.approx(0.07)
    assert summ["per_rank"][0]["kernel_sum_ms"] == pytest.approx(20.3)
    base = {"metric": "m", "value": 27900.0 * 8, "unit": "Mvoxel/s", "n_gpus": 8, "ms_per_step": 38.5,
            "config": {"workload": "eight frames", "voxels": 8 * 1024 ** 3, "per_gpu_shape": [1024, 1024, 1024]}}
    zslab = dict(summ, world=8, rccl_ranks=8, frangi_equal=True, labels_equal=True, value=180000.0, ms_per_step=23.9,
                 workload="ONE synthetic 1024x2048x2048 float32 volume ... cut into 8 Z slabs", voxels=1024 * 2048 * 2048,
                 per_gpu_owned_shape=[128, 2048, 2048], survival_fraction=0.02, labels=4000, mask_fraction_per_scale=[0.3] * 5, one_pass_scales=5)
    line = bench.merge_zslab_into_line(copy.deepcopy(base), zslab, 8)
    assert line["value"] == 180000.0 and line["ms_per_step"] == 23.9 and line["config"]["parallelism"] == "zslab8"
    assert line["replicas"]["value"] == base["value"] and line["replicas"]["ms_per_step"] == 38.5
    z = line["zslab"]
    assert z["rccl_ranks"] == 8 and z["frangi_equal"] and z["labels_equal"] and len(z["per_rank"]) == 8
    ref = z["n1_reference"]                       # the committed N = 1 profiles (both exist in this repository)
    assert ref["one_rank_alone_128x2048x2048"]["ms_per_step"] > 0 and ref["frame_1024cube"]["mvoxel_s"] > 0
    assert z["step_over_one_rank_alone"] == pytest.approx(23.9 / ref["one_rank_alone_128x2048x2048"]["ms_per_step"], abs=1e-3)
    json.dumps(line)                              # one JSON line, nothing that does not serialise
    failed = bench.merge_zslab_into_line(copy.deepcopy(base), {"error": "ncclCommInitRank: unhandled system error"}, 8)
    assert failed["value"] is None and failed["ms_per_step"] is None and failed["zslab_failed"] is True and "failed" in failed["config"]["parallelism"]
    assert failed["replicas"]["value"] == base["value"]


def test_roofline_object_has_no_rate_above_the_peak():
    """Round 6 (VERDICT r05 "next 4"): SURVEY 8(d)'s pass model is a constant for reference, not a denominator -- no group carries a rate
    against it; the fused-design bytes and the counter bytes give `design_frac` <= `counter_frac` <= 1, and `traffic_source` says where the
    counter bytes come from.  Made-up group timers of a 1024^3 step and a made-up PMC table go through bench.roofline_of."""
    sys.path.insert(0, REPO)
    import bench
    n = 1024 ** 3
    groups = {"gauss_zyx<4,4>": {"ms_total": 3 * 2.98 * 10, "launches": 30, "ms_avg": 2.98}, "gauss_zyx<3,3>": {"ms_total": 25.1, "launches": 10, "ms_avg": 2.51},
              "vesselness": {"ms_total": 131.0, "launches": 50, "ms_avg": 2.62}, "vesselness_resolve": {"ms_total": 40.0, "launches": 50, "ms_avg": 0.8},
              "sample": {"ms_total": 40.0, "launches": 130, "ms_avg": 0.31}, "mask_volume": {"ms_total": 15.0, "launches": 10, "ms_avg": 1.5},
              "label": {"ms_total": 32.5, "launches": 10, "ms_avg": 3.25}}
    kb = lambda b_per_voxel: b_per_voxel * n / 1024.0                     # noqa: E731
    rows = [{"kernel": "void hessian_v_kernel<2, 8, 2, 1>", "launches": 5, "fetch_size_kb_per_launch": kb(3.1), "write_size_kb_per_launch": kb(1.12)},
            {"kernel": "void gauss_zyx_kernel<4, 4>", "launches": 3, "fetch_size_kb_per_launch": kb(2.38), "write_size_kb_per_launch": kb(5.9)},
            {"kernel": "void vesselness_queue_kernel<true>", "launches": 5, "fetch_size_kb_per_launch": kb(0.62), "write_size_kb_per_launch": kb(0.25)},
            {"kernel": "rl_paint_kernel", "launches": 1, "fetch_size_kb_per_launch": kb(0.08), "write_size_kb_per_launch": kb(4.1)},
            {"kernel": "apply_bits_pos_kernel", "launches": 1, "fetch_size_kb_per_launch": kb(0.19), "write_size_kb_per_launch": kb(4.0)},
            {"kernel": "rl_threshold_pack_kernel", "launches": 1, "fetch_size_kb_per_launch": kb(0.19), "write_size_kb_per_launch": kb(0.12)},
            {"kernel": "void rl_emit_kernel<true>", "launches": 3, "fetch_size_kb_per_launch": kb(0.065), "write_size_kb_per_launch": kb(0.03)},
            {"kernel": "majority_bits_kernel", "launches": 1, "fetch_size_kb_per_launch": kb(0.07), "write_size_kb_per_launch": kb(0.12)},
            {"kernel": "pack_masked_kernel", "launches": 1, "fetch_size_kb_per_launch": kb(0.55), "write_size_kb_per_launch": kb(0.13)},
            {"kernel": "void bits_morph6_kernel<0>", "launches": 2, "fetch_size_kb_per_launch": kb(0.07), "write_size_kb_per_launch": kb(0.12)}]
    pmc = bench.PmcTable(rows, "this run", (1024, 1024, 1024))
    facts = {"queue_fraction_per_scale": [0.08, 0.04, 0.02, 0.02, 0.02], "mask_fraction_min": 0.14, "survival": 0.02, "zeroing_group": "gauss_zyx<4,4>"}
    r = bench.roofline_of(groups, (1024, 1024, 1024), 10, 38.5, pmc=pmc, facts=facts)
    assert r["kernel"] == "vesselness" and r["bound"] == "hbm" and r["traffic_source"] == "this run"
    assert r["frac"] == pytest.approx(14.0 * n / 2.62e-3 / 1e9 / 8000.0, rel=1e-3)          # the contract's figure: SURVEY 8(d)'s 14 B
    assert r["traffic"] == pytest.approx((2 * 3.1 + 1.12) * n, rel=1e-6)
    text = json.dumps(r)
    assert "alg_gbs" not in text and "frac_against_pass_model" not in text
    for name, g in r["groups"].items():
        for key in ("design_gbs", "counter_gbs"):
            assert g[key] is None or g[key] <= 8000.0, (name, key, g[key])
        if g["design_frac"] is not None and g["counter_frac"] is not None:
            assert g["design_frac"] <= g["counter_frac"] + 1e-9, name
    assert r["groups"]["gauss_zyx<4,4>"]["design_bytes_per_voxel"] == pytest.approx(8.0 + 4.0 / 3.0, abs=1e-3)      # one of its three launches also zeroes the scale maximum
    assert r["groups"]["vesselness"]["design_bytes_per_voxel"] == pytest.approx(4.25 + 28 * 0.036, abs=1e-3)
    assert r["groups"]["label"]["counter_bytes_per_voxel"] == pytest.approx((2 * 0.08 + 4.1) + (2 * 0.19 + 0.12) + 3 * (2 * 0.065 + 0.03) + (2 * 0.07 + 0.12), abs=2e-3)
    p = r["pipeline"]
    assert p["design_frac"] is not None and p["design_frac"] <= p["frac_by_counters"] <= 1.0
    fallback = bench.roofline_of(groups, (1024, 1024, 1024), 10, 38.5, pmc=None, facts=facts)
    assert fallback["traffic_source"] is None or fallback["traffic_source"].startswith("profiles/")
