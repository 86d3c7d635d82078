"""
CPU tests of the N > 1 path: slab geometry, the Z-slab Filter AND Label host logic under gloo with world_size 2
(oracle-backed context, tests/fake_ctx.py), and the Label protocol alone on up to 8 slabs (threads).  The sharded
results must equal the single-volume oracle bit for bit.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO
from oracle import nellie_oracle as orc



def _need_torch():
    """Skips without torch -- WITHOUT importing it here: the workers run in their own processes, and a torch imported into the pytest
    process brings its bundled librccl / HIP runtime along, which the library's own dlopen("librccl.so.1") then gets handed
    (ncclCommInitRank: "unhandled cuda error", 150 tests later -- round 5)."""
    import importlib.util
    if importlib.util.find_spec("torch") is None:
        pytest.skip("torch (torch.distributed.run + gloo for the worker processes) is not installed")


def test_slab_geometry():
    from nellie_amd.pipeline import FilterParams
    from nellie_amd.sharded import halo_depth, slab_geometry, slab_range
    from nellie_amd.synthetic import ANISO_03, ISO_01
    assert halo_depth(FilterParams(dim_res=ISO_01)) == 4 + (4 + 3 + 4 + 4 + 5)
    assert halo_depth(FilterParams(dim_res=ANISO_03)) == 4 + (1 + 1 + 1 + 1 + 2)
    for gnz, world in ((1024, 8), (100, 3), (64, 2)):
        ranges = [slab_range(gnz, world, r) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == gnz
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    (nzl, ny, nx), gz0, lo, hi = slab_geometry((1024, 8, 8), 8, 3, 24)
    assert (nzl, gz0, lo, hi) == (128 + 48, 3 * 128 - 24, 24, 24 + 128)
    (nzl, ny, nx), gz0, lo, hi = slab_geometry((1024, 8, 8), 8, 0, 24)
    assert (nzl, gz0, lo, hi) == (128 + 24, 0, 0, 128)
    with pytest.raises(ValueError):
        slab_geometry((64, 8, 8), 4, 1, 24)


@pytest.mark.parametrize("aniso,halo_mode", [(0, "steps"), (1, "steps"), (0, "fat"), (0, "steps+raw")])
def test_zslab_filter_and_label_world2_gloo(aniso, halo_mode, tmp_path):
    _need_torch()
    raw = "1" if halo_mode.endswith("+raw") else "0"       # raw ghost planes handed over with the frame instead of exchanged
    halo_mode = halo_mode.split("+")[0]
    from nellie_amd.synthetic import ANISO_03, ISO_01, make_volume
    port = 29600 + aniso + 2 * (halo_mode == "fat") + 4 * int(raw) + (os.getpid() % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(REPO, "tests", "dist_worker.py"), str(tmp_path), str(aniso)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, NELLIE_HALO=halo_mode, NELLIE_TEST_RAW_GHOSTS=raw))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    dr = ANISO_03 if aniso else ISO_01
    gshape = (40, 36, 44) if aniso else (64, 30, 34)
    vol = make_volume(gshape, 77)
    trace = []
    ref = orc.filter_frame(vol, dr, trace=trace)
    ref_thr = orc.frangi_threshold(ref)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(2)]
    got = np.concatenate([p["frangi"] for p in parts])
    assert parts[0]["o1"] == parts[1]["o0"]
    for p in parts:
        assert list(p["gamma"]) == [t["gamma"] for t in trace]
        assert list(p["mask_count"]) == [t["mask_count"] for t in trace]
        assert float(p["thr"]) == float(ref_thr)
    assert np.array_equal(got, ref), f"max |d| = {np.abs(got - ref).max()}"
    # Label across the two ranks (run tables over the gloo communicator): the whole-volume labels, numbering included
    ref_lab = orc.label_frame(ref, dr)
    lab = np.concatenate([p["labels"] for p in parts])
    assert np.array_equal(lab, ref_lab), f"{int((lab != ref_lab).sum())} label voxels differ"
    assert all(int(p["n_labels"]) == int(ref_lab.max()) for p in parts) and ref_lab.max() >= 1


def test_stage_api_world2_gloo_writes_the_single_rank_files(tmp_path):
    """Filter(im_info, shard=...).run() / Label(im_info, shard=...).run() as two gloo ranks (nellie_amd/engine.py: RankSlab):
    rank 0 creates the output files, both ranks write their own planes of both frames; the files hold what a single rank
    (here: the oracle) produces, bit for bit."""
    _need_torch()
    from nellie_amd.im_info import ome_tiff
    from nellie_amd.im_info.verifier import ImInfo
    from nellie_amd.synthetic import ISO_01, make_volume
    vols = np.stack([make_volume((64, 30, 34), 70 + t) for t in range(2)])
    src = str(tmp_path / "stack.ome.tif")
    ome_tiff.create(src, vols.shape, np.float32, ISO_01, "raw", data=vols)
    out_dir = str(tmp_path / "out")
    im_info = ImInfo(src, output_dir=out_dir)                      # makes the canonical copy the ranks reuse
    port = 29900 + (os.getpid() % 90)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "dist_stage_worker.py"), src, out_dir, "101.5"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    fr = np.asarray(im_info.get_memmap(im_info.pipeline_paths["im_preprocessed"], read_mode="r"))
    lab = np.load(os.path.join(out_dir, "labels_plain.npy"))
    lab_thr = np.asarray(im_info.get_memmap(im_info.pipeline_paths["im_instance_label"], read_mode="r"))     # the second Label run
    assert fr.shape == vols.shape == lab.shape == lab_thr.shape and fr.dtype == np.float32 and lab.dtype == np.int32
    for t in range(2):
        ref = orc.filter_frame(vols[t], ISO_01)
        assert np.array_equal(fr[t], ref), f"t={t}: {int((fr[t] != ref).sum())} voxels differ"
        assert np.array_equal(lab[t], orc.label_frame(ref, ISO_01)) and lab[t].max() >= 1
        # Label(threshold=101.5) on the slabs: every rank masks the planes it owns with the original image
        want = orc.label_frame(ref, ISO_01, original=vols[t], threshold=101.5)
        assert np.array_equal(lab_thr[t], want) and want.max() >= 1 and not np.array_equal(want, lab[t])
    assert np.array_equal(np.asarray(im_info.get_memmap(im_info.im_path, read_mode="r")), vols), "input file was modified"


def test_engine_plans():
    """Which engine a frame gets (nellie_amd/engine.py): one context up to its index range, slabs beyond it or over several GPUs."""
    from nellie_amd.engine import ShardSpec, plan_engine, slabs_needed
    from nellie_amd.pipeline import FilterParams
    from nellie_amd.synthetic import ISO_01
    p = FilterParams(dim_res=ISO_01)
    assert plan_engine((1024, 1024, 1024), p) == ("single", 1)
    assert plan_engine((320, 2048, 2048), p) == ("single", 1)
    assert plan_engine((600, 2048, 2048), p) == ("local-slabs", 2)          # 2.5e9 voxels: beyond one context
    assert plan_engine((1024, 2048, 2048), p, devices=list(range(8))) == ("local-slabs", 8)
    assert plan_engine((1024, 2048, 2048), p) == ("local-slabs", 3)
    assert plan_engine((512, 512), p, devices=[0, 1]) == ("single", 1)      # 2-D images never shard
    assert plan_engine((64, 64, 64), p, shard=ShardSpec(rank=1, world=4)) == ("rank-slab", 4)
    for shape, w in (((1024, 2048, 2048), 3), ((600, 2048, 2048), 2)):
        owned = -(-shape[0] // w)
        assert (owned + 18) * shape[1] * shape[2] < 2 ** 31 and slabs_needed(shape, 9) == w
    with pytest.raises(MemoryError):
        slabs_needed((64, 40000, 40000), 9)                                 # a plane alone does not fit


def _label_slabs_with_threads(frangi, thr, min_area, world, dr):
    """The Z-slab Label protocol of nellie_amd/sharded.py on `world` oracle-backed contexts, one thread per rank."""
    import threading
    from comms import ThreadComm, ThreadGroup
    from fake_ctx import OracleCtx
    from nellie_amd.pipeline import FilterParams
    from nellie_amd.sharded import ShardedFramePipeline, slab_range
    gshape = frangi.shape
    group = ThreadGroup(world)
    out, errs = [None] * world, []

    def worker(rank):
        try:
            o0, o1 = slab_range(gshape[0], world, rank)
            pipe = ShardedFramePipeline(gshape, rank, world, lambda ctx: ThreadComm(group, rank), FilterParams(dim_res=dr), halo=1,
                                        ctx_factory=lambda shp, dev, g0, gn, ow: OracleCtx(shp, dev, g0, gn, ow))
            pipe.upload_frangi(frangi[o0:o1])
            n = pipe.label(thr, min_area)
            out[rank] = (pipe.download_labels(), n)
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)
            group.barrier.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]
    return np.concatenate([o[0] for o in out]), [o[1] for o in out]


@pytest.mark.parametrize("world", [1, 2, 3, 4])
def test_zslab_label_protocol_golden(world):
    """Label without replication: slab trees joined through the run tables of the shared planes == the reference's labels
    of the whole volume (cavities, face contacts, objects of 65 / 66 / 67 voxels, diagonal-only contacts)."""
    from conftest import load_golden
    for name in ("labelonly_24x48x48", "labelonly_aniso_24x48x48"):
        g = load_golden(name)
        lab, counts = _label_slabs_with_threads(g["frangi"], float(g["label_thr"]), int(g["min_area_pixels"]), world, g["dim_res_dict"])
        assert np.array_equal(lab, g["labels"]), f"{name}: {int((lab != g['labels']).sum())} voxels differ at world {world}"
        assert counts == [int(g["labels"].max())] * world


@pytest.mark.parametrize("world,seed", [(2, 0), (3, 1), (5, 2), (8, 3)])
def test_zslab_label_protocol_random(world, seed):
    """Random blobs, tubes and shells thrown across the interfaces (U shapes through a neighbour, cavities cut by an
    interface, objects that fall under the area limit only on one side): sharded == oracle on the whole volume."""
    rng = np.random.default_rng(seed)
    shape = (8 * world + 3, 26, 31)
    zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    fr = np.zeros(shape, np.float32)
    for _ in range(14):
        c = [rng.uniform(0, s) for s in shape]
        r = rng.uniform(1.5, 5.5)
        d = np.sqrt((zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2)
        fr[(d < r) & (d > r - rng.uniform(1.2, 3.0))] = 1.0                   # shells: cavities
    for _ in range(10):
        y0, x0 = rng.integers(0, shape[1]), rng.integers(0, shape[2])
        z0, z1 = sorted(rng.integers(0, shape[0], 2))
        fr[z0:z1 + 1, y0:y0 + 2, x0:x0 + 2] = 1.0                           # tubes along Z
    fr[rng.random(shape) < 0.02] = 1.0                                       # specks
    ref = orc.get_labels(fr, 0.5, 12)[1]
    lab, counts = _label_slabs_with_threads(fr, 0.5, 12, world, {"X": 0.1, "Y": 0.1, "Z": 0.1, "T": 1.0})
    assert np.array_equal(lab, ref), f"{int((lab != ref).sum())} voxels differ"
    assert counts == [int(ref.max())] * world


def test_library_join_equals_the_numpy_model():
    """nl_host_slab_join (host C++ inside the library, what every rank runs on the gathered tables) against the numpy model
    sharded.join_slab_tables: same nodes, values, components and numbering, on random tables of 1..8 ranks with trees that span
    several planes and several ranks (the library is loaded, no GPU is touched)."""
    from nellie_amd import hipnative, sharded
    hipnative.load()
    rng = np.random.default_rng(7)
    for world in (1, 2, 3, 8):
        for trial in range(6):
            shared = [int(rng.integers(0, 40)) for _ in range(2 * (world - 1))]        # entries of the 2 shared planes per interface
            blobs = []
            for r in range(world):
                n = [shared[2 * (r - 1)] if r > 0 else 0, shared[2 * (r - 1) + 1] if r > 0 else 0,
                     shared[2 * r] if r + 1 < world else 0, shared[2 * r + 1] if r + 1 < world else 0]
                trees = rng.integers(0, 5000, size=max(1, sum(n) // 3 + 1))               # few trees: many entries share one
                val_of = {int(t): int(rng.integers(0, 2 ** 31 - 1)) for t in trees}
                roots = [rng.choice(trees, size=k).astype(np.int32) for k in n]
                vals = [np.array([val_of[int(t)] for t in rr], np.int32) for rr in roots]
                blobs.append(sharded.pack_slab_tables(roots, vals, nruns=5000))
            ref = sharded.join_slab_tables([sharded.unpack_slab_tables(b) for b in blobs])
            rank, root, val, comp, ncomp = hipnative.host_slab_join(blobs)
            assert ncomp == ref.ncomp and np.array_equal(rank, ref.rank) and np.array_equal(root, ref.root)
            assert np.array_equal(val, ref.val) and np.array_equal(comp, ref.comp)
    # ranks that disagree about a shared plane are an error, not a silent mis-join
    a = sharded.pack_slab_tables([[], [], [1, 2], [3]], [[], [], [0, 0], [0]])
    b = sharded.pack_slab_tables([[5], [6], [], []], [[0], [0], [], []])
    with pytest.raises(Exception, match="disagree"):
        hipnative.host_slab_join([a, b])
