"""
CPU tests of the N > 1 path: slab geometry, and the Z-slab Filter host logic under gloo with world_size 2
(oracle-backed context, tests/fake_ctx.py).  The sharded result must equal the single-volume oracle bit for bit.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO
from oracle import nellie_oracle as orc


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


@pytest.mark.parametrize("aniso", [0, 1])
def test_zslab_filter_world2_gloo(aniso, tmp_path):
    pytest.importorskip("torch")
    from nellie_amd.synthetic import ANISO_03, ISO_01, make_volume
    port = 29600 + aniso + (os.getpid() % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(REPO, "tests", "dist_worker.py"), str(tmp_path), str(aniso)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
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
