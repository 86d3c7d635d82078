"""CPU tests of host-side decisions of nellie_amd/pipeline.py that need no device (an oracle-backed context stands in for the library)."""
import numpy as np
import pytest


def _pipe(shape):
    from fake_ctx import OracleCtx
    from nellie_amd import pipeline as pl
    return pl.FramePipeline(shape, ctx=OracleCtx(shape))


def test_steps_that_fit_ahead():
    """_step_fits_ahead (round 5): nl_gauss_step_ahead takes cascade steps that write at most two of the three ping-pong volumes -- a Z
    pass and a fused Y+X pass (equal in-plane radii up to nl_ctx_info("gauss_yx_max_r") that fit the Y axis).  A context that does not
    say what it fuses (no such key) gets no step ahead at all."""
    from nellie_amd import pipeline as pl
    pipe = _pipe((24, 40, 48))
    w1, w4, w13 = pl.gaussian_weights(0.4), pl.gaussian_weights(1.3), pl.gaussian_weights(4.3)
    assert [(len(w) - 1) // 2 for w in (w1, w4, w13)] == [1, 4, 13]
    assert pipe._yx_max_r == 0                                   # the stand-in context has no "gauss_yx_max_r"
    assert not pipe._step_fits_ahead([w1, w4, w4])               # ... so Y and X count as two passes: three volumes
    assert pipe._step_fits_ahead([None, w4, w4]) and pipe._step_fits_ahead([w1, None, w4])
    pipe._yx_max_r = 12
    assert pipe._step_fits_ahead([w1, w4, w4]) and pipe._step_fits_ahead([w13, w4, w4])      # any Z radius: one pass
    assert not pipe._step_fits_ahead([w1, w13, w13])             # radius 13 has no fused Y+X kernel
    assert not pipe._step_fits_ahead([w1, w4, w13])              # unequal in-plane radii
    assert pipe._step_fits_ahead([w1, w13, None]) and pipe._step_fits_ahead([None, w13, w13])
    thin = _pipe((24, 3, 48))
    thin._yx_max_r = 12
    assert not thin._step_fits_ahead([w1, w4, w4])               # radius 4 beyond a Y axis of 3 rows: the generic passes
    flat = _pipe((1, 40, 48))                                    # one plane (a 2-D image is held that way): no Z weights
    flat._yx_max_r = 12
    assert flat._step_fits_ahead([None, w4, w4]) and flat._step_fits_ahead([None, w13, w13])


def test_chain_ahead_rule(monkeypatch):
    """pipeline._chain_ahead: below 2^26 voxels on a single context, never on a Z slab, NELLIE_CHAIN_AHEAD overrides."""
    from nellie_amd import pipeline as pl
    from nellie_amd.sharded import ShardedFramePipeline
    pipe = _pipe((8, 16, 16))
    assert pipe._chain_ahead(1 << 20) and pipe._chain_ahead((1 << 26) - 1) and not pipe._chain_ahead(1 << 26)
    pipe._chain_ahead_env = "0"
    assert not pipe._chain_ahead(1 << 20)
    pipe._chain_ahead_env = "1"
    assert pipe._chain_ahead(1 << 30)
    assert ShardedFramePipeline._chain_ahead(object.__new__(ShardedFramePipeline), 1 << 20) is False


def test_engine_plan_looks_at_free_hbm():
    """Round 6 (VERDICT r05 "next 5"; nellie/utils/adaptive_run.py:88-113 is the reference's ladder): plan_engine / slabs_needed weigh the
    bytes a context needs (35 B/voxel + the resident input) against the free HBM of the devices the slabs land on -- test doubles answer
    both questions here -- spread over the GPUs named, and raise MemoryError("... out of memory ...") with the figures when nothing fits."""
    from nellie_amd import engine
    from nellie_amd.pipeline import FilterParams
    from nellie_amd.synthetic import ISO_01
    from nellie_amd.utils import adaptive_run
    p = FilterParams(dim_res=ISO_01)
    shape = (512, 1024, 1024)
    bytes_of = lambda s: 39 * int(s[0]) * int(s[1]) * int(s[2])          # noqa: E731
    roomy = lambda d: 288 << 30                                           # noqa: E731
    assert engine.plan_engine(shape, p, free_of=roomy, bytes_of=bytes_of) == ("single", 1)
    tight = lambda d: 12 << 30                                            # noqa: E731   (the frame needs 19.5 GiB)
    with pytest.raises(MemoryError) as exc:
        engine.plan_engine(shape, p, free_of=tight, bytes_of=bytes_of)
    assert adaptive_run.is_oom_error(exc.value) and "GiB free" in str(exc.value) and "needs" in str(exc.value)
    # two GPUs with 12 GiB free each take one slab each (10.1 GiB with the ghost planes) ...
    assert engine.plan_engine(shape, p, devices=[0, 1], free_of=tight, bytes_of=bytes_of) == ("local-slabs", 2)
    # ... but not when the second one is nearly full: the message names the device that is short
    uneven = lambda d: (12 << 30) if d == 0 else (2 << 30)               # noqa: E731
    with pytest.raises(MemoryError) as exc:
        engine.plan_engine(shape, p, devices=[0, 1], free_of=uneven, bytes_of=bytes_of)
    assert "GPU 1: needs" in str(exc.value)
    assert engine.plan_engine(shape, p, devices=[0, 2, 4, 6], free_of=lambda d: 6 << 30, bytes_of=bytes_of) == ("local-slabs", 4)
    # a device that cannot be asked (no GPU here, a CPU double) never blocks a plan
    assert engine.plan_engine(shape, p, free_of=lambda d: None, bytes_of=bytes_of) == ("single", 1)
    # the index range still decides first: a frame beyond 2^31 voxels is cut whatever the memory says
    kind, w = engine.plan_engine((1024, 2048, 2048), p, free_of=roomy, bytes_of=bytes_of)
    assert kind == "local-slabs" and w >= 3
    fits, need, free = engine.memory_plan(shape, 9, 2, [0, 0], free_of=tight, bytes_of=bytes_of)
    assert not fits and need[0] == 2 * bytes_of((256 + 9, 1024, 1024))     # two slabs on ONE GPU: more memory than the whole frame, not less


def test_device_strings():
    """filtering.py:117-120 / adaptive_run.py:14-20 plus "hip", the name INTEGRATION.md's dispatch forwards."""
    from nellie_amd.utils import adaptive_run
    assert [adaptive_run.normalize_device(d) for d in ("auto", "cpu", "gpu", "cuda", "hip", "HIP", None)] == ["auto", "cpu", "gpu", "gpu", "gpu", "gpu", "auto"]
    with pytest.raises(ValueError):
        adaptive_run.normalize_device("tpu")
