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
