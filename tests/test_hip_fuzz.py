"""
A fixed-seed slice of the randomised differential runs (tools/fuzz_parity.py, fuzz_slabs.py, fuzz_stages.py) in the GPU
suite: shapes no hand-written case has (thin, prime, ragged rows), every input dtype, five spacings, textures from empty to
dense -- the 3-D hot path and the "next" rows against the oracle, Z slabs against one context.  The full runs (thousands of
cases) are in profiles/r04_fuzz_*.txt.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


def test_fuzz_slice_hot_path_vs_oracle(hip):
    import fuzz_parity as F
    rng = np.random.default_rng(2024)
    results = [F.one_case(rng, i) for i in range(60)]
    bad = [r for r in results if not r["ok"]]
    assert not bad, bad[:3]
    assert sum(r["result"] == "equal" for r in results) >= 40         # most cases meet the suite's own bars


def test_fuzz_slice_stages_vs_oracle(hip):
    import fuzz_stages as S
    rng = np.random.default_rng(2025)
    for i in range(60):
        info = (S.case_markers, S.case_network, S.case_2d, S.case_label)[i % 4](rng, i)      # an AssertionError names the stage and the count
        assert info["ok"], info


def test_fuzz_slice_slabs_vs_one_context(hip):
    import fuzz_slabs as Z
    rng = np.random.default_rng(2026)
    results = [Z.one_case(rng, i) for i in range(12)]
    bad = [r for r in results if not r["ok"]]
    assert not bad, bad[:3]
    assert sum(r["result"] == "identical" for r in results) >= 6


def test_fuzz_slice_stage_classes_vs_oracle(hip):
    """Filter / Label / Markers classes with random keywords, stacks of 1-3 frames, one context or `devices=[0, 0(, 0)]`."""
    import fuzz_api as A
    rng = np.random.default_rng(2027)
    results = [A.one_case(rng, i) for i in range(10)]
    bad = [r for r in results if not r["ok"]]
    assert not bad, bad[:3]


def test_fuzz_slice_through_files(hip, tmp_path):
    """OME-TIFF -> FileInfo / ImInfo -> run(markers=True) and run_streamed() -> every product reopened from its file."""
    import fuzz_files as FF
    rng = np.random.default_rng(2028)
    results = [FF.one_case(rng, i, str(tmp_path)) for i in range(6)]
    bad = [r for r in results if not r["ok"]]
    assert not bad, bad[:3]
