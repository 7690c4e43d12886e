"""Committed golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py from the CPU oracle).

CPU: the oracle must reproduce its frozen answers bit for bit (guards the checker itself against drift).
GPU: the HIP path, through the C ABI, must match the frozen answers within the parity tolerance (1e-5 RMS, the bound
BASELINE.json states) — these run without building or loading the oracle."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402

FX = np.load(os.path.join(HERE, "golden", "effects.npz"))
GR = np.load(os.path.join(HERE, "golden", "graphs.npz"))


def close(a, b, rms_tol=1e-5, max_tol=1e-4):
    assert a.shape == b.shape and np.isfinite(a).all()
    d = a.astype(np.float64) - b.astype(np.float64)
    assert float(np.sqrt(np.mean(d * d))) <= rms_tol
    assert float(np.abs(d).max()) <= max_tol


@pytest.mark.parametrize("case", mg.EFFECT_CASES, ids=[c[0] for c in mg.EFFECT_CASES])
def test_oracle_reproduces_effect_vectors(case):
    import oracle

    x, y = mg.run_effect(oracle.OracleEffect, case)
    assert np.array_equal(x, FX[case[0] + "_in"])
    assert np.array_equal(y, FX[case[0] + "_out"])
    assert not np.array_equal(x, y)


@pytest.mark.parametrize("case", mg.GRAPH_CASES, ids=[c[0] for c in mg.GRAPH_CASES])
def test_oracle_reproduces_graph_vectors(case):
    import oracle

    assert np.array_equal(mg.run_graph(oracle.OracleGraph, case), GR[case[0]])
    assert np.abs(GR[case[0]]).max() > 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("case", mg.EFFECT_CASES, ids=[c[0] for c in mg.EFFECT_CASES])
def test_gpu_effect_matches_golden(case):
    import phonic_amd

    x, y = mg.run_effect(lambda kind, params, seeds: phonic_amd.Effect(kind, params, seeds), case)
    assert np.array_equal(x, FX[case[0] + "_in"])
    close(y, FX[case[0] + "_out"])


@pytest.mark.gpu
@pytest.mark.parametrize("case", mg.GRAPH_CASES, ids=[c[0] for c in mg.GRAPH_CASES])
def test_gpu_graph_matches_golden(case):
    from phonic_amd.graph import Graph

    close(mg.run_graph(lambda sr, ch, mf: Graph(sr, ch, mf, 0), case), GR[case[0]])
