"""tools/knn_model.py is the CPU model of the Hilbert-chunk kNN kernel that design decisions of glim_amd/csrc/knn_chunks.hip were priced with (it reproduces
the kernel's per-wavefront work counters exactly, profiles/r02/probe/knn_model_results.txt).  Two things are pinned here, on the CPU:
the model answers exactly like the oracle, and so does its step-by-step emulation of the per-lane threshold selection of the chunk kernels
(knn_chunks.hip / knn_pairs.hip, the shipped path for k <= 10) -- on the distributions the GPU tests use for the kernel itself (ties everywhere, duplicates, far offset, two scales, fewer
points than a chunk)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _cases():
    rng = np.random.default_rng(5)
    return {
        "lattice": np.stack(np.meshgrid(np.arange(14), np.arange(14), np.arange(10), indexing="ij"), -1).reshape(-1, 3) * 0.25,
        "identical": np.tile([[1.0, 2.0, 3.0]], (700, 1)),
        "offset": rng.uniform(-1, 1, (2500, 3)) + [1e5, -2e5, 3e4],
        "two_scales": np.vstack([rng.normal(size=(1500, 3)) * 0.01, rng.uniform(-50, 50, (1200, 3))]),
        "few": rng.uniform(-1, 1, (70, 3)),
        "duplicates": np.repeat(rng.uniform(-1, 1, (300, 3)), 9, axis=0),
    }


@pytest.mark.parametrize("name", list(_cases()))
def test_chunk_model_and_staged_selection_are_exact(orc, name):
    import knn_model as km

    pts = _cases()[name].astype(np.float32)
    ref = orc.knn(pts.astype(np.float64), km.K, method="brute")
    _, rounds0, out0 = km.run(pts, "index")
    np.testing.assert_array_equal(out0, ref)
    with np.errstate(over="ignore"):
        _, rounds1, out1 = km.run(pts, "index", select_bits=(8, 0))  # selection in every scan that accepted anything
    np.testing.assert_array_equal(out1, ref)
    assert rounds1.sum() <= rounds0.sum()  # dropping candidates can only shorten the lock-step insertion loops


@pytest.mark.parametrize("name", ["lattice", "offset", "few", "duplicates"])
def test_pair_lane_model_and_staged_selection_are_exact(orc, name):
    """The same for the pair-lane kernel (tools/knn_pair_model.py): two lists per query over disjoint candidates, each lane selecting over its OWN
    list, merged at the end."""
    import knn_pair_model as pm

    pts = _cases()[name].astype(np.float32)
    ref = orc.knn(pts.astype(np.float64), pm.K, method="brute")
    rounds0, out0 = pm.run(pts)
    np.testing.assert_array_equal(out0, ref)
    rounds1, out1 = pm.run(pts, select=(8, 0))
    np.testing.assert_array_equal(out1, ref)
    assert rounds1.sum() <= rounds0.sum()
