"""tools/knn_qgroup_model.py is the CPU model of the query-group kNN kernel (glim_amd/csrc/knn_qgroup.hip, the default kNN kernel): the same chunk
walk, the same conservative FP32 tests (deflated box gaps against the inflated FP32 image of the bound; the FP32 distance image that gates the exact
evaluation), the same acceptance order with the ballot re-taken after every insertion.  Pinned here, on the CPU: the model answers exactly like the
oracle's brute-force search -- i.e. the pruning argument of the kernel loses no neighbour and no tie order -- on the distributions the GPU tests use
for the kernel itself (ties everywhere, identical points, a far offset, two scales, fewer points than two chunks, duplicates), with one and with
two queries per wavefront, for list sizes 10 and 5."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _cases():
    rng = np.random.default_rng(5)
    return {
        "lattice": np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(8), indexing="ij"), -1).reshape(-1, 3) * 0.25,
        "identical": np.tile([[1.0, 2.0, 3.0]], (500, 1)),
        "offset": rng.uniform(-1, 1, (1500, 3)) + [1e5, -2e5, 3e4],
        "two_scales": np.vstack([rng.normal(size=(900, 3)) * 0.01, rng.uniform(-50, 50, (700, 3))]),
        "few": rng.uniform(-1, 1, (70, 3)),
        "duplicates": np.repeat(rng.uniform(-1, 1, (200, 3)), 9, axis=0),
    }


@pytest.mark.parametrize("name", list(_cases()))
def test_query_group_model_is_exact(orc, name):
    import knn_qgroup_model as qm

    pts = _cases()[name].astype(np.float32)
    for k in (10, 5):
        ref = orc.knn(pts.astype(np.float64), k, method="brute")
        for q in (1, 2):
            got, counters = qm.run(pts, k, q)
            np.testing.assert_array_equal(got, ref, err_msg=f"{name} k={k} {q} queries per wavefront")
            assert (counters[:, 0] >= 1).all()  # every wavefront scanned at least its own chunk


def test_query_group_model_counts_what_the_kernel_counts():
    """The kernel's own counters (diag knn_debug, profiles/r04/probe/knn_qgroup.txt: 4.7 chunk scans, 3.9 exact query-chunk evaluations and 7.6
    insertions per two-query wavefront on a 10 000-point scan subset) are reproduced by the model on a smaller subset of the same scan within the
    spread between cloud sizes: the model walks like the kernel."""
    import knn_qgroup_model as qm
    from glim_amd import synth

    scan = synth.scan(synth.Scene.default(), synth.arc_trajectory(1)[0], synth.lidar_directions(64, 512), 0)[:, :3]
    pts = scan[np.sort(np.random.default_rng(0).choice(len(scan), 2048, replace=False))].astype(np.float32)
    got, counters = qm.run(pts, 10, 2)
    np.testing.assert_array_equal(got, qm.brute(pts, 10))
    scans, exact, inserts = counters[:, 0].mean(), counters[:, 1].mean(), counters[:, 2].mean()
    assert 3.5 <= scans <= 6.0 and 3.0 <= exact <= 5.0 and 5.5 <= inserts <= 10.0, (scans, exact, inserts)
