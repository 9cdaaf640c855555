"""Committed golden fixture (tests/golden/vgicp_small.npz, made by tests/golden/make_golden.py from the oracle).

CPU: the oracle still reproduces it (regression pin; NOT a reference pin -- DESIGN.md section 2).
GPU: the HIP path reproduces it without the oracle being consulted at run time.
"""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(os.path.join(HERE, "golden", "vgicp_small.npz")))


@pytest.mark.parametrize("tag,res", [("r050", 0.5), ("r100", 1.0)])
def test_oracle_reproduces_golden(orc, gold, tag, res):
    vm = orc.VoxelMap(res).insert(gold["target_points"], gold["target_covs"].astype(np.float64))
    coords, counts, means, covs = vm.voxels()
    np.testing.assert_array_equal(coords, gold[f"{tag}_voxel_coords"])
    np.testing.assert_array_equal(counts, gold[f"{tag}_voxel_counts"])
    np.testing.assert_allclose(means, gold[f"{tag}_voxel_means"], rtol=0, atol=1e-13)
    L = orc.vgicp_linearize(vm, gold["source_points"], gold["source_covs"].astype(np.float64), gold["delta"], num_threads=2, want_corr=True)
    assert L["num_inliers"] == int(gold[f"{tag}_num_inliers"])
    np.testing.assert_array_equal(L["corr"], gold[f"{tag}_corr"])
    for k in ("H_tt", "H_ss", "H_ts", "b_t", "b_s"):
        np.testing.assert_allclose(L[k], gold[f"{tag}_{k}"], rtol=1e-11, atol=1e-8)
    np.testing.assert_allclose(L["error"], float(gold[f"{tag}_error"]), rtol=1e-12)
    assert orc.overlap(vm, gold["source_points"], gold["delta"]) == float(gold[f"{tag}_overlap"])
    nb = orc.knn(gold["source_points"], 10)
    np.testing.assert_array_equal(nb, gold["source_neighbors"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag,res", [("r050", 0.5), ("r100", 1.0)])
def test_hip_reproduces_golden(gold, tag, res):
    from glim_amd import api

    ctx = api.Context(0, 1)
    tg = api.PointCloudGPU.clone(gold["target_points"], gold["target_covs"], ctx=ctx)
    sg = api.PointCloudGPU.clone(gold["source_points"], gold["source_covs"], gold["source_normals"], ctx=ctx)
    vm = api.GaussianVoxelMapGPU(res, ctx=ctx).insert(tg)
    coords, counts, means, covs = vm.voxels()
    og, orr = np.lexsort(coords.T[::-1]), np.lexsort(gold[f"{tag}_voxel_coords"].T[::-1])
    np.testing.assert_array_equal(coords[og], gold[f"{tag}_voxel_coords"][orr])
    np.testing.assert_array_equal(counts[og], gold[f"{tag}_voxel_counts"][orr])
    np.testing.assert_allclose(means[og], gold[f"{tag}_voxel_means"][orr], rtol=2e-7, atol=1e-7)
    np.testing.assert_allclose(covs[og], gold[f"{tag}_voxel_covs"][orr], rtol=2e-7, atol=1e-7)
    fset = api.NonlinearFactorSetGPU(ctx)
    fset.add(api.IntegratedVGICPFactorGPU(0, 1, vm, sg))
    got = fset.linearize({0: np.eye(4), 1: gold["delta"]})[0]
    assert got["num_inliers"] == int(gold[f"{tag}_num_inliers"])
    corr = fset.correspondences(0, gold["delta"])
    np.testing.assert_array_equal(corr[:, :3], gold[f"{tag}_corr"][:, :3])
    np.testing.assert_array_equal(corr[:, 3] > 0, gold[f"{tag}_corr"][:, 3] >= 0)
    for k in ("H_tt", "H_ss", "H_ts"):
        np.testing.assert_allclose(got[k], gold[f"{tag}_{k}"], rtol=0, atol=2e-4 * np.abs(gold[f"{tag}_{k}"]).max())
    d_got = np.linalg.solve(got["H_ss"], -got["b_s"])
    d_ref = np.linalg.solve(gold[f"{tag}_H_ss"], -gold[f"{tag}_b_s"])
    assert np.abs(d_got - d_ref).max() < 1e-4
    assert api.overlap_gpu(vm, sg, gold["delta"]) == float(gold[f"{tag}_overlap"])
    np.testing.assert_array_equal(sg.find_neighbors(10), gold["source_neighbors"])
