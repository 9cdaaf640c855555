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


# ------------------------------------------------------------------------------------------------------------------
# scan preprocessing, deskewing, submap merge, GICP factor (tests/golden/frontend_small.npz)
# ------------------------------------------------------------------------------------------------------------------
def _recipe():
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden_frontend", os.path.join(HERE, "golden", "make_golden_frontend.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def front():
    return dict(np.load(os.path.join(HERE, "golden", "frontend_small.npz")))


def test_oracle_reproduces_frontend_golden(orc, gold, front):
    mg = _recipe()
    out = mg.compute(orc, front, mg.gicp_pair_from(gold))
    assert set(out) <= set(front)
    for k, v in out.items():
        if k.startswith("gicp_") and k not in ("gicp_corr", "gicp_num_inliers"):
            np.testing.assert_allclose(v, front[k], rtol=1e-11, atol=1e-8, err_msg=k)  # OpenMP reduction order
        elif k.startswith("deskew_"):
            np.testing.assert_allclose(v, front[k], rtol=0, atol=1e-12, err_msg=k)  # libm sin / cos in the pose table
        else:
            np.testing.assert_array_equal(v, front[k], err_msg=k)  # integer / order / sequential-sum work: bit-exact


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["pre_random", "pre_voxel"])
def test_hip_preprocess_reproduces_golden(front, tag):
    from glim_amd import api

    mg = _recipe()
    ctx = api.Context(0, 1)
    kw = mg.PRE_RANDOM if tag == "pre_random" else mg.PRE_VOXEL
    g = api.PointCloudGPU.preprocess(front["raw_points"], front["raw_times"], front["raw_intensities"], api.preprocess_params(**kw), ctx=ctx)
    got = g.download_frame()
    assert g.size() == len(front[f"{tag}_points"]) > 100
    np.testing.assert_array_equal(got["points"], front[f"{tag}_points"])
    np.testing.assert_array_equal(got["times"], front[f"{tag}_times"])
    np.testing.assert_array_equal(got["intensities"], front[f"{tag}_intensities"])
    np.testing.assert_array_equal(got["neighbors"], front[f"{tag}_neighbors"])


@pytest.mark.gpu
def test_hip_deskew_reproduces_golden(front):
    from glim_amd import api

    mg = _recipe()
    ctx = api.Context(0, 1)
    pre = api.PointCloudGPU.preprocess(front["raw_points"], front["raw_times"], front["raw_intensities"], api.preprocess_params(**mg.PRE_RANDOM), ctx=ctx)
    Til = front["T_imu_lidar"]
    for key, kw in (("deskew_constvel", dict(linear_vel=mg.LINEAR_VEL, angular_vel=mg.ANGULAR_VEL)),
                    ("deskew_imu", dict(imu_times=front["imu_times"], imu_poses=list(front["imu_poses"]), stamp=100.0))):
        xyz, _, _ = pre.deskew(Til, **kw).download(covs=False, normals=False)
        ref = front[key]
        ulp = np.spacing(np.abs(ref.astype(np.float32))).astype(np.float64)
        assert xyz.shape == ref.shape and np.all(np.abs(xyz.astype(np.float64) - ref) <= ulp), key  # FP64 transform stored as FP32


@pytest.mark.gpu
def test_hip_merge_reproduces_golden(front):
    from glim_amd import api

    mg = _recipe()
    ctx = api.Context(0, 1)
    poses, pts, covs = list(front["key_poses"]), list(front["key_points"]), [c.astype(np.float64) for c in front["key_covs"]]
    gp, gc = api.merge_frames(poses, pts, covs, mg.MERGE_RES, ctx=ctx).download_merged()
    np.testing.assert_array_equal(gp, front["merge_all_points"])
    np.testing.assert_array_equal(gc, front["merge_all_covs"])
    gp, gc = api.merge_frames(poses, pts, covs, mg.MERGE_RES, target_num_points=mg.MERGE_TARGET, seed=mg.MERGE_SEED, ctx=ctx).download_merged()
    np.testing.assert_array_equal(gp, front["merge_target_points"])
    np.testing.assert_array_equal(gc, front["merge_target_covs"])


@pytest.mark.gpu
def test_hip_gicp_reproduces_golden(gold, front):
    from glim_amd import api

    mg = _recipe()
    ctx = api.Context(0, 1)
    tg = api.PointCloudGPU.clone(gold["target_points"], gold["target_covs"], ctx=ctx)
    sg = api.PointCloudGPU.clone(gold["source_points"], gold["source_covs"], ctx=ctx)
    f = api.IntegratedGICPFactor(0, 1, tg, sg, max_correspondence_distance=mg.GICP_MAX_D)
    values = {0: np.eye(4), 1: gold["delta"]}
    got = f.linearize(values)
    np.testing.assert_array_equal(f.correspondences(values), front["gicp_corr"])
    assert got["num_inliers"] == int(front["gicp_num_inliers"]) > 50
    np.testing.assert_allclose(got["error"], float(front["gicp_error"]), rtol=2e-4)
    for k in ("H_tt", "H_ss", "H_ts"):
        np.testing.assert_allclose(got[k], front[f"gicp_{k}"], rtol=0, atol=2e-4 * np.abs(front[f"gicp_{k}"]).max())
    lam = 1e-6 * np.trace(front["gicp_H_ss"]) / 6
    d_got = np.linalg.solve(got["H_ss"] + lam * np.eye(6), -got["b_s"])
    d_ref = np.linalg.solve(front["gicp_H_ss"] + lam * np.eye(6), -front["gicp_b_s"])
    assert np.abs(d_got - d_ref).max() < 1e-4
    f.close()
