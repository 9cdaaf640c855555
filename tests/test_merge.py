"""Submap merge: gtsam_points::merge_frames as called at src/glim/mapping/sub_mapping.cpp:480-497.
Oracle pins on CPU (independent numpy restatement), HIP parity on the GPU."""
import numpy as np
import pytest


def keyframes(orc, n_frames=5, rings=32, az=256, seed=0):
    """A short arc of LiDAR keyframes with kNN covariances (f32-representable inputs) and their poses in the submap origin."""
    from glim_amd import synth

    scene = synth.Scene.default()
    dirs = synth.lidar_directions(rings, az)
    poses = synth.arc_trajectory(n_frames, step=0.8, yaw_step_deg=3.0)
    origin = np.linalg.inv(poses[n_frames // 2])
    pts, covs, rel = [], [], []
    for i, T in enumerate(poses):
        p = synth.scan(scene, T, dirs, frame_id=seed + i).astype(np.float64)
        _, c = orc.covariances(p, orc.knn(p, 10))
        pts.append(p)
        covs.append(c.astype(np.float32).astype(np.float64))
        rel.append(origin @ T)
    return rel, pts, covs


def np_merge(poses, pts, covs, res):
    P = np.concatenate([p @ T[:3, :3].T + T[:3, 3] for T, p in zip(poses, pts)])
    C = np.concatenate([np.einsum("ij,njk,lk->nil", T[:3, :3], c, T[:3, :3]) for T, c in zip(poses, covs)])
    key = np.floor(P / res).astype(np.int64) + 1048576
    k = key[:, 0] | (key[:, 1] << 21) | (key[:, 2] << 42)
    uk, inv, cnt = np.unique(k, return_inverse=True, return_counts=True)
    MP = np.stack([np.bincount(inv, weights=P[:, a]) / cnt for a in range(3)], 1)
    MC = np.stack([np.bincount(inv, weights=C[:, a, b]) / cnt for a in range(3) for b in range(3)], 1).reshape(-1, 3, 3)
    return MP, MC


def test_oracle_merge_matches_numpy_restatement(orc):
    poses, pts, covs = keyframes(orc)
    mp, mc = orc.merge_frames(poses, pts, covs, 0.3, target_num_points=-1, block_size=0)
    rp, rc = np_merge(poses, pts, covs, 0.3)
    assert len(mp) == len(rp) > 3000
    np.testing.assert_allclose(mp, rp, rtol=0, atol=1e-9)
    np.testing.assert_allclose(mc, rc, rtol=0, atol=1e-9)
    assert np.array_equal(mc, np.transpose(mc, (0, 2, 1)))  # exactly symmetric
    assert np.all(np.linalg.eigvalsh(mc)[:, 0] > 0)  # means of PSD matrices with a 1e-3 floor stay PD
    # 1024-entry averaging blocks only split voxels (a few more points), never move them
    mp2, _ = orc.merge_frames(poses, pts, covs, 0.3, block_size=1024)
    assert len(mp) < len(mp2) <= len(mp) + sum(len(p) for p in pts) // 1024 + 1
    # target size: exactly floor(m * (target / m)) points, a subset in the original order, seed dependent
    m = len(mp2)
    tp, tc = orc.merge_frames(poses, pts, covs, 0.3, target_num_points=2000, seed=4)
    assert len(tp) == int(m * (2000 / m)) and len(tc) == len(tp)
    idx = [np.flatnonzero(np.all(mp2 == q, axis=1))[0] for q in tp[:50]]
    assert np.all(np.diff(idx) > 0)
    tp2, _ = orc.merge_frames(poses, pts, covs, 0.3, target_num_points=2000, seed=5)
    assert not np.array_equal(tp, tp2)
    # a target above the merged size changes nothing
    ap, _ = orc.merge_frames(poses, pts, covs, 0.3, target_num_points=10**7)
    np.testing.assert_array_equal(ap, mp2)


def test_oracle_merge_single_frame_identity_pose(orc):
    poses, pts, covs = keyframes(orc, n_frames=1)
    mp, mc = orc.merge_frames([np.eye(4)], pts, covs, 1e-4, block_size=0)  # voxels far smaller than the point spacing: nothing merges
    order = np.lexsort((pts[0][:, 0], pts[0][:, 1], pts[0][:, 2]))
    assert len(mp) == len(pts[0])
    np.testing.assert_array_equal(np.sort(mp, axis=0), np.sort(pts[0], axis=0))
    assert len(order) == len(mp)
    assert orc.merge_frames([], [], [], 0.5)[0].shape == (0, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("target,block", [(-1, 1024), (-1, 0), (20000, 1024), (10**7, 1024)])
def test_hip_merge_matches_oracle(orc, target, block):
    """Bit-exact merged points and covariances (FP64), same order; the FP32 device cloud is their rounding and feeds the factor path."""
    from glim_amd import api

    ctx = api.Context(0, 1)
    poses, pts, covs = keyframes(orc, n_frames=15, rings=64, az=512)
    rp, rc = orc.merge_frames(poses, pts, covs, 0.25, target_num_points=target, seed=3, block_size=block)
    g = api.merge_frames(poses, pts, covs, 0.25, target_num_points=target, seed=3, block_size=block, ctx=ctx)
    gp, gc = g.download_merged()
    assert g.size() == len(rp) > 10000
    np.testing.assert_array_equal(gp, rp)
    np.testing.assert_array_equal(gc, rc)
    xyz, c32, _ = g.download(covs=True, normals=False)
    np.testing.assert_array_equal(xyz, rp.astype(np.float32))
    np.testing.assert_array_equal(c32, rc.astype(np.float32))
    # the merged submap is what global mapping turns into a voxel map and a factor source (global_mapping.cpp:253-266)
    vm = api.GaussianVoxelMapGPU(1.0, ctx=ctx).insert(g)
    assert vm.voxelmap_info()["num_voxels"] > 500
    fset = api.NonlinearFactorSetGPU(ctx)
    fset.add(api.IntegratedVGICPFactorGPU(np.eye(4), 1, vm, g))
    assert fset.linearize({1: np.eye(4)})[0]["num_inliers"] == g.size()


@pytest.mark.gpu
def test_hip_merge_edge_cases(orc):
    from glim_amd import api

    ctx = api.Context(0, 1)
    assert api.merge_frames([], [], [], 0.5, ctx=ctx).size() == 0
    poses, pts, covs = keyframes(orc, n_frames=2)
    # an empty keyframe among the inputs, and non-finite points are dropped
    pts2 = [pts[0].copy(), np.zeros((0, 3)), pts[1]]
    pts2[0][7] = np.nan
    covs2 = [covs[0], np.zeros((0, 3, 3)), covs[1]]
    poses2 = [poses[0], np.eye(4), poses[1]]
    rp, rc = orc.merge_frames(poses2, pts2, covs2, 0.5)
    gp, gc = api.merge_frames(poses2, pts2, covs2, 0.5, ctx=ctx).download_merged()
    np.testing.assert_array_equal(gp, rp)
    np.testing.assert_array_equal(gc, rc)
    with pytest.raises(api.GlimAmdError):
        api.merge_frames(poses, pts, covs, 0.0, ctx=ctx)
