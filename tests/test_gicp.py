"""GICP factor: gtsam_points::IntegratedGICPFactor as constructed at sub_mapping.cpp:202,
global_mapping.cpp:400, global_mapping_pose_graph.cpp:393.  Oracle pins on CPU, HIP parity on the GPU."""
import numpy as np
import pytest

POSE_TOL = 1e-4  # BASELINE.json north_star: pose delta within 1e-4 m / 1e-4 rad per Gauss-Newton iteration


def gn_step(L, lam=0.0):
    return np.linalg.solve(L["H_ss"] + lam * np.eye(6), -L["b_s"])


def np_gicp(tp, tc, sp, sc, T, max_d):
    """Independent restatement: scipy kd-tree correspondences, dense J^T M J accumulation."""
    from scipy.spatial import cKDTree

    R, t = T[:3, :3], T[:3, 3]
    q = sp @ R.T + t
    d, j = cKDTree(tp).query(q, k=1)
    ok = d <= max_d
    H, b, e = np.zeros((6, 6)), np.zeros(6), 0.0

    def hat(v):
        return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])

    for i in np.flatnonzero(ok):
        M = np.linalg.inv(tc[j[i]] + R @ sc[i] @ R.T)
        r = tp[j[i]] - q[i]
        Js = np.hstack([R @ hat(sp[i]), -R])
        H += Js.T @ M @ Js
        b += Js.T @ M @ r
        e += r @ M @ r
    return dict(H_ss=H, b_s=b, error=e, num_inliers=int(ok.sum()), corr=np.where(ok, j, -1))


def test_oracle_gicp_matches_numpy_restatement(orc, small_pair):
    t, s = small_pair["target"], small_pair["source"]
    T = small_pair["delta"] @ orc.se3_exp([0.01, -0.02, 0.005, 0.05, 0.02, -0.01])
    for max_d in (1.0, 0.3):
        ref = np_gicp(t["points"], t["covs"], s["points"], s["covs"], T, max_d)
        got = orc.gicp_linearize(t["points"], t["covs"], s["points"], s["covs"], T, max_d, want_corr=True)
        assert got["num_inliers"] == ref["num_inliers"] > 1000
        assert (got["corr"] != ref["corr"]).mean() < 1e-3  # kd-tree vs exact search may differ on exact ties only
        np.testing.assert_allclose(got["H_ss"], ref["H_ss"], rtol=1e-9, atol=1e-9 * np.abs(ref["H_ss"]).max())
        np.testing.assert_allclose(got["b_s"], ref["b_s"], rtol=1e-9, atol=1e-9 * np.abs(ref["b_s"]).max())
        np.testing.assert_allclose(got["error"], ref["error"], rtol=1e-10)
        e, n = orc.gicp_error(t["points"], t["covs"], s["points"], s["covs"], T, max_d)
        assert n == got["num_inliers"] and abs(e - got["error"]) <= 1e-12 * abs(e)
    # binary blocks obey the adjoint identity J_t = -J_s Ad(delta^-1), like the VGICP factor
    from glim_amd import api

    iu = np.triu_indices(6)
    compact = np.concatenate([[got["num_inliers"], got["error"]], got["H_ss"][iu], got["b_s"]])
    ex = api.expand_compact(compact, T, api.FACTOR_BINARY)
    for k in ("H_tt", "H_ts", "b_t"):
        np.testing.assert_allclose(ex[k], got[k], rtol=1e-9, atol=1e-6 * np.abs(got[k]).max())


def test_oracle_gicp_finite_difference_gradient(orc, small_pair):
    """d/dxi sum r^T M r with M and the correspondences frozen == 2 b_s."""
    t, s = small_pair["target"], small_pair["source"]
    sp, sc = s["points"][::7], s["covs"][::7]
    T = small_pair["delta"] @ orc.se3_exp([0.004, -0.003, 0.002, 0.02, 0.01, -0.01])
    L = orc.gicp_linearize(t["points"], t["covs"], sp, sc, T, 0.7, want_corr=True)
    corr = L["corr"]
    R = T[:3, :3]

    def frozen_cost(Tp):
        q = sp @ Tp[:3, :3].T + Tp[:3, 3]
        c = 0.0
        for i in np.flatnonzero(corr >= 0):
            M = np.linalg.inv(t["covs"][corr[i]] + R @ sc[i] @ R.T)
            r = t["points"][corr[i]] - q[i]
            c += r @ M @ r
        return c

    g = np.zeros(6)
    eps = 1e-6
    for k in range(6):
        d = np.zeros(6)
        d[k] = eps
        g[k] = (frozen_cost(T @ orc.se3_exp(d)) - frozen_cost(T @ orc.se3_exp(-d))) / (2 * eps)
    np.testing.assert_allclose(g, 2.0 * L["b_s"], rtol=1e-5, atol=1e-6 * np.abs(L["b_s"]).max())


# ------------------------------------------------------------------------------------------------------------------


def clouds(orc, api, ctx, rings=48, az=512):
    from glim_amd import synth

    scene = synth.Scene.default()
    dirs = synth.lidar_directions(rings, az)
    poses = synth.arc_trajectory(2, step=0.6, yaw_step_deg=3.0)
    out = []
    for i, T in enumerate(poses):
        p = synth.scan(scene, T, dirs, frame_id=i).astype(np.float64)
        g = api.PointCloudGPU.clone(p, ctx=ctx)
        g.find_neighbors(10, download=False)
        g.estimate_covariances(10)
        _, c, _ = g.download(covs=True, normals=False)
        out.append((p, c.astype(np.float64), g))
    return out, np.linalg.inv(poses[0]) @ poses[1]


@pytest.mark.gpu
@pytest.mark.parametrize("max_d", [1.0, 0.5, 0.05])
def test_hip_gicp_matches_oracle(orc, max_d):
    """Correspondences bit-exact; H, b, error within the FP32-accumulation tolerance of the VGICP factor; GN step within 1e-4."""
    from glim_amd import api

    ctx = api.Context(0, 1)
    ((tp, tc, tg), (sp, sc, sg)), delta = clouds(orc, api, ctx)
    T = delta @ orc.se3_exp([0.004, -0.003, 0.002, 0.03, 0.02, -0.01])
    for binary in (False, True):
        f = api.IntegratedGICPFactor(0 if binary else np.eye(4), 1, tg, sg, max_correspondence_distance=max_d)
        values = {0: np.eye(4), 1: T}
        got = f.linearize(values)
        ref = orc.gicp_linearize(tp, tc, sp, sc, T, max_d, want_corr=True)
        np.testing.assert_array_equal(f.correspondences(values), ref["corr"])
        assert got["num_inliers"] == ref["num_inliers"] > 50
        scale = np.abs(ref["H_ss"]).max()
        np.testing.assert_allclose(got["error"], ref["error"], rtol=2e-4)
        np.testing.assert_allclose(got["H_ss"], ref["H_ss"], rtol=0, atol=2e-4 * scale)
        np.testing.assert_allclose(got["b_s"], ref["b_s"], rtol=0, atol=2e-4 * np.abs(ref["b_s"]).max() + 1e-6 * scale)
        if binary:
            for k in ("H_tt", "H_ts"):
                np.testing.assert_allclose(got[k], ref[k], rtol=0, atol=2e-4 * np.abs(ref[k]).max())
            np.testing.assert_allclose(got["b_t"], ref["b_t"], rtol=0, atol=2e-4 * np.abs(ref["b_t"]).max() + 1e-6 * scale)
        else:
            assert not np.any(got["H_tt"]) and not np.any(got["H_ts"]) and not np.any(got["b_t"])
        lam = 1e-6 * np.trace(ref["H_ss"]) / 6
        assert np.abs(gn_step(got, lam) - gn_step(ref, lam)).max() < POSE_TOL
        e = f.error(values)
        assert abs(e - ref["error"]) <= 2e-4 * ref["error"]
        assert f.inlier_fraction() == ref["num_inliers"] / len(sp)
        f.close()


@pytest.mark.gpu
def test_hip_gicp_alignment_converges_like_the_oracle(orc):
    """global_mapping.cpp:400-420: a few Gauss-Newton / LM iterations of the factor from a perturbed pose; every iterate within 1e-4."""
    from glim_amd import api

    ctx = api.Context(0, 1)
    ((tp, tc, tg), (sp, sc, sg)), delta = clouds(orc, api, ctx, rings=32, az=384)
    f = api.IntegratedGICPFactor(np.eye(4), 1, tg, sg, max_correspondence_distance=0.5)
    T = delta @ orc.se3_exp([0.01, -0.01, 0.01, 0.08, -0.05, 0.03])
    for it in range(6):
        got = f.linearize({1: T})
        ref = orc.gicp_linearize(tp, tc, sp, sc, T, 0.5)
        assert got["num_inliers"] == ref["num_inliers"]
        lam = 1e-6 * np.trace(ref["H_ss"]) / 6
        dg, dr = gn_step(got, lam), gn_step(ref, lam)
        assert np.abs(dg - dr).max() < POSE_TOL
        T = T @ orc.se3_exp(dr)
    err = np.linalg.inv(delta) @ T
    assert np.linalg.norm(err[:3, 3]) < 0.02 and np.abs(err[:3, :3] - np.eye(3)).max() < 2e-3


@pytest.mark.gpu
def test_hip_gicp_edge_cases(orc):
    from glim_amd import api

    ctx = api.Context(0, 1)
    ((tp, tc, tg), (sp, sc, sg)), delta = clouds(orc, api, ctx, rings=16, az=128)
    # far away: no correspondences -> zero information, never NaN
    far = delta.copy()
    far[0, 3] += 1e3
    f = api.IntegratedGICPFactor(np.eye(4), 1, tg, sg)
    L = f.linearize({1: far})
    assert L["num_inliers"] == 0 and L["error"] == 0.0 and not np.any(L["H_ss"]) and not np.any(L["b_s"])
    assert np.all(f.correspondences({1: far}) == -1)
    # a pre-built index shared by two factors (global_mapping_pose_graph.cpp:393 passes candidate.target->tree)
    g = api.IntegratedGICPFactor(np.eye(4), 1, tg, sg, target_tree=f.target_tree, max_correspondence_distance=0.25)
    assert 0 < g.linearize({1: delta})["num_inliers"] <= f.linearize({1: delta})["num_inliers"]
    # empty source / empty target
    empty = api.PointCloudGPU.clone(np.zeros((0, 3)), covs=np.zeros((0, 3, 3)), ctx=ctx)
    assert api.IntegratedGICPFactor(np.eye(4), 1, tg, empty).linearize({1: delta})["num_inliers"] == 0
    assert api.IntegratedGICPFactor(np.eye(4), 1, empty, sg).linearize({1: delta})["num_inliers"] == 0
    # a source without covariances is a state error, not a crash
    bare = api.PointCloudGPU.clone(sp, ctx=ctx)
    with pytest.raises(api.GlimAmdError):
        api.IntegratedGICPFactor(np.eye(4), 1, tg, bare).linearize({1: delta})
    # a radius far beyond what the index was sized for is refused (bounded ring walk), not searched incompletely
    g.set_max_correspondence_distance(100.0)
    with pytest.raises(api.GlimAmdError):
        g.linearize({1: delta})
    g.set_max_correspondence_distance(5.0)  # 5 x the hint the shared index was built with: fine, and still exact
    np.testing.assert_array_equal(g.correspondences({1: delta}), orc.gicp_linearize(tp, tc, sp, sc, delta, 5.0, want_corr=True)["corr"])
