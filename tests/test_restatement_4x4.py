"""A SECOND CPU restatement of SURVEY rows a4 / a6, structured differently from oracle/vgicp_oracle.c on purpose ("one author, one
file"): vectorised NumPy, voxels in a Python dict keyed by the integer coordinate, and the algebra written with the explicit 4 x 4 homogeneous
matrices SURVEY Appendix B.5 / 8a row a6 describe for upstream's CPU factor --

    M_i = (C_B + delta C_A delta^T) as 4 x 4 with element (3, 3) set to 1, inverted, (3, 3) set back to 0;
    J_t = [ -hat(q) | I ; 0 ],  J_s = delta [ hat(p) | -I ; 0 ]   (4 x 6),   r = mu_B - q  (homogeneous, w = 0)

-- instead of the C oracle's 3 x 3 blocks, rotation-factored Jacobians and per-thread accumulators.  Nothing here calls into libvgicp_oracle except
to be COMPARED with it: on a configs[1]-sized pair (131 072-pt spinning-LiDAR scans, 0.5 m voxels) and on the edge cases the GPU tests use
(tests/test_gpu_edge_cases.py: points on voxel faces / signed zeros, one voxel of identical points, a map 100 km from the origin).
This does not pin the oracle to gtsam_points (only reference-held vectors could: DESIGN.md 2) -- it removes the single point of failure."""
import numpy as np
import pytest


def hat_rows(v):
    """n x 3 -> n x 3 x 3 skew matrices, hat(a) b = a x b."""
    o = np.zeros((len(v), 3, 3))
    o[:, 0, 1], o[:, 0, 2] = -v[:, 2], v[:, 1]
    o[:, 1, 0], o[:, 1, 2] = v[:, 2], -v[:, 0]
    o[:, 2, 0], o[:, 2, 1] = -v[:, 1], v[:, 0]
    return o


def coords_of(p3, res):
    """fast_floor(p * (1 / r)) per axis: the same f64 expression on the same operands; floor == fast_floor for in-range values."""
    return np.floor(p3 * (1.0 / res)).astype(np.int64)


def voxel_dict(points, covs33, res):
    """GaussianVoxelMapCPU::insert as a dict {(cx, cy, cz): (n, mean4, cov4x4)}: homogeneous sums, divided by n at the end (B.4)."""
    p = np.asarray(points, dtype=np.float64)[:, :3]
    n = len(p)
    p4 = np.concatenate([p, np.ones((n, 1))], axis=1)
    c4 = np.zeros((n, 4, 4))
    c4[:, :3, :3] = np.asarray(covs33, dtype=np.float64)[:, :3, :3]
    c = coords_of(p, res)
    uniq, inv = np.unique(c, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    cnt = np.bincount(inv, minlength=len(uniq))
    msum = np.zeros((len(uniq), 4))
    csum = np.zeros((len(uniq), 4, 4))
    np.add.at(msum, inv, p4)
    np.add.at(csum, inv, c4)
    return {tuple(int(x) for x in u): (int(k), m / k, S / k) for u, k, m, S in zip(uniq, cnt, msum, csum)}


def linearize_4x4(vox, res, points, covs33, delta):
    p = np.asarray(points, dtype=np.float64)[:, :3]
    n = len(p)
    D = np.asarray(delta, dtype=np.float64).reshape(4, 4)
    p4 = np.concatenate([p, np.ones((n, 1))], axis=1)
    q4 = p4 @ D.T
    c = coords_of(q4[:, :3], res)
    keys = [tuple(int(x) for x in row) for row in c]
    hit = np.array([k in vox for k in keys])
    idx = np.nonzero(hit)[0]
    out = {"coords": c, "hit": hit, "num_inliers": int(hit.sum())}
    Z6, z6 = np.zeros((6, 6)), np.zeros(6)
    if len(idx) == 0:
        out.update(H_tt=Z6, H_ss=Z6.copy(), H_ts=Z6.copy(), b_t=z6, b_s=z6.copy(), error=0.0)
        return out
    muB = np.stack([vox[keys[i]][1] for i in idx])
    CB = np.stack([vox[keys[i]][2] for i in idx])
    CA = np.zeros((len(idx), 4, 4))
    CA[:, :3, :3] = np.asarray(covs33, dtype=np.float64)[idx][:, :3, :3]
    S = CB + D @ CA @ D.T  # (broadcast: delta C_A delta^T, the last row / column stay zero)
    S[:, 3, 3] = 1.0
    M = np.linalg.inv(S)
    M[:, 3, 3] = 0.0
    r = muB - q4[idx]  # w component: 1 - 1 = 0
    Jt = np.zeros((len(idx), 4, 6))
    Jt[:, :3, :3] = -hat_rows(q4[idx, :3])
    Jt[:, :3, 3:] = np.eye(3)
    Js0 = np.zeros((len(idx), 4, 6))
    Js0[:, :3, :3] = hat_rows(p[idx])
    Js0[:, :3, 3:] = -np.eye(3)
    Dl = D.copy()
    Dl[:3, 3] = 0.0  # the Jacobian is a map of DIRECTIONS (w = 0 columns): only the linear part of delta acts
    Js = Dl @ Js0
    MJt, MJs, Mr = M @ Jt, M @ Js, np.einsum("nij,nj->ni", M, r)
    out["H_tt"] = np.einsum("nki,nkj->ij", Jt, MJt)
    out["H_ss"] = np.einsum("nki,nkj->ij", Js, MJs)
    out["H_ts"] = np.einsum("nki,nkj->ij", Jt, MJs)
    out["b_t"] = np.einsum("nki,nk->i", Jt, Mr)
    out["b_s"] = np.einsum("nki,nk->i", Js, Mr)
    out["error"] = float(np.einsum("ni,ni->", r, Mr))
    # the same sums over ABSOLUTE values: what the additions were made of.  A block that cancels (b of a cloud matched against itself) is known to
    # 1e-16 of THIS, not of its own size -- and the C oracle adds its per-thread sums in an order that changes from run to run (guided schedule)
    aJt, aJs, aMJt, aMJs, aMr = np.abs(Jt), np.abs(Js), np.abs(MJt), np.abs(MJs), np.abs(Mr)
    out["abs"] = {"H_tt": np.einsum("nki,nkj->ij", aJt, aMJt).max(), "H_ss": np.einsum("nki,nkj->ij", aJs, aMJs).max(), "H_ts": np.einsum("nki,nkj->ij", aJt, aMJs).max(),
                  "b_t": np.einsum("nki,nk->i", aJt, aMr).max(), "b_s": np.einsum("nki,nk->i", aJs, aMr).max()}
    return out


def compare(orc, points_t, covs_t, points_s, covs_s, res, delta, rtol=1e-9):
    vox = voxel_dict(points_t, covs_t, res)
    ref_map = orc.VoxelMap(res).insert(points_t, covs_t)
    rc, rn, rm, rC = ref_map.voxels()
    assert len(vox) == ref_map.num_voxels() == len(rc)
    for c, n, m, C in zip(rc, rn, rm, rC):
        k, m4, C4 = vox[tuple(int(x) for x in c)]
        assert k == n
        np.testing.assert_allclose(m4[:3], m[:3], rtol=1e-13, atol=1e-13 * (1.0 + np.abs(m[:3]).max()))
        np.testing.assert_allclose(C4[:3, :3], C[:3, :3], rtol=1e-12, atol=1e-15)
    got = linearize_4x4(vox, res, points_s, covs_s, delta)
    ref = orc.vgicp_linearize(ref_map, points_s, covs_s, delta, want_corr=True)
    np.testing.assert_array_equal(got["coords"], ref["corr"][:, :3])  # bit-exact voxel coordinates ...
    np.testing.assert_array_equal(got["hit"], ref["corr"][:, 3] >= 0)  # ... and correspondences
    assert got["num_inliers"] == ref["num_inliers"]
    for k in ("H_tt", "H_ss", "H_ts", "b_t", "b_s"):
        scale = max(1e-300, np.abs(ref[k]).max())
        assert np.abs(got[k] - ref[k]).max() <= rtol * scale + 1e-13 * got["abs"][k], (k, np.abs(got[k] - ref[k]).max() / scale)
    assert got["error"] == pytest.approx(ref["error"], rel=rtol, abs=1e-12)
    return got, ref


def test_config1_sized_pair_agrees_with_the_c_oracle(orc):
    """128 rings x 1024 azimuths = 131 072 points per scan, 0.5 m voxels, target 0.5 m / 2 deg away (the M1 workload)."""
    from glim_amd import synth

    scene = synth.Scene.default()
    dirs = synth.lidar_directions(128, 1024)
    poses = synth.arc_trajectory(2)
    tgt, src = synth.scan(scene, poses[0], dirs, 0), synth.scan(scene, poses[1], dirs, 1)
    assert len(src) == 131072
    _, ct = orc.covariances(tgt, orc.knn(tgt, 10))
    _, cs = orc.covariances(src, orc.knn(src, 10))
    ct, cs = ct.astype(np.float32).astype(np.float64), cs.astype(np.float32).astype(np.float64)
    delta = synth.relative_pose(poses[0], poses[1]) @ orc.se3_exp([0.004, -0.003, 0.002, 0.03, -0.02, 0.01])
    got, ref = compare(orc, tgt, ct, src, cs, 0.5, delta)
    assert got["num_inliers"] > 60000
    step = np.abs(np.linalg.solve(got["H_ss"], -got["b_s"]) - np.linalg.solve(ref["H_ss"], -ref["b_s"])).max()
    assert step < 1e-10


def test_points_on_voxel_faces_and_signed_zero(orc):
    res = 0.5
    base = np.array([0.0, -0.0, 0.5, -0.5, 1.0, -1.0, 1.5, 2.0, -2.0, 3.0, 1e-30, -1e-30, 0.49999997, 0.50000006, -0.49999997, -0.50000006], dtype=np.float32)
    ulps = np.concatenate([base, np.nextafter(base, np.float32(np.inf)), np.nextafter(base, np.float32(-np.inf))])
    rng = np.random.default_rng(11)
    pts = np.stack([rng.permutation(ulps), rng.permutation(ulps), rng.permutation(ulps)], axis=1).astype(np.float32)
    covs = np.tile(np.eye(3), (len(pts), 1, 1))
    for shift in ([0, 0, 0], [0.5, -0.5, 1.0], [0.25, 0.25, 0.25], [-1e-7, 1e-7, 0.0]):
        T = np.eye(4)
        T[:3, 3] = shift
        compare(orc, pts, covs, pts, covs, res, T)


def test_duplicate_points_and_single_voxel(orc):
    p = np.tile(np.array([[1.3, -2.2, 0.7]], dtype=np.float32), (500, 1))
    covs = np.tile(np.diag([0.3, 0.2, 0.1]), (500, 1, 1))
    got, _ = compare(orc, p, covs, p, covs, 1.0, np.eye(4))
    assert got["num_inliers"] == 500 and got["error"] < 1e-20


def test_far_from_origin(orc, small_pair):
    t, s = small_pair["target"], small_pair["source"]
    off = np.array([1.0e5, -7.5e4, 2.0e3])
    tp = (t["points"].astype(np.float64) + off).astype(np.float32)
    T = small_pair["delta"].copy()
    T[:3, 3] += off
    got, _ = compare(orc, tp, t["covs"], s["points"], s["covs"], 0.5, T, rtol=1e-7)  # (H_tt carries |q|^2 ~ 1e10: conditioning, not arithmetic)
    assert got["num_inliers"] > 100


def test_binary_blocks_follow_from_the_source_block_by_the_adjoint(orc, small_pair):
    """What the device path relies on: J_t = -J_s Ad(delta^-1), hence H_tt, H_ts, b_t from the 6 x 6 source block -- checked on THIS
    restatement's explicitly accumulated target blocks (the C ABI's glim_amd_expand_compact is checked against the oracle in tests/test_abi_cpu.py)."""
    t, s = small_pair["target"], small_pair["source"]
    T = small_pair["delta"] @ orc.se3_exp([0.01, -0.02, 0.005, 0.05, 0.02, -0.01])
    got = linearize_4x4(voxel_dict(t["points"], t["covs"], 0.5), 0.5, s["points"], s["covs"], T)
    R, tt = T[:3, :3], T[:3, 3]
    Ad = np.zeros((6, 6))  # Adjoint(delta^-1) in [omega; v] order
    Ad[:3, :3] = R.T
    Ad[3:, 3:] = R.T
    Ad[3:, :3] = -R.T @ np.array([[0, -tt[2], tt[1]], [tt[2], 0, -tt[0]], [-tt[1], tt[0], 0.0]])
    np.testing.assert_allclose(got["H_tt"], Ad.T @ got["H_ss"] @ Ad, rtol=1e-9, atol=1e-9 * np.abs(got["H_tt"]).max())
    np.testing.assert_allclose(got["H_ts"], -Ad.T @ got["H_ss"], rtol=1e-9, atol=1e-9 * np.abs(got["H_ts"]).max())
    np.testing.assert_allclose(got["b_t"], -Ad.T @ got["b_s"], rtol=1e-9, atol=1e-9 * np.abs(got["b_t"]).max())
