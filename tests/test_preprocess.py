"""Scan preprocessing: oracle pins on CPU (independent numpy / scipy restatements of
src/glim/preprocess/cloud_preprocessor.cpp:92-188 and of the gtsam_points samplers it calls), HIP parity on the GPU."""
import numpy as np
import pytest


def raw_scan(seed=0, n=40000, nonfinite=True):
    """A LiDAR-like raw frame in the sensor frame: float32-representable points, firing times in [0, 0.1), intensities."""
    from glim_amd import synth

    scene = synth.Scene.default()
    dirs = synth.lidar_directions(32, n // 32)
    pose = synth.arc_trajectory(2)[0]
    pts = synth.scan(scene, pose, dirs, frame_id=seed).astype(np.float64)
    rng = np.random.default_rng(seed + 100)
    times = np.sort(rng.uniform(0.0, 0.1, len(pts)))
    perm = rng.permutation(len(pts))  # drivers do not always deliver in firing order
    pts, times = pts[perm], times[perm]
    inten = rng.uniform(0, 255, len(pts)).astype(np.float32).astype(np.float64)
    if nonfinite:
        pts[17] = [np.nan, 0.0, 1.0]
        pts[4242] = [1.0, np.inf, 1.0]
        pts[5] = [3.0e7, 0.0, 0.0]  # outside the 21-bit voxel range at 1 m
    return pts, times, inten


def np_keys(pts, res):
    t = pts * (1.0 / res)
    ok = np.all(np.isfinite(pts), axis=1) & np.all((t >= -1048576.0) & (t < 1048576.0), axis=1)
    c = np.floor(np.where(ok[:, None], t, 0.0)).astype(np.int64) + 1048576
    key = (c[:, 0] | (c[:, 1] << 21) | (c[:, 2] << 42)).astype(np.uint64)
    return key, ok


def test_sampling_key_and_hash(orc):
    import ctypes as C

    pts, _, _ = raw_scan()
    key, ok = np_keys(pts, 0.7)
    L = orc.lib()
    for i in list(range(0, len(pts), 997)) + [17, 4242, 5]:
        p4 = np.array([*pts[i], 1.0])
        got = L.orc_sampling_key(p4.ctypes.data_as(C.POINTER(C.c_double)), 1.0 / 0.7)
        assert got == (int(key[i]) if ok[i] else 0xFFFFFFFFFFFFFFFF)
    h = np.array([L.orc_sample_hash(7, i) for i in range(20000)], dtype=np.uint64)
    assert len(np.unique(h)) == len(h)
    bits = ((h[:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1)).mean(0)
    assert np.all(np.abs(bits - 0.5) < 0.02)  # every output bit is balanced


@pytest.mark.parametrize("res", [0.25, 1.0])
def test_voxelgrid_matches_numpy_restatement(orc, res):
    pts, times, inten = raw_scan()
    key, ok = np_keys(pts, res)
    # unsplit voxels: plain per-voxel means, ascending key
    op, ot, oi = orc.voxelgrid_sampling(pts, times, inten, res, block_size=0)
    uk, inv, cnt = np.unique(key[ok], return_inverse=True, return_counts=True)
    assert len(op) == len(uk)
    ref = np.stack([np.bincount(inv, weights=pts[ok][:, a]) / cnt for a in range(3)], 1)
    np.testing.assert_allclose(op, ref, rtol=0, atol=1e-9)
    np.testing.assert_allclose(ot, np.bincount(inv, weights=times[ok]) / cnt, rtol=0, atol=1e-12)
    np.testing.assert_allclose(oi, np.bincount(inv, weights=inten[ok]) / cnt, rtol=0, atol=1e-9)
    # 1024-entry blocks: a voxel that straddles a block boundary of the sorted order yields one point per block
    order = np.lexsort((np.arange(ok.sum()), key[ok]))
    sk = key[ok][order]
    heads = np.r_[True, sk[1:] != sk[:-1]] | (np.arange(len(sk)) % 1024 == 0)
    op2, ot2, _ = orc.voxelgrid_sampling(pts, times, inten, res, block_size=1024)
    assert len(op2) == heads.sum() > len(uk)
    seg = np.cumsum(heads) - 1
    ref2 = np.stack([np.bincount(seg, weights=pts[ok][order][:, a]) / np.bincount(seg) for a in range(3)], 1)
    np.testing.assert_allclose(op2, ref2, rtol=0, atol=1e-9)
    # exact sequential sums in (key, index) order -- the rule the HIP path reproduces bit for bit
    for s in (0, len(uk) // 2, heads.sum() - 1):
        members = order[seg == s]
        acc = np.zeros(3)
        for j in members:
            acc = acc + pts[ok][j]
        np.testing.assert_array_equal(op2[s], acc / float(len(members)))


def test_randomgrid_properties(orc):
    pts, _, _ = raw_scan(nonfinite=True)
    n = len(pts)
    key, ok = np_keys(pts, 1.0)
    num_voxels = len(np.unique(key[ok]))
    # (a) no cap: every voxel keeps min(count, ppv) points, indices ascending and unique, invalid points never selected
    rate = 0.5
    ppv = int(np.ceil(rate * n / num_voxels))
    idx = orc.randomgrid_sampling(pts, 1.0, rate, seed=3)
    assert np.all(np.diff(idx) > 0) and np.all(ok[idx])
    uk, cnt = np.unique(key[ok], return_counts=True)
    sk, scnt = np.unique(key[idx], return_counts=True)
    assert np.array_equal(uk, sk)
    expected = np.minimum(cnt, ppv)
    if expected.sum() <= int(n * rate * 1.2):
        assert np.array_equal(scnt, expected)
    # (b) the cap: exactly floor(1.2 rate n) survivors
    rate = 10000 / n
    idx = orc.randomgrid_sampling(pts, 0.25, rate, seed=3)
    nv = len(np.unique(np_keys(pts, 0.25)[0][ok]))
    assert nv > 0.2 * rate * n  # so that ceil() overshoots and the cap engages
    assert len(idx) == int(n * rate * 1.2)
    # (c) rate >= 0.99: unchanged
    assert np.array_equal(orc.randomgrid_sampling(pts, 1.0, 0.995), np.arange(n))
    # (d) seeds give different, equally sized samples; inside one big voxel every point is equally likely
    a, b = orc.randomgrid_sampling(pts, 1.0, 0.1, seed=1), orc.randomgrid_sampling(pts, 1.0, 0.1, seed=2)
    assert len(a) == len(b) and not np.array_equal(a, b)
    big = uk[np.argmax(cnt)]
    members = np.flatnonzero(ok & (key == big))
    hits = np.zeros(n)
    trials = 300
    for s in range(trials):
        hits[orc.randomgrid_sampling(pts, 1.0, 0.1, seed=1000 + s)] += 1
    p = hits[members] / trials
    expect = min(1.0, int(np.ceil(0.1 * n / num_voxels)) / len(members))
    assert abs(p.mean() - expect) < 0.02 and p.std() < 3.5 * np.sqrt(expect * (1 - expect) / trials) + 0.01


def test_find_inliers_matches_scipy(orc):
    from scipy.spatial import cKDTree

    rng = np.random.default_rng(5)
    pts = np.r_[rng.normal(size=(3000, 3)) * [5, 5, 0.2], rng.uniform(-40, 40, size=(60, 3))]
    d, _ = cKDTree(pts).query(pts, k=8)
    md = d.mean(1)
    thresh = md.mean() + 1.5 * np.sqrt((md**2).mean() - md.mean() ** 2)
    ref = np.flatnonzero(md < thresh)
    got = orc.find_inliers(pts, 8, 1.5)
    assert np.array_equal(got, ref)
    assert 0 < len(pts) - len(got) < 200


def np_preprocess(orc, pts, times, inten, prm):
    """Independent numpy restatement of preprocess_impl on top of the oracle samplers."""
    n = len(pts)
    if prm.use_random_grid_downsampling:
        rate = prm.downsample_target / n if prm.downsample_target > 0 else prm.downsample_rate
        idx = orc.randomgrid_sampling(pts, prm.downsample_resolution, rate, prm.seed)
        P, T, I = pts[idx], times[idx], inten[idx]
    else:
        P, T, I = orc.voxelgrid_sampling(pts, times, inten, prm.downsample_resolution, prm.voxelgrid_block_size)
    d2 = (P[:, 0] * P[:, 0] + P[:, 2] * P[:, 2]) + P[:, 1] * P[:, 1]
    keep = np.all(np.isfinite(P), axis=1) & (d2 > prm.distance_near_thresh**2) & (d2 < prm.distance_far_thresh**2)
    if prm.enable_cropbox_filter:
        Q = P
        if prm.crop_bbox_frame_imu:
            T12 = np.array(prm.T_imu_lidar[:]).reshape(3, 4)
            Q = P @ T12[:, :3].T + T12[:, 3]
        lo, hi = np.array(prm.crop_bbox_min[:]), np.array(prm.crop_bbox_max[:])
        keep &= ~(np.all(Q >= lo, axis=1) & np.all(Q <= hi, axis=1))
    sel = np.flatnonzero(keep)
    sel = sel[np.lexsort((sel, T[sel]))]
    P, T, I = P[sel], T[sel], I[sel]
    if prm.global_shutter:
        T = np.zeros_like(T)
    if prm.enable_outlier_removal:
        k = orc.find_inliers(P, prm.outlier_removal_k, prm.outlier_std_mul_factor)
        P, T, I = P[k], T[k], I[k]
    return P, T, I


@pytest.mark.parametrize(
    "kw",
    [
        dict(),  # shipped config_preprocess.json
        dict(use_random_grid_downsampling=0, downsample_resolution=0.5),
        dict(downsample_target=0, downsample_rate=0.3, downsample_resolution=0.5, distance_near_thresh=2.0, distance_far_thresh=25.0),
        dict(enable_cropbox_filter=1, crop_bbox_min=(-6.0, -4.0, -3.0), crop_bbox_max=(9.0, 4.0, 3.0), global_shutter=1),
        dict(enable_cropbox_filter=1, crop_bbox_frame_imu=1, crop_bbox_min=(-6.0, -4.0, -3.0), crop_bbox_max=(9.0, 4.0, 3.0)),
        dict(enable_outlier_removal=1, outlier_removal_k=8, outlier_std_mul_factor=1.0, use_random_grid_downsampling=0, downsample_resolution=0.4),
    ],
)
def test_oracle_preprocess_matches_numpy_restatement(orc, kw):
    pts, times, inten = raw_scan(seed=1)
    if kw.get("crop_bbox_frame_imu"):
        kw = dict(kw, T_imu_lidar=orc.se3_exp([0.02, -0.01, 0.5, 0.3, -0.2, 0.1]))
    prm = orc.preprocess_params(seed=11, **kw)
    out = orc.preprocess(pts, times, inten, prm)
    P, T, I = np_preprocess(orc, pts, times, inten, prm)
    assert 100 < len(P) < len(pts)
    np.testing.assert_array_equal(out["points"], P)
    np.testing.assert_array_equal(out["times"], T)
    np.testing.assert_array_equal(out["intensities"], I)
    assert np.all(np.diff(out["times"]) >= 0)
    np.testing.assert_array_equal(out["neighbors"], orc.knn(P, prm.k_correspondences))
    if not kw:  # shipped config: about 10 000 points survive
        assert 7000 < len(P) <= 12000


def test_oracle_preprocess_empty_and_tiny(orc):
    prm = orc.preprocess_params()
    out = orc.preprocess(np.zeros((0, 3)), np.zeros(0), None, prm)
    assert out["points"].shape == (0, 3) and out["neighbors"].shape == (0, 10)
    pts = np.array([[1.0, 2.0, 0.5], [np.nan, 0, 0], [0.1, 0.1, 0.1], [3.0, -2.0, 0.2]])
    out = orc.preprocess(pts, np.array([0.03, 0.01, 0.0, 0.02]), None, orc.preprocess_params(downsample_target=0, downsample_rate=1.0))
    np.testing.assert_array_equal(out["points"], pts[[3, 0]])  # NaN and too-near points dropped, the rest sorted by time
    np.testing.assert_array_equal(out["neighbors"][:, :2], [[0, 1], [1, 0]])
