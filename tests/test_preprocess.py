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


# ------------------------------------------------------------------------------------------------------------------
# HIP parity (GPU box)
# ------------------------------------------------------------------------------------------------------------------


@pytest.mark.gpu
@pytest.mark.parametrize("n,bits", [(0, 64), (1, 64), (777, 13), (2048, 8), (2049, 21), (200000, 64), (200000, 37), (50000, 0), (600001, 24)])  # last: > 256 blocks, the separate-offsets path
def test_hip_radix_sort_is_stable_and_exact(n, bits):
    from glim_amd import api

    ctx = api.Context(0, 1)
    rng = np.random.default_rng(n + bits)
    keys = rng.integers(0, 2**63, size=n, dtype=np.uint64) if n else np.zeros(0, dtype=np.uint64)
    if n > 10:
        keys[::7] = keys[3]  # many duplicates: stability is visible in the value order
    ko, vo = api.debug_sort_pairs(keys, None, bits, ctx=ctx)
    mask = np.uint64((1 << bits) - 1) if bits < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    order = np.argsort(keys & mask, kind="stable")
    np.testing.assert_array_equal(vo, order.astype(np.uint32))
    np.testing.assert_array_equal(ko, keys[order])
    if n:  # explicit values travel with their keys
        vals = rng.permutation(n).astype(np.uint32)
        ko2, vo2 = api.debug_sort_pairs(keys, vals, bits, ctx=ctx)
        np.testing.assert_array_equal(vo2, vals[order])


HIP_CASES = [
    dict(),  # shipped config_preprocess.json: random grid, 1 m, target 10 000 (the 1.2x cap engages)
    dict(downsample_target=0, downsample_rate=0.3, downsample_resolution=0.5, distance_near_thresh=2.0, distance_far_thresh=25.0),
    dict(downsample_target=0, downsample_rate=0.995),  # sampler passes the cloud through
    dict(use_random_grid_downsampling=0, downsample_resolution=0.5),
    dict(use_random_grid_downsampling=0, downsample_resolution=1.0, voxelgrid_block_size=0, k_correspondences=5),
    dict(use_random_grid_downsampling=0, downsample_resolution=0.02),  # 40 key bits
    dict(enable_cropbox_filter=1, crop_bbox_min=(-6.0, -4.0, -3.0), crop_bbox_max=(9.0, 4.0, 3.0), global_shutter=1),
    dict(enable_cropbox_filter=1, crop_bbox_frame_imu=1, crop_bbox_min=(-6.0, -4.0, -3.0), crop_bbox_max=(9.0, 4.0, 3.0)),
    dict(enable_outlier_removal=1, outlier_removal_k=8, outlier_std_mul_factor=1.0, downsample_target=20000),
    # 5 cm voxels: (nearly) every point is alone in its voxel and survives the selection -> far more candidates for the cap than the counting rank
    # of the fast path accepts: it raises its flag and the call is repeated on the sorting path
    dict(downsample_target=0, downsample_rate=0.2, downsample_resolution=0.05),
]


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["fast", "sorting"])
@pytest.mark.parametrize("case", range(len(HIP_CASES)))
def test_hip_preprocess_matches_oracle(orc, case, path):
    """Bit-exact: surviving points (FP64), times, intensities, their order, and the neighbour lists -- on the random-grid fast path (one sort +
    counting ranks, preprocess.hip) and on the general sorting path (pp_fast=0; also what the fast path falls back to)."""
    from glim_amd import api

    kw = dict(HIP_CASES[case])
    if kw.get("crop_bbox_frame_imu"):
        kw["T_imu_lidar"] = orc.se3_exp([0.02, -0.01, 0.5, 0.3, -0.2, 0.1])
    ctx = api.Context(0, 1)
    ctx.set_diag("" if path == "fast" else "pp_fast=0")
    pts, times, inten = raw_scan(seed=2, n=131072)
    ref = orc.preprocess(pts, times, inten, orc.preprocess_params(seed=5, **kw))
    g = api.PointCloudGPU.preprocess(pts, times, inten, api.preprocess_params(seed=5, **kw), ctx=ctx)
    got = g.download_frame()
    assert g.size() == len(ref["points"]) > 100
    np.testing.assert_array_equal(got["points"], ref["points"])
    np.testing.assert_array_equal(got["times"], ref["times"])
    np.testing.assert_array_equal(got["intensities"], ref["intensities"])
    xyz, _, _ = g.download(covs=False, normals=False)
    np.testing.assert_array_equal(xyz, ref["points"].astype(np.float32))
    # kNN runs on the FP32 image of the cloud (what the factor path consumes); identical to the oracle whenever the surviving
    # points are FP32-representable (samplers that select points); for averaged points compare against the oracle on that image
    nb_ref = ref["neighbors"] if kw.get("use_random_grid_downsampling", 1) else orc.knn(xyz.astype(np.float64), got["k_neighbors"])
    np.testing.assert_array_equal(got["neighbors"], nb_ref)


@pytest.mark.gpu
def test_hip_preprocess_edge_cases(orc):
    from glim_amd import api

    ctx = api.Context(0, 1)
    g = api.PointCloudGPU.preprocess(np.zeros((0, 3)), np.zeros(0), None, ctx=ctx)
    assert g.size() == 0 and g.download_frame()["points"].shape == (0, 3)
    # every point invalid / filtered out
    bad = np.full((500, 3), np.nan)
    assert api.PointCloudGPU.preprocess(bad, np.zeros(500), None, ctx=ctx).size() == 0
    near = np.random.default_rng(0).normal(size=(500, 3)) * 0.05
    for mode in (0, 1):
        assert api.PointCloudGPU.preprocess(near, np.zeros(500), None, api.preprocess_params(use_random_grid_downsampling=mode), ctx=ctx).size() == 0
    # a single surviving point; no intensities
    one = np.array([[4.0, 1.0, 0.5]])
    g = api.PointCloudGPU.preprocess(one, np.array([0.01]), None, ctx=ctx)
    fr = g.download_frame()
    np.testing.assert_array_equal(fr["points"], one)
    assert fr["intensities"] is None and fr["neighbors"].tolist() == [[0] * 10]
    # all points in one voxel with identical time stamps: order = original index
    rng = np.random.default_rng(1)
    blob = (rng.uniform(0.1, 0.9, size=(3000, 3)) + [5.0, 0.0, 0.0]).astype(np.float32).astype(np.float64)
    prm = dict(downsample_target=0, downsample_rate=0.5, seed=9)
    ref = orc.preprocess(blob, np.zeros(3000), None, orc.preprocess_params(**prm))
    got = api.PointCloudGPU.preprocess(blob, np.zeros(3000), None, api.preprocess_params(**prm), ctx=ctx).download_frame()
    np.testing.assert_array_equal(got["points"], ref["points"])
    assert len(ref["points"]) == 1500
    # invalid arguments
    with pytest.raises(api.GlimAmdError):
        api.PointCloudGPU.preprocess(blob, np.zeros(3000), None, api.preprocess_params(downsample_resolution=0.0), ctx=ctx)


@pytest.mark.gpu
def test_hip_preprocess_then_deskew_then_covariances(orc):
    """The reference order (odometry_estimation_imu.cpp:313-320): preprocess -> deskew -> covariances from the RAW scan's neighbours."""
    from glim_amd import api

    ctx = api.Context(0, 1)
    pts, times, inten = raw_scan(seed=3, n=131072, nonfinite=False)
    prm = dict(downsample_target=30000, seed=1)
    ref = orc.preprocess(pts, times, inten, orc.preprocess_params(**prm))
    Til = orc.se3_exp([0.05, 0.02, -0.1, 0.2, -0.1, 0.05])
    lv, av = [4.0, -2.0, 0.3], [0.1, -0.2, 1.5]
    ref_desk = orc.deskew(ref["points"], ref["times"], Til, linear_vel=lv, angular_vel=av)
    pre = api.PointCloudGPU.preprocess(pts, times, inten, api.preprocess_params(**prm), ctx=ctx)
    desk = pre.deskew(Til, linear_vel=lv, angular_vel=av)
    xyz, _, _ = desk.download(covs=False, normals=False)
    ref32 = ref_desk.astype(np.float32)
    assert np.all(np.abs(xyz.astype(np.float64) - ref_desk) <= np.spacing(np.abs(ref32)).astype(np.float64))
    desk.estimate_covariances(10)
    _, covs, _ = desk.download(covs=True, normals=False)
    _, ref_covs = orc.covariances(xyz.astype(np.float64), ref["neighbors"])
    # (ill-conditioned neighbourhoods flip the smallest eigenvector: tests/test_gpu_parity.py masks them the same way)
    assert (np.abs(covs - ref_covs).max(axis=(1, 2)) < 1e-4).mean() > 0.9
    # a cloud that did not come from preprocess() cannot be deskewed this way
    with pytest.raises(api.GlimAmdError):
        api.PointCloudGPU.clone(pts[:100], ctx=ctx).deskew(Til)
