"""Deskewing: oracle pins on CPU (analytic cases + an independent scipy restatement of
src/glim/common/cloud_deskewing.cpp), HIP parity on the GPU."""
import numpy as np
import pytest


def make_case(seed=0, n=5000):
    rng = np.random.default_rng(seed)
    pts = (rng.normal(size=(n, 3)) * [15, 10, 2]).astype(np.float32).astype(np.float64)
    times = np.sort(rng.uniform(0.0, 0.1, n))
    times[: n // 50] = 0.0  # many points share the first time stamp
    return pts, times


def scipy_deskew_imu(orc, pts, times, Til, imu_times, imu_poses, stamp):
    from scipy.spatial.transform import Rotation, Slerp

    table, idx = [], []
    for t in times:  # cloud_deskewing.cpp:79-84
        if not table or t - table[-1] > 1e-4:
            table.append(t)
        idx.append(len(table) - 1)
    Tli = np.linalg.inv(Til)
    TT, cur, T0w = [], 0, None
    for i, tt in enumerate(table):
        time = stamp + tt
        while cur < len(imu_times) - 1 and imu_times[cur + 1] < time:
            cur += 1
        if i == 0:
            T0w = np.linalg.inv(imu_poses[cur])
        if cur + 1 >= len(imu_times):
            Tw1 = imu_poses[cur]
        else:
            p = max(0.0, min(1.0, (time - imu_times[cur]) / (imu_times[cur + 1] - imu_times[cur])))
            R = Slerp([0, 1], Rotation.from_matrix([imu_poses[cur][:3, :3], imu_poses[cur + 1][:3, :3]]))(p).as_matrix()
            Tw1 = np.eye(4)
            Tw1[:3, :3] = R
            Tw1[:3, 3] = (1 - p) * imu_poses[cur][:3, 3] + p * imu_poses[cur + 1][:3, 3]
        TT.append(Tli @ T0w @ Tw1 @ Til)
    return np.array([(TT[j] @ np.append(p, 1.0))[:3] for p, j in zip(pts, idx)]), len(table)


def imu_track(orc, seed=1):
    rng = np.random.default_rng(seed)
    imu_times = 100.0 + np.array([-0.02, 0.013, 0.031, 0.058, 0.09, 0.13])
    poses = [orc.se3_exp(rng.normal(size=6) * [0.3, 0.3, 0.3, 2, 2, 2])]
    for _ in imu_times[1:]:
        poses.append(poses[-1] @ orc.se3_exp(rng.normal(size=6) * [0.02, 0.02, 0.05, 0.05, 0.05, 0.02]))
    return imu_times, poses


def test_oracle_zero_velocity_is_identity(orc):
    pts, times = make_case()
    Til = orc.se3_exp([0.1, -0.2, 0.05, 0.3, 0.1, -0.2])
    np.testing.assert_allclose(orc.deskew(pts, times, Til), pts, atol=1e-12)
    # no IMU poses -> the IMU form falls back to zero velocity (cloud_deskewing.cpp:66-68)
    np.testing.assert_allclose(orc.deskew(pts, times, Til, imu_times=[], imu_poses=[]), pts, atol=1e-12)


def test_oracle_constant_velocity_closed_form(orc):
    """Pure translation velocity v in the IMU frame, identity extrinsic: T_lidar0_lidar1 = Exp(dt v)^-1 => p - dt v."""
    pts, times = make_case()
    v = np.array([3.0, -1.0, 0.5])
    out = orc.deskew(pts, times, np.eye(4), linear_vel=v, angular_vel=[0, 0, 0])
    # time is quantised to the 0.1 ms table
    table, q = [], []
    for t in times:
        if not table or t - table[-1] > 1e-4:
            table.append(t)
        q.append(table[-1])
    np.testing.assert_allclose(out, pts - np.array(q)[:, None] * v[None, :], atol=1e-12)
    # pure rotation about z: points rotate by -w dt
    w = np.array([0.0, 0.0, 2.0])
    out = orc.deskew(pts, times, np.eye(4), linear_vel=[0, 0, 0], angular_vel=w)
    ang = -np.array(q) * 2.0
    ref = np.stack([np.cos(ang) * pts[:, 0] - np.sin(ang) * pts[:, 1], np.sin(ang) * pts[:, 0] + np.cos(ang) * pts[:, 1], pts[:, 2]], 1)
    np.testing.assert_allclose(out, ref, atol=1e-11)


def test_oracle_imu_form_matches_scipy_restatement(orc):
    pts, times = make_case(3)
    Til = orc.se3_exp([0.05, 0.02, -0.1, 0.2, -0.1, 0.05])
    imu_times, poses = imu_track(orc)
    out = orc.deskew(pts, times, Til, imu_times=imu_times, imu_poses=poses, stamp=100.0)
    ref, table_size = scipy_deskew_imu(orc, pts, times, Til, imu_times, poses, 100.0)
    assert 300 < table_size < 1100
    np.testing.assert_allclose(out, ref, atol=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["imu", "constvel", "zero"])
def test_hip_deskew_matches_oracle(orc, form):
    from glim_amd import api

    ctx = api.Context(0, 1)
    pts, times = make_case(5, n=131072)
    Til = orc.se3_exp([0.05, 0.02, -0.1, 0.2, -0.1, 0.05])
    if form == "imu":
        imu_times, poses = imu_track(orc)
        kw = dict(imu_times=imu_times, imu_poses=poses, stamp=100.0)
    elif form == "constvel":
        kw = dict(linear_vel=[4.0, -2.0, 0.3], angular_vel=[0.1, -0.2, 1.5])
    else:
        kw = {}
    ref = orc.deskew(pts, times, Til, **kw)
    g = api.PointCloudGPU.clone_deskewed(pts, times, Til, ctx=ctx, **kw)
    assert g.size() == len(pts)
    xyz, _, _ = g.download(covs=False, normals=False)
    # FP64 transform on the device, stored as FP32: within one FP32 ulp of the oracle's FP64 result
    ref32 = ref.astype(np.float32)
    ulp = np.spacing(np.abs(ref32))
    assert np.all(np.abs(xyz.astype(np.float64) - ref) <= ulp.astype(np.float64))
    assert (xyz == ref32).mean() > 0.9999  # separate roundings on both sides (no FMA contraction): equal bar host-side table differences
    if form == "zero":
        np.testing.assert_array_equal(xyz, pts.astype(np.float32))
    # the deskewed cloud feeds the rest of the path: kNN + covariances + voxel map + factor run on it
    g.find_neighbors(10, download=False)
    g.estimate_covariances(10)
    assert api.GaussianVoxelMapGPU(0.5, ctx=ctx).insert(g).voxelmap_info()["num_voxels"] > 100
    empty = api.PointCloudGPU.clone_deskewed(np.zeros((0, 3)), np.zeros(0), Til, ctx=ctx)
    assert empty.size() == 0


@pytest.mark.parametrize("form", ["constvel", "constvel_still", "imu", "imu_short"])
def test_library_deskew_table_reproduces_the_oracle_bit_for_bit(orc, form):
    """Host logic of glim_amd/csrc/deskew.hip (no device needed): the time table and the per-entry transforms the deskewing kernel gathers from,
    applied here with numpy in the kernel's operation order (three 4-term sums, separate roundings), give the oracle's -- hence the compiled
    reference's (tests/test_ref.py) -- FP64 deskewed points bit for bit, and so does the IMU-frame step on top."""
    import ctypes as C

    from glim_amd import _lib

    rng = np.random.default_rng(12)
    n = 5000
    pts = rng.uniform(-40, 40, (n, 3)).astype(np.float32).astype(np.float64)
    times = np.sort(rng.uniform(0, 0.1, n))
    times[200:260] = times[200]
    Til = orc.se3_exp([0.3, -0.2, 0.4, 0.5, -0.3, 1.2])
    it = 100.0 + np.sort(rng.uniform(-0.05, 0.15, 9))
    ip = [orc.se3_exp(rng.normal(size=6))]
    for _ in it[1:]:
        ip.append(ip[-1] @ orc.se3_exp(rng.normal(size=6) * 0.05))
    kw = {"constvel": dict(linear_vel=[4.0, -2.0, 0.3], angular_vel=[0.1, -0.2, 1.5]), "constvel_still": dict(linear_vel=[0.5, 0, 0], angular_vel=[0, 0, 0]),
          "imu": dict(imu_times=it, imu_poses=ip, stamp=100.0), "imu_short": dict(imu_times=it[:1], imu_poses=ip[:1], stamp=100.0)}[form]
    want = orc.deskew(pts, times, Til, **kw)
    L = _lib.lib()
    dp = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))
    T12 = np.ascontiguousarray(Til[:3, :4].reshape(12))
    if "imu_times" in kw:
        imu_t = np.ascontiguousarray(kw["imu_times"], dtype=np.float64)
        imu_p = np.ascontiguousarray(np.stack([P[:3, :4].reshape(12) for P in kw["imu_poses"]]))
        n_imu, lv, av = len(imu_t), None, None
    else:
        imu_t = imu_p = None
        n_imu, lv, av = 0, np.array(kw["linear_vel"], dtype=np.float64), np.array(kw["angular_vel"], dtype=np.float64)
    entry = np.zeros(n, dtype=np.int32)
    table = np.zeros((n, 12))
    size = C.c_int32()
    rc = L.glim_amd_debug_deskew_table(n, dp(times), dp(T12), n_imu, dp(imu_t), dp(imu_p), float(kw.get("stamp", 0.0)), dp(lv), dp(av),
                                       entry.ctypes.data_as(C.POINTER(C.c_int32)), dp(table), n, C.byref(size))
    assert rc == 0 and 0 < size.value <= n

    def apply(T, p):  # Isometry3d * Vector4d in the kernel's order
        return np.stack([((T[:, 4 * r] * p[:, 0] + T[:, 4 * r + 1] * p[:, 1]) + T[:, 4 * r + 2] * p[:, 2]) + T[:, 4 * r + 3] * 1.0 for r in range(3)], axis=1)

    got = apply(table[entry], pts)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(apply(np.tile(T12, (n, 1)), got), orc.transform_points(Til, want))
