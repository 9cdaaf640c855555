"""ctypes binding of the CPU oracle (oracle/vgicp_oracle.{h,c}).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see vgicp_oracle.h).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; the product package `glim_amd` must never do so.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvgicp_oracle.so")


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
        os.path.getmtime(os.path.join(_HERE, f)) for f in ("vgicp_oracle.c", "vgicp_oracle.h")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Linearized6(C.Structure):
    _fields_ = [
        ("num_inliers", C.c_int64),
        ("error", C.c_double),
        ("H_tt", C.c_double * 36),
        ("H_ss", C.c_double * 36),
        ("H_ts", C.c_double * 36),
        ("b_t", C.c_double * 6),
        ("b_s", C.c_double * 6),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        vp = C.c_void_p
        L.orc_fast_floor.restype = C.c_int32
        L.orc_fast_floor.argtypes = [C.c_double]
        L.orc_transform_point.argtypes = [dp, dp, dp]
        L.orc_se3_exp.argtypes = [dp, dp]
        L.orc_pose_compose.argtypes = [dp, dp, dp]
        L.orc_pose_inverse.argtypes = [dp, dp]
        L.orc_knn_bruteforce.argtypes = [dp, C.c_int, C.c_int, ip, C.c_int]
        L.orc_knn_grid.argtypes = [dp, C.c_int, C.c_int, C.c_double, ip, C.c_int]
        L.orc_covariance_estimate.restype = C.c_int
        L.orc_covariance_estimate.argtypes = [dp, C.c_int, ip, C.c_int, C.c_int, dp, dp, C.c_int]
        L.orc_eigen3_direct.argtypes = [dp, dp, dp]
        L.orc_voxelmap_create.restype = vp
        L.orc_voxelmap_create.argtypes = [C.c_double]
        L.orc_voxelmap_destroy.argtypes = [vp]
        L.orc_voxelmap_insert.argtypes = [vp, dp, dp, C.c_int]
        L.orc_voxelmap_num_voxels.restype = C.c_int
        L.orc_voxelmap_num_voxels.argtypes = [vp]
        L.orc_voxelmap_resolution.restype = C.c_double
        L.orc_voxelmap_resolution.argtypes = [vp]
        L.orc_voxelmap_get.argtypes = [vp, C.c_int, ip, ip, dp, dp]
        L.orc_voxelmap_lookup.restype = C.c_int
        L.orc_voxelmap_lookup.argtypes = [vp, ip]
        L.orc_voxelmap_round_to_f32.argtypes = [vp]
        L.orc_vgicp_linearize.restype = C.c_int
        L.orc_vgicp_linearize.argtypes = [vp, dp, dp, C.c_int, dp, C.c_int, C.POINTER(Linearized6), ip]
        L.orc_vgicp_error.restype = C.c_double
        L.orc_vgicp_error.argtypes = [vp, dp, dp, C.c_int, dp, C.c_int, C.POINTER(C.c_int64)]
        L.orc_vgicp_error_frozen.restype = C.c_double
        L.orc_vgicp_error_frozen.argtypes = [vp, dp, dp, C.c_int, dp, dp, C.c_int, C.POINTER(C.c_int64)]
        L.orc_overlap.restype = C.c_double
        L.orc_overlap.argtypes = [C.POINTER(vp), dp, C.c_int, dp, C.c_int, C.c_int]
        L.orc_solve6.restype = C.c_int
        L.orc_solve6.argtypes = [dp, dp, C.c_double, dp]
        L.orc_gn_align.restype = C.c_int
        L.orc_gn_align.argtypes = [vp, dp, dp, C.c_int, dp, C.c_int, C.c_double, C.c_int, dp]
        L.orc_deskew_constvel.restype = C.c_int
        L.orc_deskew_constvel.argtypes = [dp, dp, dp, dp, dp, C.c_int, dp]
        L.orc_deskew_imu.restype = C.c_int
        L.orc_deskew_imu.argtypes = [dp, dp, dp, C.c_int, C.c_double, dp, dp, C.c_int, dp]
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


# ---- layout helpers (reference layouts: Vector4d points, column-major Matrix4d covariances) ---------


def points4(xyz):
    """N x 3 -> N x 4 homogeneous (w = 1), float64."""
    xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    out = np.ones((xyz.shape[0], 4), dtype=np.float64)
    out[:, :3] = xyz
    return out


def covs16(c33):
    """N x 3 x 3 -> N x 16 (Matrix4d, zero last row/col), float64."""
    c33 = np.asarray(c33, dtype=np.float64).reshape(-1, 3, 3)
    out = np.zeros((c33.shape[0], 4, 4), dtype=np.float64)
    out[:, :3, :3] = np.transpose(c33, (0, 2, 1))  # column-major storage of a symmetric block
    return out.reshape(-1, 16)


def covs33(c16):
    c = np.asarray(c16, dtype=np.float64).reshape(-1, 4, 4)
    return np.transpose(c[:, :3, :3], (0, 2, 1)).copy()


def pose12(T):
    """4x4 or 3x4 -> 12 doubles row-major [R | t]."""
    T = np.asarray(T, dtype=np.float64)
    return np.ascontiguousarray(T[:3, :4]).reshape(12)


def pose44(T12):
    T = np.eye(4)
    T[:3, :4] = np.asarray(T12, dtype=np.float64).reshape(3, 4)
    return T


# ---- functional wrappers ---------------------------------------------------------------------------


def max_threads():
    return lib().orc_max_threads()


def se3_exp(xi):
    xi = _f64(xi, (6,))
    out = np.zeros(12)
    lib().orc_se3_exp(_dp(xi), _dp(out))
    return pose44(out)


def transform_point(T, p):
    T = pose12(T)
    p = _f64(p, (3,))
    q = np.zeros(3)
    lib().orc_transform_point(_dp(T), _dp(p), _dp(q))
    return q


def knn(points_xyz, k, num_threads=0, method="auto", cell=0.0):
    p4 = points4(points_xyz)
    n = p4.shape[0]
    out = np.zeros((n, k), dtype=np.int32)
    if n == 0:
        return out
    if method == "brute" or (method == "auto" and n <= 4096):
        lib().orc_knn_bruteforce(_dp(p4), n, k, _ip(out), num_threads)
    else:
        lib().orc_knn_grid(_dp(p4), n, k, float(cell), _ip(out), num_threads)
    return out


def covariances(points_xyz, neighbors, k_neighbors=None, num_threads=0):
    """returns (normals N x 3, covs N x 3 x 3) per cloud_covariance_estimation.cpp:43-122."""
    p4 = points4(points_xyz)
    n = p4.shape[0]
    nb = np.ascontiguousarray(neighbors, dtype=np.int32).reshape(n, -1)
    k_corr = nb.shape[1]
    k_nbr = k_corr if k_neighbors is None else int(k_neighbors)
    normals = np.zeros((n, 4))
    covs = np.zeros((n, 16))
    rc = lib().orc_covariance_estimate(_dp(p4), n, _ip(nb), k_corr, k_nbr, _dp(normals), _dp(covs), num_threads)
    if rc != 0:
        raise ValueError("orc_covariance_estimate failed")
    return normals[:, :3].copy(), covs33(covs)


def eigen3(m):
    m = _f64(m, (9,))
    ev = np.zeros(3)
    V = np.zeros(9)
    lib().orc_eigen3_direct(_dp(m), _dp(ev), _dp(V))
    return ev, V.reshape(3, 3).T.copy()  # columns = eigenvectors


class VoxelMap:
    """GaussianVoxelMapCPU semantics (first-touch voxel order, mean of means / mean of covs)."""

    def __init__(self, resolution):
        self._h = C.c_void_p(lib().orc_voxelmap_create(float(resolution)))

    def __del__(self):
        try:
            if self._h:
                lib().orc_voxelmap_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def insert(self, points_xyz, covs_33):
        p4 = points4(points_xyz)
        c16 = covs16(covs_33)
        lib().orc_voxelmap_insert(self._h, _dp(p4), _dp(c16), p4.shape[0])
        return self

    @property
    def resolution(self):
        return lib().orc_voxelmap_resolution(self._h)

    def num_voxels(self):
        return lib().orc_voxelmap_num_voxels(self._h)

    def round_to_f32(self):
        lib().orc_voxelmap_round_to_f32(self._h)
        return self

    def voxels(self):
        """returns (coords V x 3 int32, counts V, means V x 3, covs V x 3 x 3) in first-touch order."""
        v = self.num_voxels()
        coords = np.zeros((v, 3), dtype=np.int32)
        counts = np.zeros(v, dtype=np.int32)
        means = np.zeros((v, 4))
        covs = np.zeros((v, 16))
        cnt = C.c_int32()
        for i in range(v):
            lib().orc_voxelmap_get(self._h, i, _ip(coords[i]), C.byref(cnt), _dp(means[i]), _dp(covs[i]))
            counts[i] = cnt.value
        return coords, counts, means[:, :3].copy(), covs33(covs)

    def lookup(self, coord):
        c = np.ascontiguousarray(coord, dtype=np.int32).reshape(3)
        return lib().orc_voxelmap_lookup(self._h, _ip(c))


def _lin_to_dict(L):
    return {
        "num_inliers": int(L.num_inliers),
        "error": float(L.error),
        "H_tt": np.array(L.H_tt).reshape(6, 6),
        "H_ss": np.array(L.H_ss).reshape(6, 6),
        "H_ts": np.array(L.H_ts).reshape(6, 6),
        "b_t": np.array(L.b_t),
        "b_s": np.array(L.b_s),
    }


def vgicp_linearize(vmap, src_xyz, src_covs33, delta, num_threads=0, want_corr=False):
    p4 = points4(src_xyz)
    c16 = covs16(src_covs33)
    n = p4.shape[0]
    T = pose12(delta)
    L = Linearized6()
    corr = np.zeros((n, 4), dtype=np.int32) if want_corr else None
    rc = lib().orc_vgicp_linearize(vmap._h, _dp(p4), _dp(c16), n, _dp(T), num_threads, C.byref(L),
                                   _ip(corr) if want_corr else None)
    if rc != 0:
        raise ValueError("orc_vgicp_linearize failed")
    out = _lin_to_dict(L)
    if want_corr:
        out["corr"] = corr
    return out


def vgicp_error(vmap, src_xyz, src_covs33, delta, num_threads=0, delta_lin=None):
    p4 = points4(src_xyz)
    c16 = covs16(src_covs33)
    n = p4.shape[0]
    T = pose12(delta)
    ninl = C.c_int64()
    if delta_lin is None:
        e = lib().orc_vgicp_error(vmap._h, _dp(p4), _dp(c16), n, _dp(T), num_threads, C.byref(ninl))
    else:
        Tl = pose12(delta_lin)
        e = lib().orc_vgicp_error_frozen(vmap._h, _dp(p4), _dp(c16), n, _dp(Tl), _dp(T), num_threads, C.byref(ninl))
    return float(e), int(ninl.value)


def overlap(vmaps, src_xyz, deltas, num_threads=0):
    if isinstance(vmaps, VoxelMap):
        vmaps, deltas = [vmaps], [deltas]
    p4 = points4(src_xyz)
    hs = (C.c_void_p * len(vmaps))(*[m._h for m in vmaps])
    T = np.concatenate([pose12(d) for d in deltas])
    return float(lib().orc_overlap(hs, _dp(T), len(vmaps), _dp(p4), p4.shape[0], num_threads))


def solve6(H, b, lam=0.0):
    H = _f64(H, (36,))
    b = _f64(b, (6,))
    x = np.zeros(6)
    rc = lib().orc_solve6(_dp(H), _dp(b), float(lam), _dp(x))
    if rc != 0:
        raise np.linalg.LinAlgError("H + lambda I not positive definite")
    return x


def gn_align(vmap, src_xyz, src_covs33, T_init, max_iters=8, lam=0.0, num_threads=0):
    p4 = points4(src_xyz)
    c16 = covs16(src_covs33)
    T = pose12(T_init).copy()
    deltas = np.zeros((max_iters, 6))
    it = lib().orc_gn_align(vmap._h, _dp(p4), _dp(c16), p4.shape[0], _dp(T), max_iters, float(lam), num_threads, _dp(deltas))
    return pose44(T), deltas[:it].copy()


def deskew(points_xyz, times, T_imu_lidar, imu_times=None, imu_poses=None, stamp=0.0, linear_vel=(0, 0, 0), angular_vel=(0, 0, 0)):
    """CloudDeskewing::deskew (cloud_deskewing.cpp): IMU-pose form when imu_times/imu_poses are given, else constant velocity.
    Returns N x 3 float64."""
    p4 = points4(points_xyz)
    n = p4.shape[0]
    t = _f64(times, (n,))
    Til = pose12(T_imu_lidar)
    out = np.zeros((n, 4))
    if imu_times is not None and len(imu_times) > 0:
        it = _f64(imu_times, (-1,))
        ip = np.ascontiguousarray(np.stack([pose12(P) for P in imu_poses]))
        lib().orc_deskew_imu(_dp(Til), _dp(it), _dp(ip), len(it), float(stamp), _dp(t), _dp(p4), n, _dp(out))
    else:
        lv, av = _f64(linear_vel, (3,)), _f64(angular_vel, (3,))
        lib().orc_deskew_constvel(_dp(Til), _dp(lv), _dp(av), _dp(t), _dp(p4), n, _dp(out))
    return out[:, :3].copy()
