"""ctypes binding of the CPU oracle (oracle/vgicp_oracle.{h,c}).

TEST INFRASTRUCTURE ONLY -- parity partly pinned: covariance + deskewing against the reference's own compiled code (oracle/_ref), the
gtsam_points rows unpinned (see vgicp_oracle.h).  Only tests/, __graft_entry__.smoke()
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
        os.path.getmtime(os.path.join(_HERE, f)) for f in ("vgicp_oracle.c", "preprocess_oracle.c", "vgicp_oracle.h")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


# ---- oracle/_ref: the reference's own translation units (cloud_covariance_estimation.cpp, cloud_deskewing.cpp) compiled from
# /root/reference against stand-in headers (oracle/Makefile target `ref`).  Present in this container and -- as a prebuilt, git-ignored
# .so that travels with the snapshot -- on the GPU box; /root/reference itself is never read at run time.
_REF_PATH = os.path.join(_HERE, "_ref", "libglim_ref.so")
_REFERENCE_ROOT = "/root/reference"
_ref = None


def build_ref(force=False):
    """Compile oracle/_ref/libglim_ref.so when the reference tree is present; returns the path or None."""
    if not os.path.isdir(os.path.join(_REFERENCE_ROOT, "src", "glim", "common")):
        return _REF_PATH if os.path.exists(_REF_PATH) else None
    subprocess.check_call(["make", "-C", _HERE, "ref"] + (["-B"] if force else ["-s"]), stdout=subprocess.DEVNULL)
    return _REF_PATH


def build_twin():
    """oracle/_ref/libglim_twin.so: the same C entry points as libglim_ref.so over the HIP-backed twins of the three in-tree translation units
    (adapters/glim/cloud_*_hip.cpp) -- tests/test_twins.py runs both side by side.  Needs the reference's headers and libglim_amd.so: built where
    /root/reference exists, the prebuilt .so travels to the GPU box.  Returns the path or None."""
    path = os.path.join(_HERE, "_ref", "libglim_twin.so")
    if os.path.isdir(os.path.join(_REFERENCE_ROOT, "include", "glim")):
        subprocess.check_call(["make", "-C", _HERE, "twin", "-s"], stdout=subprocess.DEVNULL)
    return path if os.path.exists(path) else None


def ref_lib():
    """The compiled reference translation units, or None when neither the prebuilt .so nor /root/reference exists."""
    global _ref
    if _ref is None:
        path = build_ref()
        if path is None or not os.path.exists(path):
            return None
        L = C.CDLL(path)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        L.ref_covariance_estimate.restype = C.c_int
        L.ref_covariance_estimate.argtypes = [dp, C.c_int, ip, C.c_int, C.c_int, dp, dp, C.c_int]
        L.ref_deskew_constvel.argtypes = [dp, dp, dp, dp, dp, C.c_int, dp]
        L.ref_deskew_imu.argtypes = [dp, dp, dp, C.c_int, C.c_double, dp, dp, C.c_int, dp]
        if hasattr(L, "ref_frontend"):
            L.ref_frontend.restype = C.c_int
            L.ref_frontend.argtypes = [dp, dp, dp, C.c_int, C.c_double, dp, dp, dp, dp, C.c_int, ip, C.c_int, C.c_int, dp, dp, dp, C.c_int]
        if hasattr(L, "ref_preprocess"):
            L.ref_preprocess.restype = C.c_int
            L.ref_preprocess.argtypes = [dp, dp, dp, C.c_int, C.POINTER(PreprocessParams), dp, dp, dp, ip, dp, C.c_int]
        _ref = L
    return _ref


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


class PreprocessParams(C.Structure):
    """orc_preprocess_params (defaults: the shipped config/config_preprocess.json)."""

    _fields_ = [
        ("distance_near_thresh", C.c_double),
        ("distance_far_thresh", C.c_double),
        ("use_random_grid_downsampling", C.c_int32),
        ("downsample_target", C.c_int32),
        ("downsample_resolution", C.c_double),
        ("downsample_rate", C.c_double),
        ("global_shutter", C.c_int32),
        ("enable_outlier_removal", C.c_int32),
        ("outlier_removal_k", C.c_int32),
        ("outlier_std_mul_factor", C.c_double),
        ("enable_cropbox_filter", C.c_int32),
        ("crop_bbox_frame_imu", C.c_int32),
        ("crop_bbox_min", C.c_double * 3),
        ("crop_bbox_max", C.c_double * 3),
        ("T_imu_lidar", C.c_double * 12),
        ("k_correspondences", C.c_int32),
        ("voxelgrid_block_size", C.c_int32),
        ("seed", C.c_uint64),
    ]


def preprocess_params(**kw):
    p = PreprocessParams()
    d = dict(
        distance_near_thresh=0.5, distance_far_thresh=100.0, use_random_grid_downsampling=1, downsample_target=10000,
        downsample_resolution=1.0, downsample_rate=0.1, global_shutter=0, enable_outlier_removal=0, outlier_removal_k=10,
        outlier_std_mul_factor=1.0, enable_cropbox_filter=0, crop_bbox_frame_imu=0, k_correspondences=10, voxelgrid_block_size=1024, seed=0,
    )
    d.update({k: v for k, v in kw.items() if k not in ("crop_bbox_min", "crop_bbox_max", "T_imu_lidar")})
    for k, v in d.items():
        setattr(p, k, v)
    p.crop_bbox_min[:] = list(kw.get("crop_bbox_min", (-1.0, -1.0, -1.0)))
    p.crop_bbox_max[:] = list(kw.get("crop_bbox_max", (1.0, 1.0, 1.0)))
    p.T_imu_lidar[:] = list(pose12(kw.get("T_imu_lidar", np.eye(4))))
    return p


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
        L.orc_voxelmap_set_lru.argtypes = [vp, C.c_int, C.c_int]
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
        L.orc_transform_points.restype = C.c_int
        L.orc_transform_points.argtypes = [dp, dp, C.c_int, dp]
        L.orc_deskew_imu.restype = C.c_int
        L.orc_deskew_imu.argtypes = [dp, dp, dp, C.c_int, C.c_double, dp, dp, C.c_int, dp]
        pp = C.POINTER(PreprocessParams)
        L.orc_sample_hash.restype = C.c_uint64
        L.orc_sample_hash.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_sampling_key.restype = C.c_uint64
        L.orc_sampling_key.argtypes = [dp, C.c_double]
        L.orc_voxelgrid_sampling.restype = C.c_int
        L.orc_voxelgrid_sampling.argtypes = [dp, dp, dp, C.c_int, C.c_double, C.c_int, dp, dp, dp]
        L.orc_randomgrid_sampling.restype = C.c_int
        L.orc_randomgrid_sampling.argtypes = [dp, C.c_int, C.c_double, C.c_double, C.c_uint64, ip]
        L.orc_find_inliers.restype = C.c_int
        L.orc_find_inliers.argtypes = [dp, C.c_int, C.c_int, C.c_double, ip, C.c_int]
        L.orc_preprocess_keep.restype = C.c_int
        L.orc_preprocess_keep.argtypes = [dp, pp]
        L.orc_preprocess.restype = C.c_int
        L.orc_preprocess.argtypes = [dp, dp, dp, C.c_int, pp, dp, dp, dp, ip, C.c_int]
        L.orc_gicp_linearize.restype = C.c_int
        L.orc_gicp_linearize.argtypes = [dp, dp, C.c_int, dp, dp, C.c_int, dp, C.c_double, C.c_int, C.POINTER(Linearized6), ip]
        L.orc_gicp_error.restype = C.c_double
        L.orc_gicp_error.argtypes = [dp, dp, C.c_int, dp, dp, C.c_int, dp, C.c_double, C.c_int, C.POINTER(C.c_int64)]
        L.orc_median_distance.restype = C.c_double
        L.orc_median_distance.argtypes = [dp, C.c_int, C.c_int]
        L.orc_adaptive_resolution.restype = C.c_double
        L.orc_adaptive_resolution.argtypes = [C.c_double] * 5
        L.orc_merge_frames.restype = C.c_int
        L.orc_merge_frames.argtypes = [C.c_int, dp, C.POINTER(dp), C.POINTER(dp), ip, C.c_double, C.c_int, C.c_int, C.c_uint64, dp, dp]
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


_fast = None


def fast_lib():
    """A second build of the SAME restatement for bench.py's `cpu_baseline` timing only: -O3 -march=native (SURVEY.md 8d "CPU baseline
    timing"), compiled on the box it runs on into oracle/_fast/ (git-ignored).  The checker above keeps its bit-exact flags
    (-O2 -ffp-contract=off -march=x86-64-v3); nothing is ever CHECKED against this build.  Returns None when gcc is unavailable."""
    global _fast
    if _fast is None:
        out_dir = os.path.join(_HERE, "_fast")
        path = os.path.join(out_dir, "libvgicp_oracle_native.so")
        try:
            os.makedirs(out_dir, exist_ok=True)
            subprocess.check_call(["gcc", "-O3", "-march=native", "-std=c11", "-fopenmp", "-fPIC", "-shared", os.path.join(_HERE, "vgicp_oracle.c"),
                                   os.path.join(_HERE, "preprocess_oracle.c"), "-o", path, "-lm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            L = C.CDLL(path)
        except Exception:
            return None
        dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p
        L.orc_voxelmap_create.restype = vp
        L.orc_voxelmap_create.argtypes = [C.c_double]
        L.orc_voxelmap_destroy.argtypes = [vp]
        L.orc_voxelmap_insert.argtypes = [vp, dp, dp, C.c_int]
        L.orc_vgicp_linearize.restype = C.c_int
        L.orc_vgicp_linearize.argtypes = [vp, dp, dp, C.c_int, dp, C.c_int, C.POINTER(Linearized6), ip]
        L.orc_max_threads.restype = C.c_int
        _fast = L
    return _fast


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


def covariances(points_xyz, neighbors, k_neighbors=None, num_threads=0, ref=False):
    """returns (normals N x 3, covs N x 3 x 3) per cloud_covariance_estimation.cpp:43-122.  ref=True: run the reference's own compiled
    CloudCovarianceEstimation::estimate (oracle/_ref) instead of the restatement."""
    p4 = points4(points_xyz)
    n = p4.shape[0]
    nb = np.ascontiguousarray(neighbors, dtype=np.int32).reshape(n, -1)
    k_corr = nb.shape[1]
    k_nbr = k_corr if k_neighbors is None else int(k_neighbors)
    normals = np.zeros((n, 4))
    covs = np.zeros((n, 16))
    fn = ref_lib().ref_covariance_estimate if ref else lib().orc_covariance_estimate
    rc = fn(_dp(p4), n, _ip(nb), k_corr, k_nbr, _dp(normals), _dp(covs), num_threads)
    if rc != 0:
        raise ValueError("covariance_estimate failed")
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

    def set_lru_horizon(self, horizon, clear_cycle=10):
        """GaussianVoxelMapCPU::set_lru_horizon (odometry_estimation_cpu.cpp:67) + lru_clear_cycle; horizon <= 0: no eviction."""
        lib().orc_voxelmap_set_lru(self._h, int(horizon), int(clear_cycle))
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


def vgicp_linearize_sv(vmap, src_xyz, src_covs33, src_normals, delta, force=None, num_threads=0, want_corr=False):
    """orc_vgicp_linearize_sv: the factor with surface validation ON (predicate of DESIGN.md 4.8, FP64; upstream unverified).  Returns the
    linearisation dict plus `s` (the per-point predicate value (R n) . q; a point is dropped when s > 0) and, on request, `corr`."""
    p4, c16 = points4(src_xyz), covs16(src_covs33)
    n4 = np.zeros_like(p4)
    n4[:, :3] = np.asarray(src_normals, dtype=np.float64)
    n = p4.shape[0]
    L = Linearized6()
    corr = np.zeros((n, 4), dtype=np.int32) if want_corr else None
    s = np.zeros(n)
    f = None if force is None else np.ascontiguousarray(force, dtype=np.int8)
    fn = lib().orc_vgicp_linearize_sv
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int8),
                   C.c_int, C.POINTER(Linearized6), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    rc = fn(vmap._h, _dp(p4), _dp(c16), _dp(n4), n, _dp(pose12(delta)), f.ctypes.data_as(C.POINTER(C.c_int8)) if f is not None else None, num_threads,
            C.byref(L), _ip(corr) if want_corr else None, _dp(s))
    if rc != 0:
        raise ValueError("orc_vgicp_linearize_sv failed")
    out = _lin_to_dict(L)
    out["s"] = s
    if want_corr:
        out["corr"] = corr
    return out


def vgicp_error_frozen_sv(vmap, src_xyz, src_covs33, src_normals, delta_lin, delta_eval, force=None, num_threads=0):
    """error at delta_eval over the correspondences (voxel hit AND surface validation) frozen at delta_lin; returns (error, inliers)."""
    p4, c16 = points4(src_xyz), covs16(src_covs33)
    n4 = np.zeros_like(p4)
    n4[:, :3] = np.asarray(src_normals, dtype=np.float64)
    f = None if force is None else np.ascontiguousarray(force, dtype=np.int8)
    fn = lib().orc_vgicp_error_frozen_sv
    fn.restype = C.c_double
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                   C.POINTER(C.c_int8), C.c_int, C.POINTER(C.c_int64)]
    ninl = C.c_int64()
    e = fn(vmap._h, _dp(p4), _dp(c16), _dp(n4), p4.shape[0], _dp(pose12(delta_lin)), _dp(pose12(delta_eval)),
           f.ctypes.data_as(C.POINTER(C.c_int8)) if f is not None else None, num_threads, C.byref(ninl))
    return float(e), int(ninl.value)


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


def deskew(points_xyz, times, T_imu_lidar, imu_times=None, imu_poses=None, stamp=0.0, linear_vel=(0, 0, 0), angular_vel=(0, 0, 0), ref=False):
    """CloudDeskewing::deskew (cloud_deskewing.cpp): IMU-pose form when imu_times/imu_poses are given, else constant velocity.
    Returns N x 3 float64.  ref=True: the reference's own compiled CloudDeskewing (oracle/_ref)."""
    L = ref_lib() if ref else lib()
    f_imu = L.ref_deskew_imu if ref else L.orc_deskew_imu
    f_cv = L.ref_deskew_constvel if ref else L.orc_deskew_constvel
    p4 = points4(points_xyz)
    n = p4.shape[0]
    t = _f64(times, (n,))
    Til = pose12(T_imu_lidar)
    out = np.zeros((n, 4))
    if imu_times is not None and len(imu_times) > 0:
        it = _f64(imu_times, (-1,))
        ip = np.ascontiguousarray(np.stack([pose12(P) for P in imu_poses]))
        f_imu(_dp(Til), _dp(it), _dp(ip), len(it), float(stamp), _dp(t), _dp(p4), n, _dp(out))
    else:
        lv, av = _f64(linear_vel, (3,)), _f64(angular_vel, (3,))
        f_cv(_dp(Til), _dp(lv), _dp(av), _dp(t), _dp(p4), n, _dp(out))
    return out[:, :3].copy()


def transform_points(T, points_xyz):
    """`pt = T * pt` for every point (Isometry3d * Vector4d): the IMU-frame step after deskewing (odometry_estimation_imu.cpp:314-316)."""
    p4 = points4(points_xyz)
    out = np.zeros_like(p4)
    lib().orc_transform_points(_dp(pose12(T)), _dp(p4), p4.shape[0], _dp(out))
    return out[:, :3].copy()


def frontend(points_xyz, times, neighbors, T_imu_lidar, imu_times=None, imu_poses=None, stamp=0.0, linear_vel=(0, 0, 0), angular_vel=(0, 0, 0),
             to_imu_frame=True, ref=False, num_threads=0):
    """The chain between preprocessing and create_frame (odometry_estimation_imu.cpp:313-320): deskew -> pt = T_imu_lidar * pt -> covariance
    estimation from the RAW scan's neighbours, on FP64 points throughout.  Returns (points N x 3, normals N x 3, covs N x 3 x 3).
    ref=True: the three steps on the reference's own compiled objects (oracle/_ref `ref_frontend`); otherwise the restatement's functions."""
    nb = np.ascontiguousarray(neighbors, dtype=np.int32)
    n, k = nb.shape
    if ref:
        p4 = points4(points_xyz)
        t = _f64(times, (n,))
        op, on, oc = np.zeros((n, 4)), np.zeros((n, 4)), np.zeros((n, 16))
        if imu_times is not None and len(imu_times) > 0:
            it = _f64(imu_times, (-1,))
            ip = np.ascontiguousarray(np.stack([pose12(P) for P in imu_poses]))
            n_imu, lv, av = len(it), np.zeros(3), np.zeros(3)
        else:
            it, ip, n_imu = np.zeros(1), np.zeros((1, 12)), -1
            lv, av = _f64(linear_vel, (3,)), _f64(angular_vel, (3,))
        rc = ref_lib().ref_frontend(_dp(pose12(T_imu_lidar)), _dp(it), _dp(ip), n_imu, float(stamp), _dp(lv), _dp(av), _dp(t), _dp(p4), n, _ip(nb), k,
                                    int(bool(to_imu_frame)), _dp(op), _dp(on), _dp(oc), int(num_threads))
        if rc != 0:
            raise ValueError("ref_frontend failed")
        return op[:, :3].copy(), on[:, :3].copy(), covs33(oc)
    d = deskew(points_xyz, times, T_imu_lidar, imu_times=imu_times, imu_poses=imu_poses, stamp=stamp, linear_vel=linear_vel, angular_vel=angular_vel)
    if to_imu_frame:
        d = transform_points(T_imu_lidar, d)
    normals, covs = covariances(d, nb, num_threads=num_threads)
    return d, normals, covs


# ---- scan preprocessing ----------------------------------------------------------


def voxelgrid_sampling(points_xyz, times, intensities, resolution, block_size=1024):
    p4 = points4(points_xyz)
    n = p4.shape[0]
    t = _f64(times, (n,))
    it = _f64(intensities, (n,)) if intensities is not None else None
    op, ot, oi = np.zeros((max(n, 1), 4)), np.zeros(max(n, 1)), np.zeros(max(n, 1))
    m = lib().orc_voxelgrid_sampling(_dp(p4), _dp(t), _dp(it) if it is not None else None, n, float(resolution), int(block_size), _dp(op), _dp(ot), _dp(oi))
    return op[:m, :3].copy(), ot[:m].copy(), (oi[:m].copy() if it is not None else None)


def randomgrid_sampling(points_xyz, resolution, rate, seed=0):
    p4 = points4(points_xyz)
    n = p4.shape[0]
    idx = np.zeros(max(n, 1), dtype=np.int32)
    m = lib().orc_randomgrid_sampling(_dp(p4), n, float(resolution), float(rate), int(seed), _ip(idx))
    return idx[:m].copy()


def find_inliers(points_xyz, k, std_mul, num_threads=0):
    p4 = points4(points_xyz)
    n = p4.shape[0]
    idx = np.zeros(max(n, 1), dtype=np.int32)
    m = lib().orc_find_inliers(_dp(p4), n, int(k), float(std_mul), _ip(idx), int(num_threads))
    return idx[:m].copy()


def preprocess(points_xyz, times, intensities=None, params=None, neighbors=True, num_threads=0, ref=False):
    """CloudPreprocessor::preprocess_impl.  Returns dict(points N'x3, times, intensities|None, neighbors|None).
    ref=True: the reference's own cloud_preprocessor.cpp (oracle/_ref, ref_preprocess_shim.cpp) instead of the restatement; its three
    gtsam_points sampling calls are answered by the restatement's functions, everything else is the reference's code (also returns
    `scan_end_time`, for stamp = 0, and `k_neighbors`)."""
    prm = params if params is not None else preprocess_params()
    p4 = points4(points_xyz)
    n = p4.shape[0]
    t = _f64(times, (n,))
    it = _f64(intensities, (n,)) if intensities is not None else None
    op, ot, oi = np.zeros((max(n, 1), 4)), np.zeros(max(n, 1)), np.zeros(max(n, 1))
    nb = np.zeros((max(n, 1), prm.k_correspondences), dtype=np.int32) if neighbors else None
    if ref:
        meta = np.zeros(2)
        m = ref_lib().ref_preprocess(_dp(p4), _dp(t), _dp(it) if it is not None else None, n, C.byref(prm), _dp(op), _dp(ot), _dp(oi),
                                     _ip(nb) if nb is not None else None, _dp(meta), int(num_threads))
        if m < 0:
            raise ValueError("the reference preprocessor threw")
        return dict(points=op[:m, :3].copy(), times=ot[:m].copy(), intensities=oi[:m].copy() if it is not None else None,
                    neighbors=nb[:m].copy() if nb is not None else None, scan_end_time=float(meta[0]), k_neighbors=int(meta[1]))
    m = lib().orc_preprocess(_dp(p4), _dp(t), _dp(it) if it is not None else None, n, C.byref(prm), _dp(op), _dp(ot), _dp(oi),
                             _ip(nb) if nb is not None else None, int(num_threads))
    return dict(points=op[:m, :3].copy(), times=ot[:m].copy(), intensities=oi[:m].copy() if it is not None else None,
                neighbors=nb[:m].copy() if nb is not None else None)


def merge_frames(poses, frames_points, frames_covs, resolution, target_num_points=-1, seed=0, block_size=1024):
    """gtsam_points::merge_frames (sub_mapping.cpp:480-497).  frames_points[f]: N_f x 3, frames_covs[f]: N_f x 3 x 3.
    Returns (points M x 3, covs M x 3 x 3)."""
    nf = len(poses)
    P12 = np.ascontiguousarray(np.stack([pose12(T) for T in poses])) if nf else np.zeros((1, 12))
    p4s = [points4(p) for p in frames_points]
    c16s = [covs16(c) for c in frames_covs]
    sizes = np.array([len(p) for p in p4s], dtype=np.int32)
    dp = C.POINTER(C.c_double)
    pp = (dp * nf)(*[_dp(p) for p in p4s])
    cp = (dp * nf)(*[_dp(c) for c in c16s])
    total = int(sizes.sum())
    op, oc = np.zeros((max(total, 1), 4)), np.zeros((max(total, 1), 16))
    m = lib().orc_merge_frames(nf, _dp(P12), pp, cp, _ip(sizes), float(resolution), int(block_size), int(target_num_points), int(seed), _dp(op), _dp(oc))
    return op[:m, :3].copy(), covs33(oc[:m])


def gicp_linearize(tgt_xyz, tgt_covs33, src_xyz, src_covs33, delta, max_correspondence_distance=1.0, num_threads=0, want_corr=False):
    """gtsam_points::IntegratedGICPFactor::linearize (sub_mapping.cpp:202-203): exact nearest target point within the distance."""
    tp, tc, sp, sc = points4(tgt_xyz), covs16(tgt_covs33), points4(src_xyz), covs16(src_covs33)
    L = Linearized6()
    corr = np.zeros(len(sp), dtype=np.int32) if want_corr else None
    rc = lib().orc_gicp_linearize(_dp(tp), _dp(tc), len(tp), _dp(sp), _dp(sc), len(sp), _dp(pose12(delta)), float(max_correspondence_distance),
                                  num_threads, C.byref(L), _ip(corr) if want_corr else None)
    if rc != 0:
        raise ValueError("orc_gicp_linearize failed")
    out = _lin_to_dict(L)
    if want_corr:
        out["corr"] = corr
    return out


def gicp_error(tgt_xyz, tgt_covs33, src_xyz, src_covs33, delta, max_correspondence_distance=1.0, num_threads=0):
    tp, tc, sp, sc = points4(tgt_xyz), covs16(tgt_covs33), points4(src_xyz), covs16(src_covs33)
    ninl = C.c_int64()
    e = lib().orc_gicp_error(_dp(tp), _dp(tc), len(tp), _dp(sp), _dp(sc), len(sp), _dp(pose12(delta)), float(max_correspondence_distance), num_threads,
                             C.byref(ninl))
    return float(e), int(ninl.value)


def median_distance(points_xyz, max_scan_count=256):
    p4 = points4(points_xyz)
    return float(lib().orc_median_distance(_dp(p4), len(p4), int(max_scan_count)))


def adaptive_resolution(dist_median, r0, rmax, dmin, dmax):
    return float(lib().orc_adaptive_resolution(float(dist_median), float(r0), float(rmax), float(dmin), float(dmax)))
