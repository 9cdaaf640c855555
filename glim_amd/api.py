"""Host-side mirror of the gtsam_points GPU interface GLIM uses for the VGICP hot path.

Same names and argument meaning as the reference-side classes so that tests read like the reference's call sites:

    frame     = PointCloudGPU.clone(points, covs)                    # odometry_estimation_gpu.cpp:96
    voxelmap  = GaussianVoxelMapGPU(resolution); voxelmap.insert(frame)   # :103-104
    factor    = IntegratedVGICPFactorGPU(target_key, source_key, voxelmap, frame)   # :144  (binary)
    factor    = IntegratedVGICPFactorGPU(fixed_target_pose, source_key, voxelmap, frame)   # :161  (unary)
    factor.set_enable_surface_validation(True)                       # :145
    fset      = NonlinearFactorSetGPU(); fset.add(factor); fset.linearize(values)   # :383-386
    overlap   = overlap_gpu(voxelmap, frame, delta)                  # :248

Everything below is a thin ctypes shim over the C ABI (include/glim_amd.h); all arithmetic happens in the HIP kernels.
`Values` is a plain dict {key: 4x4 pose}; a linearised factor is returned as the HessianFactor ingredients.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FACTOR_BINARY, FACTOR_SURFACE_VALIDATION, GlimAmdError, Linearized6, PreprocessParams, check, lib  # noqa: F401

_default_ctx = None
STREAM_LEGACY = 1  # hipStreamLegacy ((hipStream_t)1): the null stream, as an explicit handle
STREAM_PER_THREAD = 2  # hipStreamPerThread


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def pose12(T):
    T = np.asarray(T, dtype=np.float64)
    if T.size == 12:
        return np.ascontiguousarray(T).reshape(12)
    return np.ascontiguousarray(T[:3, :4]).reshape(12)


def device_count():
    return lib().glim_amd_device_count()


class Context:
    """Device + stream pool (gtsam_points::CUDAStream / StreamTempBufferRoundRobin)."""

    def __init__(self, device=0, num_streams=1, external_stream=None, priority=0):
        """priority: 0 default, 1 = the device's greatest stream priority (the odometry's context), -1 = its least."""
        h = C.c_void_p()
        ext = None
        if external_stream is not None:
            # 0 is the legacy default stream (what torch.cuda.current_stream().cuda_stream returns for torch's default stream): the C ABI
            # takes the HIP handle for it, hipStreamLegacy, because NULL means "create private streams"
            ext = C.c_void_p(int(external_stream) if int(external_stream) != 0 else STREAM_LEGACY)
        check(lib().glim_amd_ctx_create_ex(int(device), int(num_streams), ext, int(priority), C.byref(h)), "glim_amd_ctx_create_ex")
        self._h = h
        self.device = device

    def synchronize(self):
        check(lib().glim_amd_ctx_synchronize(self._h), "glim_amd_ctx_synchronize")

    def set_diag(self, key_values=""):
        """Diagnostic switches of this context ("key=value,key=value"; "" restores the process defaults): include/glim_amd.h."""
        check(lib().glim_amd_ctx_set_diag(self._h, key_values.encode()), "glim_amd_ctx_set_diag")

    def diag(self, key_values=""):
        """Context manager: the process defaults + `key_values` inside the block, the process defaults again on the way out -- whatever
        happens inside.  (set_diag ADDS to the switches in force; tests and measurements that flip kernel paths use this instead, so that a
        failure between two set_diag calls cannot leave a shared context in the wrong mode.)"""
        import contextlib

        @contextlib.contextmanager
        def scope():
            self.set_diag("")
            try:
                if key_values:
                    self.set_diag(key_values)
                yield self
            finally:
                if self._h:
                    self.set_diag("")

        return scope()

    def get_diag(self):
        buf = C.create_string_buffer(1024)
        check(lib().glim_amd_ctx_get_diag(self._h, buf, 1024), "glim_amd_ctx_get_diag")
        return dict(kv.split("=", 1) for kv in buf.value.decode().split(","))

    def device_info(self):
        name = C.create_string_buffer(256)
        free, total, cus = C.c_size_t(), C.c_size_t(), C.c_int()
        check(lib().glim_amd_device_info(self._h, name, 256, C.byref(free), C.byref(total), C.byref(cus)), "glim_amd_device_info")
        return {"name": name.value.decode(), "free_bytes": free.value, "total_bytes": total.value, "num_cus": cus.value}

    def close(self):
        """Destroys the context.  Refused (GlimAmdError, the context stays usable) while clouds / maps / factor sets made from it are alive."""
        if self._h:
            check(lib().glim_amd_ctx_destroy(self._h), "glim_amd_ctx_destroy")
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0, 1)
    return _default_ctx


class PointCloudGPU:
    """gtsam_points::PointCloudGPU -- device-resident FP32 SoA copy of a frame."""

    def __init__(self, handle, ctx):
        self._h = handle
        self.ctx = ctx

    @staticmethod
    def clone(points, covs=None, normals=None, ctx=None):
        """points: N x 3 or N x 4 (homogeneous); covs: N x 3 x 3 or N x 4 x 4 (Matrix4d); normals: N x 3 / N x 4.
        float64 inputs travel in the reference's Vector4d / Matrix4d layout and are packed on the device; float32 inputs use
        the compact entry point."""
        ctx = ctx or default_context()
        points = np.asarray(points)
        n = points.shape[0]
        h = C.c_void_p()
        if points.dtype == np.float32 and (covs is None or np.asarray(covs).dtype == np.float32):
            xyz = np.ascontiguousarray(points[:, :3], dtype=np.float32)
            c = None if covs is None else np.ascontiguousarray(np.asarray(covs, dtype=np.float32)[:, :3, :3]).reshape(n, 9)
            nr = None if normals is None else np.ascontiguousarray(np.asarray(normals, dtype=np.float32)[:, :3])
            check(lib().glim_amd_cloud_create_f32(ctx._h, n, _fp(xyz), _fp(c), _fp(nr), C.byref(h)), "glim_amd_cloud_create_f32")
        else:
            p4 = np.ones((n, 4), dtype=np.float64)
            p4[:, : min(4, points.shape[1])] = points[:, :4]
            p4[:, 3] = 1.0
            c16 = None
            if covs is not None:
                covs = np.asarray(covs, dtype=np.float64)
                m = np.zeros((n, 4, 4))
                m[:, :3, :3] = covs[:, :3, :3]
                c16 = np.ascontiguousarray(np.transpose(m, (0, 2, 1))).reshape(n, 16)  # column-major Matrix4d
            n4 = None
            if normals is not None:
                normals = np.asarray(normals, dtype=np.float64)
                n4 = np.zeros((n, 4))
                n4[:, :3] = normals[:, :3]
            check(lib().glim_amd_cloud_create(ctx._h, n, _dp(p4), _dp(c16), _dp(n4), C.byref(h)), "glim_amd_cloud_create")
        return PointCloudGPU(h, ctx)

    @staticmethod
    def clone_packed(points4, covs16=None, normals4=None, ctx=None):
        """PointCloudGPU::clone of arrays that already have the reference's layout (n x Vector4d, n x column-major Matrix4d): nothing is
        repacked on the Python side, so timing this call times the library."""
        ctx = ctx or default_context()
        h = C.c_void_p()
        check(lib().glim_amd_cloud_create(ctx._h, len(points4), _dp(points4), _dp(covs16), _dp(normals4), C.byref(h)), "glim_amd_cloud_create")
        return PointCloudGPU(h, ctx)

    @staticmethod
    def clone_deskewed(points, times, T_imu_lidar, imu_times=None, imu_poses=None, stamp=0.0, linear_vel=None, angular_vel=None, ctx=None,
                       to_imu_frame=False):
        """CloudDeskewing::deskew (cloud_deskewing.cpp) fused with PointCloudGPU::clone: IMU-pose form when imu_times / imu_poses are
        given, constant-velocity form otherwise.  to_imu_frame: also apply `pt = T_imu_lidar * pt` (odometry_estimation_imu.cpp:314-316),
        the step both reference callers take before estimating covariances."""
        ctx = ctx or default_context()
        points = np.asarray(points, dtype=np.float64)
        n = points.shape[0]
        p4 = np.ones((n, 4), dtype=np.float64)
        p4[:, :3] = points[:, :3]
        t = np.ascontiguousarray(times, dtype=np.float64).reshape(n)
        Til = pose12(T_imu_lidar)
        it = ip = None
        n_imu = 0
        if imu_times is not None and len(imu_times) > 0:
            it = np.ascontiguousarray(imu_times, dtype=np.float64)
            ip = np.ascontiguousarray(np.stack([pose12(P) for P in imu_poses]))
            n_imu = len(it)
        lv = None if linear_vel is None else np.ascontiguousarray(linear_vel, dtype=np.float64)
        av = None if angular_vel is None else np.ascontiguousarray(angular_vel, dtype=np.float64)
        h = C.c_void_p()
        check(lib().glim_amd_cloud_create_deskewed(ctx._h, n, _dp(p4), _dp(t), _dp(Til), n_imu, _dp(it), _dp(ip), float(stamp), _dp(lv), _dp(av),
                                                   int(bool(to_imu_frame)), C.byref(h)), "glim_amd_cloud_create_deskewed")
        return PointCloudGPU(h, ctx)

    @staticmethod
    def preprocess(points, times, intensities=None, params=None, ctx=None):
        """CloudPreprocessor::preprocess (cloud_preprocessor.cpp:92-188) on the device: returns the preprocessed cloud (points, times,
        intensities, kNN resident in HBM); `download_frame()` gives the PreprocessedFrame fields."""
        ctx = ctx or default_context()
        prm = params if params is not None else preprocess_params()
        points = np.asarray(points, dtype=np.float64)
        n = points.shape[0]
        if points.ndim == 2 and points.shape[1] == 4 and points.flags.c_contiguous:
            p4 = points  # already the reference's Vector4d layout: handed over as it is (a 131 072-point copy costs as much as the whole call)
        else:
            p4 = np.ones((n, 4), dtype=np.float64)
            p4[:, : points.shape[1]] = points
        t = np.ascontiguousarray(times, dtype=np.float64).reshape(n)
        it = None if intensities is None else np.ascontiguousarray(intensities, dtype=np.float64).reshape(n)
        h = C.c_void_p()
        check(lib().glim_amd_preprocess(ctx._h, n, _dp(p4), _dp(t), _dp(it), C.byref(prm), C.byref(h)), "glim_amd_preprocess")
        g = PointCloudGPU(h, ctx)
        g._k = prm.k_correspondences
        g._has_intensities = it is not None
        return g

    def download_frame(self):
        """PreprocessedFrame fields (include/glim/preprocess/preprocessed_frame.hpp:26-39) of a preprocessed cloud."""
        n, k = self.size(), getattr(self, "_k", 0)
        p4, t = np.zeros((n, 4)), np.zeros(n)
        it = np.zeros(n) if getattr(self, "_has_intensities", False) else None
        nb = np.zeros((n, k), dtype=np.int32) if k > 0 else None
        check(lib().glim_amd_cloud_download_frame(self._h, _dp(p4), _dp(t), _dp(it), _ip(nb)), "glim_amd_cloud_download_frame")
        return dict(points=p4[:, :3].copy(), times=t, intensities=it, neighbors=nb, k_neighbors=k)

    def save_compact(self, directory):
        """gtsam_points::PointCloud::save_compact (sub_map.cpp:62): *_compact.bin files inside `directory`."""
        check(lib().glim_amd_cloud_save_compact(self._h, str(directory).encode()), "glim_amd_cloud_save_compact")

    @staticmethod
    def load_compact(directory, ctx=None):
        """gtsam_points::PointCloudCPU::load (sub_map.cpp:142) + clone."""
        ctx = ctx or default_context()
        h = C.c_void_p()
        check(lib().glim_amd_cloud_load_compact(ctx._h, str(directory).encode(), C.byref(h)), "glim_amd_cloud_load_compact")
        return PointCloudGPU(h, ctx)

    def download_merged(self):
        """The merged submap as PointCloudCPU holds it: (points N x 3, covs N x 3 x 3), exact FP64."""
        n = self.size()
        p4, c16 = np.zeros((n, 4)), np.zeros((n, 16))
        check(lib().glim_amd_cloud_download_merged(self._h, _dp(p4), _dp(c16)), "glim_amd_cloud_download_merged")
        return p4[:, :3].copy(), np.transpose(c16.reshape(n, 4, 4)[:, :3, :3], (0, 2, 1)).copy()

    def download_points64(self):
        """The exact FP64 points a preprocessed / deskewed / merged cloud keeps next to its FP32 image (N x 3)."""
        p4 = np.zeros((self.size(), 4))
        check(lib().glim_amd_cloud_download_frame(self._h, _dp(p4), None, None, None), "glim_amd_cloud_download_frame")
        return p4[:, :3].copy()

    def deskew(self, T_imu_lidar, imu_times=None, imu_poses=None, stamp=0.0, linear_vel=None, angular_vel=None, to_imu_frame=False):
        """CloudDeskewing::deskew of a preprocessed cloud that is already on the device; the raw-scan neighbours are carried over.
        to_imu_frame: followed by `pt = T_imu_lidar * pt` (odometry_estimation_imu.cpp:314-316, sub_mapping.cpp:368-370)."""
        Til = pose12(T_imu_lidar)
        it = ip = None
        n_imu = 0
        if imu_times is not None and len(imu_times) > 0:
            it = np.ascontiguousarray(imu_times, dtype=np.float64)
            ip = np.ascontiguousarray(np.stack([pose12(P) for P in imu_poses]))
            n_imu = len(it)
        lv = None if linear_vel is None else np.ascontiguousarray(linear_vel, dtype=np.float64)
        av = None if angular_vel is None else np.ascontiguousarray(angular_vel, dtype=np.float64)
        h = C.c_void_p()
        check(lib().glim_amd_cloud_deskew(self._h, _dp(Til), n_imu, _dp(it), _dp(ip), float(stamp), _dp(lv), _dp(av), int(bool(to_imu_frame)), C.byref(h)),
              "glim_amd_cloud_deskew")
        g = PointCloudGPU(h, self.ctx)
        g._k = getattr(self, "_k", 0)
        return g

    def size(self):
        n = C.c_int64()
        check(lib().glim_amd_cloud_size(self._h, C.byref(n)), "glim_amd_cloud_size")
        return n.value

    def memory_usage_gpu(self):
        b = C.c_size_t()
        check(lib().glim_amd_cloud_memory_usage(self._h, C.byref(b)), "glim_amd_cloud_memory_usage")
        return b.value

    def find_neighbors(self, k, download=True):
        """CloudPreprocessor::find_neighbors on the device; returns N x k int32 when download."""
        out = np.zeros((self.size(), k), dtype=np.int32) if download else None
        check(lib().glim_amd_cloud_find_neighbors(self._h, int(k), _ip(out)), "glim_amd_cloud_find_neighbors")
        return out

    def set_neighbors(self, neighbors):
        nb = np.ascontiguousarray(neighbors, dtype=np.int32)
        check(lib().glim_amd_cloud_set_neighbors(self._h, nb.shape[1], _ip(nb)), "glim_amd_cloud_set_neighbors")

    def profile_neighbors(self, k=10, iters=20):
        """(wall ms per find_neighbors call, HIP-event ms of the query-group kernel inside it): glim_amd_cloud_profile_neighbors."""
        a, b = C.c_float(), C.c_float()
        check(lib().glim_amd_cloud_profile_neighbors(self._h, int(k), int(iters), C.byref(a), C.byref(b)), "glim_amd_cloud_profile_neighbors")
        return a.value, b.value

    def estimate_covariances(self, k_neighbors):
        """CloudCovarianceEstimation::estimate on the device (fills covs + normals)."""
        check(lib().glim_amd_cloud_estimate_covariances(self._h, int(k_neighbors)), "glim_amd_cloud_estimate_covariances")

    def download(self, covs=True, normals=True):
        n = self.size()
        xyz = np.zeros((n, 3), dtype=np.float32)
        c = np.zeros((n, 9), dtype=np.float32) if covs else None
        nr = np.zeros((n, 3), dtype=np.float32) if normals else None
        check(lib().glim_amd_cloud_download(self._h, _fp(xyz), _fp(c), _fp(nr), None), "glim_amd_cloud_download")
        return xyz, (c.reshape(n, 3, 3) if covs else None), nr

    def close(self):
        if self._h:
            lib().glim_amd_cloud_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GaussianVoxelMapGPU:
    """gtsam_points::GaussianVoxelMapGPU(resolution, init_num_buckets, max_bucket_scan_count, target_points_drop_rate)."""

    def __init__(self, resolution, init_num_buckets=8192 * 2, max_bucket_scan_count=10, target_points_drop_rate=1e-3, ctx=None):
        self.ctx = ctx or default_context()
        h = C.c_void_p()
        check(lib().glim_amd_voxelmap_create(self.ctx._h, float(resolution), int(init_num_buckets), int(max_bucket_scan_count),
                                             float(target_points_drop_rate), C.byref(h)), "glim_amd_voxelmap_create")
        self._h = h
        self._frame = None

    def voxel_resolution(self):
        return self.voxelmap_info()["voxel_resolution"]

    def insert(self, frame):
        check(lib().glim_amd_voxelmap_insert(self._h, frame._h), "glim_amd_voxelmap_insert")
        return self

    def set_lru_horizon(self, lru_horizon, lru_clear_cycle=10):
        """GaussianVoxelMapCPU::set_lru_horizon (odometry_estimation_cpu.cpp:67): voxels untouched for more than `lru_horizon` inserts are dropped
        every `lru_clear_cycle` inserts; <= 0 switches eviction off (default)."""
        check(lib().glim_amd_voxelmap_set_lru_horizon(self._h, int(lru_horizon), int(lru_clear_cycle)), "glim_amd_voxelmap_set_lru_horizon")
        return self

    def voxelmap_info(self):
        nv, nb, res, by = C.c_int32(), C.c_int32(), C.c_double(), C.c_size_t()
        check(lib().glim_amd_voxelmap_info(self._h, C.byref(nv), C.byref(nb), C.byref(res), C.byref(by)), "glim_amd_voxelmap_info")
        return {"num_voxels": nv.value, "num_buckets": nb.value, "voxel_resolution": res.value, "bytes": by.value}

    def voxels(self):
        """(coords V x 3, counts V, means V x 3, covs V x 3 x 3), unspecified order."""
        v = self.voxelmap_info()["num_voxels"]
        coords = np.zeros((v, 3), dtype=np.int32)
        counts = np.zeros(v, dtype=np.int32)
        means = np.zeros((v, 3), dtype=np.float32)
        covs = np.zeros((v, 9), dtype=np.float32)
        check(lib().glim_amd_voxelmap_download(self._h, _ip(coords), _ip(counts), _fp(means), _fp(covs)), "glim_amd_voxelmap_download")
        return coords, counts, means, covs.reshape(v, 3, 3)

    def close(self):
        if self._h:
            lib().glim_amd_voxelmap_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def frame_create(points4, covs16, normals4, resolutions, ctx=None):
    """glim_amd_frame_create: PointCloudGPU::clone + one GaussianVoxelMapGPU per resolution as ONE submission with one synchronise (create_frame of
    the GPU odometry).  Arrays in the reference's layout (n x Vector4d, n x column-major Matrix4d, n x Vector4d).  Returns (cloud, [maps])."""
    ctx = ctx or default_context()
    res = np.ascontiguousarray(resolutions, dtype=np.float64)
    hc = C.c_void_p()
    hm = (C.c_void_p * len(res))()
    check(lib().glim_amd_frame_create(ctx._h, len(points4), _dp(points4), _dp(covs16), _dp(normals4), len(res), _dp(res), C.byref(hc),
                                      C.cast(hm, C.POINTER(C.c_void_p))), "glim_amd_frame_create")
    maps = []
    for h in hm:
        m = GaussianVoxelMapGPU.__new__(GaussianVoxelMapGPU)
        m.ctx, m._h, m._frame = ctx, C.c_void_p(h), None
        maps.append(m)
    return PointCloudGPU(hc, ctx), maps


def _lin_to_dict(L):
    # one buffer view of the 976-byte struct (int64 + 121 doubles) instead of five element-wise ctypes -> numpy conversions
    a = np.frombuffer(L, dtype=np.float64, count=122).copy()
    return {
        "num_inliers": int(L.num_inliers),
        "error": float(a[1]),
        "H_tt": a[2:38].reshape(6, 6),
        "H_ss": a[38:74].reshape(6, 6),
        "H_ts": a[74:110].reshape(6, 6),
        "b_t": a[110:116],
        "b_s": a[116:122],
    }


class IntegratedVGICPFactorGPU:
    """gtsam_points::IntegratedVGICPFactorGPU.

    IntegratedVGICPFactorGPU(target_key, source_key, target_voxelmap, source)            binary
    IntegratedVGICPFactorGPU(fixed_target_pose(4x4), source_key, target_voxelmap, source)  unary
    """

    def __init__(self, target, source_key, target_voxelmap, source, stream=None, temp_buffer=None):
        self.is_binary = not (isinstance(target, np.ndarray) or isinstance(target, (list, tuple)))
        if self.is_binary:
            self.target_key = target
            self.fixed_target_pose = None
        else:
            self.target_key = None
            self.fixed_target_pose = np.array(target, dtype=np.float64).reshape(4, 4)
        self.source_key = source_key
        self.target_voxelmap = target_voxelmap
        self.source = source
        self.enable_surface_validation = False
        self._linearized = None  # filled by NonlinearFactorSetGPU (store_linearized in the reference's batch protocol)
        self._own_set = None

    # -- reference API ------------------------------------------------------------------------------------------
    def set_enable_surface_validation(self, enable):
        self.enable_surface_validation = bool(enable)
        self._own_set = None

    def keys(self):
        return [self.target_key, self.source_key] if self.is_binary else [self.source_key]

    def dim(self):
        return 6

    def get_fixed_target_pose(self):
        return self.fixed_target_pose

    def memory_usage(self):
        return 976  # one glim_amd_linearized6 on the host

    def memory_usage_gpu(self):
        return self.source.memory_usage_gpu() + self.target_voxelmap.voxelmap_info()["bytes"]

    def clone(self):
        f = IntegratedVGICPFactorGPU(self.target_key if self.is_binary else self.fixed_target_pose, self.source_key,
                                     self.target_voxelmap, self.source)
        f.enable_surface_validation = self.enable_surface_validation
        return f

    def flags(self):
        return (FACTOR_BINARY if self.is_binary else 0) | (FACTOR_SURFACE_VALIDATION if self.enable_surface_validation else 0)

    def calc_delta(self, values):
        """T_target_source at `values` (dict key -> 4x4)."""
        Ts = np.asarray(values[self.source_key], dtype=np.float64)
        Tt = np.asarray(values[self.target_key], dtype=np.float64) if self.is_binary else self.fixed_target_pose
        return np.linalg.inv(Tt) @ Ts

    def _single_set(self):
        if self._own_set is None:
            self._own_set = NonlinearFactorSetGPU(self.source.ctx)
            self._own_set.add(self)
        return self._own_set

    def linearize(self, values):
        """HessianFactor ingredients.  Uses the batch result when a NonlinearFactorSetGPU linearised this factor at the
        same values; otherwise performs its own upload/launch/download (the reference's slow path)."""
        delta = self.calc_delta(values)
        if self._linearized is not None and np.array_equal(self._linearized[0], delta):
            return self._linearized[1]
        self._single_set().linearize(values)
        return self._linearized[1]

    def error(self, values):
        return self._single_set().error(values)[0]

    def inlier_fraction(self):
        if self._linearized is None:
            return 0.0
        return self._linearized[1]["num_inliers"] / max(1, self.source.size())


class NonlinearFactorSetGPU:
    """gtsam_points::NonlinearFactorSetGPU: linearises every added GPU factor in one fused launch."""

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()
        h = C.c_void_p()
        check(lib().glim_amd_factor_set_create(self.ctx._h, C.byref(h)), "glim_amd_factor_set_create")
        self._h = h
        self.factors = []

    def add(self, factor):
        if not isinstance(factor, IntegratedVGICPFactorGPU):
            return False  # non-GPU factors are ignored, like the reference
        idx = C.c_int32()
        check(lib().glim_amd_factor_set_add(self._h, factor.target_voxelmap._h, factor.source._h, factor.flags(), C.byref(idx)),
              "glim_amd_factor_set_add")
        self.factors.append(factor)
        return True

    def clear(self):
        check(lib().glim_amd_factor_set_clear(self._h), "glim_amd_factor_set_clear")
        self.factors = []

    def size(self):
        n = C.c_int32()
        check(lib().glim_amd_factor_set_size(self._h, C.byref(n)), "glim_amd_factor_set_size")
        return n.value

    def _poses(self, values):
        return np.ascontiguousarray(np.stack([pose12(f.calc_delta(values)) for f in self.factors])) if self.factors else np.zeros((0, 12))

    def linearize(self, values):
        n = len(self.factors)
        if n == 0:
            return []
        T = self._poses(values)
        out = (Linearized6 * n)()
        check(lib().glim_amd_factor_set_linearize(self._h, _dp(T), out), "glim_amd_factor_set_linearize")
        res = []
        for f, L, t in zip(self.factors, out, T):
            d = _lin_to_dict(L)
            delta = np.eye(4)
            delta[:3, :4] = t.reshape(3, 4)
            f._linearized = (delta, d)
            res.append(d)
        return res

    def linearize_poses(self, T_target_source):
        """Same with explicit n x 12 poses (no Values indirection)."""
        n = len(self.factors)
        T = np.ascontiguousarray(np.asarray(T_target_source, dtype=np.float64).reshape(n, 12))
        out = (Linearized6 * n)()
        check(lib().glim_amd_factor_set_linearize(self._h, _dp(T), out), "glim_amd_factor_set_linearize")
        return [_lin_to_dict(L) for L in out]

    def error(self, values, values_lin=None):
        n = len(self.factors)
        if n == 0:
            return []
        Te = self._poses(values)
        Tl = self._poses(values_lin) if values_lin is not None else None
        err = np.zeros(n)
        inl = np.zeros(n, dtype=np.int64)
        check(lib().glim_amd_factor_set_error(self._h, _dp(Tl), _dp(Te), _dp(err), inl.ctypes.data_as(C.POINTER(C.c_int64))),
              "glim_amd_factor_set_error")
        self.last_error_inliers = inl
        return list(err)

    def correspondences(self, index, delta):
        n = self.factors[index].source.size()
        corr = np.zeros((n, 4), dtype=np.int32)
        T = pose12(delta)
        check(lib().glim_amd_factor_set_correspondences(self._h, int(index), _dp(T), _ip(corr)), "glim_amd_factor_set_correspondences")
        return corr

    def cull_stats(self, reset=False):
        """(trips culled by the pre-pass, trips that hold points) since the last reset, or None when the plan has no pre-cull (glim_amd_factor_set_cull_stats)."""
        a, b = C.c_uint64(), C.c_uint64()
        rc = lib().glim_amd_factor_set_cull_stats(self._h, C.byref(a), C.byref(b), int(bool(reset)))
        if rc == -6:
            return None
        check(rc, "glim_amd_factor_set_cull_stats")
        return a.value, b.value

    def linearize_device_async(self, T_target_source, out_device_ptr, row_offset=0):
        T = T_target_source
        if not (isinstance(T, np.ndarray) and T.dtype == np.float64 and T.flags.c_contiguous and T.size == 12 * len(self.factors)):
            T = np.ascontiguousarray(np.asarray(T_target_source, dtype=np.float64).reshape(len(self.factors), 12))
        # raw addresses: building ctypes pointer objects costs several microseconds per call, which matters in launch-rate loops
        rc = lib().glim_amd_factor_set_linearize_device_async(self._h, C.cast(T.ctypes.data, C.POINTER(C.c_double)), C.c_void_p(out_device_ptr),
                                                              int(row_offset))
        if rc != 0:
            check(rc, "glim_amd_factor_set_linearize_device_async")

    def profile(self, T_target_source, iters=20):
        """(ms per fused VGICP kernel launch, ms per device-resident linearise) measured with HIP events on the set's stream."""
        T = np.ascontiguousarray(np.asarray(T_target_source, dtype=np.float64).reshape(len(self.factors), 12))
        a, b = C.c_float(), C.c_float()
        check(lib().glim_amd_factor_set_profile(self._h, _dp(T), int(iters), C.byref(a), C.byref(b)), "glim_amd_factor_set_profile")
        return a.value, b.value

    def trip_stats(self, reset=False):
        """(skipped wavefront trips since the last reset, trips per evaluation) of the general factor kernel: glim_amd_factor_set_trip_stats."""
        a, b = C.c_uint64(), C.c_uint64()
        check(lib().glim_amd_factor_set_trip_stats(self._h, C.byref(a), C.byref(b), int(bool(reset))), "glim_amd_factor_set_trip_stats")
        return a.value, b.value

    def profile_sync(self, T_target_source, iters=200):
        """milliseconds per synchronous linearize() call measured inside the library (no binding overhead)."""
        T = np.ascontiguousarray(np.asarray(T_target_source, dtype=np.float64).reshape(len(self.factors), 12))
        a = C.c_float()
        check(lib().glim_amd_factor_set_profile_sync(self._h, _dp(T), int(iters), C.byref(a)), "glim_amd_factor_set_profile_sync")
        return a.value

    def linearize_repeat(self, pose_sets, iters):
        """EXACTLY `iters` synchronous linearize() calls issued from C (no binding overhead, no warm-up, no clock: the caller times it), call i at
        pose_sets[i % len(pose_sets)]; returns the last call's records (glim_amd_factor_set_linearize_repeat)."""
        nf = len(self.factors)
        T = np.ascontiguousarray(np.asarray(pose_sets, dtype=np.float64).reshape(-1, nf, 12))
        out = (Linearized6 * nf)()
        check(lib().glim_amd_factor_set_linearize_repeat(self._h, _dp(T), T.shape[0], int(iters), out), "glim_amd_factor_set_linearize_repeat")
        return [_lin_to_dict(L) for L in out]

    def profile_lm(self, T_target_source, iters=20):
        """(ms per synchronous linearize(), ms per synchronous error()) of the whole set, timed inside the library."""
        T = np.ascontiguousarray(np.asarray(T_target_source, dtype=np.float64).reshape(len(self.factors), 12))
        a, b = C.c_float(), C.c_float()
        check(lib().glim_amd_factor_set_profile_lm(self._h, _dp(T), int(iters), C.byref(a), C.byref(b)), "glim_amd_factor_set_profile_lm")
        return a.value, b.value

    def close(self):
        if self._h:
            lib().glim_amd_factor_set_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiDeviceCost:
    """Single-process, multi-device evaluation of a multi-scan VGICP cost (glim_amd_multi_*, include/glim_amd.h): clouds and voxel maps
    replicated on every device, the factor list sharded, one RCCL all-gather of the compact records per evaluation."""

    def __init__(self, devices=(0,), virtual=False):
        """virtual=True (tests / one-GPU boxes only, glim_amd_diag.h): a device ordinal may be listed several times -- every entry is a "virtual
        device" with its own context, host thread, shard and gathered array; the exchange is a same-device stand-in for ncclAllGather."""
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        h = C.c_void_p()
        if virtual:
            check(lib().glim_amd_debug_multi_create_virtual(_ip(dev), len(dev), C.byref(h)), "glim_amd_debug_multi_create_virtual")
        else:
            check(lib().glim_amd_multi_create(_ip(dev), len(dev), C.byref(h)), "glim_amd_multi_create")
        self._h = h
        self.factors = []

    def set_gather_mode(self, mode):
        """1 (default): the all-gather completes behind the call; 2: the call waits for it; 0: no exchange (glim_amd_multi_set_gather_mode)."""
        check(lib().glim_amd_multi_set_gather_mode(self._h, int(mode)), "glim_amd_multi_set_gather_mode")

    def wait_gather(self):
        check(lib().glim_amd_multi_wait_gather(self._h), "glim_amd_multi_wait_gather")

    def gathered_device(self, device=0):
        """(device pointer, rows) of the complete device-resident record array of one device (glim_amd_multi_gathered_device)."""
        ptr, rows = C.c_void_p(), C.c_int64()
        check(lib().glim_amd_multi_gathered_device(self._h, int(device), C.byref(ptr), C.byref(rows)), "glim_amd_multi_gathered_device")
        return ptr.value, rows.value

    def gathered_records(self, device=0, first=0, count=None):
        """What a device-side consumer on `device` reads after the exchange, copied back in factor order (glim_amd_debug_multi_gathered_download)."""
        count = self._n - first if count is None else count
        out = np.zeros((count, 29), dtype=np.float64)
        check(lib().glim_amd_debug_multi_gathered_download(self._h, int(device), int(first), int(count), _dp(out)), "glim_amd_debug_multi_gathered_download")
        return out

    def inject_failure(self, device, where):
        """One shot: the next evaluation fails on `device` before the barrier (where=1) / inside its exchange (where=2)."""
        check(lib().glim_amd_debug_multi_inject_failure(self._h, int(device), int(where)), "glim_amd_debug_multi_inject_failure")

    def set_diag(self, key_values):
        check(lib().glim_amd_debug_multi_set_diag(self._h, key_values.encode()), "glim_amd_debug_multi_set_diag")

    def info(self):
        nd, rc, nf = C.c_int32(), C.c_int32(), C.c_int64()
        check(lib().glim_amd_multi_info(self._h, C.byref(nd), C.byref(rc), C.byref(nf)), "glim_amd_multi_info")
        return {"num_devices": nd.value, "uses_rccl": bool(rc.value), "num_factors": nf.value}

    def add_cloud(self, points, covs=None, normals=None):
        points = np.asarray(points)
        n = points.shape[0]
        cid = C.c_int32()
        xyz = np.ascontiguousarray(points[:, :3], dtype=np.float32)
        c = None if covs is None else np.ascontiguousarray(np.asarray(covs, dtype=np.float32)[:, :3, :3]).reshape(n, 9)
        nr = None if normals is None else np.ascontiguousarray(np.asarray(normals, dtype=np.float32)[:, :3])
        check(lib().glim_amd_multi_add_cloud_f32(self._h, n, _fp(xyz), _fp(c), _fp(nr), C.byref(cid)), "glim_amd_multi_add_cloud_f32")
        return cid.value

    def estimate_covariances(self, cloud_id, k=10):
        check(lib().glim_amd_multi_cloud_estimate_covariances(self._h, int(cloud_id), int(k)), "glim_amd_multi_cloud_estimate_covariances")

    def add_voxelmap(self, cloud_id, resolution):
        mid = C.c_int32()
        check(lib().glim_amd_multi_add_voxelmap(self._h, int(cloud_id), float(resolution), C.byref(mid)), "glim_amd_multi_add_voxelmap")
        return mid.value

    def set_factors(self, target_map_ids, source_cloud_ids, flags=None):
        t = np.ascontiguousarray(target_map_ids, dtype=np.int32)
        s = np.ascontiguousarray(source_cloud_ids, dtype=np.int32)
        f = None if flags is None else np.ascontiguousarray(flags, dtype=np.uint32)
        check(lib().glim_amd_multi_set_factors(self._h, len(t), _ip(t), _ip(s), None if f is None else f.ctypes.data_as(C.POINTER(C.c_uint32))),
              "glim_amd_multi_set_factors")
        self._n = len(t)

    def shard(self):
        b = np.zeros(self.info()["num_devices"] + 1, dtype=np.int64)
        check(lib().glim_amd_multi_shard(self._h, b.ctypes.data_as(C.POINTER(C.c_int64))), "glim_amd_multi_shard")
        return b

    def linearize(self, T_target_source):
        T = np.ascontiguousarray(np.asarray(T_target_source, dtype=np.float64).reshape(self._n, 12))
        out = (Linearized6 * self._n)()
        tot = C.c_double()
        check(lib().glim_amd_multi_linearize(self._h, _dp(T), out, C.byref(tot)), "glim_amd_multi_linearize")
        return [_lin_to_dict(L) for L in out], tot.value

    def profile(self, T_target_source, iters=10):
        T = np.ascontiguousarray(np.asarray(T_target_source, dtype=np.float64).reshape(self._n, 12))
        ms = C.c_float()
        check(lib().glim_amd_multi_profile(self._h, _dp(T), int(iters), C.byref(ms)), "glim_amd_multi_profile")
        return ms.value

    def last_timing(self):
        """Per device: HIP-event ms of the last evaluation's kernels and of its collective + copy-out."""
        nd = self.info()["num_devices"]
        k, g = np.zeros(nd, dtype=np.float32), np.zeros(nd, dtype=np.float32)
        check(lib().glim_amd_multi_last_timing(self._h, k.ctypes.data_as(C.POINTER(C.c_float)), g.ctypes.data_as(C.POINTER(C.c_float))), "glim_amd_multi_last_timing")
        return k.tolist(), g.tolist()

    BREAKDOWN_FIELDS = ("post", "wake", "pose_stage", "enqueue", "barrier", "collective", "wait", "join", "scan", "total", "library_calls",
                        "device_gather", "device_copy_out")

    def last_breakdown(self, device=0):
        """Host-side account of the last evaluation on one device's thread, microseconds (glim_amd_multi_last_breakdown)."""
        us = np.zeros(len(self.BREAKDOWN_FIELDS), dtype=np.float64)
        check(lib().glim_amd_multi_last_breakdown(self._h, int(device), _dp(us), len(us)), "glim_amd_multi_last_breakdown")
        return dict(zip(self.BREAKDOWN_FIELDS, us.tolist()))

    def set_split(self, mode):
        """-1: pieces of >= 2048 factors, at most 4 (default); 0 / 1: one piece; n: n pieces.  Applies to the next set_factors."""
        check(lib().glim_amd_multi_set_split(self._h, int(mode)), "glim_amd_multi_set_split")

    def set_host_records(self, mode):
        """1 (default): the finalising kernels store every record into the host array as well; 0: device-to-host copies behind each piece.
        Applies to the next set_factors."""
        check(lib().glim_amd_multi_set_host_records(self._h, int(mode)), "glim_amd_multi_set_host_records")

    def set_one_rank_collective(self, on):
        """One device: make the (no-op) ncclAllGather in every evaluation as well (measurement aid; default off)."""
        check(lib().glim_amd_multi_set_one_rank_collective(self._h, int(bool(on))), "glim_amd_multi_set_one_rank_collective")

    def evaluate(self, T_target_source):
        """One evaluation that leaves the records in the handle (records()) and returns the total error, summed by the devices."""
        T = np.ascontiguousarray(np.asarray(T_target_source, dtype=np.float64).reshape(self._n, 12))
        tot = C.c_double()
        check(lib().glim_amd_multi_linearize(self._h, _dp(T), None, C.byref(tot)), "glim_amd_multi_linearize")
        return tot.value

    def records(self, first=0, count=None):
        """Compact 29-double records of the last evaluation, factor order (glim_amd_multi_records)."""
        count = self._n - first if count is None else count
        out = np.zeros((count, 29), dtype=np.float64)
        check(lib().glim_amd_multi_records(self._h, int(first), int(count), _dp(out)), "glim_amd_multi_records")
        return out

    def close(self):
        if self._h:
            lib().glim_amd_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def resident_stats(ctx=None):
    """Debug: the device's resident session (glim_amd_debug_resident_stats): kernel launches, requests served, alive right now."""
    ctx = ctx or default_context()
    a, b, c = C.c_uint64(), C.c_uint64(), C.c_int32()
    check(lib().glim_amd_debug_resident_stats(int(getattr(ctx, "device", 0)), C.byref(a), C.byref(b), C.byref(c)), "glim_amd_debug_resident_stats")
    return {"launches": a.value, "requests": b.value, "alive": bool(c.value)}


def plan_stats(ctx=None):
    """Debug: factor plans built for new lists, how many of them in the buffers of an evicted plan, idle plans cached (glim_amd_debug_plan_stats)."""
    ctx = ctx or default_context()
    a, b, c = C.c_uint64(), C.c_uint64(), C.c_int32()
    check(lib().glim_amd_debug_plan_stats(ctx._h, C.byref(a), C.byref(b), C.byref(c)), "glim_amd_debug_plan_stats")
    return {"built": a.value, "recycled": b.value, "cached": c.value}


def scratch_poke(ctx, word, value=0xFFFFFFFF):
    """Test hook (glim_amd_debug_scratch_poke): leave `value` in a word of the context's pinned scratch; the default stands for the sequence
    number the context's next polled voxel-map build waits for."""
    check(lib().glim_amd_debug_scratch_poke(ctx._h, int(word), int(value)), "glim_amd_debug_scratch_poke")


RESIDENT_TIMELINE_FIELDS = ("host_round_trip", "leader_published", "worker_pose_seen_min", "worker_pose_seen_median", "worker_pose_seen_max",
                            "worker_row_computed_min", "worker_row_computed_median", "worker_row_computed_max", "worker_row_published_min",
                            "worker_row_published_median", "worker_row_published_max", "finaliser_pose_seen", "finaliser_rows_summed",
                            "finaliser_record_stored", "workers_accounted", "device_span", "worker_loop_left_min", "worker_loop_left_median",
                            "worker_loop_left_max", "finaliser_group_sums_added", "finaliser_blocks_rotated", "shader_clock_mhz")


def resident_timeline(device=0, enable=True, read=True):
    """glim_amd_debug_resident_timeline: ends the device's resident session; read=True returns the last request's timeline (microseconds relative to
    the leader seeing the request; None when no stamped session has run); the stamps stay on / off for the next sessions per `enable`."""
    us = np.zeros(len(RESIDENT_TIMELINE_FIELDS), dtype=np.float64)
    rc = lib().glim_amd_debug_resident_timeline(int(device), int(bool(enable)), _dp(us) if read else None, len(us) if read else 0)
    if rc == -5 and read:
        return None
    check(rc, "glim_amd_debug_resident_timeline")
    return dict(zip(RESIDENT_TIMELINE_FIELDS, us.tolist())) if read else None


def resident_stop(ctx=None):
    """Debug: end the device's resident session now (it would idle out by itself after `resident_idle_us`)."""
    ctx = ctx or default_context()
    check(lib().glim_amd_debug_resident_stop(int(getattr(ctx, "device", 0))), "glim_amd_debug_resident_stop")


def shard_bounds(costs, world):
    """glim_amd_shard_bounds: the C implementation of the sharding rule (host only; works without a device)."""
    c = np.ascontiguousarray(costs, dtype=np.float64)
    b = np.zeros(int(world) + 1, dtype=np.int64)
    check(lib().glim_amd_shard_bounds(_dp(c) if len(c) else None, len(c), int(world), b.ctypes.data_as(C.POINTER(C.c_int64))), "glim_amd_shard_bounds")
    return [int(x) for x in b]


def shard_layout(bounds, split_mode=-1):
    """glim_amd_shard_layout: (row of every factor in the gathered array, max_rows, pieces, piece_rows) -- host only, no device needed."""
    b = np.ascontiguousarray(bounds, dtype=np.int64)
    world = len(b) - 1
    rows = np.zeros(max(1, int(b[-1])), dtype=np.int64)
    mr, pc, pr = C.c_int64(), C.c_int32(), C.c_int64()
    lp = C.POINTER(C.c_int64)
    check(lib().glim_amd_shard_layout(b.ctypes.data_as(lp), world, int(split_mode), rows.ctypes.data_as(lp), C.byref(mr), C.byref(pc), C.byref(pr)), "glim_amd_shard_layout")
    return rows[: int(b[-1])], mr.value, pc.value, pr.value


def preprocess_params(**kw):
    """glim_amd_preprocess_params with the shipped defaults (config/config_preprocess.json), fields overridden by keyword."""
    p = PreprocessParams()
    check(lib().glim_amd_preprocess_default_params(C.byref(p)), "glim_amd_preprocess_default_params")
    for k, v in kw.items():
        if k in ("crop_bbox_min", "crop_bbox_max"):
            getattr(p, k)[:] = [float(x) for x in v]
        elif k == "T_imu_lidar":
            p.T_imu_lidar[:] = list(pose12(v))
        else:
            setattr(p, k, v)
    return p


class IntegratedGICPFactor:
    """gtsam_points::IntegratedGICPFactor on the device (sub_mapping.cpp:202, global_mapping.cpp:400, global_mapping_pose_graph.cpp:393).
    Unary form: IntegratedGICPFactor(fixed_target_pose, source_key, target, source); binary: (target_key, source_key, target, source).
    `target` / `source` are PointCloudGPU with covariances; the target's search index is built once here (or passed as `target_tree`)."""

    def __init__(self, target, source_key, target_frame, source_frame, target_tree=None, max_correspondence_distance=1.0):
        self.binary = np.isscalar(target)
        self.target_key = int(target) if self.binary else None
        self.fixed_target_pose = None if self.binary else np.asarray(target, dtype=np.float64)
        self.source_key = int(source_key)
        self.target_frame, self.source_frame = target_frame, source_frame
        self.max_correspondence_distance = float(max_correspondence_distance)
        self._inliers = 0
        self._own_tree = target_tree is None
        if target_tree is None:
            h = C.c_void_p()
            check(lib().glim_amd_nn_index_create(target_frame._h, self.max_correspondence_distance, C.byref(h)), "glim_amd_nn_index_create")
            target_tree = h
        self.target_tree = target_tree

    def set_max_correspondence_distance(self, d):
        self.max_correspondence_distance = float(d)

    def calc_delta(self, values):
        Ts = np.asarray(values[self.source_key], dtype=np.float64)
        Tt = np.asarray(values[self.target_key], dtype=np.float64) if self.binary else self.fixed_target_pose
        return np.linalg.inv(Tt) @ Ts

    def linearize(self, values):
        L = Linearized6()
        T = pose12(self.calc_delta(values))
        check(lib().glim_amd_gicp_linearize(self.target_tree, self.source_frame._h, _dp(T), self.max_correspondence_distance,
                                            FACTOR_BINARY if self.binary else 0, C.byref(L)), "glim_amd_gicp_linearize")
        out = _lin_to_dict(L)
        self._inliers = out["num_inliers"]
        return out

    def error(self, values):
        e, n = C.c_double(), C.c_int64()
        T = pose12(self.calc_delta(values))
        check(lib().glim_amd_gicp_error(self.target_tree, self.source_frame._h, _dp(T), self.max_correspondence_distance, C.byref(e), C.byref(n)),
              "glim_amd_gicp_error")
        self._inliers = n.value
        return e.value

    def inlier_fraction(self):
        return self._inliers / max(1, self.source_frame.size())

    def correspondences(self, values):
        out = np.zeros(self.source_frame.size(), dtype=np.int32)
        T = pose12(self.calc_delta(values))
        check(lib().glim_amd_gicp_correspondences(self.target_tree, self.source_frame._h, _dp(T), self.max_correspondence_distance, _ip(out)),
              "glim_amd_gicp_correspondences")
        return out

    def close(self):
        if self._own_tree and self.target_tree:
            lib().glim_amd_nn_index_destroy(self.target_tree)
            self.target_tree = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def median_distance(points, max_scan_count=256):
    """gtsam_points::median_distance (odometry_estimation_gpu.cpp:91, global_mapping.cpp:239): host side, as in the reference."""
    p = np.asarray(points, dtype=np.float64)[:, :3]
    n = len(p)
    if n == 0:
        return 0.0
    step = 1 if n < max_scan_count else n // max_scan_count
    d = np.sort(np.sqrt((p[::step] ** 2).sum(1)))
    return float(d[len(d) // 2])


def adaptive_voxel_resolution(dist_median, voxel_resolution, voxel_resolution_max, voxel_resolution_dmin, voxel_resolution_dmax):
    """base_resolution of odometry_estimation_gpu.cpp:92-93 / global_mapping.cpp:240-241."""
    p = max(0.0, min(1.0, (dist_median - voxel_resolution_dmin) / (voxel_resolution_dmax - voxel_resolution_dmin)))
    return voxel_resolution + p * (voxel_resolution_max - voxel_resolution)


def _pack_frames(poses, frames_points, frames_covs):
    """Host layouts of the reference for a list of frames: poses n x 12, Vector4d points, column-major Matrix4d covariances."""
    nf = len(poses)
    P12 = np.ascontiguousarray(np.stack([pose12(T) for T in poses])) if nf else np.zeros((1, 12))
    p4s, c16s = [], []
    for p, c in zip(frames_points, frames_covs):
        p = np.asarray(p, dtype=np.float64).reshape(-1, np.shape(p)[-1] if np.ndim(p) == 2 else 3)
        n = p.shape[0]
        p4 = np.ones((n, 4))
        p4[:, : p.shape[1]] = p
        c = np.asarray(c, dtype=np.float64)
        c16 = np.zeros((n, 4, 4))
        if c.size == n * 9:
            c16[:, :3, :3] = np.transpose(c.reshape(n, 3, 3), (0, 2, 1))
        else:
            c16[:] = c.reshape(n, 4, 4)
        p4s.append(p4)
        c16s.append(np.ascontiguousarray(c16.reshape(n, 16)))
    sizes = np.array([len(p) for p in p4s], dtype=np.int64)
    dp = C.POINTER(C.c_double)
    pp = (dp * max(nf, 1))(*[_dp(p) for p in p4s])
    cp = (dp * max(nf, 1))(*[_dp(c) for c in c16s])
    return dict(nf=nf, P12=P12, p4s=p4s, c16s=c16s, sizes=sizes, pp=pp, cp=cp)


def merge_frames(poses, frames_points, frames_covs, downsample_resolution, target_num_points=-1, seed=0, block_size=1024, ctx=None, packed=None):
    """gtsam_points::merge_frames (sub_mapping.cpp:480-497) on the device.  poses[f]: T_origin_frame; frames_points[f]: N_f x 3 / x 4;
    frames_covs[f]: N_f x 3 x 3 / 4 x 4.  Returns the merged PointCloudGPU (points + covariances)."""
    ctx = ctx or default_context()
    k = packed if packed is not None else _pack_frames(poses, frames_points, frames_covs)
    h = C.c_void_p()
    check(lib().glim_amd_merge_frames(ctx._h, k["nf"], _dp(k["P12"]), k["pp"], k["cp"], k["sizes"].ctypes.data_as(C.POINTER(C.c_int64)),
                                      float(downsample_resolution), int(target_num_points), int(block_size), int(seed), C.byref(h)), "glim_amd_merge_frames")
    return PointCloudGPU(h, ctx)


def debug_sort_pairs(keys, vals=None, bits=64, ctx=None):
    """The stable device radix sort behind the preprocessing (parity / debug)."""
    ctx = ctx or default_context()
    k = np.ascontiguousarray(keys, dtype=np.uint64)
    v = None if vals is None else np.ascontiguousarray(vals, dtype=np.uint32)
    ko, vo = np.zeros_like(k), np.zeros(len(k), dtype=np.uint32)
    u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    check(lib().glim_amd_debug_sort_pairs(ctx._h, len(k), int(bits), k.ctypes.data_as(u64p), None if v is None else v.ctypes.data_as(u32p),
                                          ko.ctypes.data_as(u64p), vo.ctypes.data_as(u32p)), "glim_amd_debug_sort_pairs")
    return ko, vo


def expand_compact(compact, T_target_source, flags):
    c = np.ascontiguousarray(compact, dtype=np.float64).reshape(_lib.COMPACT_DOUBLES)
    T = pose12(T_target_source)
    L = Linearized6()
    check(lib().glim_amd_expand_compact(_dp(c), _dp(T), int(flags), C.byref(L)), "glim_amd_expand_compact")
    return _lin_to_dict(L)


def overlap_gpu(target_voxelmaps, source, deltas, ctx=None):
    """gtsam_points::overlap_gpu: single (voxelmap, source, delta) or multi-target (voxelmaps[], source, deltas[])."""
    if isinstance(target_voxelmaps, GaussianVoxelMapGPU):
        target_voxelmaps, deltas = [target_voxelmaps], [deltas]
    ctx = ctx or source.ctx
    hs = (C.c_void_p * len(target_voxelmaps))(*[m._h.value for m in target_voxelmaps])
    T = np.ascontiguousarray(np.stack([pose12(d) for d in deltas]))
    out = C.c_double()
    check(lib().glim_amd_overlap(ctx._h, len(target_voxelmaps), hs, _dp(T), source._h, C.byref(out)), "glim_amd_overlap")
    return out.value


overlap_auto = overlap_gpu


def _overlap_args(queries):
    nt = np.array([len(q[0]) for q in queries], dtype=np.int32)
    maps = [m._h.value for q in queries for m in q[0]]
    hs = (C.c_void_p * len(maps))(*maps)
    srcs = (C.c_void_p * len(queries))(*[q[1]._h.value for q in queries])
    T = np.ascontiguousarray(np.stack([pose12(d) for q in queries for d in q[2]]))
    return nt, hs, srcs, T


def overlap_profile(queries, iters=200, ctx=None):
    """microseconds per glim_amd_overlap_batch call answering `queries` ([(voxelmaps[], source, deltas[]), ...]), timed inside the library."""
    ctx = ctx or queries[0][1].ctx
    nt, hs, srcs, T = _overlap_args(queries)
    us = C.c_float()
    check(lib().glim_amd_overlap_profile(ctx._h, len(queries), _ip(nt), hs, _dp(T), srcs, int(iters), C.byref(us)), "glim_amd_overlap_profile")
    return us.value


def profile_fresh_sets(factors, T_target_source, iters=200, ctx=None):
    """GLIM's live pattern -- a fresh NonlinearFactorSetGPU per linearisation: create, add every factor, linearize, destroy -- timed inside
    the library; microseconds per iteration.  factors: IntegratedVGICPFactorGPU objects; T_target_source: n x 12."""
    ctx = ctx or factors[0].source.ctx
    n = len(factors)
    maps = (C.c_void_p * n)(*[f.target_voxelmap._h.value for f in factors])
    srcs = (C.c_void_p * n)(*[f.source._h.value for f in factors])
    flags = (C.c_uint32 * n)(*[f.flags() for f in factors])
    T = np.ascontiguousarray(T_target_source, dtype=np.float64)
    us = C.c_float()
    check(lib().glim_amd_factor_set_profile_fresh(ctx._h, n, maps, srcs, flags, _dp(T), int(iters), C.byref(us)), "glim_amd_factor_set_profile_fresh")
    return us.value


def profile_fresh_sets_samples(factors, T_target_source, iters=200, gap_us=0.0, ctx=None):
    """profile_fresh_sets with every iteration timed on its own: microseconds per iteration (array of `iters`); `gap_us` of host work between
    two iterations."""
    ctx = ctx or factors[0].source.ctx
    n = len(factors)
    maps = (C.c_void_p * n)(*[f.target_voxelmap._h.value for f in factors])
    srcs = (C.c_void_p * n)(*[f.source._h.value for f in factors])
    flags = (C.c_uint32 * n)(*[f.flags() for f in factors])
    T = np.ascontiguousarray(T_target_source, dtype=np.float64)
    out = np.zeros(int(iters), dtype=np.float32)
    check(lib().glim_amd_factor_set_profile_fresh_samples(ctx._h, n, maps, srcs, flags, _dp(T), int(iters), float(gap_us), out.ctypes.data_as(C.POINTER(C.c_float))),
          "glim_amd_factor_set_profile_fresh_samples")
    return out


def overlap_gpu_batch(queries, ctx=None):
    """Many overlap_gpu calls in ONE launch (the keyframe-selection loops of odometry_estimation_gpu.cpp:262-281 issue K of them back to
    back): queries = [(target_voxelmaps[], source, deltas[]), ...] -> list of overlaps."""
    if not queries:
        return []
    ctx = ctx or queries[0][1].ctx
    nt, hs, srcs, T = _overlap_args(queries)
    out = np.zeros(len(queries), dtype=np.float64)
    check(lib().glim_amd_overlap_batch(ctx._h, len(queries), _ip(nt), hs, _dp(T), srcs, _dp(out)), "glim_amd_overlap_batch")
    return out.tolist()
