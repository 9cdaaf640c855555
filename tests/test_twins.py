"""Drop-in reach of the kernels that are NOT behind a gtsam_points GPU header: GLIM's in-tree front end and the two
gtsam_points CPU entry points the mapping modules call.

  * glim::CloudCovarianceEstimation / glim::CloudDeskewing / glim::CloudPreprocessor -- adapters/glim/cloud_*_hip.cpp are translation units with the
    reference's OWN class interfaces (its headers, its member signatures) implemented on the C ABI: a HIP build of libglim compiles them in place of
    src/glim/common/cloud_covariance_estimation.cpp, cloud_deskewing.cpp and src/glim/preprocess/cloud_preprocessor.cpp, and every caller
    (odometry_estimation_imu.cpp:189,313-320, sub_mapping.cpp:364-374, glim_ros) reaches K1 / K2 / K8 / K9 without an edit.
  * gtsam_points::IntegratedGICPFactor and gtsam_points::merge_frames -- shim headers at gtsam_points' own paths
    (adapters/gtsam_points_hip/gtsam_points/factors/integrated_gicp_factor.hpp, .../types/point_cloud_cpu.hpp): sub_mapping.cpp:202,480-497 and
    global_mapping.cpp:400 reach K10 / K11, the reference sources compiled unmodified.

CPU tests (where /root/reference exists): the compiles, and `nm` on the objects.  GPU test: the SAME C entry points (oracle/ref_shim.cpp,
ref_preprocess_shim.cpp) built once over the reference's translation units (oracle/_ref/libglim_ref.so) and once over the twins
(oracle/_ref/libglim_twin.so, `make -C oracle twin`), run side by side on the same arrays."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
INCLUDES = ["-I" + os.path.join(ROOT, "adapters", "gtsam_points_hip"), "-I" + os.path.join(ROOT, "adapters", "gtsam"), "-I" + os.path.join(ROOT, "include"),
            "-I" + os.path.join(ROOT, "tests", "cpp", "glim_standin"), "-I" + os.path.join(REF, "include")]
have_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "glim", "mapping")), reason="the reference tree is not present on this machine")


def _nm(obj):
    return subprocess.run(["nm", "-C", obj], capture_output=True, text=True, check=True).stdout


@have_reference
@pytest.mark.parametrize("unit, defines, needs", [
    ("cloud_covariance_estimation_hip.cpp",
     ["T glim::CloudCovarianceEstimation::estimate(", "T glim::CloudCovarianceEstimation::regularize(", "T glim::CloudCovarianceEstimation::CloudCovarianceEstimation(int)"],
     ["U glim_amd_cloud_estimate_covariances", "U glim_amd_cloud_set_neighbors", "U glim_amd_cloud_create_exact"]),
    ("cloud_deskewing_hip.cpp", ["T glim::CloudDeskewing::deskew(Eigen::Isometry3d const&, Eigen::Matrix<double, 3, 1> const&", "T glim::CloudDeskewing::deskew(Eigen::Isometry3d const&, std::vector<double"],
     ["U glim_amd_cloud_create_deskewed", "U glim_amd_cloud_download_frame"]),
    ("cloud_preprocessor_hip.cpp",
     ["T glim::CloudPreprocessor::preprocess(", "T glim::CloudPreprocessor::preprocess_impl(", "T glim::CloudPreprocessor::find_neighbors(", "T glim::CloudPreprocessorParams::CloudPreprocessorParams()"],
     ["U glim_amd_preprocess", "U glim_amd_cloud_find_neighbors", "U glim_amd_cloud_download_frame"]),
    ("merge_frames_hip.cpp", ["T gtsam_points::merge_frames(std::vector<Eigen::Isometry3d"], ["U glim_amd_merge_frames", "U glim_amd_cloud_download_merged"]),
])
def test_twin_translation_units_define_the_reference_interfaces_over_the_c_abi(tmp_path, unit, defines, needs):
    """Compiled against GLIM's REAL headers (/root/reference/include): the member signatures are the reference's, the bodies call the C ABI."""
    obj = str(tmp_path / "twin.o")
    subprocess.check_call(["g++", "-std=c++17", "-O0", "-w", "-c"] + INCLUDES + [os.path.join(ROOT, "adapters", "glim", unit), "-o", obj])
    syms = _nm(obj)
    for d in defines + needs:
        assert d in syms, (unit, d)


@have_reference
@pytest.mark.parametrize("unit, wants", [
    ("sub_mapping", ["U glim_amd_gicp_linearize", "U glim_amd_merge_frames", "U glim_amd_nn_index_create", "gtsam_points::merge_frames_hip(",
                     "gtsam_points::IntegratedGICPFactor::IntegratedGICPFactor(unsigned long, unsigned long, std::shared_ptr<gtsam_points::PointCloud const> const&"]),
    ("global_mapping", ["U glim_amd_gicp_linearize", "U glim_amd_gicp_error", "U glim_amd_nn_index_create",
                        "gtsam_points::IntegratedGICPFactor::IntegratedGICPFactor(unsigned long, unsigned long, std::shared_ptr<gtsam_points::PointCloud const> const&"]),
])
def test_reference_mapping_sources_reach_gicp_and_merge_without_an_edit(tmp_path, unit, wants):
    """sub_mapping.cpp:202 (between factors), :496 (merge_frames) and global_mapping.cpp:400 (between factors), compiled UNMODIFIED with the shim tree in
    front: the objects reference the device GICP factor and the device merge."""
    obj = str(tmp_path / (unit + ".o"))
    subprocess.check_call(["g++", "-std=c++17", "-O0", "-w", "-c", "-DGTSAM_POINTS_USE_CUDA"] + INCLUDES + [os.path.join(REF, "src", "glim", "mapping", unit + ".cpp"), "-o", obj])
    syms = _nm(obj)
    for w in wants:
        assert w in syms, (unit, w)


# ---- side-by-side run of the twins and the compiled reference (GPU) ------------------------------------------------------------------------
def _twin_lib():
    path = os.path.join(ROOT, "oracle", "_ref", "libglim_twin.so")
    if os.path.isdir(os.path.join(REF, "include", "glim")):
        from glim_amd import _lib

        if not os.path.exists(_lib.LIB_PATH):
            _lib.build()
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "twin", "-s"], stdout=subprocess.DEVNULL)
    if not os.path.exists(path):
        return None
    from oracle.oracle import PreprocessParams

    L = C.CDLL(path)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    L.ref_covariance_estimate.restype = C.c_int
    L.ref_covariance_estimate.argtypes = [dp, C.c_int, ip, C.c_int, C.c_int, dp, dp, C.c_int]
    L.ref_deskew_constvel.argtypes = [dp, dp, dp, dp, dp, C.c_int, dp]
    L.ref_deskew_imu.argtypes = [dp, dp, dp, C.c_int, C.c_double, dp, dp, C.c_int, dp]
    L.ref_frontend.restype = C.c_int
    L.ref_frontend.argtypes = [dp, dp, dp, C.c_int, C.c_double, dp, dp, dp, dp, C.c_int, ip, C.c_int, C.c_int, dp, dp, dp, C.c_int]
    L.ref_preprocess.restype = C.c_int
    L.ref_preprocess.argtypes = [dp, dp, dp, C.c_int, C.POINTER(PreprocessParams), dp, dp, dp, ip, dp, C.c_int]
    return L


def test_twin_library_builds_and_exports_the_same_entry_points():
    if not os.path.isdir(os.path.join(REF, "include", "glim")):
        pytest.skip("the reference headers are not present on this machine")
    L = _twin_lib()
    assert L is not None
    for name in ("ref_covariance_estimate", "ref_deskew_constvel", "ref_deskew_imu", "ref_frontend", "ref_preprocess"):
        assert hasattr(L, name)
    syms = subprocess.run(["nm", "-CD", os.path.join(ROOT, "oracle", "_ref", "libglim_twin.so")], capture_output=True, text=True, check=True).stdout
    for w in ("U glim_amd_cloud_estimate_covariances", "U glim_amd_preprocess", "U glim_amd_cloud_create_deskewed", "T glim::CloudCovarianceEstimation::estimate("):
        assert w in syms, w


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


@pytest.mark.gpu
def test_twins_run_like_the_compiled_reference(orc):
    """glim::CloudPreprocessor::preprocess, glim::CloudDeskewing::deskew (both overloads), glim::CloudCovarianceEstimation::estimate and the
    composed front end of odometry_estimation_imu.cpp:313-320 -- the reference's translation units and their HIP twins behind the same C entry
    points, on a raw 131 072-pt scan: surviving points, times, intensities and neighbour lists equal, deskewed points bit-exact in FP64,
    normals / covariances within the 1e-5 gate (FP32 storage on the device)."""
    from glim_amd import synth
    from oracle import oracle as o

    ref, twin = o.ref_lib(), _twin_lib()
    if ref is None or twin is None:
        pytest.skip("oracle/_ref libraries are not available (neither prebuilt nor buildable here)")
    scene = synth.Scene.default()
    T = synth.arc_trajectory(1)[0]
    pts = synth.scan(scene, T, synth.lidar_directions(128, 1024), 7)
    n = len(pts)
    rng = np.random.default_rng(4)
    times = np.sort(rng.uniform(0.0, 0.1, n))
    inten = rng.uniform(0.0, 255.0, n)
    p4 = np.ones((n, 4))
    p4[:, :3] = pts[:, :3]
    prm = o.preprocess_params(seed=11, k_correspondences=10)

    def preprocess(L):
        op, ot, oi = np.zeros((n, 4)), np.zeros(n), np.zeros(n)
        on = np.zeros((n, 10), dtype=np.int32)
        meta = np.zeros(2)
        m = L.ref_preprocess(_dp(p4), _dp(times), _dp(inten), n, C.byref(prm), _dp(op), _dp(ot), _dp(oi), _ip(on), _dp(meta), 2)
        assert m > 100
        return op[:m].copy(), ot[:m].copy(), oi[:m].copy(), on[:m].copy(), meta.copy()

    a, b = preprocess(ref), preprocess(twin)
    assert len(a[0]) == len(b[0])
    for x, y, what in zip(a, b, ("points", "times", "intensities", "neighbors", "scan_end_time / k")):
        np.testing.assert_array_equal(x, y, err_msg=what)
    q4, qt, _, qn, _ = a
    m = len(q4)
    # deskew: constant twist and IMU poses
    T_imu_lidar = o.pose12(o.se3_exp([0.02, -0.01, 0.5, 0.3, -0.2, 0.1]))
    v, w = np.array([1.5, -0.3, 0.1]), np.array([0.05, -0.1, 0.4])
    imu_t = np.linspace(-0.01, 0.12, 28)
    imu_T = np.stack([o.pose12(o.se3_exp(np.array([0.03, -0.02, 0.3, 1.2, -0.4, 0.05]) * t * 8)) for t in imu_t])

    def deskews(L):
        c, i = np.zeros((m, 4)), np.zeros((m, 4))
        L.ref_deskew_constvel(_dp(T_imu_lidar), _dp(v), _dp(w), _dp(qt), _dp(q4), m, _dp(c))
        L.ref_deskew_imu(_dp(T_imu_lidar), _dp(imu_t), _dp(imu_T), len(imu_t), 0.0, _dp(qt), _dp(q4), m, _dp(i))
        return c, i

    (rc, ri), (tc, ti) = deskews(ref), deskews(twin)
    np.testing.assert_array_equal(rc, tc, err_msg="constant-twist deskew")
    np.testing.assert_array_equal(ri, ti, err_msg="IMU-pose deskew")
    assert np.abs(ri[:, :3] - q4[:, :3]).max() > 1e-3  # (the motion does move the points)

    # covariance + normal, and the composed front end
    def covs(L):
        nr, cv = np.zeros((m, 4)), np.zeros((m, 16))
        assert L.ref_covariance_estimate(_dp(q4), m, _ip(qn), 10, 10, _dp(nr), _dp(cv), 2) == 0
        fp, fn, fc = np.zeros((m, 4)), np.zeros((m, 4)), np.zeros((m, 16))
        assert L.ref_frontend(_dp(T_imu_lidar), _dp(imu_t), _dp(imu_T), len(imu_t), 0.0, None, None, _dp(qt), _dp(q4), m, _ip(qn), 10, 1, _dp(fp), _dp(fn), _dp(fc), 2) == 0
        return nr, cv, fp, fn, fc

    r, t = covs(ref), covs(twin)
    np.testing.assert_array_equal(r[2], t[2], err_msg="front end: IMU-frame points")
    for (rn, rcv, pp), (tn, tcv), what in (((r[0], r[1], q4), (t[0], t[1]), "estimate"), ((r[3], r[4], r[2]), (t[3], t[4]), "front end")):
        # normals: equal within the FP32 storage of the device; the SIGN is `p . n <= 0` (cloud_covariance_estimation.cpp:98-101), which a point seen
        # at grazing incidence (|p . n| below FP32 resolution) may resolve either way -- the covariance I - 0.999 n n^T does not depend on it
        same, flipped = np.abs(rn - tn).max(axis=1), np.abs(rn + tn).max(axis=1)
        assert np.minimum(same, flipped).max() < 1e-5, what
        flips = flipped < same
        grazing = np.abs(np.einsum("ij,ij->i", pp[:, :3], rn[:, :3])) <= 1e-5 * np.linalg.norm(pp[:, :3], axis=1)
        assert not (flips & ~grazing).any(), (what, int(flips.sum()))
        rel = np.abs(rcv - tcv).max(axis=1) / np.abs(rcv).max(axis=1)
        assert (rel > 1e-5).sum() == 0, (what, float(rel.max()))
    print(f"twins vs compiled reference on {n} raw -> {m} points: preprocess equal, deskew bit-exact, covariance max rel "
          f"{float((np.abs(r[1] - t[1]).max(axis=1) / np.abs(r[1]).max(axis=1)).max()):.2e}")
