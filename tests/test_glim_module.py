"""The outer plugin ABI: GLIM's own GPU module sources -- odometry_estimation_gpu.cpp, sub_mapping.cpp, global_mapping.cpp -- compiled
UNMODIFIED against the drop-in include tree adapters/gtsam_points_hip, and that tree driven on the GPU the way the modules drive gtsam_points.

  * compile test (only where /root/reference exists): /root/reference/src/glim/odometry/odometry_estimation_gpu.cpp + adapters/glim/
    odometry_estimation_hip_create.cpp -> object files that reference this library's symbols and export create_odometry_estimation_module.
    Third-party headers are stand-ins (tests/cpp/glim_standin: Eigen / GTSAM / spdlog / OpenCV / the CPU half of gtsam_points are not
    installed here); GLIM's headers are the real ones.  Linking needs libglim + GTSAM and is out of reach in this image.
  * run test (GPU): tests/cpp/test_shim.cpp -- PointCloudGPU::clone, GaussianVoxelMapGPU(res, 8192 * 2, 10, 1e-3, *stream), the six-argument
    IntegratedVGICPFactorGPU constructors, NonlinearFactorSetGPU, overlap_gpu -- against the plain C ABI on the same data.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
INCLUDES = ["-I" + os.path.join(ROOT, "adapters", "gtsam_points_hip"), "-I" + os.path.join(ROOT, "adapters", "gtsam"), "-I" + os.path.join(ROOT, "include"),
            "-I" + os.path.join(ROOT, "tests", "cpp", "glim_standin")]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "glim", "odometry")), reason="the reference tree is not present on this machine")
def test_reference_odometry_module_source_compiles_unmodified_against_the_hip_headers(tmp_path):
    src = os.path.join(REF, "src", "glim", "odometry", "odometry_estimation_gpu.cpp")
    obj = str(tmp_path / "odometry_estimation_gpu.o")
    subprocess.check_call(["g++", "-std=c++17", "-O0", "-w", "-c"] + INCLUDES + ["-I" + os.path.join(REF, "include"), src, "-o", obj])
    syms = subprocess.run(["nm", "-C", obj], capture_output=True, text=True, check=True).stdout
    # the module's GPU call sites resolved to this library: six-argument factor constructors, overlap_gpu, the C ABI underneath
    assert "gtsam_points::IntegratedVGICPFactorGPU::IntegratedVGICPFactorGPU(unsigned long, unsigned long, std::shared_ptr<gtsam_points::GaussianVoxelMap const> const&" in syms
    assert "gtsam_points::IntegratedVGICPFactorGPU::IntegratedVGICPFactorGPU(gtsam::Pose3 const&" in syms
    assert "U glim_amd_factor_set_linearize" in syms and "U glim_amd_overlap" in syms and "U glim_amd_voxelmap_insert" in syms and "U glim_amd_cloud_create" in syms
    assert "glim::OdometryEstimationGPU::create_factors" in syms and "glim::OdometryEstimationGPU::update_keyframes_overlap" in syms
    create = str(tmp_path / "create.o")
    subprocess.check_call(["g++", "-std=c++17", "-O0", "-w", "-c"] + INCLUDES + ["-I" + os.path.join(REF, "include"),
                                                                                os.path.join(ROOT, "adapters", "glim", "odometry_estimation_hip_create.cpp"), "-o", create])
    assert " T create_odometry_estimation_module" in subprocess.run(["nm", create], capture_output=True, text=True, check=True).stdout


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "glim", "mapping")), reason="the reference tree is not present on this machine")
@pytest.mark.parametrize("unit, create, symbol, members", [
    ("sub_mapping", "sub_mapping_hip_create.cpp", "create_sub_mapping_module", ["glim::SubMapping::insert_keyframe", "glim::SubMapping::create_submap"]),
    ("global_mapping", "global_mapping_hip_create.cpp", "create_global_mapping_module",
     ["glim::GlobalMapping::create_matching_cost_factors", "glim::GlobalMapping::insert_submap"]),
])
def test_reference_mapping_module_sources_compile_unmodified_against_the_hip_headers(tmp_path, unit, create, symbol, members):
    """sub_mapping.cpp / global_mapping.cpp are part of libglim; their GPU branches sit inside #ifdef GTSAM_POINTS_USE_CUDA (sub_mapping.cpp:86-87,
    165-171, 300-310, 393-399; global_mapping.cpp:110, 253-266, 322-335, 448-466, 743-748, 860).  A HIP build defines that macro and puts the
    drop-in tree first: the reference translation units compile without an edit and their GPU call sites resolve to this library."""
    src = os.path.join(REF, "src", "glim", "mapping", unit + ".cpp")
    obj = str(tmp_path / (unit + ".o"))
    subprocess.check_call(["g++", "-std=c++17", "-O0", "-w", "-c", "-DGTSAM_POINTS_USE_CUDA"] + INCLUDES + ["-I" + os.path.join(REF, "include"), src, "-o", obj])
    syms = subprocess.run(["nm", "-C", obj], capture_output=True, text=True, check=True).stdout
    # the GPU branches were compiled (not preprocessed away) and resolved to the shim tree / the C ABI underneath
    assert "gtsam_points::IntegratedVGICPFactorGPU::IntegratedVGICPFactorGPU(unsigned long, unsigned long, std::shared_ptr<gtsam_points::GaussianVoxelMap const> const&" in syms
    assert "gtsam_points::PointCloudGPU::clone(gtsam_points::PointCloud const&, CUstream_st*)" in syms
    assert "gtsam_points::GaussianVoxelMapGPU::GaussianVoxelMapGPU(float, int, int, double, CUstream_st*)" in syms
    assert "gtsam_points::StreamTempBufferRoundRobin::StreamTempBufferRoundRobin(int)" in syms
    assert "U glim_amd_cloud_create" in syms and "U glim_amd_voxelmap_insert" in syms and "U glim_amd_factor_set_add" in syms
    if unit == "global_mapping":
        assert "gtsam_points::overlap_gpu(" in syms and "U glim_amd_overlap" in syms
    for m in members:
        assert m in syms, m
    entry = str(tmp_path / "create.o")
    subprocess.check_call(["g++", "-std=c++17", "-O0", "-w", "-c", "-DGTSAM_POINTS_USE_CUDA"] + INCLUDES + ["-I" + os.path.join(REF, "include"),
                                                                                                          os.path.join(ROOT, "adapters", "glim", create), "-o", entry])
    assert f" T {symbol}" in subprocess.run(["nm", entry], capture_output=True, text=True, check=True).stdout


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "glim", "util")), reason="the reference tree is not present on this machine")
@pytest.mark.parametrize("unit, wants", [
    ("util/debug.cpp", ["gtsam_points::cuda_device_names", "U glim_amd_device_count", "U glim_amd_device_info"]),
    ("viewer/memory_monitor.cpp", ["gtsam_points::cuda_mem_get_info", "U glim_amd_device_info"]),
])
def test_reference_device_info_call_sites_compile_unmodified(tmp_path, unit, wants):
    """The two remaining non-viewer files that name the GPU library: the system-info dump (src/glim/util/debug.cpp:81-89, cuda_device_names) and
    the memory monitor (src/glim/viewer/memory_monitor.cpp:36-47, cuda_mem_get_info) resolve to glim_amd_device_info through the shim tree."""
    obj = str(tmp_path / "unit.o")
    subprocess.check_call(["g++", "-std=c++17", "-O0", "-w", "-c", "-DGTSAM_POINTS_USE_CUDA"] + INCLUDES + ["-I" + os.path.join(REF, "include"),
                                                                                                          os.path.join(REF, "src", "glim", unit), "-o", obj])
    syms = subprocess.run(["nm", "-C", obj], capture_output=True, text=True, check=True).stdout
    for w in wants:
        assert w in syms, w


def _build_shim_test(tmp_path):
    from glim_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    exe = str(tmp_path / "test_shim")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-w"] + INCLUDES + [os.path.join(ROOT, "tests", "cpp", "test_shim.cpp"), "-o", exe,
                                                                           "-L" + os.path.join(ROOT, "glim_amd"), "-lglim_amd", "-Wl,-rpath," + os.path.join(ROOT, "glim_amd"),
                                                                           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    return exe


def test_shim_tree_compiles_and_links(tmp_path):
    _build_shim_test(tmp_path)


@pytest.mark.gpu
def test_shim_tree_runs_like_the_reference_call_sites(tmp_path):
    exe = _build_shim_test(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "test_shim OK" in out.stdout


# ---- the CPU-NAMED classes: gtsam_points/factors/integrated_vgicp_factor.hpp and gtsam_points/types/gaussian_voxelmap_cpu.hpp
# of the shim tree put GaussianVoxelMapCPU / IntegratedVGICPFactor on the device, so that the module BASELINE configs[0] names
# (config_odometry_cpu.json -> odometry_estimation_cpu.cpp), the loop-closure validator (global_mapping_pose_graph.cpp) and the enable_gpu = false
# branches of sub_mapping.cpp / global_mapping.cpp reach the HIP path without an edit.

@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "glim", "odometry")), reason="the reference tree is not present on this machine")
@pytest.mark.parametrize("unit, wants", [
    ("odometry/odometry_estimation_cpu.cpp",
     ["U glim_amd_voxelmap_set_lru_horizon", "U glim_amd_voxelmap_insert", "U glim_amd_factor_set_linearize", "U glim_amd_factor_set_error",
      "W gtsam_points::GaussianVoxelMapCPU::GaussianVoxelMapCPU(double)",
      "W gtsam_points::IntegratedVGICPFactor::IntegratedVGICPFactor(gtsam::Pose3 const&, unsigned long, std::shared_ptr<gtsam_points::GaussianVoxelMap const> const&",
      # the iVox GICP branch (registration_type "GICP") is NOT this library's path: it keeps resolving to libgtsam_points' class template
      "U gtsam_points::IntegratedGICPFactor_<gtsam_points::IncrementalVoxelMap<gtsam_points::FlatContainer>, gtsam_points::PointCloud>::IntegratedGICPFactor_(",
      "glim::OdometryEstimationCPU::create_factors", "glim::OdometryEstimationCPU::update_target"]),
    ("mapping/global_mapping_pose_graph.cpp",
     ["U glim_amd_voxelmap_insert", "U glim_amd_factor_set_linearize", "U glim_amd_gicp_linearize", "U glim_amd_nn_index_create",
      "W gtsam_points::GaussianVoxelMapCPU::GaussianVoxelMapCPU(double)",
      "W gtsam_points::IntegratedVGICPFactor::IntegratedVGICPFactor(gtsam::Pose3 const&, unsigned long, std::shared_ptr<gtsam_points::GaussianVoxelMap const> const&",
      "glim::GlobalMappingPoseGraph::"]),
])
def test_reference_cpu_named_call_sites_compile_unmodified_and_reach_the_hip_library(tmp_path, unit, wants):
    obj = str(tmp_path / "unit.o")
    subprocess.check_call(["g++", "-std=c++17", "-O0", "-w", "-c"] + INCLUDES + ["-I" + os.path.join(REF, "include"), os.path.join(REF, "src", "glim", unit), "-o", obj])
    syms = subprocess.run(["nm", "-C", obj], capture_output=True, text=True, check=True).stdout
    for w in wants:
        assert w in syms, w


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "glim", "mapping")), reason="the reference tree is not present on this machine")
def test_the_cpu_branches_of_the_mapping_modules_resolve_to_the_device_classes(tmp_path):
    """Compiled WITHOUT GTSAM_POINTS_USE_CUDA (a CPU-only libglim): sub_mapping.cpp:291,409 and global_mapping.cpp:275,341,457,757,867 construct
    GaussianVoxelMapCPU / IntegratedVGICPFactor -- with the shim tree in front, the device-backed ones."""
    for unit in ("sub_mapping", "global_mapping"):
        obj = str(tmp_path / (unit + ".o"))
        subprocess.check_call(["g++", "-std=c++17", "-O0", "-w", "-c"] + INCLUDES + ["-I" + os.path.join(REF, "include"),
                                                                                    os.path.join(REF, "src", "glim", "mapping", unit + ".cpp"), "-o", obj])
        syms = subprocess.run(["nm", "-C", obj], capture_output=True, text=True, check=True).stdout
        assert "W gtsam_points::GaussianVoxelMapCPU::GaussianVoxelMapCPU(double)" in syms
        assert "gtsam_points::IntegratedVGICPFactor::IntegratedVGICPFactor(unsigned long, unsigned long, std::shared_ptr<gtsam_points::GaussianVoxelMap const> const&" in syms
        assert "U glim_amd_voxelmap_insert" in syms and "U glim_amd_factor_set_linearize" in syms
        assert "gtsam_points::GaussianVoxelMapGPU::GaussianVoxelMapGPU(float" not in syms  # (the GPU-named branch was preprocessed away)


def _build_cpu_names_test(tmp_path):
    from glim_amd import _lib
    from oracle import oracle

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    oracle.build()
    exe = str(tmp_path / "test_shim_cpu_names")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-w"] + INCLUDES + [os.path.join(ROOT, "tests", "cpp", "test_shim_cpu_names.cpp"), "-o", exe,
                                                                           "-L" + os.path.join(ROOT, "glim_amd"), "-lglim_amd", "-L" + os.path.join(ROOT, "oracle"), "-lvgicp_oracle",
                                                                           "-Wl,-rpath," + os.path.join(ROOT, "glim_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
                                                                           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    return exe


def test_cpu_named_shims_compile_and_link(tmp_path):
    _build_cpu_names_test(tmp_path)


@pytest.mark.gpu
def test_cpu_odometry_loop_through_the_cpu_named_shims_tracks_the_oracle(tmp_path):
    """tests/cpp/test_shim_cpu_names.cpp: GaussianVoxelMapCPU + set_lru_horizon + per-frame insert, IntegratedVGICPFactor(Pose3(), X(i), map, frame), the
    optimiser loop of odometry_estimation_cpu.cpp:112-149 on BASELINE configs[0]'s sizes (16 384-pt frame, 1.0 m voxels, <= 8 iterations): every
    Gauss-Newton step within 1e-4 of the CPU oracle's, the final pose within 1e-4, error() with correspondences at the evaluation pose."""
    exe = _build_cpu_names_test(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "test_shim_cpu_names OK" in out.stdout
