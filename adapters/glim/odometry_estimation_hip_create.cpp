// libodometry_estimation_hip.so -- the HIP-backed twin of GLIM's libodometry_estimation_gpu.so.
//
// The module's logic is GLIM's own src/glim/odometry/odometry_estimation_gpu.cpp, compiled UNMODIFIED with adapters/gtsam_points_hip in front of
// the include path (its gtsam_points GPU types then resolve to this library); this file adds the plugin entry point
// (src/glim/odometry/odometry_estimation_gpu_create.cpp:3-6 exports the same symbol) and registers the HIP linearisation hook, which in a live
// system glim_ros registers for the CUDA build.  Select it with  "so_name": "libodometry_estimation_hip.so"
// in config_odometry.json (config/config_odometry_gpu.json:41).  Build wiring: adapters/glim/CMakeLists.txt.
#include <glim/odometry/odometry_estimation_gpu.hpp>

#include <glim_amd_gtsam.hpp>

extern "C" glim::OdometryEstimationBase* create_odometry_estimation_module() {
  static const bool hook_registered = (glim_amd::register_linearization_hook(), true);
  (void)hook_registered;
  glim::OdometryEstimationGPUParams params;
  // the module's CUDAStream / StreamTempBufferRoundRobin members (odometry_estimation_gpu.cpp:76-77) are created inside this constructor call:
  // the odometry's stream pools get the device's greatest stream priority, so that its 25 us linearisations are dispatched ahead of the
  // mapping threads' millisecond kernels (async_sub_mapping.cpp:8, async_global_mapping.cpp:24 run beside it on the same device)
  glim_amd::ScopedStreamPriority odometry_first(1);
  return new glim::OdometryEstimationGPU(params);
}
