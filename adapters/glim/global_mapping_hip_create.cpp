// libglobal_mapping_hip.so -- plugin entry of GLIM's global-mapping module for a HIP build.
//
// GLIM's GlobalMapping (src/glim/mapping/global_mapping.cpp) is part of libglim itself; its GPU branches -- StreamTempBufferRoundRobin(64)
// (:110), PointCloudGPU::clone (:253, :260, :743), GaussianVoxelMapGPU (:265-266, :747-748), overlap_gpu / overlap_auto (:322, :448) and the
// six-argument IntegratedVGICPFactorGPU (:335, :466, :860) -- sit inside #ifdef GTSAM_POINTS_USE_CUDA.  A HIP build of libglim compiles that
// file UNMODIFIED with -DGTSAM_POINTS_USE_CUDA and adapters/gtsam_points_hip in front of the include path (tests/test_glim_module.py).
// This file is the twin of src/glim/mapping/global_mapping_create.cpp:3-6 (same exported symbol) that also registers the HIP
// linearisation hook.  Select it with "so_name": "libglobal_mapping_hip.so" in config_global_mapping.json.
#include <glim/mapping/global_mapping.hpp>

#include <glim_amd_gtsam.hpp>

extern "C" glim::GlobalMappingBase* create_global_mapping_module() {
  static const bool hook_registered = (glim_amd::register_linearization_hook(), true);
  (void)hook_registered;
  glim::GlobalMappingParams params;
  return new glim::GlobalMapping(params);
}
