// libsub_mapping_hip.so -- plugin entry of GLIM's sub-mapping module for a HIP build.
//
// GLIM's SubMapping (src/glim/mapping/sub_mapping.cpp) is part of libglim itself; its GPU branches -- PointCloudGPU::clone (:168, :393),
// GaussianVoxelMapGPU (:398-399), StreamTempBufferRoundRobin(8) + the six-argument IntegratedVGICPFactorGPU (:86-87, :300-310) -- sit inside
// #ifdef GTSAM_POINTS_USE_CUDA.  A HIP build of libglim compiles that file UNMODIFIED with -DGTSAM_POINTS_USE_CUDA and
// adapters/gtsam_points_hip in front of the include path (tests/test_glim_module.py does exactly that compile).  This file is the twin of
// src/glim/mapping/sub_mapping_create.cpp:3-6 (same exported symbol) that also registers the HIP linearisation hook, which glim_ros
// registers for the CUDA build.  Select it with "so_name": "libsub_mapping_hip.so" in config_sub_mapping.json.
#include <glim/mapping/sub_mapping.hpp>

#include <glim_amd_gtsam.hpp>

extern "C" glim::SubMappingBase* create_sub_mapping_module() {
  static const bool hook_registered = (glim_amd::register_linearization_hook(), true);
  (void)hook_registered;
  glim::SubMappingParams params;
  return new glim::SubMapping(params);
}
