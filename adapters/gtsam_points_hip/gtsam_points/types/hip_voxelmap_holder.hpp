// gtsam_points/types/hip_voxelmap_holder.hpp -- NOT an upstream header: the one thing the device-backed voxel maps of this tree have in common.
// Both class names GLIM constructs -- GaussianVoxelMapGPU (odometry_estimation_gpu.cpp:103, sub_mapping.cpp:398, global_mapping.cpp:265,747) and
// GaussianVoxelMapCPU (odometry_estimation_cpu.cpp:66, sub_mapping.cpp:409, global_mapping.cpp:275,757, global_mapping_pose_graph.cpp:276) -- hold a
// glim_amd::GaussianVoxelMapGPU (a glim_amd_voxelmap handle); factors and overlap_* reach it through this interface, whatever the name.
#pragma once

#include <memory>
#include <stdexcept>

#include <gtsam_points/types/gaussian_voxelmap.hpp>

#include <glim_amd_gtsam.hpp>

namespace gtsam_points {

struct HipVoxelMapHolder {
  virtual ~HipVoxelMapHolder() {}
  virtual glim_amd::GaussianVoxelMapGPU::ConstPtr device() const = 0;
};

inline glim_amd::GaussianVoxelMapGPU::ConstPtr device_map(const GaussianVoxelMap::ConstPtr& voxelmap) {
  const auto* holder = dynamic_cast<const HipVoxelMapHolder*>(voxelmap.get());
  if (!holder) throw std::runtime_error("a HIP factor / overlap_gpu needs a device-backed voxel map (GaussianVoxelMapGPU, or GaussianVoxelMapCPU of this include tree)");
  return holder->device();
}

}  // namespace gtsam_points
