// gtsam_points/factors/integrated_vgicp_factor_gpu.hpp, MI355X edition: the constructors GLIM calls, unchanged --
//   IntegratedVGICPFactorGPU(target_key, source_key, voxelmap, frame, stream, buffer)        odometry_estimation_gpu.cpp:144, sub_mapping.cpp:307,
//                                                                                             global_mapping.cpp:335,466,860
//   IntegratedVGICPFactorGPU(fixed_target_pose, source_key, voxelmap, frame, stream, buffer)  odometry_estimation_gpu.cpp:161
// taking the base-class pointers the call sites hold (GaussianVoxelMap::ConstPtr / PointCloud::ConstPtr) and the trailing (stream, buffer)
// pair: the stream handle names the context (stream pool) of the module's StreamTempBufferRoundRobin, on which a set of such factors runs.  Everything else -- error / linearize / clone /
// set_enable_surface_validation / get_fixed_target_pose / memory_usage[_gpu] / the batch protocol -- is glim_amd::IntegratedVGICPFactorHIP.
#pragma once

#include <memory>

#include <gtsam_points/cuda/stream_temp_buffer_roundrobin.hpp>
#include <gtsam_points/types/gaussian_voxelmap_gpu.hpp>
#include <gtsam_points/types/point_cloud_gpu.hpp>

#include <glim_amd_gtsam.hpp>

namespace gtsam_points {

class IntegratedVGICPFactorGPU : public glim_amd::IntegratedVGICPFactorHIP {
public:
  using shared_ptr = std::shared_ptr<IntegratedVGICPFactorGPU>;

  IntegratedVGICPFactorGPU(gtsam::Key target_key, gtsam::Key source_key, const GaussianVoxelMap::ConstPtr& target, const PointCloud::ConstPtr& source,
                           CUstream_st* stream = nullptr, std::shared_ptr<TempBufferManager> /*temp_buffer*/ = nullptr)
  : glim_amd::IntegratedVGICPFactorHIP(target_key, source_key, device_map(target), device_cloud(source), stream ? glim_amd::context_of(stream) : nullptr) {}

  IntegratedVGICPFactorGPU(const gtsam::Pose3& fixed_target_pose, gtsam::Key source_key, const GaussianVoxelMap::ConstPtr& target, const PointCloud::ConstPtr& source,
                           CUstream_st* stream = nullptr, std::shared_ptr<TempBufferManager> /*temp_buffer*/ = nullptr)
  : glim_amd::IntegratedVGICPFactorHIP(fixed_target_pose, source_key, device_map(target), device_cloud(source), stream ? glim_amd::context_of(stream) : nullptr) {}
};

}  // namespace gtsam_points
