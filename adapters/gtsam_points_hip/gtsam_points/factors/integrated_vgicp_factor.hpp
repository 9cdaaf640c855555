// gtsam_points/factors/integrated_vgicp_factor.hpp, MI355X edition: the CPU-NAMED VGICP factor on the device, with the constructors GLIM calls --
//   IntegratedVGICPFactor(fixed_target_pose, source_key, voxelmap, frame)     odometry_estimation_cpu.cpp:107, global_mapping_pose_graph.cpp:406
//   IntegratedVGICPFactor(target_key, source_key, voxelmap, frame)            sub_mapping.cpp:291, global_mapping.cpp:341,457,867
// and set_num_threads (odometry_estimation_cpu.cpp:108; a no-op: the factor is one fused kernel), inlier_fraction
// (global_mapping_pose_graph.cpp:417), error / linearize / clone / dim of gtsam::NonlinearFactor.  `dynamic_cast<IntegratedVGICPFactor*>` at
// global_mapping.cpp:586 sees this class (the factors it inspects were made here).
// Semantics are the CPU factor's, not the GPU factor's: error(values) finds its correspondences AT `values` (the GPU factor
// evaluates with the correspondences frozen at its last linearisation point) -- which is what gtsam's LM loop of
// odometry_estimation_cpu.cpp:116-149 compares between iterations.
// The voxel map must be device-backed: a GaussianVoxelMapCPU of this include tree or a GaussianVoxelMapGPU.
#pragma once

#include <memory>

#include <gtsam_points/types/gaussian_voxelmap_cpu.hpp>
#include <gtsam_points/types/point_cloud_gpu.hpp>

#include <glim_amd_gtsam.hpp>

namespace gtsam_points {

class IntegratedVGICPFactor : public glim_amd::IntegratedVGICPFactorHIP {
public:
  using shared_ptr = std::shared_ptr<IntegratedVGICPFactor>;

  IntegratedVGICPFactor(gtsam::Key target_key, gtsam::Key source_key, const GaussianVoxelMap::ConstPtr& target, const PointCloud::ConstPtr& source)
  : glim_amd::IntegratedVGICPFactorHIP(target_key, source_key, device_map(target), device_cloud(source), nullptr) {
    set_frozen_error(false);
  }
  IntegratedVGICPFactor(const gtsam::Pose3& fixed_target_pose, gtsam::Key source_key, const GaussianVoxelMap::ConstPtr& target, const PointCloud::ConstPtr& source)
  : glim_amd::IntegratedVGICPFactorHIP(fixed_target_pose, source_key, device_map(target), device_cloud(source), nullptr) {
    set_frozen_error(false);
  }
  gtsam::NonlinearFactor::shared_ptr clone() const override {
    auto f = std::make_shared<IntegratedVGICPFactor>(*this);
    f->reset_impl_clone();
    return f;
  }
  void set_num_threads(int /*n*/) {}
};

}  // namespace gtsam_points
