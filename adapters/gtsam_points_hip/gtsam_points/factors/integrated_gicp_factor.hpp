// gtsam_points/factors/integrated_gicp_factor.hpp, MI355X edition: the constructors GLIM calls, unchanged --
//   IntegratedGICPFactor(target_key, source_key, target_frame, source_frame)                   sub_mapping.cpp:202, global_mapping.cpp:400
//   IntegratedGICPFactor(target_key, source_key, target_frame, source_frame, target_tree)       global_mapping_pose_graph.cpp:393
//   IntegratedGICPFactor(fixed_target_pose, source_key, target_frame, source_frame)
// taking the base-class pointers the call sites hold (gtsam_points::PointCloud::ConstPtr).  A frame made by PointCloudGPU::clone is used where it
// lies in HBM; any other frame (the between-factor keyframes of sub_mapping.cpp:202 are PointCloudCPU) is uploaded once, by the constructor.
// error / linearize / clone / set_max_correspondence_distance / set_num_threads / inlier_fraction: glim_amd::IntegratedGICPFactorHIP, i.e.
// glim_amd_gicp_linearize / glim_amd_gicp_error on the device (exact nearest target point through a uniform grid index + the VGICP algebra).
//
// With this directory in front of the include path GLIM's sub_mapping.cpp and global_mapping.cpp reach the device GICP factor WITHOUT AN EDIT
// (`dynamic_cast<gtsam_points::IntegratedGICPFactor*>` at global_mapping.cpp:584 sees this class: the factors it inspects were made here).
// The target's search structure: upstream passes a gtsam_points::NearestNeighborSearch (a KdTree of the target); here any such pointer is
// accepted and ignored -- the device index (glim_amd_nn_index) is built from the target frame itself, once per factor.
//
// The REAL header stays reachable (round 6): upstream defines a class TEMPLATE IntegratedGICPFactor_<TargetFrame, SourceFrame> and the alias
// `using IntegratedGICPFactor = IntegratedGICPFactor_<>`; the CPU odometry instantiates the template over an iVox target
// (odometry_estimation_cpu.cpp:94-99, registration_type "GICP"), which is not this library's path.  So the real header is pulled in with
// #include_next under a one-identifier rename of the ALIAS only (`IntegratedGICPFactor_` is a different token and untouched): every
// `IntegratedGICPFactor_<iVox, PointCloud>` keeps compiling against libgtsam_points, and the plain name `IntegratedGICPFactor` is the device class below.
#pragma once

#define IntegratedGICPFactor IntegratedGICPFactorUpstreamAlias
#include_next <gtsam_points/factors/integrated_gicp_factor.hpp>
#undef IntegratedGICPFactor

#include <memory>

#include <gtsam_points/types/point_cloud_gpu.hpp>

#include <glim_amd_gtsam.hpp>

namespace gtsam_points {

class NearestNeighborSearch;  // gtsam_points/ann/nearest_neighbor_search.hpp (CPU); only ever passed through

class IntegratedGICPFactor : public glim_amd::IntegratedGICPFactorHIP {
public:
  using shared_ptr = std::shared_ptr<IntegratedGICPFactor>;

  IntegratedGICPFactor(gtsam::Key target_key, gtsam::Key source_key, const PointCloud::ConstPtr& target, const PointCloud::ConstPtr& source)
  : glim_amd::IntegratedGICPFactorHIP(target_key, source_key, device_cloud(target), device_cloud(source)) {}

  IntegratedGICPFactor(gtsam::Key target_key, gtsam::Key source_key, const PointCloud::ConstPtr& target, const PointCloud::ConstPtr& source,
                       const std::shared_ptr<const NearestNeighborSearch>& /*target_tree: the device index replaces it*/)
  : glim_amd::IntegratedGICPFactorHIP(target_key, source_key, device_cloud(target), device_cloud(source)) {}

  IntegratedGICPFactor(const gtsam::Pose3& fixed_target_pose, gtsam::Key source_key, const PointCloud::ConstPtr& target, const PointCloud::ConstPtr& source)
  : glim_amd::IntegratedGICPFactorHIP(fixed_target_pose, source_key, device_cloud(target), device_cloud(source)) {}

  IntegratedGICPFactor(const gtsam::Pose3& fixed_target_pose, gtsam::Key source_key, const PointCloud::ConstPtr& target, const PointCloud::ConstPtr& source,
                       const std::shared_ptr<const NearestNeighborSearch>& /*target_tree*/)
  : glim_amd::IntegratedGICPFactorHIP(fixed_target_pose, source_key, device_cloud(target), device_cloud(source)) {}
};

}  // namespace gtsam_points
