#pragma once
#include <gtsam_points/types/point_cloud.hpp>
#include <vector>
namespace gtsam_points {
struct PointCloudCPU : public PointCloud {
  using Ptr = std::shared_ptr<PointCloudCPU>;
  using ConstPtr = std::shared_ptr<const PointCloudCPU>;
  static Ptr clone(const PointCloud& frame);
  std::vector<Eigen::Vector4d> points_storage, normals_storage;
  std::vector<Eigen::Matrix4d> covs_storage;
};
double median_distance(const PointCloud::ConstPtr& frame, size_t max_scan_count);
}  // namespace gtsam_points
