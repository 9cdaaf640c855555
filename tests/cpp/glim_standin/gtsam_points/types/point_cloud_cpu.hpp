#pragma once
#include <gtsam_points/types/point_cloud.hpp>
#include <Eigen/Geometry>
#include <random>
#include <vector>
namespace gtsam_points {
struct PointCloudCPU : public PointCloud {
  using Ptr = std::shared_ptr<PointCloudCPU>;
  using ConstPtr = std::shared_ptr<const PointCloudCPU>;
  PointCloudCPU() {}
  template <class T, int D>
  PointCloudCPU(const std::vector<Eigen::Matrix<T, D, 1>>& points);
  template <class T, int D>
  PointCloudCPU(const Eigen::Matrix<T, D, 1>* points, int num_points);
  static Ptr clone(const PointCloud& frame);
  static Ptr load(const std::string& path);
  template <class T>
  void add_times(const std::vector<T>&);
  template <class T>
  void add_times(const T*, int);
  template <class T, int D>
  void add_points(const std::vector<Eigen::Matrix<T, D, 1>>&);
  template <class T, int D>
  void add_normals(const std::vector<Eigen::Matrix<T, D, 1>>&);
  template <class T, int D>
  void add_covs(const std::vector<Eigen::Matrix<T, D, D>>&);
  template <class T>
  void add_intensities(const std::vector<T>&);
  template <class T>
  void add_intensities(const T*, int);
  std::vector<Eigen::Vector4d> points_storage, normals_storage;
  std::vector<Eigen::Matrix4d> covs_storage;
};
double median_distance(const PointCloud::ConstPtr& frame, size_t max_scan_count);
PointCloudCPU::Ptr random_sampling(const PointCloud::ConstPtr& frame, double sampling_rate, std::mt19937& mt);
PointCloudCPU::Ptr transform(const PointCloud::ConstPtr& frame, const Eigen::Isometry3d& T);  // odometry_estimation_cpu.cpp:184
PointCloudCPU::Ptr merge_frames(const std::vector<Eigen::Isometry3d>& poses, const std::vector<PointCloud::ConstPtr>& frames, double downsample_resolution);
PointCloudCPU::Ptr merge_frames(const std::vector<Eigen::Isometry3d>& poses, const std::vector<PointCloud::ConstPtr>& frames, double downsample_resolution,
                                int target_num_points);
PointCloud::Ptr merge_frames_auto(const std::vector<Eigen::Isometry3d>& poses, const std::vector<PointCloud::ConstPtr>& frames, double downsample_resolution);
}  // namespace gtsam_points
