#pragma once
#include <Eigen/Core>
#include <cstddef>
#include <memory>
#include <string>
namespace gtsam_points {
struct PointCloud {  // raw-pointer view
  using Ptr = std::shared_ptr<PointCloud>;
  using ConstPtr = std::shared_ptr<const PointCloud>;
  virtual ~PointCloud() {}
  size_t num_points = 0;
  double* times = nullptr;
  Eigen::Vector4d* points = nullptr;
  Eigen::Vector4d* normals = nullptr;
  Eigen::Matrix4d* covs = nullptr;
  double* intensities = nullptr;
  float* times_gpu = nullptr;
  Eigen::Matrix<float, 3, 1>* points_gpu = nullptr;
  size_t size() const { return num_points; }
  bool has_points() const { return points != nullptr; }
  bool has_covs() const { return covs != nullptr; }
  bool has_normals() const { return normals != nullptr; }
  bool has_times() const { return times != nullptr; }
  bool has_intensities() const { return intensities != nullptr; }
  void save(const std::string& path) const;
  void save_compact(const std::string& path) const;
};
}  // namespace gtsam_points
