#pragma once
#include <gtsam_points/types/point_cloud.hpp>
#include <memory>
#include <string>
namespace gtsam_points {
class GaussianVoxelMap {
public:
  using Ptr = std::shared_ptr<GaussianVoxelMap>;
  using ConstPtr = std::shared_ptr<const GaussianVoxelMap>;
  virtual ~GaussianVoxelMap() {}
  virtual double voxel_resolution() const = 0;
  virtual void insert(const PointCloud& frame) = 0;
  virtual void save_compact(const std::string& path) const = 0;
};
}  // namespace gtsam_points
