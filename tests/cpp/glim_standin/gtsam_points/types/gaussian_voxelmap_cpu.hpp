#pragma once
#include <gtsam_points/types/gaussian_voxelmap.hpp>
#include <gtsam_points/util/vector3i_hash.hpp>
namespace gtsam_points {
class GaussianVoxelMapCPU : public GaussianVoxelMap {
public:
  using Ptr = std::shared_ptr<GaussianVoxelMapCPU>;
  using ConstPtr = std::shared_ptr<const GaussianVoxelMapCPU>;
  explicit GaussianVoxelMapCPU(double resolution);
  double voxel_resolution() const override;
  void insert(const PointCloud& frame) override;
  void save_compact(const std::string& path) const override;
  static Ptr load(const std::string& path);
  void set_lru_horizon(int);
  void set_lru_clear_cycle(int);
};
}  // namespace gtsam_points
