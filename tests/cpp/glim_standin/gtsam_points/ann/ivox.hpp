#pragma once
// stand-in (declarations only) of gtsam_points' incremental voxel map family as GLIM's CPU odometry names it
// (include/glim/odometry/odometry_estimation_cpu.hpp:5-12: `using iVox = IncrementalVoxelMap<FlatContainer>`)
#include <memory>
#include <vector>
#include <Eigen/Core>
#include <gtsam_points/ann/nearest_neighbor_search.hpp>
#include <gtsam_points/types/point_cloud.hpp>
namespace gtsam_points {
struct FlatContainer {
  struct Setting {
    void set_min_dist_in_cell(double);
  };
};
template <typename VoxelContents>
class IncrementalVoxelMap : public NearestNeighborSearch {
public:
  using Ptr = std::shared_ptr<IncrementalVoxelMap>;
  explicit IncrementalVoxelMap(double resolution);
  typename VoxelContents::Setting& voxel_insertion_setting();
  void set_lru_horizon(int);
  void set_lru_clear_cycle(int);
  void set_neighbor_voxel_mode(int);
  void insert(const PointCloud& frame);
  std::vector<Eigen::Vector4d> voxel_points() const;
};
using iVox = IncrementalVoxelMap<FlatContainer>;
}  // namespace gtsam_points
