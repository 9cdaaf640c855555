#pragma once
#include <Eigen/Core>
#include <gtsam_points/ann/nearest_neighbor_search.hpp>
namespace gtsam_points {
class KdTree : public NearestNeighborSearch {
public:
  KdTree(const Eigen::Vector4d* points, int num_points, int build_num_threads = 1);
};
}  // namespace gtsam_points
