// Stand-in for gtsam_points/ann/kdtree.hpp (a nanoflann kd-tree upstream): an exhaustive search with the oracle's distance expression
// ((dx*dx + dy*dy) + dz*dz on the first three coordinates) and (distance, index) order -- the kd-tree leaves exact ties to its traversal.
#pragma once
#include <Eigen/Core>
#include <algorithm>
#include <cstddef>
#include <utility>
#include <vector>
namespace gtsam_points {
class KdTree {
public:
  KdTree(const Eigen::Vector4d* points, int num_points) : points_(points), n_(num_points) {}
  size_t knn_search(const double* pt, size_t k, size_t* k_indices, double* k_sq_dists) const {
    std::vector<std::pair<double, size_t>> d((size_t)n_);
    for (int i = 0; i < n_; i++) {
      const double dx = pt[0] - points_[i][0], dy = pt[1] - points_[i][1], dz = pt[2] - points_[i][2];
      const double xx = dx * dx, yy = dy * dy, zz = dz * dz;
      d[(size_t)i] = std::make_pair((xx + yy) + zz, (size_t)i);
    }
    const size_t m = std::min(k, (size_t)n_);
    std::partial_sort(d.begin(), d.begin() + (std::ptrdiff_t)m, d.end());
    for (size_t j = 0; j < m; j++) {
      k_indices[j] = d[j].second;
      k_sq_dists[j] = d[j].first;
    }
    return m;
  }

private:
  const Eigen::Vector4d* points_;
  int n_;
};
}  // namespace gtsam_points
