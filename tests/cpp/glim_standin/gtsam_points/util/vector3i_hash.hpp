#pragma once
#include <Eigen/Core>
#include <cstddef>
namespace gtsam_points {
struct Vector3iHash {
  size_t operator()(const Eigen::Vector3i& x) const { return (size_t)((x[0] * 73856093) ^ (x[1] * 19349669) ^ (x[2] * 83492791)); }
};
}  // namespace gtsam_points
