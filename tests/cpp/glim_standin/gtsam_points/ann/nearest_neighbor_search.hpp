#pragma once
#include <cstddef>
namespace gtsam_points {
class NearestNeighborSearch {
public:
  virtual ~NearestNeighborSearch() {}
};
}  // namespace gtsam_points
