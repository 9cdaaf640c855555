#pragma once
#include <memory>
namespace gtsam_points {
template <class T>
using shared_ptr = std::shared_ptr<T>;
}  // namespace gtsam_points
namespace gtsam {
using std::make_shared;  // GTSAM >= 4.3 (std::shared_ptr) flavour of the migration header
}  // namespace gtsam
