#pragma once
#include <gtsam/geometry/Pose3.h>
namespace gtsam {
class NavState {
public:
  NavState();
  NavState(const Pose3&, const Vector3&);
  Pose3 pose() const;
  Vector3 velocity() const;
};
}  // namespace gtsam
