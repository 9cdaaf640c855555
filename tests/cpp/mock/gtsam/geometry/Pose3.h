#pragma once
#include <Eigen/Core>
namespace gtsam {
class Pose3 {
public:
  Pose3() = default;
  explicit Pose3(const Eigen::Matrix4d& T) : T_(T) {}
  Eigen::Matrix4d matrix() const { return T_; }

private:
  Eigen::Matrix4d T_ = Eigen::Matrix4d::Identity();
};
}  // namespace gtsam
