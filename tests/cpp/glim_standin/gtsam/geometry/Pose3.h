#pragma once
#include <Eigen/Core>
#include <gtsam/base/Matrix.h>
namespace gtsam {
class Pose3 {
public:
  Pose3() : m_(Eigen::Matrix4d::Identity()) {}
  explicit Pose3(const Eigen::Matrix4d& m) : m_(m) {}
  Eigen::Matrix4d matrix() const { return m_; }
  static Pose3 Expmap(const Vector6&);

private:
  Eigen::Matrix4d m_;
};
}  // namespace gtsam
