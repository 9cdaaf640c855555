#pragma once
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <gtsam/base/Matrix.h>
#include <gtsam/geometry/Rot3.h>
namespace gtsam {
class Pose3 {
public:
  Pose3() : m_(Eigen::Matrix4d::Identity()) {}
  explicit Pose3(const Eigen::Matrix4d& m) : m_(m) {}
  Pose3(const Rot3&, const Eigen::Vector3d&);
  Eigen::Matrix4d matrix() const { return m_; }
  static Pose3 Expmap(const Vector6&);
  static Vector6 Logmap(const Pose3&);
  static Pose3 Identity();
  Eigen::Vector3d translation() const;
  Rot3 rotation() const;
  Pose3 inverse() const;
  Pose3 operator*(const Pose3&) const;
  Pose3 between(const Pose3&) const;

private:
  Eigen::Matrix4d m_;
};
}  // namespace gtsam
