#pragma once
#include <Eigen/Core>
#include <Eigen/Geometry>
namespace gtsam {
class Rot3 {
public:
  Rot3();
  explicit Rot3(const Eigen::Matrix3d&);
  Eigen::Quaterniond toQuaternion() const;
  Eigen::Matrix3d matrix() const;
};
}  // namespace gtsam
