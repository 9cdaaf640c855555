// stand-in for gtsam/geometry/Pose3.h (see ../../Eigen/Core): gtsam::Vector6 and gtsam::Pose3::Expmap / matrix(), restated from GTSAM 4.2
// (Pose3::Expmap: R = Rot3::Expmap(omega) through SO3's ExpmapFunctor; t = (omega x v - R (omega x v) + omega (omega . v)) / theta^2).
#pragma once
#include <limits>

#include "../../Eigen/Core"

namespace gtsam {

typedef Eigen::Vector6d Vector6;
typedef Eigen::Vector3d Vector3;
typedef Eigen::Matrix3d Matrix3;

class Pose3 {
public:
  Pose3() : R_(Matrix3::Identity()) {}
  Pose3(const Matrix3& R, const Vector3& t) : R_(R), t_(t) {}
  static Pose3 Expmap(const Vector6& xi) {
    const Vector3 omega(xi(0), xi(1), xi(2)), v(xi(3), xi(4), xi(5));
    const Matrix3 R = RotExpmap(omega);
    const double theta2 = omega.dot(omega);
    if (theta2 > std::numeric_limits<double>::epsilon()) {
      const Vector3 t_parallel = omega * omega.dot(v);
      const Vector3 omega_cross_v = omega.cross(v);
      const Vector3 t = (omega_cross_v - R * omega_cross_v + t_parallel) / theta2;
      return Pose3(R, t);
    }
    return Pose3(R, v);
  }
  Eigen::Matrix4d matrix() const {
    Eigen::Matrix4d m = Eigen::Matrix4d::Identity();
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) m(r, c) = R_(r, c);
      m(r, 3) = t_[r];
    }
    return m;
  }

private:
  // so3::ExpmapFunctor(omega).expmap(): I + sin(theta) K + (1 - cos(theta)) K^2 with K = hat(omega) / theta and
  // 1 - cos(theta) evaluated as 2 sin^2(theta / 2); first-order form I + hat(omega) when theta^2 <= epsilon
  static Matrix3 RotExpmap(const Vector3& omega) {
    const double theta2 = omega.dot(omega);
    Matrix3 W;
    W(0, 0) = 0.0; W(0, 1) = -omega[2]; W(0, 2) = omega[1];
    W(1, 0) = omega[2]; W(1, 1) = 0.0; W(1, 2) = -omega[0];
    W(2, 0) = -omega[1]; W(2, 1) = omega[0]; W(2, 2) = 0.0;
    if (theta2 <= std::numeric_limits<double>::epsilon()) return Matrix3::Identity() + W;
    const double theta = std::sqrt(theta2);
    const double sin_theta = std::sin(theta);
    const double s2 = std::sin(theta / 2.0);
    const double one_minus_cos = 2.0 * s2 * s2;
    const Matrix3 K = W / theta;
    const Matrix3 KK = K * K;
    return Matrix3::Identity() + sin_theta * K + one_minus_cos * KK;
  }
  Matrix3 R_;
  Vector3 t_;
};

}  // namespace gtsam
