#pragma once
#include <gtsam/base/Matrix.h>
namespace gtsam {
namespace imuBias {
class ConstantBias {
public:
  ConstantBias();
  explicit ConstantBias(const Vector6&);
  ConstantBias(const Vector3& acc, const Vector3& gyro);
  Vector6 vector() const;
};
}  // namespace imuBias
}  // namespace gtsam
