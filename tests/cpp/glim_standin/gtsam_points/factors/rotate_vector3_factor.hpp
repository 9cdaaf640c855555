#pragma once
#include <gtsam/linear/NoiseModel.h>
#include <gtsam/nonlinear/NonlinearFactor.h>
namespace gtsam_points {
class RotateVector3Factor : public gtsam::NonlinearFactor {
public:
  RotateVector3Factor(gtsam::Key, gtsam::Key, const gtsam::Vector3&, const gtsam::SharedNoiseModel&);
  size_t dim() const override;
  double error(const gtsam::Values&) const override;
  std::shared_ptr<gtsam::GaussianFactor> linearize(const gtsam::Values&) const override;
  shared_ptr clone() const override;
};
}  // namespace gtsam_points
