#pragma once
#include <gtsam/geometry/Rot3.h>
#include <gtsam/linear/NoiseModel.h>
#include <gtsam/nonlinear/NonlinearFactor.h>
namespace gtsam {
template <class POSE>
class PoseRotationPrior : public NonlinearFactor {
public:
  PoseRotationPrior(Key, const Rot3&, const SharedNoiseModel&);
  PoseRotationPrior(Key, const POSE&, const SharedNoiseModel&);
  size_t dim() const override;
  double error(const Values&) const override;
  std::shared_ptr<GaussianFactor> linearize(const Values&) const override;
  shared_ptr clone() const override;
};
}  // namespace gtsam
