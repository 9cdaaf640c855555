#pragma once
#include <gtsam/linear/NoiseModel.h>
#include <gtsam/nonlinear/NonlinearFactor.h>
namespace gtsam {
template <class POSE>
class PoseTranslationPrior : public NonlinearFactor {
public:
  PoseTranslationPrior(Key, const Eigen::Vector3d&, const SharedNoiseModel&);
  PoseTranslationPrior(Key, const POSE&, const SharedNoiseModel&);
  size_t dim() const override;
  double error(const Values&) const override;
  std::shared_ptr<GaussianFactor> linearize(const Values&) const override;
  shared_ptr clone() const override;
};
}  // namespace gtsam
