#pragma once
#include <gtsam/linear/NoiseModel.h>
#include <gtsam/nonlinear/NonlinearFactor.h>
namespace gtsam {
template <class T>
class BetweenFactor : public NonlinearFactor {
public:
  BetweenFactor(Key, Key, const T&, const SharedNoiseModel& = nullptr);
  size_t dim() const override;
  double error(const Values&) const override;
  std::shared_ptr<GaussianFactor> linearize(const Values&) const override;
  shared_ptr clone() const override;
  const T& measured() const;
};
template <class T>
class PriorFactor : public NonlinearFactor {
public:
  PriorFactor(Key, const T&, const SharedNoiseModel& = nullptr);
  size_t dim() const override;
  double error(const Values&) const override;
  std::shared_ptr<GaussianFactor> linearize(const Values&) const override;
  shared_ptr clone() const override;
};
}  // namespace gtsam
