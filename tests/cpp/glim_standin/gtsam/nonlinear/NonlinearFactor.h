#pragma once
#include <gtsam/inference/Key.h>
#include <gtsam/linear/HessianFactor.h>
#include <gtsam/nonlinear/Values.h>
#include <memory>
namespace gtsam {
class NonlinearFactor {
public:
  typedef std::shared_ptr<NonlinearFactor> shared_ptr;
  NonlinearFactor() {}
  explicit NonlinearFactor(const KeyVector& keys) : keys_(keys) {}
  virtual ~NonlinearFactor() {}
  const KeyVector& keys() const { return keys_; }
  virtual size_t dim() const = 0;
  virtual double error(const Values&) const = 0;
  virtual std::shared_ptr<GaussianFactor> linearize(const Values&) const = 0;
  virtual shared_ptr clone() const = 0;

protected:
  KeyVector keys_;
};
}  // namespace gtsam
