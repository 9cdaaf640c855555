#pragma once
#include <gtsam/linear/HessianFactor.h>
#include <gtsam/nonlinear/Values.h>
#include <memory>
namespace gtsam {
class NonlinearFactor {
public:
  using shared_ptr = std::shared_ptr<NonlinearFactor>;
  NonlinearFactor() {}
  template <typename CONTAINER>
  NonlinearFactor(const CONTAINER& keys) : keys_(keys.begin(), keys.end()) {}
  virtual ~NonlinearFactor() {}
  const KeyVector& keys() const { return keys_; }
  virtual double error(const Values& c) const = 0;
  virtual size_t dim() const = 0;
  virtual std::shared_ptr<GaussianFactor> linearize(const Values& c) const = 0;
  virtual shared_ptr clone() const = 0;

protected:
  KeyVector keys_;
};
}  // namespace gtsam
