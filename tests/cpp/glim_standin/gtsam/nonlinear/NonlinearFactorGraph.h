#pragma once
#include <gtsam/nonlinear/NonlinearFactor.h>
#include <memory>
#include <vector>
namespace gtsam {
class NonlinearFactorGraph {
public:
  typedef std::vector<NonlinearFactor::shared_ptr>::const_iterator const_iterator;
  template <class F>
  void add(const std::shared_ptr<F>& f) { factors_.push_back(f); }
  void push_back(const NonlinearFactor::shared_ptr& f) { factors_.push_back(f); }
  const_iterator begin() const { return factors_.begin(); }
  const_iterator end() const { return factors_.end(); }
  size_t size() const { return factors_.size(); }
  std::shared_ptr<GaussianFactorGraph> linearize(const Values&) const;

private:
  std::vector<NonlinearFactor::shared_ptr> factors_;
};
}  // namespace gtsam
