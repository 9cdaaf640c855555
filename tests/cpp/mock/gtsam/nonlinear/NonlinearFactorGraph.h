#pragma once
#include <gtsam/nonlinear/NonlinearFactor.h>
#include <vector>
namespace gtsam {
class NonlinearFactorGraph {
public:
  void push_back(const NonlinearFactor::shared_ptr& f) { factors_.push_back(f); }
  void add(const NonlinearFactor::shared_ptr& f) { factors_.push_back(f); }
  size_t size() const { return factors_.size(); }
  std::vector<NonlinearFactor::shared_ptr>::const_iterator begin() const { return factors_.begin(); }
  std::vector<NonlinearFactor::shared_ptr>::const_iterator end() const { return factors_.end(); }
  const NonlinearFactor::shared_ptr& operator[](size_t i) const { return factors_[i]; }

private:
  std::vector<NonlinearFactor::shared_ptr> factors_;
};
}  // namespace gtsam
