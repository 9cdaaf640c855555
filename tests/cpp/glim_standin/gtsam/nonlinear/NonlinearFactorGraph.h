#pragma once
#include <gtsam/nonlinear/NonlinearFactor.h>
#include <map>
#include <memory>
#include <vector>
namespace gtsam {
class NonlinearFactorGraph {
public:
  typedef std::vector<NonlinearFactor::shared_ptr>::const_iterator const_iterator;
  typedef std::vector<NonlinearFactor::shared_ptr>::iterator iterator;
  iterator begin() { return factors_.begin(); }
  iterator end() { return factors_.end(); }
  iterator erase(iterator a, iterator b) { return factors_.erase(a, b); }
  iterator erase(iterator a) { return factors_.erase(a); }
  NonlinearFactorGraph rekey(const std::map<Key, Key>&) const;
  void remove(size_t i);
  void replace(size_t i, const NonlinearFactor::shared_ptr&);
  template <class F>
  void add(const std::shared_ptr<F>& f) { factors_.push_back(f); }
  void add(const NonlinearFactorGraph&);
  void add(const std::vector<NonlinearFactor::shared_ptr>& fs) { for (const auto& f : fs) factors_.push_back(f); }
  void push_back(const NonlinearFactor::shared_ptr& f) { factors_.push_back(f); }
  template <class F, class... A>
  void emplace_shared(A&&... a) { factors_.push_back(std::make_shared<F>(std::forward<A>(a)...)); }
  const_iterator begin() const { return factors_.begin(); }
  const_iterator end() const { return factors_.end(); }
  size_t size() const { return factors_.size(); }
  bool empty() const { return factors_.empty(); }
  void resize(size_t n) { factors_.resize(n); }
  void reserve(size_t n) { factors_.reserve(n); }
  const NonlinearFactor::shared_ptr& operator[](size_t i) const { return factors_[i]; }
  NonlinearFactor::shared_ptr& operator[](size_t i) { return factors_[i]; }
  const NonlinearFactor::shared_ptr& at(size_t i) const { return factors_[i]; }
  std::shared_ptr<GaussianFactorGraph> linearize(const Values&) const;
  double error(const Values&) const;

private:
  std::vector<NonlinearFactor::shared_ptr> factors_;
};
}  // namespace gtsam
