#pragma once
#include <gtsam/nonlinear/NonlinearFactorGraph.h>
#include <functional>
#include <memory>
#include <vector>
namespace gtsam_points {
class NonlinearFactorSet {  // the batch protocol the Ext optimisers drive before graph.linearize()
public:
  virtual ~NonlinearFactorSet() {}
  virtual int size() const = 0;
  virtual void clear() = 0;
  virtual void clear_counts() = 0;
  virtual bool add(std::shared_ptr<gtsam::NonlinearFactor> factor) = 0;
  virtual void add(const gtsam::NonlinearFactorGraph& factors) = 0;
  virtual void linearize(const gtsam::Values& linearization_point) = 0;
  virtual void error(const gtsam::Values& values) = 0;
  virtual std::vector<gtsam::GaussianFactor::shared_ptr> calc_linear_factors(const gtsam::Values& linearization_point) = 0;
};
struct LinearizationHook {
  static std::vector<std::function<std::shared_ptr<NonlinearFactorSet>()>>& hooks() {
    static std::vector<std::function<std::shared_ptr<NonlinearFactorSet>()>> h;
    return h;
  }
  static void register_hook(const std::function<std::shared_ptr<NonlinearFactorSet>()>& hook) { hooks().push_back(hook); }
};
}  // namespace gtsam_points
