#pragma once
#include <gtsam/nonlinear/NonlinearFactorGraph.h>
#include <gtsam/nonlinear/Values.h>
namespace gtsam {
class LevenbergMarquardtParams {
public:
  void setAbsoluteErrorTol(double);
  void setRelativeErrorTol(double);
  void setMaxIterations(int);
};
class LevenbergMarquardtOptimizer {
public:
  LevenbergMarquardtOptimizer(const NonlinearFactorGraph&, const Values&, const LevenbergMarquardtParams&);
  const Values& optimize();
};
}  // namespace gtsam
