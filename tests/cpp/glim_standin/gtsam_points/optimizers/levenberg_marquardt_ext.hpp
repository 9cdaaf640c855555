#pragma once
#include <functional>
#include <string>
#include <gtsam/nonlinear/LevenbergMarquardtOptimizer.h>
namespace gtsam_points {
struct LevenbergMarquardtOptimizationStatus {
  std::string to_string() const;
  std::string to_short_string() const;
};
class LevenbergMarquardtExtParams : public gtsam::LevenbergMarquardtParams {
public:
  void set_verbose();
  void setlambdaInitial(double);
  std::function<void(const LevenbergMarquardtOptimizationStatus&, const gtsam::Values&)> callback;
  std::function<void(const gtsam::Values&)> status_msg_callback;
  std::function<bool(const gtsam::Values&)> termination_criteria;  // odometry_estimation_cpu.cpp:121
};
class LevenbergMarquardtOptimizerExt {
public:
  LevenbergMarquardtOptimizerExt(const gtsam::NonlinearFactorGraph&, const gtsam::Values&, const LevenbergMarquardtExtParams&);
  const gtsam::Values& optimize();
};
}  // namespace gtsam_points
