#pragma once
#include <string>
#include <gtsam/nonlinear/NonlinearFactorGraph.h>
#include <gtsam/nonlinear/Values.h>
#include <exception>
namespace gtsam {
class IndeterminantLinearSystemException : public std::exception {
public:
  Key nearbyVariable() const;
  const char* what() const noexcept override;
};
struct ISAM2GaussNewtonParams {
  explicit ISAM2GaussNewtonParams(double wildfire = 0.001);
};
struct ISAM2DoglegParams {
  ISAM2DoglegParams();
};
struct ISAM2Params {
  void setRelinearizeSkip(int);
  void setRelinearizeThreshold(double);
  void setOptimizationParams(const ISAM2GaussNewtonParams&);
  void setOptimizationParams(const ISAM2DoglegParams&);
  bool enableRelinearization = true;
  int relinearizeSkip = 1;
  double relinearizeThreshold = 0.1;
};
}  // namespace gtsam
namespace gtsam_points {
using gtsam::ISAM2Params;
struct ISAM2ResultExt {
  std::string to_string() const;
  double delta = 0.0;
  int num_lin_gpu = 0, num_lin_cpu = 0, num_factors = 0, num_values = 0;
};
class ISAM2Ext {
public:
  ISAM2Ext();
  explicit ISAM2Ext(const ISAM2Params&);
  virtual ~ISAM2Ext() {}
  virtual ISAM2ResultExt update(const gtsam::NonlinearFactorGraph& = gtsam::NonlinearFactorGraph(), const gtsam::Values& = gtsam::Values());
  virtual gtsam::Values calculateEstimate() const;
  template <class T>
  T calculateEstimate(gtsam::Key) const;
  const gtsam::NonlinearFactorGraph& getFactorsUnsafe() const;
  bool valueExists(gtsam::Key) const;
  bool empty() const;
  const gtsam::Values& getLinearizationPoint() const;
  ISAM2Params params() const;
};
}  // namespace gtsam_points
