#pragma once
#include <gtsam_points/optimizers/incremental_fixed_lag_smoother_ext.hpp>
namespace gtsam_points {
class IncrementalFixedLagSmootherExtWithFallback : public IncrementalFixedLagSmootherExt {
public:
  gtsam::Values calculateEstimate() const;
};
}  // namespace gtsam_points
