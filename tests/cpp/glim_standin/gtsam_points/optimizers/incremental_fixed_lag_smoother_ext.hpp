#pragma once
#include <gtsam/nonlinear/Values.h>
namespace gtsam_points {
class IncrementalFixedLagSmootherExt {
public:
  virtual ~IncrementalFixedLagSmootherExt() {}
};
}  // namespace gtsam_points
