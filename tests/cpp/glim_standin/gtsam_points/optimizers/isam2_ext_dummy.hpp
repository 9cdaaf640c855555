#pragma once
#include <gtsam_points/optimizers/isam2_ext.hpp>
namespace gtsam_points {
class ISAM2ExtDummy : public ISAM2Ext {
public:
  explicit ISAM2ExtDummy(const ISAM2Params&);
};
}  // namespace gtsam_points
