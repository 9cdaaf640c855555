#pragma once
#include <gtsam_points/factors/integrated_matching_cost_factor.hpp>
namespace gtsam_points {
class IntegratedVGICPFactor : public IntegratedMatchingCostFactor {
public:
  IntegratedVGICPFactor(gtsam::Key, gtsam::Key, const GaussianVoxelMap::ConstPtr&, const PointCloud::ConstPtr&);
  IntegratedVGICPFactor(const gtsam::Pose3&, gtsam::Key, const GaussianVoxelMap::ConstPtr&, const PointCloud::ConstPtr&);
};
}  // namespace gtsam_points
