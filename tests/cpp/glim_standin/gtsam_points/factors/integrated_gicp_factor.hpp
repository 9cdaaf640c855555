#pragma once
#include <gtsam_points/factors/integrated_matching_cost_factor.hpp>
namespace gtsam_points {
class NearestNeighborSearch;
class IntegratedGICPFactor : public IntegratedMatchingCostFactor {
public:
  IntegratedGICPFactor(gtsam::Key, gtsam::Key, const PointCloud::ConstPtr&, const PointCloud::ConstPtr&);
  IntegratedGICPFactor(gtsam::Key, gtsam::Key, const PointCloud::ConstPtr&, const PointCloud::ConstPtr&, const std::shared_ptr<const NearestNeighborSearch>&);
  IntegratedGICPFactor(const gtsam::Pose3&, gtsam::Key, const PointCloud::ConstPtr&, const PointCloud::ConstPtr&);
  void set_max_correspondence_distance(double);
};
}  // namespace gtsam_points
