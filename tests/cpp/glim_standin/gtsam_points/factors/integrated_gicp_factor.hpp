#pragma once
// upstream's shape: a class template over the target / source frame types + the alias GLIM's mapping modules name
#include <gtsam_points/factors/integrated_matching_cost_factor.hpp>
namespace gtsam_points {
class NearestNeighborSearch;
template <typename TargetFrame = PointCloud, typename SourceFrame = PointCloud>
class IntegratedGICPFactor_ : public IntegratedMatchingCostFactor {
public:
  IntegratedGICPFactor_(gtsam::Key, gtsam::Key, const std::shared_ptr<const TargetFrame>&, const std::shared_ptr<const SourceFrame>&);
  IntegratedGICPFactor_(gtsam::Key, gtsam::Key, const std::shared_ptr<const TargetFrame>&, const std::shared_ptr<const SourceFrame>&,
                        const std::shared_ptr<const NearestNeighborSearch>&);
  IntegratedGICPFactor_(const gtsam::Pose3&, gtsam::Key, const std::shared_ptr<const TargetFrame>&, const std::shared_ptr<const SourceFrame>&);
  IntegratedGICPFactor_(const gtsam::Pose3&, gtsam::Key, const std::shared_ptr<const TargetFrame>&, const std::shared_ptr<const SourceFrame>&,
                        const std::shared_ptr<const NearestNeighborSearch>&);
  void set_max_correspondence_distance(double);
};
using IntegratedGICPFactor = IntegratedGICPFactor_<>;
}  // namespace gtsam_points
