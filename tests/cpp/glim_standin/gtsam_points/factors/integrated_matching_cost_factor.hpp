#pragma once
#include <gtsam/geometry/Pose3.h>
#include <gtsam/nonlinear/NonlinearFactor.h>
#include <gtsam_points/types/gaussian_voxelmap.hpp>
#include <gtsam_points/types/point_cloud.hpp>
#include <gtsam_points/util/gtsam_migration.hpp>
namespace gtsam_points {
class IntegratedMatchingCostFactor : public gtsam::NonlinearFactor {
public:
  size_t dim() const override;
  double error(const gtsam::Values&) const override;
  std::shared_ptr<gtsam::GaussianFactor> linearize(const gtsam::Values&) const override;
  shared_ptr clone() const override;
  void set_num_threads(int);
  double inlier_fraction() const;
  Eigen::Isometry3d get_fixed_target_pose() const;
};
}  // namespace gtsam_points
