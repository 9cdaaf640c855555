// merge_frames_hip.cpp -- gtsam_points::merge_frames under its ORIGINAL name, backed by the device merge (glim_amd_merge_frames, preprocess.hip K10),
// for builds that prefer link-time interposition to the one-identifier macro of adapters/gtsam_points_hip/gtsam_points/types/point_cloud_cpu.hpp:
// compile with -DGLIM_AMD_NO_MERGE_FRAMES_MACRO and link this object into libglim in front of libgtsam_points -- sub_mapping.cpp:496 then binds to
// the definitions below.
#define GLIM_AMD_NO_MERGE_FRAMES_MACRO
#include <gtsam_points/types/point_cloud_cpu.hpp>

namespace gtsam_points {

PointCloudCPU::Ptr merge_frames(const std::vector<Eigen::Isometry3d>& poses, const std::vector<PointCloud::ConstPtr>& frames, double downsample_resolution) {
  return merge_frames_hip(poses, frames, downsample_resolution, -1);
}

PointCloudCPU::Ptr merge_frames(const std::vector<Eigen::Isometry3d>& poses, const std::vector<PointCloud::ConstPtr>& frames, double downsample_resolution,
                                int target_num_points) {
  return merge_frames_hip(poses, frames, downsample_resolution, target_num_points);
}

}  // namespace gtsam_points
