// gtsam_points/types/point_cloud_cpu.hpp, MI355X edition: the REAL header (PointCloudCPU and the CPU samplers stay what they are -- it is
// pulled in with #include_next from wherever the build finds gtsam_points) plus ONE redirection:
//
//   gtsam_points::merge_frames(poses, frames, downsample_resolution[, target_num_points])      sub_mapping.cpp:496
//
// resolves to the device merge (glim_amd_merge_frames: transform + voxel-grid average of points AND covariances + optional uniform sample,
// preprocess.hip K10) in every translation unit that sees this header, so that GLIM's sub_mapping.cpp builds its submap on the GPU WITHOUT AN
// EDIT.  The redirection is a one-identifier macro placed AFTER the real declarations: `merge_frames` -> `merge_frames_hip`, an inline function
// of the same namespace and signatures (merge_frames_gpu / merge_frames_auto are different identifiers and untouched).  The result is a
// PointCloudCPU with points and covariances, as upstream returns; voxel membership and averages follow the reference's rule bit for bit
// (tests/test_merge.py), the optional random sample uses a counter-based generator instead of std::mt19937.
// A build that prefers link-time interposition over a macro links adapters/glim/merge_frames_hip.cpp (the same function under the original
// name) in front of libgtsam_points and defines GLIM_AMD_NO_MERGE_FRAMES_MACRO.
#pragma once

#include_next <gtsam_points/types/point_cloud_cpu.hpp>

#include <cstring>
#include <vector>

#include <glim_amd_gtsam.hpp>
#include <glim_amd/glim_preprocess_compat.hpp>

namespace gtsam_points {

inline PointCloudCPU::Ptr merge_frames_hip(const std::vector<Eigen::Isometry3d>& poses, const std::vector<PointCloud::ConstPtr>& frames, double downsample_resolution,
                                           int target_num_points = -1) {
  std::vector<glim_amd::Isometry3d> T(poses.size());
  std::vector<glim_amd::FrameView> views(frames.size());
  for (std::size_t i = 0; i < poses.size(); i++) T[i] = glim_amd::to_iso(poses[i]);
  for (std::size_t i = 0; i < frames.size(); i++) {
    views[i].points4 = reinterpret_cast<const double*>(frames[i]->points);  // Vector4d / column-major Matrix4d arrays: the C ABI's input layout
    views[i].covs16 = reinterpret_cast<const double*>(frames[i]->covs);
    views[i].size = (std::int64_t)frames[i]->size();
  }
  const glim_amd::MergedFrame merged = glim_amd::merge_frames(T, views, downsample_resolution, target_num_points);
  std::vector<Eigen::Vector4d> points(merged.size());
  std::vector<Eigen::Matrix4d> covs(merged.size());
  static_assert(sizeof(Eigen::Vector4d) == 4 * sizeof(double) && sizeof(Eigen::Matrix4d) == 16 * sizeof(double), "dense fixed-size Eigen storage");
  if (merged.size()) {
    std::memcpy(points.data(), merged.points.data(), merged.size() * sizeof(Eigen::Vector4d));
    std::memcpy(covs.data(), merged.covs.data(), merged.size() * sizeof(Eigen::Matrix4d));
  }
  auto out = std::make_shared<PointCloudCPU>();
  out->add_points(points);
  out->add_covs(covs);
  return out;
}

}  // namespace gtsam_points

#ifndef GLIM_AMD_NO_MERGE_FRAMES_MACRO
#define merge_frames merge_frames_hip
#endif
