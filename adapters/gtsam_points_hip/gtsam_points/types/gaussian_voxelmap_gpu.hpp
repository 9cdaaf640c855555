// gtsam_points/types/gaussian_voxelmap_gpu.hpp, MI355X edition: GaussianVoxelMapGPU(resolution, init_num_buckets, max_bucket_scan_count,
// target_points_drop_rate, stream) + insert(frame) (odometry_estimation_gpu.cpp:103-104; sub_mapping.cpp:398-399; global_mapping.cpp:265-266,
// 747-748), VoxelMapInfo (viewer/standard_viewer_mem.cpp:76-77) and overlap_gpu / overlap_auto (odometry_estimation_gpu.cpp:231,248,265,279,326;
// sub_mapping.cpp:252-253; global_mapping.cpp:322,448).
#pragma once

#include <memory>
#include <stdexcept>
#include <vector>

#include <Eigen/Core>
#include <Eigen/Geometry>
#include <gtsam_points/types/gaussian_voxelmap.hpp>
#include <gtsam_points/types/hip_voxelmap_holder.hpp>
#include <gtsam_points/types/point_cloud_gpu.hpp>

struct CUstream_st;

namespace gtsam_points {

struct VoxelMapInfo {
  int num_voxels;
  int num_buckets;
  int max_bucket_scan_count;
  float voxel_resolution;
};

class GaussianVoxelMapGPU : public GaussianVoxelMap, public HipVoxelMapHolder {
public:
  using Ptr = std::shared_ptr<GaussianVoxelMapGPU>;
  using ConstPtr = std::shared_ptr<const GaussianVoxelMapGPU>;

  // init_num_buckets / max_bucket_scan_count / target_points_drop_rate are accepted and ignored: the table is sized from the cloud and no
  // point is ever dropped (include/glim_amd.h)
  GaussianVoxelMapGPU(float resolution, int init_num_buckets = 8192 * 2, int max_bucket_scan_count = 10, double target_points_drop_rate = 1e-3,
                      CUstream_st* stream = nullptr)
  : impl_(std::make_shared<glim_amd::GaussianVoxelMapGPU>(resolution, init_num_buckets, max_bucket_scan_count, target_points_drop_rate,
                                                          glim_amd::context_of(stream))) {
    voxelmap_info.num_voxels = 0;
    voxelmap_info.num_buckets = 0;
    voxelmap_info.max_bucket_scan_count = max_bucket_scan_count;
    voxelmap_info.voxel_resolution = resolution;
  }
  ~GaussianVoxelMapGPU() override {}

  double voxel_resolution() const override { return impl_->voxel_resolution(); }
  void insert(const PointCloud& frame) override {
    const auto* gpu = dynamic_cast<const PointCloudGPU*>(&frame);
    if (gpu) impl_->insert(*gpu->device());
    else impl_->insert(*glim_amd::clone(frame));
    const auto info = impl_->voxelmap_info();
    voxelmap_info.num_voxels = info.num_voxels;
    voxelmap_info.num_buckets = info.num_buckets;
  }
  void save_compact(const std::string& /*path*/) const { throw std::runtime_error("GaussianVoxelMapGPU::save_compact: not supported (upstream neither)"); }
  size_t memory_usage_gpu() const { return impl_->voxelmap_info().bytes; }
  glim_amd::GaussianVoxelMapGPU::ConstPtr device() const override { return impl_; }

  VoxelMapInfo voxelmap_info;  // standard_viewer_mem.cpp:77 reads num_voxels / num_buckets

private:
  std::shared_ptr<glim_amd::GaussianVoxelMapGPU> impl_;
};

// (stream: accepted for the signature; the call runs on the context of the target map, which is the calling module's own)
inline double overlap_gpu(const GaussianVoxelMap::ConstPtr& target, const PointCloud::ConstPtr& source, const Eigen::Isometry3d& delta, CUstream_st* /*stream*/ = nullptr) {
  return glim_amd::overlap_gpu(device_map(target), device_cloud(source), delta);
}
inline double overlap_gpu(const std::vector<GaussianVoxelMap::ConstPtr>& targets, const PointCloud::ConstPtr& source, const std::vector<Eigen::Isometry3d>& deltas,
                          CUstream_st* /*stream*/ = nullptr) {
  std::vector<glim_amd::GaussianVoxelMapGPU::ConstPtr> t;
  for (const auto& m : targets) t.push_back(device_map(m));
  return glim_amd::overlap_gpu(t, device_cloud(source), deltas);
}
// The CPU overlap of the (CPU-only) gtsam_points install this tree is layered over -- libgtsam_points defines it; declared here so that
// overlap_auto can fall back to it for a map that is NOT device-backed (a map of some other GaussianVoxelMap subclass).
double overlap(const GaussianVoxelMap::ConstPtr& target, const PointCloud::ConstPtr& source, const Eigen::Isometry3d& delta);

// overlap_auto dispatches on where the map lives, as upstream does: a device-backed target -- GaussianVoxelMapGPU, and the GaussianVoxelMapCPU of
// this include tree, which a libglim run with enable_gpu = false creates (global_mapping.cpp:275) and hands to this call unconditionally at
// sub_mapping.cpp:253 and global_mapping.cpp:322,448 -- takes the device path; anything else falls back to the CPU overlap.
inline double overlap_auto(const GaussianVoxelMap::ConstPtr& target, const PointCloud::ConstPtr& source, const Eigen::Isometry3d& delta) {
  if (dynamic_cast<const HipVoxelMapHolder*>(target.get())) return overlap_gpu(target, source, delta);
  return overlap(target, source, delta);
}

}  // namespace gtsam_points
