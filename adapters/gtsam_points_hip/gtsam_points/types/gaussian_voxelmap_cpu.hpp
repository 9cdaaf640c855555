// gtsam_points/types/gaussian_voxelmap_cpu.hpp, MI355X edition: the CPU-NAMED voxel map, device-backed.
//   GaussianVoxelMapCPU(resolution) + set_lru_horizon(lru_thresh) + insert(frame) per frame    odometry_estimation_cpu.cpp:63-68, :177-191
//   GaussianVoxelMapCPU(resolution) + insert(frame) once                                        sub_mapping.cpp:409-410, global_mapping.cpp:275-276,757-758,
//                                                                                               global_mapping_pose_graph.cpp:276-277
// With this header in front of the real one, the module BASELINE configs[0] names (config_odometry_cpu.json, registration_type VGICP), the
// loop-closure validator of global_mapping_pose_graph.cpp and the enable_gpu = false branches of sub_mapping.cpp / global_mapping.cpp build their
// maps with glim_amd_voxelmap_* WITHOUT AN EDIT: a second insert() re-opens the voxels of the first (GaussianVoxelMapCPU semantics, which is
// what glim_amd_voxelmap_insert implements) and set_lru_horizon reaches glim_amd_voxelmap_set_lru_horizon.
// Not offered (no GLIM caller outside the viewers): save_compact / load, the per-voxel accessors of IncrementalVoxelMap.  voxel_points() -- the
// visualisation cloud of odometry_estimation_cpu.cpp:212 -- returns the voxel means.
#pragma once

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <Eigen/Core>
#include <gtsam_points/types/gaussian_voxelmap.hpp>
#include <gtsam_points/types/hip_voxelmap_holder.hpp>
#include <gtsam_points/types/point_cloud_gpu.hpp>
#include <gtsam_points/util/vector3i_hash.hpp>  // (upstream's header pulls it in, and global_mapping.cpp:291 relies on that)

namespace gtsam_points {

class GaussianVoxelMapCPU : public GaussianVoxelMap, public HipVoxelMapHolder {
public:
  using Ptr = std::shared_ptr<GaussianVoxelMapCPU>;
  using ConstPtr = std::shared_ptr<const GaussianVoxelMapCPU>;

  explicit GaussianVoxelMapCPU(double resolution) : impl_(std::make_shared<glim_amd::GaussianVoxelMapGPU>((float)resolution)), resolution_(resolution) {}
  ~GaussianVoxelMapCPU() override {}

  double voxel_resolution() const override { return resolution_; }
  // every insert ADDS to the voxels already there (the CPU odometry inserts every frame into one map: odometry_estimation_cpu.cpp:189)
  void insert(const PointCloud& frame) override {
    const auto* gpu = dynamic_cast<const PointCloudGPU*>(&frame);
    if (gpu) impl_->insert(*gpu->device());
    else impl_->insert(*glim_amd::clone(frame));
  }
  // IncrementalVoxelMap::set_lru_horizon / set_lru_clear_cycle (odometry_estimation_cpu.cpp:67; config_odometry_cpu.json "lru_thresh": 100)
  void set_lru_horizon(int lru_horizon) {
    lru_horizon_ = lru_horizon;
    impl_->set_lru_horizon(lru_horizon_, lru_clear_cycle_);
  }
  void set_lru_clear_cycle(int lru_clear_cycle) {
    lru_clear_cycle_ = lru_clear_cycle;
    impl_->set_lru_horizon(lru_horizon_, lru_clear_cycle_);
  }
  size_t num_voxels() const { return (size_t)impl_->voxelmap_info().num_voxels; }
  // the voxel means as a point list (odometry_estimation_cpu.cpp:212: visualisation of the target model)
  std::vector<Eigen::Vector4d> voxel_points() const {
    const auto info = impl_->voxelmap_info();
    std::vector<float> means((size_t)info.num_voxels * 3);
    if (info.num_voxels > 0) glim_amd::check(glim_amd_voxelmap_download(impl_->handle(), nullptr, nullptr, means.data(), nullptr), "GaussianVoxelMapCPU::voxel_points");
    std::vector<Eigen::Vector4d> out((size_t)info.num_voxels);
    for (size_t i = 0; i < out.size(); i++) {
      out[i](0) = means[3 * i];
      out[i](1) = means[3 * i + 1];
      out[i](2) = means[3 * i + 2];
      out[i](3) = 1.0;
    }
    return out;
  }
  void save_compact(const std::string& /*path*/) const override { throw std::runtime_error("GaussianVoxelMapCPU (HIP-backed)::save_compact: not supported"); }
  static Ptr load(const std::string& /*path*/) { throw std::runtime_error("GaussianVoxelMapCPU (HIP-backed)::load: not supported"); }

  glim_amd::GaussianVoxelMapGPU::ConstPtr device() const override { return impl_; }

private:
  std::shared_ptr<glim_amd::GaussianVoxelMapGPU> impl_;
  double resolution_;
  int lru_horizon_ = 0, lru_clear_cycle_ = 10;
};

}  // namespace gtsam_points

// overlap_gpu / overlap_auto (upstream declares the overlap family next to the CPU map as well)
#include <gtsam_points/types/gaussian_voxelmap_gpu.hpp>
