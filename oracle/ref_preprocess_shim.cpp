// oracle/ref_preprocess_shim.cpp -- C entry point around the REFERENCE's own CloudPreprocessor (test infrastructure, part of oracle/_ref).
//
// /root/reference/src/glim/preprocess/cloud_preprocessor.cpp is compiled UNMODIFIED from where it lies.  It needs three things this file
// supplies: (1) glim::Config / GlobalConfig, whose real implementation parses JSON files with nlohmann::json (not installed): here every
// param() returns the default the caller passes, and the shim then sets the public fields of CloudPreprocessorParams from its arguments;
// (2) the definitions of the PreprocessCallbacks slots (src/glim/preprocess/callbacks.cpp upstream); (3) a C signature that mirrors
// orc_preprocess (vgicp_oracle.h) so that tests run the restatement and the reference code side by side.  The gtsam_points calls of the
// reference are answered as described in ref_standin/gtsam_points/types/point_cloud_cpu.hpp.
#include <glim/preprocess/callbacks.hpp>
#include <glim/preprocess/cloud_preprocessor.hpp>
#include <glim/util/config.hpp>

#include <cstdint>
#include <cstring>
#include <stdexcept>

#include "vgicp_oracle.h"

namespace glim {

Config::Config(const std::string&) {}
Config::~Config() {}
template <typename T>
T Config::param(const std::string&, const std::string&, const T& default_value) const {
  return default_value;
}
template bool Config::param<bool>(const std::string&, const std::string&, const bool&) const;
template int Config::param<int>(const std::string&, const std::string&, const int&) const;
template double Config::param<double>(const std::string&, const std::string&, const double&) const;
template std::string Config::param<std::string>(const std::string&, const std::string&, const std::string&) const;
template Eigen::Vector3d Config::param<Eigen::Vector3d>(const std::string&, const std::string&, const Eigen::Vector3d&) const;
template Eigen::Isometry3d Config::param<Eigen::Isometry3d>(const std::string&, const std::string&, const Eigen::Isometry3d&) const;
std::string GlobalConfig::get_config_path(const std::string&) { return std::string(); }

CallbackSlot<void(const RawPoints::ConstPtr& points)> PreprocessCallbacks::on_raw_points_received;
CallbackSlot<void(gtsam_points::PointCloudCPU::Ptr& points)> PreprocessCallbacks::on_preprocessing_begin;
CallbackSlot<void(gtsam_points::PointCloudCPU::Ptr& points)> PreprocessCallbacks::on_downsampling_finished;
CallbackSlot<void(gtsam_points::PointCloudCPU::Ptr& points)> PreprocessCallbacks::on_filtering_finished;

}  // namespace glim

extern "C" {

// glim::CloudPreprocessor(params).preprocess(raw_points)   cloud_preprocessor.cpp:74-188 (+ find_neighbors :190-221)
// Same arguments and outputs as orc_preprocess; meta[0] = stamp + last time (PreprocessedFrame::scan_end_time for stamp = 0), meta[1] = k_neighbors.
int ref_preprocess(const double* points4, const double* times, const double* intensities, int n, const orc_preprocess_params* prm, double* out_points4,
                   double* out_times, double* out_intensities, int32_t* out_neighbors, double* meta, int num_threads) {
  glim::CloudPreprocessorParams p;  // Config stub: the defaults of cloud_preprocessor.cpp:26-58
  p.global_shutter = prm->global_shutter != 0;
  p.distance_near_thresh = prm->distance_near_thresh;
  p.distance_far_thresh = prm->distance_far_thresh;
  p.use_random_grid_downsampling = prm->use_random_grid_downsampling != 0;
  p.downsample_resolution = prm->downsample_resolution;
  p.downsample_target = prm->downsample_target;
  p.downsample_rate = prm->downsample_rate;
  p.enable_outlier_removal = prm->enable_outlier_removal != 0;
  p.outlier_removal_k = prm->outlier_removal_k;
  p.outlier_std_mul_factor = prm->outlier_std_mul_factor;
  p.enable_cropbox_filter = prm->enable_cropbox_filter != 0;
  p.crop_bbox_frame = prm->crop_bbox_frame_imu ? "imu" : "lidar";
  p.crop_bbox_min = Eigen::Vector3d(prm->crop_bbox_min[0], prm->crop_bbox_min[1], prm->crop_bbox_min[2]);
  p.crop_bbox_max = Eigen::Vector3d(prm->crop_bbox_max[0], prm->crop_bbox_max[1], prm->crop_bbox_max[2]);
  p.T_imu_lidar = Eigen::Isometry3d::Identity();
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) p.T_imu_lidar.matrix()(r, c) = prm->T_imu_lidar[4 * r + c];
  p.k_correspondences = prm->k_correspondences;
  p.num_threads = num_threads > 0 ? num_threads : 1;
  gtsam_points::RefSamplingContext::get().seed = prm->seed;
  gtsam_points::RefSamplingContext::get().voxelgrid_block_size = prm->voxelgrid_block_size;

  auto raw = std::make_shared<glim::RawPoints>();
  raw->stamp = 0.0;
  raw->times.assign(times, times + n);
  raw->points.resize((size_t)n);
  for (int i = 0; i < n; i++) raw->points[(size_t)i] = Eigen::Vector4d(points4[4 * (size_t)i], points4[4 * (size_t)i + 1], points4[4 * (size_t)i + 2], points4[4 * (size_t)i + 3]);
  if (intensities) raw->intensities.assign(intensities, intensities + n);

  glim::CloudPreprocessor pre(p);
  glim::PreprocessedFrame::Ptr out;
  try {
    out = pre.preprocess(raw);
  } catch (const std::exception&) {
    return -1;
  }
  const int m = out->size();
  for (int i = 0; i < m; i++)
    for (int k = 0; k < 4; k++) out_points4[4 * (size_t)i + k] = out->points[(size_t)i][k];
  std::memcpy(out_times, out->times.data(), sizeof(double) * (size_t)m);
  if (out_intensities && !out->intensities.empty()) std::memcpy(out_intensities, out->intensities.data(), sizeof(double) * (size_t)m);
  if (out_neighbors) std::memcpy(out_neighbors, out->neighbors.data(), sizeof(int32_t) * (size_t)m * (size_t)out->k_neighbors);
  if (meta) {
    meta[0] = out->scan_end_time;
    meta[1] = (double)out->k_neighbors;
  }
  return m;
}

}  // extern "C"
