// glim_preprocess_compat.hpp -- header-only C++17 mirror, over the C ABI (glim_amd.h), of the per-scan front end GLIM runs
// before the VGICP factors: glim::CloudPreprocessor (include/glim/preprocess/cloud_preprocessor.hpp,
// src/glim/preprocess/cloud_preprocessor.cpp) and glim::CloudDeskewing (include/glim/common/cloud_deskewing.hpp).  Same class and
// member names, same argument meaning; the call sites translate one to one:
//
//   glim_amd::CloudPreprocessor preprocessor(params);                       // glim_ros / offline tools construct it from the config
//   auto frame = preprocessor.preprocess(raw_points);                        // PreprocessedFrame: times, points, intensities, neighbors
//   auto deskewed = deskewing.deskew(frame, T_imu_lidar, imu_times, imu_poses, raw->stamp, CloudDeskewing::Frame::IMU);
//                                                                            // odometry_estimation_imu.cpp:313-316: deskew + `pt = T_imu_lidar * pt`
//   deskewed->estimate_covariances(k);                                      // :320  (neighbours of the raw scan, carried over; FP64 points)
//
// Differences from the reference, all on purpose:
//   * the preprocessed cloud also stays on the device (PreprocessedFrame::gpu), so deskewing, covariance estimation, the voxel map
//     and the factors consume it without another upload; the host vectors are filled for the callers that read them;
//   * random-grid sampling draws from a counter-based generator (CloudPreprocessorParams::seed) instead of a std::mt19937 member:
//     the same scan and seed give the same sample on every run and every device.
// Eigen is not available in this image: Vector4d = std::array<double, 4>, Isometry3d = 12 doubles (gtsam_points_compat.hpp).
#pragma once

#include <array>
#include <string>
#include <vector>

#include "gtsam_points_compat.hpp"

namespace glim_amd {

using Vector4d = std::array<double, 4>;
using Vector3d = std::array<double, 3>;

// glim::RawPoints (include/glim/util/raw_points.hpp:12-27)
struct RawPoints {
  using Ptr = std::shared_ptr<RawPoints>;
  using ConstPtr = std::shared_ptr<const RawPoints>;
  int size() const { return (int)points.size(); }
  double stamp = 0.0;
  std::vector<double> times;
  std::vector<double> intensities;
  std::vector<Vector4d> points;
};

// glim::PreprocessedFrame (include/glim/preprocess/preprocessed_frame.hpp:14-39) + the device-resident cloud
struct PreprocessedFrame {
  using Ptr = std::shared_ptr<PreprocessedFrame>;
  using ConstPtr = std::shared_ptr<const PreprocessedFrame>;
  int size() const { return (int)points.size(); }
  double stamp = 0.0;
  double scan_end_time = 0.0;
  std::vector<double> times;
  std::vector<double> intensities;
  std::vector<Vector4d> points;
  int k_neighbors = 0;
  std::vector<int> neighbors;
  RawPoints::ConstPtr raw_points;
  PointCloudGPU::Ptr gpu;  // the same points (FP32 + exact FP64), times, intensities and neighbours in HBM
};

// glim::CloudPreprocessorParams (cloud_preprocessor.hpp:14-41); defaults = config/config_preprocess.json
struct CloudPreprocessorParams {
  double distance_near_thresh = 0.5;
  double distance_far_thresh = 100.0;
  bool global_shutter = false;
  bool use_random_grid_downsampling = true;
  double downsample_resolution = 1.0;
  int downsample_target = 10000;
  double downsample_rate = 0.1;
  bool enable_outlier_removal = false;
  int outlier_removal_k = 10;
  double outlier_std_mul_factor = 1.0;
  bool enable_cropbox_filter = false;
  std::string crop_bbox_frame = "lidar";
  Vector3d crop_bbox_min{{-1.0, -1.0, -1.0}};
  Vector3d crop_bbox_max{{1.0, 1.0, 1.0}};
  Isometry3d T_imu_lidar;
  int k_correspondences = 10;
  int num_threads = 2;              // unused on the device
  std::uint64_t seed = 0;           // replaces the std::mt19937 member of the reference (cloud_preprocessor.hpp:74)
  int voxelgrid_block_size = 1024;  // gtsam_points::voxelgrid_sampling averages inside blocks of 1024 sorted points

  glim_amd_preprocess_params c_params() const {
    if (crop_bbox_frame != "lidar" && crop_bbox_frame != "imu") throw std::runtime_error("Unsupported crop bbox frame: " + crop_bbox_frame);  // cloud_preprocessor.cpp:49,159
    glim_amd_preprocess_params p;
    check(glim_amd_preprocess_default_params(&p), "preprocess_default_params");
    p.distance_near_thresh = distance_near_thresh;
    p.distance_far_thresh = distance_far_thresh;
    p.use_random_grid_downsampling = use_random_grid_downsampling;
    p.downsample_target = downsample_target;
    p.downsample_resolution = downsample_resolution;
    p.downsample_rate = downsample_rate;
    p.global_shutter = global_shutter;
    p.enable_outlier_removal = enable_outlier_removal;
    p.outlier_removal_k = outlier_removal_k;
    p.outlier_std_mul_factor = outlier_std_mul_factor;
    p.enable_cropbox_filter = enable_cropbox_filter;
    p.crop_bbox_frame_imu = crop_bbox_frame == "imu";
    for (int a = 0; a < 3; a++) {
      p.crop_bbox_min[a] = crop_bbox_min[a];
      p.crop_bbox_max[a] = crop_bbox_max[a];
    }
    for (int i = 0; i < 12; i++) p.T_imu_lidar[i] = T_imu_lidar.m[i];
    p.k_correspondences = k_correspondences;
    p.voxelgrid_block_size = voxelgrid_block_size;
    p.seed = seed;
    return p;
  }
};

// PointCloudGPU built by the C ABI calls of this header (friend-free: goes through the public adopt() below)
inline PointCloudGPU::Ptr adopt_cloud(glim_amd_cloud* h, Context ctx) { return PointCloudGPU::adopt(h, std::move(ctx)); }

// glim::CloudPreprocessor (cloud_preprocessor.hpp:47-78)
class CloudPreprocessor {
public:
  explicit CloudPreprocessor(const CloudPreprocessorParams& params = CloudPreprocessorParams(), Context ctx = nullptr)
  : params(params), ctx_(ctx ? ctx : StreamTempBufferRoundRobin::default_instance()) {}
  virtual ~CloudPreprocessor() {}

  // cloud_preprocessor.cpp:74-188
  virtual PreprocessedFrame::Ptr preprocess(const RawPoints::ConstPtr& raw_points) {
    const std::int64_t n = raw_points->size();
    if ((std::int64_t)raw_points->times.size() != n) throw std::runtime_error("CloudPreprocessor: times / points size mismatch");
    const bool has_int = !raw_points->intensities.empty();
    glim_amd_preprocess_params p = params.c_params();
    p.seed = params.seed + frame_count_++;  // a fresh, reproducible sample per frame (the reference advances its mt19937)
    glim_amd_cloud* h = nullptr;
    check(glim_amd_preprocess(ctx_->context(), n, n ? raw_points->points[0].data() : nullptr, raw_points->times.data(),
                              has_int ? raw_points->intensities.data() : nullptr, &p, &h),
          "CloudPreprocessor::preprocess");
    auto frame = std::make_shared<PreprocessedFrame>();
    frame->gpu = adopt_cloud(h, ctx_);
    const std::size_t m = frame->gpu->size();
    frame->stamp = raw_points->stamp;
    frame->times.resize(m);
    frame->points.resize(m);
    if (has_int) frame->intensities.resize(m);
    frame->k_neighbors = params.k_correspondences;
    frame->neighbors.resize(m * (std::size_t)params.k_correspondences);
    check(glim_amd_cloud_download_frame(h, m ? frame->points[0].data() : nullptr, frame->times.data(), has_int ? frame->intensities.data() : nullptr,
                                        params.k_correspondences > 0 ? frame->neighbors.data() : nullptr),
          "CloudPreprocessor::download");
    frame->scan_end_time = m ? raw_points->stamp + frame->times[m - 1] : raw_points->stamp;  // cloud_preprocessor.cpp:171
    frame->raw_points = raw_points;
    return frame;
  }

  CloudPreprocessorParams params;

private:
  Context ctx_;
  std::uint64_t frame_count_ = 0;
};

// glim::CloudDeskewing (cloud_deskewing.hpp:11-54): both forms, applied to the device-resident preprocessed frame; the result is
// the deskewed device cloud with the raw scan's neighbour lists, ready for estimate_covariances() (odometry_estimation_imu.cpp:313-320).
//
// `out` (no default: the call site has to say it): which frame the result is expressed in.  CloudDeskewing::deskew itself returns LiDAR-frame
// points (Frame::LIDAR), and BOTH of its callers move every point into the IMU frame right away -- `for (auto& pt : deskewed) pt =
// T_imu_lidar * pt;` (odometry_estimation_imu.cpp:314-316, sub_mapping.cpp:368-370) -- BEFORE they estimate covariances: Frame::IMU fuses that
// loop into the deskewing kernel as a second FP64 product, and is what a port of those two call sites must pass (the result is a device
// cloud, so the host loop cannot run on it afterwards; with Frame::LIDAR the normals would face the LiDAR origin and every map / factor built
// from the cloud would live in the LiDAR frame).
class CloudDeskewing {
public:
  enum class Frame { LIDAR = 0, IMU = 1 };
  PointCloudGPU::Ptr deskew(const PreprocessedFrame& frame, const Isometry3d& T_imu_lidar, const Vector3d& linear_vel, const Vector3d& angular_vel,
                            Frame out) const {
    glim_amd_cloud* h = nullptr;
    check(glim_amd_cloud_deskew(frame.gpu->handle(), T_imu_lidar.m.data(), 0, nullptr, nullptr, 0.0, linear_vel.data(), angular_vel.data(),
                                out == Frame::IMU ? 1 : 0, &h),
          "CloudDeskewing::deskew");
    return adopt_cloud(h, frame.gpu->context());
  }
  PointCloudGPU::Ptr deskew(const PreprocessedFrame& frame, const Isometry3d& T_imu_lidar, const std::vector<double>& imu_times,
                            const std::vector<Isometry3d>& imu_poses, double stamp, Frame out) const {
    std::vector<double> poses(12 * imu_poses.size());
    for (std::size_t i = 0; i < imu_poses.size(); i++) std::memcpy(&poses[12 * i], imu_poses[i].m.data(), 12 * sizeof(double));
    glim_amd_cloud* h = nullptr;
    check(glim_amd_cloud_deskew(frame.gpu->handle(), T_imu_lidar.m.data(), (std::int32_t)imu_times.size(), imu_times.data(), poses.data(), stamp, nullptr,
                                nullptr, out == Frame::IMU ? 1 : 0, &h),
          "CloudDeskewing::deskew");
    return adopt_cloud(h, frame.gpu->context());
  }
};

// gtsam_points::merge_frames(poses, frames, downsample_resolution, target_num_points) as called at sub_mapping.cpp:480-497.
// A frame is what gtsam_points::PointCloud exposes: raw pointers to Vector4d points and Matrix4d covariances.
struct FrameView {
  const double* points4 = nullptr;  // n x Vector4d
  const double* covs16 = nullptr;   // n x Matrix4d, column-major
  std::int64_t size = 0;
};
struct MergedFrame {  // gtsam_points::PointCloudCPU fields of the merged submap + the device-resident cloud
  std::vector<Vector4d> points;
  std::vector<std::array<double, 16>> covs;
  PointCloudGPU::Ptr gpu;
  std::size_t size() const { return points.size(); }
};
inline MergedFrame merge_frames(const std::vector<Isometry3d>& poses, const std::vector<FrameView>& frames, double downsample_resolution,
                                int target_num_points = -1, std::uint64_t seed = 0, Context ctx = nullptr, bool download = true) {
  if (poses.size() != frames.size()) throw std::runtime_error("merge_frames: poses / frames size mismatch");
  ctx = ctx ? ctx : StreamTempBufferRoundRobin::default_instance();
  const std::size_t nf = frames.size();
  std::vector<double> poses12(12 * nf);
  std::vector<const double*> pp(nf), cp(nf);
  std::vector<std::int64_t> sizes(nf);
  for (std::size_t f = 0; f < nf; f++) {
    std::memcpy(&poses12[12 * f], poses[f].m.data(), 12 * sizeof(double));
    pp[f] = frames[f].points4;
    cp[f] = frames[f].covs16;
    sizes[f] = frames[f].size;
  }
  glim_amd_cloud* h = nullptr;
  check(glim_amd_merge_frames(ctx->context(), (std::int32_t)nf, poses12.data(), pp.data(), cp.data(), sizes.data(), downsample_resolution, target_num_points,
                              1024, seed, &h),
        "merge_frames");
  MergedFrame out;
  out.gpu = adopt_cloud(h, ctx);
  if (download) {
    const std::size_t m = out.gpu->size();
    out.points.resize(m);
    out.covs.resize(m);
    check(glim_amd_cloud_download_merged(h, m ? out.points[0].data() : nullptr, m ? out.covs[0].data() : nullptr), "merge_frames download");
  }
  return out;
}

}  // namespace glim_amd
