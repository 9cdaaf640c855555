// cloud_deskewing_hip.cpp -- the HIP-backed twin of GLIM's src/glim/common/cloud_deskewing.cpp.
//
// Same header (include/glim/common/cloud_deskewing.hpp), same class, same two deskew() overloads: a HIP build of libglim compiles THIS file in place of
// the reference's, and odometry_estimation_imu.cpp:313, sub_mapping.cpp:364-367 and the CT odometry reach the device kernel (deskew.hip, K8) without an
// edit.  Per call: points + per-point times go up, the host builds the time table the reference builds (constant twist: cloud_deskewing.cpp:22-45; IMU
// poses: :70-124, GTSAM's Pose3::Expmap in its own operation order), one kernel applies the table entry of every point in FP64, and the LiDAR-frame
// points come back bit-exact (tests/test_ref.py, tests/test_twins.py).  Callers that move the result into the IMU frame and estimate covariances next
// (odometry_estimation_imu.cpp:314-320) can keep everything on the device through glim_amd::CloudDeskewing (glim_preprocess_compat.hpp) instead.
#include <glim/common/cloud_deskewing.hpp>

#include <cstdint>
#include <vector>

#include <glim_amd/gtsam_points_compat.hpp>

namespace glim {

namespace {

void pose12(const Eigen::Isometry3d& T, double* out) {  // row-major 3x4 of the 4x4 (column-major dense storage in Eigen::Isometry3d::matrix())
  const auto& m = T.matrix();
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) out[4 * r + c] = m(r, c);
}

std::vector<Eigen::Vector4d> deskew_on_device(const Eigen::Isometry3d& T_imu_lidar, int n_imu, const double* imu_times, const double* imu_poses12, double stamp,
                                              const double* linear_vel3, const double* angular_vel3, const std::vector<double>& times,
                                              const std::vector<Eigen::Vector4d>& points) {
  static_assert(sizeof(Eigen::Vector4d) == 4 * sizeof(double), "dense fixed-size Eigen storage");
  std::vector<Eigen::Vector4d> out(points.size());
  if (points.empty()) return out;
  double T12[12];
  pose12(T_imu_lidar, T12);
  glim_amd::Context ctx = glim_amd::StreamTempBufferRoundRobin::default_instance();
  glim_amd_cloud* cloud = nullptr;
  glim_amd::check(glim_amd_cloud_create_deskewed(ctx->context(), (std::int64_t)points.size(), reinterpret_cast<const double*>(points.data()), times.data(), T12, n_imu,
                                                 imu_times, imu_poses12, stamp, linear_vel3, angular_vel3, /*to_imu_frame=*/0, &cloud),
                  "CloudDeskewing::deskew");
  const int rc = glim_amd_cloud_download_frame(cloud, reinterpret_cast<double*>(out.data()), nullptr, nullptr, nullptr);  // the exact FP64 points
  (void)glim_amd_cloud_destroy(cloud);
  glim_amd::check(rc, "CloudDeskewing::deskew download");
  return out;
}

}  // namespace

CloudDeskewing::CloudDeskewing() {}

CloudDeskewing::~CloudDeskewing() {}

std::vector<Eigen::Vector4d> CloudDeskewing::deskew(const Eigen::Isometry3d& T_imu_lidar, const Eigen::Vector3d& linear_vel, const Eigen::Vector3d& angular_vel,
                                                    const std::vector<double>& times, const std::vector<Eigen::Vector4d>& points) {
  const double v[3] = {linear_vel[0], linear_vel[1], linear_vel[2]}, w[3] = {angular_vel[0], angular_vel[1], angular_vel[2]};
  return deskew_on_device(T_imu_lidar, 0, nullptr, nullptr, 0.0, v, w, times, points);
}

std::vector<Eigen::Vector4d> CloudDeskewing::deskew(const Eigen::Isometry3d& T_imu_lidar, const std::vector<double>& imu_times,
                                                    const std::vector<Eigen::Isometry3d>& imu_poses, const double stamp, const std::vector<double>& times,
                                                    const std::vector<Eigen::Vector4d>& points) {
  std::vector<double> poses12(12 * imu_poses.size());
  for (std::size_t i = 0; i < imu_poses.size(); i++) pose12(imu_poses[i], &poses12[12 * i]);
  return deskew_on_device(T_imu_lidar, (int)imu_times.size(), imu_times.data(), poses12.data(), stamp, nullptr, nullptr, times, points);
}

}  // namespace glim
