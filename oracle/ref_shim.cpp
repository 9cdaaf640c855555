// oracle/ref_shim.cpp -- C entry points around the REFERENCE's own translation units (test infrastructure).
//
// oracle/_ref/libglim_ref.so = /root/reference/src/glim/common/cloud_covariance_estimation.cpp + cloud_deskewing.cpp compiled UNMODIFIED from
// where they lie (oracle/Makefile, target `ref`) against the stand-in headers of oracle/ref_standin/ (Eigen, GTSAM, gtsam_points and spdlog are not
// installed in this image), plus this file.  The signatures mirror orc_covariance_estimate / orc_deskew_* (vgicp_oracle.h) so the tests can
// run the restatement and the reference code side by side on the same arrays.  (cloud_preprocessor.cpp: ref_preprocess_shim.cpp.)
#include <glim/common/cloud_covariance_estimation.hpp>
#include <glim/common/cloud_deskewing.hpp>

#include <cstdint>
#include <cstring>
#include <vector>

namespace {
Eigen::Isometry3d pose_from12(const double* T) {
  Eigen::Isometry3d P = Eigen::Isometry3d::Identity();
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) P.matrix()(r, c) = T[4 * r + c];
  return P;
}
std::vector<Eigen::Vector4d> points_from4(const double* p, int n) {
  std::vector<Eigen::Vector4d> v((size_t)n);
  for (int i = 0; i < n; i++) v[(size_t)i] = Eigen::Vector4d(p[4 * (size_t)i], p[4 * (size_t)i + 1], p[4 * (size_t)i + 2], p[4 * (size_t)i + 3]);
  return v;
}
void points_to4(const std::vector<Eigen::Vector4d>& v, double* out) {
  for (size_t i = 0; i < v.size(); i++)
    for (int k = 0; k < 4; k++) out[4 * i + k] = v[i][k];
}
}  // namespace

extern "C" {

// glim::CloudCovarianceEstimation(num_threads).estimate(points, neighbors, k_neighbors, normals, covs)   cloud_covariance_estimation.cpp:43-122
int ref_covariance_estimate(const double* pts4, int n, const int32_t* nbrs, int k_corr, int k_nbr, double* normals4, double* covs16, int num_threads) {
  if (n <= 0) return 0;
  if (k_nbr > k_corr || k_nbr <= 0) return -1;
  const std::vector<Eigen::Vector4d> points = points_from4(pts4, n);
  const std::vector<int> neighbors(nbrs, nbrs + (size_t)n * k_corr);
  std::vector<Eigen::Vector4d> normals;
  std::vector<Eigen::Matrix4d> covs;
  glim::CloudCovarianceEstimation est(num_threads > 0 ? num_threads : 1);
  est.estimate(points, neighbors, k_nbr, normals, covs);
  points_to4(normals, normals4);
  for (int i = 0; i < n; i++) std::memcpy(covs16 + 16 * (size_t)i, covs[(size_t)i].data(), 16 * sizeof(double));  // column-major Matrix4d
  return 0;
}

// glim::CloudDeskewing::deskew(T_imu_lidar, linear_vel, angular_vel, times, points)   cloud_deskewing.cpp:11-53
int ref_deskew_constvel(const double* T_imu_lidar12, const double* linear_vel3, const double* angular_vel3, const double* times, const double* pts4,
                        int n, double* out4) {
  if (n <= 0) return 0;
  glim::CloudDeskewing d;
  const std::vector<double> t(times, times + n);
  const auto out = d.deskew(pose_from12(T_imu_lidar12), Eigen::Vector3d(linear_vel3[0], linear_vel3[1], linear_vel3[2]),
                            Eigen::Vector3d(angular_vel3[0], angular_vel3[1], angular_vel3[2]), t, points_from4(pts4, n));
  points_to4(out, out4);
  return 0;
}

// glim::CloudDeskewing::deskew(T_imu_lidar, imu_times, imu_poses, stamp, times, points)   cloud_deskewing.cpp:55-133
int ref_deskew_imu(const double* T_imu_lidar12, const double* imu_times, const double* imu_poses12, int n_imu, double stamp, const double* times,
                   const double* pts4, int n, double* out4) {
  if (n <= 0) return 0;
  glim::CloudDeskewing d;
  const std::vector<double> t(times, times + n);
  const std::vector<double> it(imu_times, imu_times + (n_imu > 0 ? n_imu : 0));
  std::vector<Eigen::Isometry3d> poses;
  for (int i = 0; i < n_imu; i++) poses.push_back(pose_from12(imu_poses12 + 12 * (size_t)i));
  const auto out = d.deskew(pose_from12(T_imu_lidar12), it, poses, stamp, t, points_from4(pts4, n));
  points_to4(out, out4);
  return 0;
}

// The per-scan front end between preprocessing and create_frame, as OdometryEstimationIMU::insert_frame runs it
// (src/glim/odometry/odometry_estimation_imu.cpp:313-320), on the reference's own objects:
//     auto deskewed = deskewing->deskew(T_imu_lidar, pred_imu_times, pred_imu_poses, raw_frame->stamp, raw_frame->times, raw_frame->points);
//     for (auto& pt : deskewed) { pt = T_imu_lidar * pt; }
//     covariance_estimation->estimate(deskewed, raw_frame->neighbors, deskewed_normals, deskewed_covs);
// (sub_mapping.cpp:364-374 is the same chain with T_lidar_imu.inverse().)  n_imu < 0 selects the constant-velocity form of deskew().
// to_imu_frame == 0 skips the middle line (what a LiDAR-frame cloud would give; for the tests that show the difference).
int ref_frontend(const double* T_imu_lidar12, const double* imu_times, const double* imu_poses12, int n_imu, double stamp, const double* linear_vel3,
                 const double* angular_vel3, const double* times, const double* pts4, int n, const int32_t* nbrs, int k, int to_imu_frame,
                 double* out_points4, double* out_normals4, double* out_covs16, int num_threads) {
  if (n <= 0) return 0;
  glim::CloudDeskewing deskewing;
  const Eigen::Isometry3d T_imu_lidar = pose_from12(T_imu_lidar12);
  const std::vector<double> t(times, times + n);
  std::vector<Eigen::Vector4d> deskewed;
  if (n_imu < 0) {
    deskewed = deskewing.deskew(T_imu_lidar, Eigen::Vector3d(linear_vel3[0], linear_vel3[1], linear_vel3[2]),
                                Eigen::Vector3d(angular_vel3[0], angular_vel3[1], angular_vel3[2]), t, points_from4(pts4, n));
  } else {
    const std::vector<double> it(imu_times, imu_times + n_imu);
    std::vector<Eigen::Isometry3d> poses;
    for (int i = 0; i < n_imu; i++) poses.push_back(pose_from12(imu_poses12 + 12 * (size_t)i));
    deskewed = deskewing.deskew(T_imu_lidar, it, poses, stamp, t, points_from4(pts4, n));
  }
  if (to_imu_frame)
    for (auto& pt : deskewed) {
      pt = T_imu_lidar * pt;
    }
  const std::vector<int> neighbors(nbrs, nbrs + (size_t)n * k);
  std::vector<Eigen::Vector4d> normals;
  std::vector<Eigen::Matrix4d> covs;
  glim::CloudCovarianceEstimation est(num_threads > 0 ? num_threads : 1);
  est.estimate(deskewed, neighbors, normals, covs);  // the 4-argument overload: k = neighbors.size() / points.size()
  points_to4(deskewed, out_points4);
  points_to4(normals, out_normals4);
  for (int i = 0; i < n; i++) std::memcpy(out_covs16 + 16 * (size_t)i, covs[(size_t)i].data(), 16 * sizeof(double));
  return 0;
}

}  // extern "C"
