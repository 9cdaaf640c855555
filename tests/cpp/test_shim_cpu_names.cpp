// test_shim_cpu_names.cpp -- the CPU-NAMED gtsam_points classes of adapters/gtsam_points_hip driven the way GLIM's CPU odometry drives them
// (src/glim/odometry/odometry_estimation_cpu.cpp): GaussianVoxelMapCPU(resolution) + set_lru_horizon(lru_thresh) (:63-68), insert(frame) per frame
// (update_target :177-191), IntegratedVGICPFactor(gtsam::Pose3(), X(current), voxelmap, frame) + set_num_threads (:105-110), then the optimiser
// loop of :112-149 -- linearise, solve, retract, stop when the step falls below 1e-3 m / 1e-3 deg (:121-137) -- and error / inlier_fraction as
// the loop-closure validator reads them (global_mapping_pose_graph.cpp:406-417).  Every iteration's Gauss-Newton step is compared with the CPU
// oracle's on the same inputs (north_star gate: 1e-4 m / 1e-4 rad per iteration), the voxel map with the oracle's after every insert.
// Stand-in third-party headers: tests/cpp/glim_standin (test infrastructure).  Built and run by tests/test_glim_module.py.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include <gtsam/inference/Symbol.h>
#include <gtsam/linear/HessianFactor.h>
#include <gtsam/nonlinear/NonlinearFactorGraph.h>
#include <gtsam_points/factors/integrated_gicp_factor.hpp>
#include <gtsam_points/factors/integrated_vgicp_factor.hpp>
#include <gtsam_points/types/gaussian_voxelmap_cpu.hpp>
#include <gtsam_points/types/point_cloud_cpu.hpp>
#include <gtsam_points/util/gtsam_migration.hpp>

extern "C" {
#include "../../oracle/vgicp_oracle.h"
}

using gtsam::symbol_shorthand::X;

#define REQUIRE(cond)                                                        \
  do {                                                                       \
    if (!(cond)) {                                                           \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      return 1;                                                              \
    }                                                                        \
  } while (0)

namespace gtsam_points {
double overlap(const GaussianVoxelMap::ConstPtr&, const PointCloud::ConstPtr&, const Eigen::Isometry3d&) { return -1.0; }  // libgtsam_points' CPU overlap: never reached here
}

// three walls of a room corner seen from (ox, oy, yaw), PLANE-form covariances (what CloudCovarianceEstimation emits), FP32-representable points
static gtsam_points::PointCloudCPU::Ptr make_frame(int n, double ox, double oy, double yaw, unsigned seed) {
  std::mt19937_64 rng(seed);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  std::normal_distribution<double> G(0.0, 0.004);
  auto f = std::make_shared<gtsam_points::PointCloudCPU>();
  f->points_storage.resize((size_t)n);
  f->covs_storage.resize((size_t)n);
  const double c = std::cos(yaw), s = std::sin(yaw);
  for (int i = 0; i < n; i++) {
    double x, y, z, nx = 0, ny = 0, nz = 0;
    if (i % 3 == 0) { x = 12.0 * U(rng); y = 9.0 * U(rng); z = -1.5 + G(rng); nz = 1; }
    else if (i % 3 == 1) { x = 12.0 + G(rng); y = 9.0 * U(rng); z = -1.5 + 3.0 * U(rng); nx = -1; }
    else { x = 12.0 * U(rng); y = 9.0 + G(rng); z = -1.5 + 3.0 * U(rng); ny = -1; }
    const double wx = x - ox, wy = y - oy;
    f->points_storage[(size_t)i] = Eigen::Vector4d((double)(float)(c * wx + s * wy), (double)(float)(-s * wx + c * wy), (double)(float)z, 1.0);
    const double n3[3] = {c * nx + s * ny, -s * nx + c * ny, nz};
    Eigen::Matrix4d C = Eigen::Matrix4d::Zero();
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) C(a, b) = (double)(float)((a == b ? 1.0 : 0.0) - (1.0 - 1e-3) * n3[a] * n3[b]);
    f->covs_storage[(size_t)i] = C;
  }
  f->num_points = (size_t)n;
  f->points = f->points_storage.data();
  f->covs = f->covs_storage.data();
  return f;
}

static void to12(const Eigen::Matrix4d& m, double* T) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) T[4 * r + c] = m(r, c);
}
static Eigen::Matrix4d from12(const double* T) {
  Eigen::Matrix4d m = Eigen::Matrix4d::Identity();
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) m(r, c) = T[4 * r + c];
  return m;
}
static Eigen::Matrix4d pose2d(double x, double y, double yaw) {
  Eigen::Matrix4d T = Eigen::Matrix4d::Identity();
  T(0, 0) = std::cos(yaw); T(0, 1) = -std::sin(yaw); T(1, 0) = std::sin(yaw); T(1, 1) = std::cos(yaw);
  T(0, 3) = x; T(1, 3) = y;
  return T;
}
static Eigen::Matrix4d rel(const Eigen::Matrix4d& A, const Eigen::Matrix4d& B) {  // A^-1 B for rigid A
  Eigen::Matrix4d Ai = Eigen::Matrix4d::Identity(), out = Eigen::Matrix4d::Identity();
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) Ai(r, c) = A(c, r);
    Ai(r, 3) = -(A(0, r) * A(0, 3) + A(1, r) * A(1, 3) + A(2, r) * A(2, 3));
  }
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) {
      double s = (c == 3) ? Ai(r, 3) : 0.0;
      for (int k = 0; k < 3; k++) s += Ai(r, k) * B(k, c);
      out(r, c) = s;
    }
  return out;
}
// p' = T p, C' = R C R^T (what gtsam_points::transform does to the frame update_target inserts, odometry_estimation_cpu.cpp:184)
static gtsam_points::PointCloudCPU::Ptr transformed(const gtsam_points::PointCloudCPU& f, const Eigen::Matrix4d& T) {
  auto out = std::make_shared<gtsam_points::PointCloudCPU>();
  out->points_storage.resize(f.size());
  out->covs_storage.resize(f.size());
  for (size_t i = 0; i < f.size(); i++) {
    Eigen::Vector4d p = Eigen::Vector4d(0, 0, 0, 1);
    Eigen::Matrix4d C = Eigen::Matrix4d::Zero();
    for (int r = 0; r < 3; r++) {
      double s = T(r, 3);
      for (int c = 0; c < 3; c++) s += T(r, c) * f.points[i](c);
      p(r) = (double)(float)s;  // (FP32-representable, so that the oracle and the device consume identical inputs)
    }
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        double s = 0.0;
        for (int k = 0; k < 3; k++)
          for (int l = 0; l < 3; l++) s += T(a, k) * f.covs[i](k, l) * T(b, l);
        C(a, b) = (double)(float)s;
      }
    for (int a = 0; a < 3; a++)
      for (int b = a + 1; b < 3; b++) C(b, a) = C(a, b);
    out->points_storage[i] = p;
    out->covs_storage[i] = C;
  }
  out->num_points = f.size();
  out->points = out->points_storage.data();
  out->covs = out->covs_storage.data();
  return out;
}

int main() {
  if (glim_amd_device_count() < 1) {
    std::fprintf(stderr, "no HIP device\n");
    return 2;
  }
  const int N = 16384;  // BASELINE configs[0]: 16k-pt scan vs 16k-pt target, 1.0 m voxels
  // --- OdometryEstimationCPU::OdometryEstimationCPU (:63-68) ---
  const double resolution = 1.0;
  auto voxelmap = std::make_shared<gtsam_points::GaussianVoxelMapCPU>(resolution);
  voxelmap->set_lru_horizon(100);
  REQUIRE(voxelmap->voxel_resolution() == resolution);
  orc_voxelmap* ref_map = orc_voxelmap_create(resolution);
  orc_voxelmap_set_lru(ref_map, 100, 10);
  // --- update_target for the first frames (:177-191): every frame goes INTO THE SAME map, in the target (world) frame ---
  const double px[3] = {1.0, 1.4, 1.9}, py[3] = {1.0, 1.15, 1.2}, pw[3] = {0.0, 0.03, 0.07};
  std::vector<gtsam_points::PointCloudCPU::Ptr> frames;
  for (int k = 0; k < 3; k++) frames.push_back(make_frame(N, px[k], py[k], pw[k], 11 + (unsigned)k));
  for (int k = 0; k < 2; k++) {
    const auto world = transformed(*frames[(size_t)k], pose2d(px[k], py[k], pw[k]));
    voxelmap->insert(*world);
    orc_voxelmap_insert(ref_map, reinterpret_cast<const double*>(world->points), reinterpret_cast<const double*>(world->covs), (int)world->size());
    REQUIRE((int)voxelmap->num_voxels() == orc_voxelmap_num_voxels(ref_map));  // the second insert re-opened the voxels of the first
  }
  REQUIRE(voxelmap->voxel_points().size() == voxelmap->num_voxels());
  // --- create_factors (:105-110): the unary factor of the new frame against the model, from a predicted pose ---
  const gtsam_points::PointCloud::ConstPtr frame = frames[2];
  auto vgicp_factor = gtsam::make_shared<gtsam_points::IntegratedVGICPFactor>(gtsam::Pose3(), X(2), voxelmap, frame);
  vgicp_factor->set_num_threads(2);
  gtsam::NonlinearFactorGraph graph;
  graph.add(vgicp_factor);
  REQUIRE(dynamic_cast<gtsam_points::IntegratedVGICPFactor*>(graph[0].get()) != nullptr);  // global_mapping.cpp:586
  REQUIRE(vgicp_factor->dim() == 6 && vgicp_factor->keys().size() == 1 && vgicp_factor->clone() != nullptr);
  const Eigen::Matrix4d truth = pose2d(px[2], py[2], pw[2]);
  double xi[6] = {0.01, -0.02, 0.015, 0.10, -0.05, 0.02}, E[12], T[12], Tref[12];
  orc_se3_exp(xi, E);
  to12(truth, T);
  orc_pose_compose(T, E, T);  // predicted pose = truth * Exp([0.01 -0.02 0.015 rad; 0.10 -0.05 0.02 m])  
  memcpy(Tref, T, sizeof(T));
  // --- the optimiser loop (:112-149): max_iterations 8 (config_odometry_cpu.json:23) ---
  double ref_deltas[6 * 8];
  const int ref_iters = orc_gn_align(ref_map, reinterpret_cast<const double*>(frame->points), reinterpret_cast<const double*>(frame->covs), N, Tref, 8, 0.0, 0, ref_deltas);
  double worst_step = 0.0;
  int it = 0;
  for (; it < 8; it++) {
    gtsam::Values values;
    values.insert(X(2), gtsam::Pose3(from12(T)));
    // error(values): correspondences AT `values` (CPU-factor semantics) -- the oracle's recomputed error
    int64_t inl_ref = 0;
    const double e_ref = orc_vgicp_error(ref_map, reinterpret_cast<const double*>(frame->points), reinterpret_cast<const double*>(frame->covs), N, T, 0, &inl_ref);
    const double e = vgicp_factor->error(values);
    REQUIRE(std::fabs(e - e_ref) <= 3e-4 * std::fabs(e_ref) + 1e-9);
    auto lin = std::dynamic_pointer_cast<gtsam::HessianFactor>(vgicp_factor->linearize(values));
    REQUIRE(lin != nullptr && lin->keys_.size() == 1 && lin->keys_[0] == X(2));
    REQUIRE(std::fabs(vgicp_factor->inlier_fraction() - (double)inl_ref / N) < 1e-12);  // the same correspondences, point for point
    double H[36], b[6], d[6];
    for (int r = 0; r < 6; r++) {
      b[r] = -lin->g2_(r);  // HessianFactor stores g = -b
      for (int c = 0; c < 6; c++) H[6 * r + c] = lin->G22_(r, c);
    }
    REQUIRE(orc_solve6(H, b, 0.0, d) == 0);
    REQUIRE(it < ref_iters);
    for (int k = 0; k < 6; k++) worst_step = std::fmax(worst_step, std::fabs(d[k] - ref_deltas[6 * it + k]));
    orc_se3_exp(d, E);
    orc_pose_compose(T, E, T);
    const double dt = std::sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]), dr = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (dt < 1e-3 && dr < 1e-3 * M_PI / 180.0) {  // termination_criteria (:121-137)
      it++;
      break;
    }
  }
  REQUIRE(it == ref_iters);
  REQUIRE(worst_step < 1e-4);  // per iteration, against the CPU path (north_star)
  double pose_err = 0.0;
  for (int k = 0; k < 12; k++) pose_err = std::fmax(pose_err, std::fabs(T[k] - Tref[k]));
  REQUIRE(pose_err < 1e-4);
  double truth12[12], off = 0.0;
  to12(truth, truth12);
  for (int k = 0; k < 12; k++) off = std::fmax(off, std::fabs(T[k] - truth12[k]));
  REQUIRE(off < 0.02);  // and the loop did converge to the frame's true pose (noise level)
  // --- the loop-closure validator (global_mapping_pose_graph.cpp:276-277, :406-417): a map built by ONE insert, a fixed-target factor, error + inlier_fraction ---
  auto voxels = std::make_shared<gtsam_points::GaussianVoxelMapCPU>(2.0);
  voxels->insert(*frames[0]);
  auto factor = gtsam::make_shared<gtsam_points::IntegratedVGICPFactor>(gtsam::Pose3(), 0, voxels, frames[1]);
  gtsam::Values v0;
  v0.insert(0, gtsam::Pose3(rel(pose2d(px[0], py[0], pw[0]), pose2d(px[1], py[1], pw[1]))));
  REQUIRE(factor->linearize(v0) != nullptr && factor->error(v0) > 0.0 && factor->inlier_fraction() > 0.5 && factor->inlier_fraction() <= 1.0);
  // --- the mapping modules' enable_gpu = false branch: overlap_auto on a CPU-named map takes the device path (sub_mapping.cpp:253) ---
  const double ov = gtsam_points::overlap_auto(voxels, frames[1], Eigen::Isometry3d(rel(pose2d(px[0], py[0], pw[0]), pose2d(px[1], py[1], pw[1]))));
  REQUIRE(ov > 0.5 && ov <= 1.0);
  orc_voxelmap_destroy(ref_map);
  std::printf("test_shim_cpu_names OK (%d iterations, worst step difference %.2e, final pose difference %.2e, overlap %.3f)\n", it, worst_step, pose_err, ov);
  return 0;
}
