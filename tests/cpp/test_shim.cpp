// test_shim.cpp -- drives the gtsam_points GPU names of adapters/gtsam_points_hip the way GLIM's sources do (odometry_estimation_gpu.cpp:96-104,
// :137-163, :231-248, :383-386): PointCloudGPU::clone(frame), GaussianVoxelMapGPU(res, 8192 * 2, 10, 1e-3, *stream), the SIX-argument factor
// constructors with a (stream, buffer) pair from StreamTempBufferRoundRobin, NonlinearFactorSetGPU, overlap_gpu with Eigen::Isometry3d -- and checks the
// numbers against direct C-ABI calls on the same data.  Stand-in third-party headers: tests/cpp/glim_standin (test infrastructure).
// Built and run by tests/test_glim_module.py.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include <gtsam/inference/Symbol.h>
#include <gtsam/nonlinear/NonlinearFactorGraph.h>
#include <gtsam_points/cuda/cuda_device_prop.hpp>
#include <gtsam_points/cuda/cuda_memory.hpp>
#include <gtsam_points/cuda/cuda_stream.hpp>
#include <gtsam_points/cuda/nonlinear_factor_set_gpu.hpp>
#include <gtsam_points/cuda/nonlinear_factor_set_gpu_create.hpp>
#include <gtsam_points/cuda/stream_temp_buffer_roundrobin.hpp>
#include <gtsam_points/factors/integrated_vgicp_factor_gpu.hpp>
#include <gtsam_points/types/gaussian_voxelmap_gpu.hpp>
#include <gtsam_points/types/point_cloud_cpu.hpp>
#include <gtsam_points/types/point_cloud_gpu.hpp>
#include <gtsam_points/util/gtsam_migration.hpp>

using gtsam::symbol_shorthand::X;

#define REQUIRE(cond)                                                        \
  do {                                                                       \
    if (!(cond)) {                                                           \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      return 1;                                                              \
    }                                                                        \
  } while (0)

// the pieces of the third-party libraries this test needs at run time (declared by the stand-in headers)
namespace gtsam_points {
PointCloudCPU::Ptr PointCloudCPU::clone(const PointCloud& frame) {
  auto out = std::make_shared<PointCloudCPU>();
  out->num_points = frame.num_points;
  if (frame.points) out->points_storage.assign(frame.points, frame.points + frame.num_points);
  if (frame.normals) out->normals_storage.assign(frame.normals, frame.normals + frame.num_points);
  if (frame.covs) out->covs_storage.assign(frame.covs, frame.covs + frame.num_points);
  out->points = out->points_storage.empty() ? nullptr : out->points_storage.data();
  out->normals = out->normals_storage.empty() ? nullptr : out->normals_storage.data();
  out->covs = out->covs_storage.empty() ? nullptr : out->covs_storage.data();
  return out;
}
// libgtsam_points' CPU overlap (declared by gaussian_voxelmap_gpu.hpp for overlap_auto's fallback): a recording stand-in
static int g_cpu_overlap_calls = 0;
double overlap(const GaussianVoxelMap::ConstPtr&, const PointCloud::ConstPtr&, const Eigen::Isometry3d&) {
  g_cpu_overlap_calls++;
  return 0.25;
}
// a voxel map that does not live on the device (GaussianVoxelMapCPU in a real build)
struct HostOnlyMap : public GaussianVoxelMap {
  double voxel_resolution() const override { return 1.0; }
  void insert(const PointCloud&) override {}
  void save_compact(const std::string&) const override {}
};
}  // namespace gtsam_points

static gtsam_points::PointCloudCPU::Ptr make_frame(int n, double ox, double oy, double yaw, unsigned seed) {
  std::mt19937_64 rng(seed);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  std::normal_distribution<double> G(0.0, 0.004);
  auto f = std::make_shared<gtsam_points::PointCloudCPU>();
  f->points_storage.resize((size_t)n);
  f->covs_storage.resize((size_t)n);
  f->normals_storage.resize((size_t)n);
  const double c = std::cos(yaw), s = std::sin(yaw);
  for (int i = 0; i < n; i++) {
    double x, y, z, nx = 0, ny = 0, nz = 0;
    if (i % 3 == 0) { x = 8.0 * U(rng); y = 6.0 * U(rng); z = -1.5 + G(rng); nz = 1; }
    else if (i % 3 == 1) { x = 8.0 + G(rng); y = 6.0 * U(rng); z = -1.5 + 3.0 * U(rng); nx = -1; }
    else { x = 8.0 * U(rng); y = 6.0 + G(rng); z = -1.5 + 3.0 * U(rng); ny = -1; }
    const double wx = x - ox, wy = y - oy;
    auto& p = f->points_storage[(size_t)i];
    p = Eigen::Vector4d((double)(float)(c * wx + s * wy), (double)(float)(-s * wx + c * wy), (double)(float)z, 1.0);
    const Eigen::Vector4d nn(c * nx + s * ny, -s * nx + c * ny, nz, 0.0);  // world normal in the sensor frame
    f->normals_storage[(size_t)i] = nn;
    Eigen::Matrix4d C = Eigen::Matrix4d::Zero();
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) C(a, b) = (a == b ? 1.0 : 0.0) - (1.0 - 1e-3) * nn[a] * nn[b];  // the PLANE-regularised form GLIM produces
    f->covs_storage[(size_t)i] = C;
  }
  f->num_points = (size_t)n;
  f->points = f->points_storage.data();
  f->covs = f->covs_storage.data();
  f->normals = f->normals_storage.data();
  return f;
}

static Eigen::Isometry3d pose2d(double x, double y, double yaw) {
  Eigen::Isometry3d T = Eigen::Isometry3d::Identity();
  T.matrix()(0, 0) = std::cos(yaw); T.matrix()(0, 1) = -std::sin(yaw); T.matrix()(1, 0) = std::sin(yaw); T.matrix()(1, 1) = std::cos(yaw);
  T.matrix()(0, 3) = x; T.matrix()(1, 3) = y;
  return T;
}

int main() {
  if (glim_amd_device_count() < 1) {
    std::fprintf(stderr, "no HIP device\n");
    return 2;
  }
  // --- what OdometryEstimationGPU's constructor and create_frame do (odometry_estimation_gpu.cpp:76-77, :96-104) ---
  std::unique_ptr<gtsam_points::CUDAStream> stream(new gtsam_points::CUDAStream());
  std::unique_ptr<gtsam_points::StreamTempBufferRoundRobin> stream_buffer_roundrobin(new gtsam_points::StreamTempBufferRoundRobin());
  const Eigen::Isometry3d T0 = pose2d(1.0, 1.0, 0.0), T1 = pose2d(1.5, 1.2, 0.05);
  gtsam_points::PointCloud::ConstPtr frame0 = gtsam_points::PointCloudGPU::clone(*make_frame(30000, 1.0, 1.0, 0.0, 1));
  gtsam_points::PointCloud::ConstPtr frame1 = gtsam_points::PointCloudGPU::clone(*make_frame(30000, 1.5, 1.2, 0.05, 2));
  REQUIRE(frame0->size() == 30000 && frame0->points_gpu != nullptr && frame0->points != nullptr);
  std::vector<gtsam_points::GaussianVoxelMap::Ptr> voxelmaps;
  for (int i = 0; i < 2; i++) {
    auto voxelmap = std::make_shared<gtsam_points::GaussianVoxelMapGPU>(0.5 * std::pow(2.0, i), 8192 * 2, 10, 1e-3, *stream);
    voxelmap->insert(*frame0);
    REQUIRE(voxelmap->voxelmap_info.num_voxels > 100 && voxelmap->voxelmap_info.num_buckets >= voxelmap->voxelmap_info.num_voxels);
    voxelmaps.push_back(voxelmap);
  }
  // --- create_factors (:128-206): binary and unary factors with the (stream, buffer) pair ---
  gtsam::NonlinearFactorGraph factors;
  auto stream_buffer = stream_buffer_roundrobin->get_stream_buffer();
  const auto& st = stream_buffer.first;
  const auto& buffer = stream_buffer.second;
  for (const auto& voxelmap : voxelmaps) {
    auto factor = gtsam::make_shared<gtsam_points::IntegratedVGICPFactorGPU>(X(0), X(1), voxelmap, frame1, st, buffer);
    factor->set_enable_surface_validation(true);
    factors.add(factor);
  }
  const gtsam::Pose3 fixed(T0.matrix());
  auto unary = gtsam::make_shared<gtsam_points::IntegratedVGICPFactorGPU>(fixed, X(1), voxelmaps[0], frame1, st, buffer);
  factors.add(unary);
  gtsam::Values values;
  values.insert(X(0), gtsam::Pose3(T0.matrix()));
  values.insert(X(1), gtsam::Pose3(T1.matrix()));
  // --- the entropy strategy's batch (:383-386) and the hook factory (offline_viewer.cpp:29) ---
  gtsam_points::NonlinearFactorSetGPU factor_set;
  factor_set.add(factors);
  REQUIRE(factor_set.size() == 3);
  factor_set.linearize(values);
  REQUIRE(gtsam_points::create_nonlinear_factor_set_gpu() != nullptr);
  // --- numbers: the same three factors through the plain C ABI ---
  const auto dev_map0 = gtsam_points::device_map(voxelmaps[0]), dev_map1 = gtsam_points::device_map(voxelmaps[1]);
  const auto dev1 = gtsam_points::device_cloud(frame1);
  glim_amd_factor_set* cset = nullptr;
  REQUIRE(glim_amd_factor_set_create(dev1->context()->context(), &cset) == GLIM_AMD_OK);
  const uint32_t fl = GLIM_AMD_FACTOR_BINARY | GLIM_AMD_FACTOR_SURFACE_VALIDATION;
  REQUIRE(glim_amd_factor_set_add(cset, dev_map0->handle(), dev1->handle(), fl, nullptr) == GLIM_AMD_OK);
  REQUIRE(glim_amd_factor_set_add(cset, dev_map1->handle(), dev1->handle(), fl, nullptr) == GLIM_AMD_OK);
  REQUIRE(glim_amd_factor_set_add(cset, dev_map0->handle(), dev1->handle(), 0u, nullptr) == GLIM_AMD_OK);
  const Eigen::Isometry3d delta = T0.inverse() * T1;
  double T[36];
  for (int f = 0; f < 3; f++)
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 4; c++) T[12 * f + 4 * r + c] = delta.matrix()(r, c);
  glim_amd_linearized6 want[3];
  REQUIRE(glim_amd_factor_set_linearize(cset, T, want) == GLIM_AMD_OK);
  int k = 0;
  for (const auto& f : factors) {
    auto hip = std::dynamic_pointer_cast<gtsam_points::IntegratedVGICPFactorGPU>(f);
    REQUIRE(hip != nullptr);
    REQUIRE(hip->linearize(values) != nullptr);  // served from the batch result
    const double frac = hip->inlier_fraction();
    REQUIRE(std::fabs(frac - (double)want[k].num_inliers / 30000.0) < 1e-12 && frac > 0.3);
    // frozen-correspondence error at the linearisation point == the error of the linearisation
    REQUIRE(std::fabs(hip->error(values) - want[k].error) <= 1e-5 * std::fabs(want[k].error));  // (a different kernel variant: FP32 summation-order level)
    k++;
  }
  REQUIRE(unary->get_fixed_target_pose().matrix()(0, 3) == T0.matrix()(0, 3));
  REQUIRE(unary->memory_usage_gpu() > 0 && unary->clone() != nullptr);
  glim_amd_factor_set_destroy(cset);
  // --- keyframe management (:231, :248): overlap_gpu with Eigen poses, single and multi target ---
  std::vector<gtsam_points::GaussianVoxelMap::ConstPtr> keyframes_ = {voxelmaps[1], voxelmaps[0]};
  std::vector<Eigen::Isometry3d> deltas = {delta, delta};
  const double ov_multi = gtsam_points::overlap_gpu(keyframes_, frame1, deltas, *stream);
  const double ov_single = gtsam_points::overlap_gpu(voxelmaps[1], frame1, delta, *stream);
  REQUIRE(ov_multi >= ov_single && ov_single > 0.5 && ov_multi <= 1.0);
  REQUIRE(gtsam_points::overlap_auto(voxelmaps[1], frame1, delta) == ov_single && gtsam_points::g_cpu_overlap_calls == 0);
  // a map that is NOT on the device (enable_gpu = false: global_mapping.cpp:275 builds GaussianVoxelMapCPU): overlap_auto falls back to the
  // CPU overlap of libgtsam_points, as upstream's does (sub_mapping.cpp:253, global_mapping.cpp:322,448), instead of throwing
  const gtsam_points::GaussianVoxelMap::ConstPtr host_map = std::make_shared<gtsam_points::HostOnlyMap>();
  REQUIRE(gtsam_points::overlap_auto(host_map, frame1, delta) == 0.25 && gtsam_points::g_cpu_overlap_calls == 1);
  // src/glim/util/debug.cpp:84 and src/glim/viewer/memory_monitor.cpp:39: device names and memory figures through the same headers
  const std::vector<std::string> devices = gtsam_points::cuda_device_names();
  REQUIRE(!devices.empty() && !devices[0].empty());
  size_t gpu_free = 0, gpu_total = 0;
  gtsam_points::cuda_mem_get_info(&gpu_free, &gpu_total);
  REQUIRE(gpu_total > (size_t)64 << 30 && gpu_free > 0 && gpu_free <= gpu_total);  // an MI355X carries 288 GB
  std::printf("test_shim OK (inlier fractions %.3f %.3f %.3f, overlap %.3f)\n", (double)want[0].num_inliers / 30000.0, (double)want[1].num_inliers / 30000.0,
              (double)want[2].num_inliers / 30000.0, ov_single);
  return 0;
}
