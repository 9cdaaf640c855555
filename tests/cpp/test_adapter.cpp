// test_adapter.cpp -- compiles adapters/gtsam/glim_amd_gtsam.hpp against the stand-in GTSAM / Eigen / gtsam_points headers of
// tests/cpp/mock/ and drives it the way GLIM does (odometry_estimation_gpu.cpp:128-206 create_factors, :383-386 factor set +
// graph.linearize, offline_viewer.cpp:29 linearisation hook); every Hessian block is checked against the CPU oracle (test-only).
// Built by tests/test_adapter.py:  g++ -std=c++17 -Itests/cpp/mock -Iinclude -Iadapters/gtsam test_adapter.cpp -lglim_amd -lvgicp_oracle
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include <glim_amd_gtsam.hpp>

#include "../../oracle/vgicp_oracle.h"

using namespace glim_amd;

#define REQUIRE(cond)                                                        \
  do {                                                                       \
    if (!(cond)) {                                                           \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      return 1;                                                              \
    }                                                                        \
  } while (0)

// three walls of a room corner seen from a sensor at (ox, oy, yaw); FP32-representable coordinates
static std::vector<Eigen::Vector4d> make_scan(int n, double ox, double oy, double yaw, unsigned seed) {
  std::mt19937_64 rng(seed);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  std::normal_distribution<double> G(0.0, 0.005);
  std::vector<Eigen::Vector4d> pts((size_t)n);
  const double c = std::cos(yaw), s = std::sin(yaw);
  for (int i = 0; i < n; i++) {
    double x, y, z;
    if (i % 3 == 0) { x = 8.0 * U(rng); y = 6.0 * U(rng); z = -1.5 + G(rng); }
    else if (i % 3 == 1) { x = 8.0 + G(rng); y = 6.0 * U(rng); z = -1.5 + 3.0 * U(rng); }
    else { x = 8.0 * U(rng); y = 6.0 + G(rng); z = -1.5 + 3.0 * U(rng); }
    const double wx = x - ox, wy = y - oy;
    pts[(size_t)i][0] = (double)(float)(c * wx + s * wy);
    pts[(size_t)i][1] = (double)(float)(-s * wx + c * wy);
    pts[(size_t)i][2] = (double)(float)z;
    pts[(size_t)i][3] = 1.0;
  }
  return pts;
}

static gtsam::Pose3 pose2d(double x, double y, double yaw) {
  Eigen::Matrix4d T = Eigen::Matrix4d::Identity();
  T(0, 0) = std::cos(yaw); T(0, 1) = -std::sin(yaw); T(1, 0) = std::sin(yaw); T(1, 1) = std::cos(yaw);
  T(0, 3) = x; T(1, 3) = y;
  return gtsam::Pose3(T);
}

static double rel_diff(const gtsam::Matrix& G, const double* ref /* row-major 6x6 */) {
  double num = 0, den = 1e-300;
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < 6; c++) {
      num = std::fmax(num, std::fabs(G(r, c) - ref[6 * r + c]));
      den = std::fmax(den, std::fabs(ref[6 * r + c]));
    }
  return num / den;
}
static double rel_diff_neg(const gtsam::Vector& g, const double* ref_b) {  // g must be -b
  double num = 0, den = 1e-300;
  for (int r = 0; r < 6; r++) {
    num = std::fmax(num, std::fabs(g(r) + ref_b[r]));
    den = std::fmax(den, std::fabs(ref_b[r]));
  }
  return num / den;
}

// a non-HIP factor in the same graph: the set must leave it alone
struct DummyFactor : gtsam::NonlinearFactor {
  DummyFactor() : gtsam::NonlinearFactor(gtsam::KeyVector{7}) {}
  double error(const gtsam::Values&) const override { return 0.0; }
  size_t dim() const override { return 6; }
  std::shared_ptr<gtsam::GaussianFactor> linearize(const gtsam::Values&) const override { return nullptr; }
  gtsam::NonlinearFactor::shared_ptr clone() const override { return std::make_shared<DummyFactor>(*this); }
};

int main() {
  if (glim_amd_device_count() < 1) {
    std::fprintf(stderr, "no HIP device: this test must run on the GPU box\n");
    return 2;
  }
  const int n = 20000, k = 10;
  std::vector<Eigen::Vector4d> pa = make_scan(n, 1.0, 1.0, 0.05, 1), pb = make_scan(n, 1.3, 1.1, 0.08, 2);
  const gtsam::Pose3 T_world_a = pose2d(1.0, 1.0, 0.05), T_world_b = pose2d(1.3, 1.1, 0.08);

  // frames as gtsam_points::PointCloud views (odometry_estimation_gpu.cpp:96: PointCloudGPU::clone(*frame))
  gtsam_points::PointCloud view_a, view_b;
  view_a.num_points = view_b.num_points = (size_t)n;
  view_a.points = pa.data();
  view_b.points = pb.data();
  auto fa = glim_amd::clone(view_a), fb = glim_amd::clone(view_b);
  REQUIRE(fa->size() == (size_t)n && fb->size() == (size_t)n);
  fa->find_neighbors(k);
  fb->find_neighbors(k);
  fa->estimate_covariances(k);
  fb->estimate_covariances(k);
  auto vm = std::make_shared<GaussianVoxelMapGPU>(0.5f);
  vm->insert(*fa);

  // oracle on the same inputs
  std::vector<int32_t> nba((size_t)n * k), nbb((size_t)n * k);
  std::vector<double> na(4 * (size_t)n), ca(16 * (size_t)n), nb(4 * (size_t)n), cb(16 * (size_t)n);
  orc_knn_grid(pa[0].data(), n, k, 0.0, nba.data(), 0);
  orc_knn_grid(pb[0].data(), n, k, 0.0, nbb.data(), 0);
  orc_covariance_estimate(pa[0].data(), n, nba.data(), k, k, na.data(), ca.data(), 0);
  orc_covariance_estimate(pb[0].data(), n, nbb.data(), k, k, nb.data(), cb.data(), 0);
  for (auto& v : ca) v = (double)(float)v;
  for (auto& v : cb) v = (double)(float)v;
  orc_voxelmap* om = orc_voxelmap_create(0.5);
  orc_voxelmap_insert(om, pa[0].data(), ca.data(), n);

  // -- create_factors: a binary and a unary factor, surface validation switched like GLIM does
  const gtsam::Key X0 = 0, X1 = 1;
  gtsam::Values values;
  values.insert(X0, T_world_a);
  values.insert(X1, T_world_b);
  auto binary = std::make_shared<IntegratedVGICPFactorHIP>(X0, X1, vm, fb);
  auto unary = std::make_shared<IntegratedVGICPFactorHIP>(T_world_a, X1, vm, fb);
  binary->set_enable_surface_validation(false);
  REQUIRE(binary->dim() == 6 && binary->keys().size() == 2 && unary->keys().size() == 1 && unary->keys()[0] == X1);
  REQUIRE(std::fabs(unary->get_fixed_target_pose().matrix()(0, 3) - 1.0) < 1e-15);
  REQUIRE(binary->memory_usage_gpu() > (size_t)n * 16);

  const Isometry3d delta = to_iso(T_world_a).inverse() * to_iso(T_world_b);
  orc_linearized6 ref;
  orc_vgicp_linearize(om, pb[0].data(), cb.data(), n, delta.m.data(), 0, &ref, nullptr);

  // -- slow path: factor->linearize(values) on its own
  {
    auto hf = std::dynamic_pointer_cast<gtsam::HessianFactor>(binary->linearize(values));
    REQUIRE(hf && hf->keys.size() == 2 && hf->keys[0] == X0 && hf->keys[1] == X1);
    REQUIRE(rel_diff(hf->G11, ref.H_tt) < 2e-4 && rel_diff(hf->G12, ref.H_ts) < 2e-4 && rel_diff(hf->G22, ref.H_ss) < 2e-4);
    REQUIRE(rel_diff_neg(hf->g1, ref.b_t) < 1e-3 && rel_diff_neg(hf->g2, ref.b_s) < 1e-3);
    REQUIRE(std::fabs(hf->f - ref.error) < 2e-4 * ref.error);
    REQUIRE(std::fabs(binary->inlier_fraction() - (double)ref.num_inliers / n) < 1e-12);
    // the Gauss-Newton step GTSAM would take on the source pose: solve G22 x = g2  (x = -H^-1 b)
    double H[36], b[6], dg[6], dr[6];
    for (int r = 0; r < 6; r++) {
      b[r] = -hf->g2(r);
      for (int c = 0; c < 6; c++) H[6 * r + c] = hf->G22(r, c);
    }
    REQUIRE(orc_solve6(H, b, 0.0, dg) == 0 && orc_solve6(ref.H_ss, ref.b_s, 0.0, dr) == 0);
    for (int i = 0; i < 6; i++) REQUIRE(std::fabs(dg[i] - dr[i]) < 1e-4);
    auto hu = std::dynamic_pointer_cast<gtsam::HessianFactor>(unary->linearize(values));
    REQUIRE(hu && hu->keys.size() == 1 && hu->keys[0] == X1 && rel_diff(hu->G11, ref.H_ss) < 2e-4 && rel_diff_neg(hu->g1, ref.b_s) < 1e-3);
    // error at the linearisation point == the linearised error; clone() is independent and agrees
    REQUIRE(std::fabs(binary->error(values) - hf->f) < 1e-5 * hf->f);
    auto cl = std::dynamic_pointer_cast<IntegratedVGICPFactorHIP>(binary->clone());
    REQUIRE(cl && cl.get() != binary.get() && cl->impl().get() != binary->impl().get());
    auto hc = std::dynamic_pointer_cast<gtsam::HessianFactor>(cl->linearize(values));
    REQUIRE(rel_diff(hc->G22, H) < 1e-6);
  }

  // -- batch path: the set linearises every HIP factor of a graph in one launch; graph.linearize then only wraps the results
  {
    register_linearization_hook();
    REQUIRE(gtsam_points::LinearizationHook::hooks().size() == 1);
    auto set = gtsam_points::LinearizationHook::hooks()[0]();
    gtsam::NonlinearFactorGraph graph;
    auto b2 = std::make_shared<IntegratedVGICPFactorHIP>(X0, X1, vm, fb);
    auto u2 = std::make_shared<IntegratedVGICPFactorHIP>(T_world_a, X1, vm, fb);
    graph.push_back(b2);
    graph.push_back(std::make_shared<DummyFactor>());
    graph.push_back(u2);
    set->add(graph);
    REQUIRE(set->size() == 2);  // the dummy is not ours
    set->linearize(values);
    auto hb = std::dynamic_pointer_cast<gtsam::HessianFactor>(graph[0]->linearize(values));
    auto hu = std::dynamic_pointer_cast<gtsam::HessianFactor>(graph[2]->linearize(values));
    REQUIRE(hb && hu);
    REQUIRE(rel_diff(hb->G11, ref.H_tt) < 2e-4 && rel_diff(hb->G12, ref.H_ts) < 2e-4 && rel_diff(hb->G22, ref.H_ss) < 2e-4);
    REQUIRE(rel_diff(hu->G11, ref.H_ss) < 2e-4 && std::fabs(hu->f - ref.error) < 2e-4 * ref.error);
    // a different evaluation point: batch error (frozen correspondences, GPU-factor semantics) == the factor's own error there
    gtsam::Values moved;
    moved.insert(X0, T_world_a);
    moved.insert(X1, pose2d(1.31, 1.09, 0.081));
    set->error(moved);
    const double eb = b2->error(moved);  // cached by the set
    auto solo = std::make_shared<IntegratedVGICPFactorHIP>(X0, X1, vm, fb);
    solo->linearize(values);
    REQUIRE(std::fabs(solo->error(moved) - eb) < 1e-5 * eb);  // same kernel, different block partition
    // ... and differs from re-matching at the new pose only slightly (same surfaces), but is not the linearisation-point error
    REQUIRE(std::fabs(eb - hb->f) > 1e-6 * hb->f);
    const Isometry3d dm = to_iso(T_world_a).inverse() * to_iso(moved.at<gtsam::Pose3>(X1));
    const double e_frozen = orc_vgicp_error_frozen(om, pb[0].data(), cb.data(), n, delta.m.data(), dm.m.data(), 0, nullptr);
    REQUIRE(std::fabs(eb - e_frozen) < 5e-4 * e_frozen);
    // relinearising at the moved point refreshes the cache
    set->linearize(moved);
    auto hm = std::dynamic_pointer_cast<gtsam::HessianFactor>(graph[0]->linearize(moved));
    orc_linearized6 ref_m;
    orc_vgicp_linearize(om, pb[0].data(), cb.data(), n, dm.m.data(), 0, &ref_m, nullptr);
    REQUIRE(rel_diff(hm->G22, ref_m.H_ss) < 2e-4 && std::fabs(hm->f - ref_m.error) < 2e-4 * ref_m.error);
    auto lf = set->calc_linear_factors(values);
    REQUIRE(lf.size() == 2 && std::dynamic_pointer_cast<gtsam::HessianFactor>(lf[0]));
    set->clear();
    REQUIRE(set->size() == 0);
  }

  // -- GICP factor (global_mapping.cpp:400-402) through the adapter
  {
    auto tree = std::make_shared<NearestNeighborSearchGPU>(fa, 0.5);
    auto g = std::make_shared<IntegratedGICPFactorHIP>(X0, X1, fa, fb, tree);
    g->set_max_correspondence_distance(0.5);
    g->set_num_threads(2);
    auto hg = std::dynamic_pointer_cast<gtsam::HessianFactor>(g->linearize(values));
    orc_linearized6 rg;
    REQUIRE(orc_gicp_linearize(pa[0].data(), ca.data(), n, pb[0].data(), cb.data(), n, delta.m.data(), 0.5, 0, &rg, nullptr) == 0);
    REQUIRE(hg && rel_diff(hg->G22, rg.H_ss) < 2e-4 && rel_diff(hg->G11, rg.H_tt) < 2e-4 && rel_diff_neg(hg->g2, rg.b_s) < 1e-3);
    REQUIRE(std::fabs(g->error(values) - rg.error) < 2e-4 * rg.error && g->inlier_fraction() == (double)rg.num_inliers / n);
    REQUIRE(std::dynamic_pointer_cast<IntegratedGICPFactorHIP>(g->clone()) != nullptr);
  }

  // -- overlap with Eigen poses (odometry_estimation_gpu.cpp:248)
  {
    Eigen::Isometry3d d = Eigen::Isometry3d::Identity();
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 4; c++) d.matrix()(r, c) = delta.m[(size_t)(4 * r + c)];
    const double ov = glim_amd::overlap_gpu(vm, fb, d);
    REQUIRE(std::fabs(ov - (double)ref.num_inliers / n) < 1e-12);
    REQUIRE(std::fabs(glim_amd::overlap_auto(vm, fb, d) - ov) < 1e-15);
    REQUIRE(std::fabs(glim_amd::overlap_gpu(std::vector<GaussianVoxelMapGPU::ConstPtr>{vm, vm}, fb, std::vector<Eigen::Isometry3d>{d, d}) - ov) < 1e-15);
  }
  orc_voxelmap_destroy(om);
  std::printf("test_adapter OK\n");
  return 0;
}
