// test_compat.cpp -- exercises the C++ mirror of the gtsam_points GPU interface (include/glim_amd/gtsam_points_compat.hpp) the way
// GLIM's odometry does (odometry_estimation_gpu.cpp:86-107,128-206), and checks the results against the CPU oracle (test-only).
// Built and run by tests/test_gpu_cpp.py on the GPU box:  g++ -std=c++17 test_compat.cpp -lglim_amd -lvgicp_oracle
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../include/glim_amd/gtsam_points_compat.hpp"
#include "../../include/glim_amd/glim_preprocess_compat.hpp"
#include "../../oracle/vgicp_oracle.h"

using namespace glim_amd;

#define REQUIRE(cond)                                                   \
  do {                                                                  \
    if (!(cond)) {                                                      \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      return 1;                                                         \
    }                                                                   \
  } while (0)

// a corner made of three walls, seen from a sensor pose: n points, f32-representable coordinates (Vector4d layout)
static std::vector<double> make_scan(int n, double ox, double oy, double yaw, unsigned seed) {
  std::mt19937_64 rng(seed);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  std::normal_distribution<double> G(0.0, 0.005);
  std::vector<double> pts(4 * (size_t)n);
  const double c = std::cos(yaw), s = std::sin(yaw);
  for (int i = 0; i < n; i++) {
    double x, y, z;
    const int wall = i % 3;
    if (wall == 0) { x = 8.0 * U(rng); y = 6.0 * U(rng); z = -1.5 + G(rng); }        // floor
    else if (wall == 1) { x = 8.0 + G(rng); y = 6.0 * U(rng); z = -1.5 + 3.0 * U(rng); }  // front wall
    else { x = 8.0 * U(rng); y = 6.0 + G(rng); z = -1.5 + 3.0 * U(rng); }               // side wall
    const double wx = x - ox, wy = y - oy;  // world -> sensor frame (T_world_sensor^-1)
    pts[4 * i + 0] = (double)(float)(c * wx + s * wy);
    pts[4 * i + 1] = (double)(float)(-s * wx + c * wy);
    pts[4 * i + 2] = (double)(float)z;
    pts[4 * i + 3] = 1.0;
  }
  return pts;
}

static Isometry3d pose2d(double x, double y, double yaw) {
  Isometry3d T;
  T.m = {std::cos(yaw), -std::sin(yaw), 0, x, std::sin(yaw), std::cos(yaw), 0, y, 0, 0, 1, 0};
  return T;
}

static double max_rel_diff(const double* a, const double* b, int n) {
  double num = 0, den = 1e-300;
  for (int i = 0; i < n; i++) {
    num = std::fmax(num, std::fabs(a[i] - b[i]));
    den = std::fmax(den, std::fabs(b[i]));
  }
  return num / den;
}

int main() {
  if (glim_amd_device_count() < 1) {
    std::fprintf(stderr, "no HIP device: this test must run on the GPU box\n");
    return 2;
  }
  const int n = 20000, k = 10;
  const Isometry3d T_world_a = pose2d(1.0, 1.0, 0.05), T_world_b = pose2d(1.3, 1.1, 0.08);
  const std::vector<double> pa = make_scan(n, 1.0, 1.0, 0.05, 1), pb = make_scan(n, 1.3, 1.1, 0.08, 2);

  // -- OdometryEstimationGPU::create_frame: clone, (kNN + covariance on the device), two voxel-map levels
  auto fa = PointCloudGPU::clone(pa.data(), nullptr, nullptr, n);
  auto fb = PointCloudGPU::clone(pb.data(), nullptr, nullptr, n);
  const std::vector<int> nb_b = fb->find_neighbors(k);
  fa->find_neighbors(k);
  fa->estimate_covariances(k);
  fb->estimate_covariances(k);
  std::vector<GaussianVoxelMapGPU::Ptr> maps;
  for (int level = 0; level < 2; level++) {
    auto vm = std::make_shared<GaussianVoxelMapGPU>(0.5f * std::pow(2.0f, (float)level), 8192 * 2, 10, 1e-3);
    vm->insert(*fa);
    REQUIRE(vm->voxelmap_info().num_voxels > 100);
    maps.push_back(vm);
  }

  // -- create_factors: binary factor per level (+ one unary), batched through NonlinearFactorSetGPU
  const Key X0 = 0, X1 = 1;
  Values values;
  values[X0] = T_world_a;
  values[X1] = T_world_b;
  NonlinearFactorSetGPU set;
  std::vector<IntegratedVGICPFactorGPU::shared_ptr> factors;
  for (auto& vm : maps) {
    auto f = std::make_shared<IntegratedVGICPFactorGPU>(X0, X1, vm, fb);
    factors.push_back(f);
    set.add(f);
  }
  auto unary = std::make_shared<IntegratedVGICPFactorGPU>(T_world_a, X1, maps[0], fb);
  factors.push_back(unary);
  set.add(unary);
  REQUIRE(set.size() == 3);
  set.linearize(values);

  // -- oracle on the same inputs: kNN sets, covariances, voxel map, factor
  std::vector<int32_t> onb((size_t)n * k);
  orc_knn_grid(pb.data(), n, k, 0.0, onb.data(), 0);
  for (size_t i = 0; i < onb.size(); i++) REQUIRE(onb[i] == nb_b[i]);
  std::vector<double> na(4 * (size_t)n), ca(16 * (size_t)n), nbv(4 * (size_t)n), cb(16 * (size_t)n);
  std::vector<int32_t> onb_a((size_t)n * k);
  orc_knn_grid(pa.data(), n, k, 0.0, onb_a.data(), 0);
  orc_covariance_estimate(pa.data(), n, onb_a.data(), k, k, na.data(), ca.data(), 0);
  orc_covariance_estimate(pb.data(), n, onb.data(), k, k, nbv.data(), cb.data(), 0);
  for (auto& v : ca) v = (double)(float)v;  // the device stores FP32 covariances
  for (auto& v : cb) v = (double)(float)v;
  const Isometry3d delta = factors[0]->calc_delta(values);
  for (int level = 0; level < 2; level++) {
    orc_voxelmap* om = orc_voxelmap_create(0.5 * std::pow(2.0, level));
    orc_voxelmap_insert(om, pa.data(), ca.data(), n);
    REQUIRE(orc_voxelmap_num_voxels(om) == maps[level]->voxelmap_info().num_voxels);
    orc_linearized6 ref;
    orc_vgicp_linearize(om, pb.data(), cb.data(), n, delta.m.data(), 0, &ref, nullptr);
    const LinearizedSystem6& got = factors[level]->linearized();
    REQUIRE(got.num_inliers == ref.num_inliers);
    REQUIRE(got.num_inliers > n / 2);
    REQUIRE(max_rel_diff(got.H_ss, ref.H_ss, 36) < 2e-4);
    REQUIRE(max_rel_diff(got.H_tt, ref.H_tt, 36) < 2e-4);
    REQUIRE(max_rel_diff(got.H_ts, ref.H_ts, 36) < 2e-4);
    REQUIRE(max_rel_diff(got.b_s, ref.b_s, 6) < 1e-3);
    REQUIRE(std::fabs(got.error - ref.error) < 2e-4 * ref.error);
    double dg[6], dr[6];
    REQUIRE(orc_solve6(got.H_ss, got.b_s, 0.0, dg) == 0);
    REQUIRE(orc_solve6(ref.H_ss, ref.b_s, 0.0, dr) == 0);
    for (int i = 0; i < 6; i++) REQUIRE(std::fabs(dg[i] - dr[i]) < 1e-4);
    if (level == 0) {
      // unary factor with the target pose fixed == source block of the binary one, no target blocks
      const LinearizedSystem6& u = unary->linearized();
      REQUIRE(u.num_inliers == got.num_inliers);
      REQUIRE(max_rel_diff(u.H_ss, got.H_ss, 36) < 1e-12);
      for (int i = 0; i < 36; i++) REQUIRE(u.H_tt[i] == 0.0 && u.H_ts[i] == 0.0);
      // slow path (factor linearised on its own) == batch result; error() == linearised error at the same point
      auto solo = factors[0]->clone();
      const LinearizedSystem6 s = solo->linearize(values);
      REQUIRE(s.num_inliers == got.num_inliers);
      REQUIRE(max_rel_diff(s.H_ss, got.H_ss, 36) < 1e-6);
      REQUIRE(std::fabs(solo->error(values) - got.error) < 1e-6 * got.error);
      // overlap_gpu == inlier fraction (odometry_estimation_gpu.cpp:248); multi-target form counts any hit
      const double ov = overlap_gpu(maps[0], fb, delta);
      REQUIRE(std::fabs(ov - (double)got.num_inliers / n) < 1e-12);
      Isometry3d far = delta;
      far.m[3] += 1e4;
      REQUIRE(overlap_gpu(maps[0], fb, far) == 0.0);
      REQUIRE(std::fabs(overlap_gpu({maps[0], maps[1]}, fb, {far, delta}) - overlap_gpu(maps[1], fb, delta)) < 1e-12);
      REQUIRE(std::fabs(factors[0]->inlier_fraction() - ov) < 1e-12);
      // the batched form answers the same queries in one launch
      const std::vector<double> batch = overlap_gpu_batch({{{maps[0]}, fb, {delta}}, {{maps[0]}, fb, {far}}, {{maps[0], maps[1]}, fb, {far, delta}}});
      REQUIRE(batch.size() == 3 && batch[0] == ov && batch[1] == 0.0 && batch[2] == overlap_gpu(maps[1], fb, delta));
    }
    orc_voxelmap_destroy(om);
  }
  // ---- GICP factor (global_mapping.cpp:400-402): same frames, nearest-neighbour correspondences; step == oracle within 1e-4 ----
  {
    const Isometry3d gdelta = T_world_a.inverse() * T_world_b;
    Values gvalues;
    gvalues[0] = Isometry3d::Identity();
    gvalues[1] = gdelta;
    auto tree = std::make_shared<NearestNeighborSearchGPU>(fa, 0.5);
    IntegratedGICPFactor gicp(0, 1, fa, fb, tree);
    gicp.set_max_correspondence_distance(0.5);
    gicp.set_num_threads(2);
    const LinearizedSystem6& got = gicp.linearize(gvalues);
    std::vector<float> c32a((size_t)n * 9), c32b((size_t)n * 9);
    REQUIRE(glim_amd_cloud_download(fa->handle(), nullptr, c32a.data(), nullptr, nullptr) == GLIM_AMD_OK);
    REQUIRE(glim_amd_cloud_download(fb->handle(), nullptr, c32b.data(), nullptr, nullptr) == GLIM_AMD_OK);
    std::vector<double> ca16((size_t)n * 16, 0.0), cb16((size_t)n * 16, 0.0);
    for (int i = 0; i < n; i++)
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
          ca16[16 * (size_t)i + 4 * c + r] = c32a[9 * (size_t)i + 3 * r + c];
          cb16[16 * (size_t)i + 4 * c + r] = c32b[9 * (size_t)i + 3 * r + c];
        }
    orc_linearized6 ref;
    REQUIRE(orc_gicp_linearize(pa.data(), ca16.data(), n, pb.data(), cb16.data(), n, gdelta.m.data(), 0.5, 0, &ref, nullptr) == 0);
    REQUIRE(got.num_inliers == ref.num_inliers && ref.num_inliers > n / 2);
    REQUIRE(std::fabs(got.error - ref.error) <= 2e-4 * ref.error);
    REQUIRE(max_rel_diff(got.H_ss, ref.H_ss, 36) < 2e-4 && max_rel_diff(got.H_tt, ref.H_tt, 36) < 2e-4 && max_rel_diff(got.H_ts, ref.H_ts, 36) < 2e-4);
    double dg[6], dr[6];
    REQUIRE(orc_solve6(got.H_ss, got.b_s, 0.0, dg) == 0 && orc_solve6(ref.H_ss, ref.b_s, 0.0, dr) == 0);
    for (int i = 0; i < 6; i++) REQUIRE(std::fabs(dg[i] - dr[i]) < 1e-4);
    REQUIRE(std::fabs(gicp.error(gvalues) - got.error) <= 1e-9 * got.error);
    REQUIRE(gicp.inlier_fraction() == (double)ref.num_inliers / n);
    std::printf("gicp OK: %lld inliers\n", (long long)ref.num_inliers);
  }
  // ---- the per-scan front end: CloudPreprocessor::preprocess -> CloudDeskewing::deskew -> covariances (odometry_estimation_imu.cpp:300-325) ----
  {
    auto raw = std::make_shared<RawPoints>();
    const int nr = 60000;
    const std::vector<double> rp = make_scan(nr, 3.0, 2.0, 0.3, 99);
    raw->stamp = 1234.5;
    raw->points.resize(nr);
    raw->times.resize(nr);
    raw->intensities.resize(nr);
    std::mt19937_64 rng(5);
    std::uniform_real_distribution<double> U(0.0, 0.1);
    for (int i = 0; i < nr; i++) {
      raw->points[i] = {rp[4 * i], rp[4 * i + 1], rp[4 * i + 2], 1.0};
      raw->times[i] = U(rng);
      raw->intensities[i] = (double)(i % 255);
    }
    CloudPreprocessorParams params;
    params.downsample_resolution = 0.25;
    params.downsample_target = 8000;
    params.distance_near_thresh = 1.0;
    params.seed = 42;
    CloudPreprocessor preprocessor(params);
    auto frame = preprocessor.preprocess(raw);
    // the oracle on the same scan with the same parameters
    orc_preprocess_params op;
    std::memset(&op, 0, sizeof(op));
    const glim_amd_preprocess_params cp = params.c_params();
    static_assert(sizeof(orc_preprocess_params) == sizeof(glim_amd_preprocess_params), "parameter blocks mirror each other");
    std::memcpy(&op, &cp, sizeof(op));
    op.seed = 42;
    std::vector<double> ref_p(4 * (size_t)nr), ref_t(nr), ref_i(nr);
    std::vector<int32_t> ref_nb((size_t)nr * 10);
    const int m = orc_preprocess(rp.data(), raw->times.data(), raw->intensities.data(), nr, &op, ref_p.data(), ref_t.data(), ref_i.data(), ref_nb.data(), 0);
    REQUIRE(frame->size() == m && m > 5000 && m <= 9600);
    REQUIRE(frame->k_neighbors == 10 && (int)frame->neighbors.size() == 10 * m);
    for (int i = 0; i < m; i++) {
      for (int a = 0; a < 4; a++) REQUIRE(frame->points[i][a] == ref_p[4 * (size_t)i + a]);
      REQUIRE(frame->times[i] == ref_t[i] && frame->intensities[i] == ref_i[i]);
      if (i) REQUIRE(frame->times[i] >= frame->times[i - 1]);
    }
    for (size_t i = 0; i < frame->neighbors.size(); i++) REQUIRE(frame->neighbors[i] == ref_nb[i]);
    REQUIRE(frame->scan_end_time == raw->stamp + frame->times[m - 1]);
    // a second frame draws a different sample (the reference's mt19937 advances too)
    auto frame2 = preprocessor.preprocess(raw);
    REQUIRE(frame2->size() == m);
    bool differs = false;
    for (int i = 0; i < m && !differs; i++) differs = frame2->points[i] != frame->points[i];
    REQUIRE(differs);
    // deskew on the device, constant-velocity form, then covariances with the raw scan's neighbours
    const Isometry3d T_imu_lidar = pose2d(0.1, -0.05, 0.02);
    const Vector3d lv{{3.0, -1.0, 0.2}}, av{{0.05, -0.1, 0.8}};
    CloudDeskewing deskewing;
    // default: deskew() + the `pt = T_imu_lidar * pt` loop of both reference callers (odometry_estimation_imu.cpp:314-316), fused
    auto deskewed = deskewing.deskew(*frame, T_imu_lidar, lv, av, CloudDeskewing::Frame::IMU);
    REQUIRE((int)deskewed->size() == m);
    std::vector<double> ref_d(4 * (size_t)m), ref_imu(4 * (size_t)m);
    orc_deskew_constvel(T_imu_lidar.m.data(), lv.data(), av.data(), ref_t.data(), ref_p.data(), m, ref_d.data());
    orc_transform_points(T_imu_lidar.m.data(), ref_d.data(), m, ref_imu.data());
    const std::vector<float> got_d = deskewed->download_points();
    for (int i = 0; i < m; i++)
      for (int a = 0; a < 3; a++) REQUIRE(got_d[3 * (size_t)i + a] == (float)ref_imu[4 * (size_t)i + a]);
    // Frame::LIDAR: the bare CloudDeskewing::deskew value
    const std::vector<float> got_l = deskewing.deskew(*frame, T_imu_lidar, lv, av, CloudDeskewing::Frame::LIDAR)->download_points();
    for (int i = 0; i < m; i++)
      for (int a = 0; a < 3; a++) REQUIRE(got_l[3 * (size_t)i + a] == (float)ref_d[4 * (size_t)i + a]);
    deskewed->estimate_covariances(10);
    auto vm = std::make_shared<GaussianVoxelMapGPU>(0.5);
    vm->insert(*deskewed);
    REQUIRE(vm->voxelmap_info().num_voxels > 100);
    std::printf("front end OK: %d -> %d points\n", nr, m);
    // submap merge of two keyframes (sub_mapping.cpp:480-497): device result == oracle, bit for bit
    {
      const int na = 20000, nb2 = 15000;
      const std::vector<double> A = make_scan(na, 0.0, 0.0, 0.0, 7), B = make_scan(nb2, 0.4, 0.2, 0.05, 8);
      std::vector<double> CA(16 * (size_t)na, 0.0), CB(16 * (size_t)nb2, 0.0);
      std::mt19937_64 r2(3);
      std::uniform_real_distribution<double> V(0.001, 0.02);
      for (auto* CC : {&CA, &CB})
        for (size_t i = 0; i < CC->size() / 16; i++) {
          double* c = CC->data() + 16 * i;
          const double a = V(r2), b = V(r2), d = V(r2), e = 0.3 * std::sqrt(a * b);
          c[0] = (double)(float)a; c[5] = (double)(float)b; c[10] = (double)(float)d; c[1] = c[4] = (double)(float)e;
        }
      const std::vector<Isometry3d> Ts = {Isometry3d::Identity(), pose2d(0.4, 0.2, 0.05)};
      const MergedFrame merged = merge_frames(Ts, {FrameView{A.data(), CA.data(), na}, FrameView{B.data(), CB.data(), nb2}}, 0.1, 9000, 11);
      std::vector<double> poses12(24);
      std::memcpy(&poses12[0], Ts[0].m.data(), 96);
      std::memcpy(&poses12[12], Ts[1].m.data(), 96);
      const double* pp[2] = {A.data(), B.data()};
      const double* cp2[2] = {CA.data(), CB.data()};
      const int sz[2] = {na, nb2};
      std::vector<double> rp(4 * (size_t)(na + nb2)), rc(16 * (size_t)(na + nb2));
      const int mm = orc_merge_frames(2, poses12.data(), pp, cp2, sz, 0.1, 1024, 9000, 11, rp.data(), rc.data());
      REQUIRE((int)merged.size() == mm);
      REQUIRE(mm >= 8999 && mm <= 9000);  // the target sampling engaged
      for (int i = 0; i < mm; i++) {
        for (int a = 0; a < 4; a++) REQUIRE(merged.points[i][a] == rp[4 * (size_t)i + a]);
        for (int a = 0; a < 16; a++) REQUIRE(merged.covs[i][a] == rc[16 * (size_t)i + a]);
      }
      auto vm2 = std::make_shared<GaussianVoxelMapGPU>(1.0);
      vm2->insert(*merged.gpu);
      REQUIRE(vm2->voxelmap_info().num_voxels > 20);
      std::printf("merge OK: %d + %d -> %d points\n", na, nb2, mm);
      // create_frame as one submission (glim_amd::create_frame): the same maps as clone + insert, level by level
      {
        const std::vector<double> levels = {0.5, 1.0};
        const FrameGPU fr = create_frame(na, A.data(), CA.data(), nullptr, levels);
        REQUIRE((int)fr.frame->size() == na && fr.voxelmaps.size() == 2);
        auto host_cloud = PointCloudGPU::clone(A.data(), CA.data(), nullptr, na);
        for (size_t lv = 0; lv < levels.size(); lv++) {
          GaussianVoxelMapGPU ref((float)levels[lv]);
          ref.insert(*host_cloud);
          REQUIRE(ref.voxelmap_info().num_voxels == fr.voxelmaps[lv]->voxelmap_info().num_voxels);
          REQUIRE(ref.voxelmap_info().num_buckets == fr.voxelmaps[lv]->voxelmap_info().num_buckets);
        }
        // an incremental map with an LRU horizon (GaussianVoxelMapCPU::set_lru_horizon): re-inserting the same frame keeps every voxel alive
        GaussianVoxelMapGPU inc(0.5f);
        inc.set_lru_horizon(2, 2);
        for (int k = 0; k < 5; k++) inc.insert(*host_cloud);
        REQUIRE(inc.voxelmap_info().num_voxels == fr.voxelmaps[0]->voxelmap_info().num_voxels);
        std::printf("create_frame / lru OK\n");
      }
    }
  }
  std::printf("test_compat OK: %d points, inliers level0 = %lld\n", n, (long long)factors[0]->linearized().num_inliers);
  return 0;
}
