// odometry_frame_loop.cpp -- GLIM's LIVE GPU odometry frame, timed end to end as ONE unit through the plain C ABI (include/glim_amd.h), from C++ as a
// GLIM module would drive it.  Per frame, the sequence of src/glim/odometry/odometry_estimation_gpu.cpp (shipped config_odometry_gpu.json):
//   create_frame (:86-107)        a NEW 10 000-pt frame arrives with CPU covariances: PointCloudGPU::clone + voxelmap_levels (2) GaussianVoxelMapGPU::insert
//   create_factors (:128-206)     (full_connection_window_size 2 + max_num_keyframes 15) x 2 levels = 34 IntegratedVGICPFactorGPU with the new frame as
//                                 source, surface validation ON: window frames binary, keyframes unary against their fixed poses
//   optimiser                     ITERS iterations, each a FRESH NonlinearFactorSetGPU: add(34 factors); linearize (:383-385 + the linearisation hook)
//   update_keyframes_overlap      one 15-target overlap_gpu (:224-231)
//   marginalisation               the frame that leaves the window gives up its cloud and its two voxel maps
// Input: a binary scene written by bench.py (`--workload odometry_frame`): header {int32 frames, int32 points, int32 keyframes, int32 window, double res0},
// then per frame points4 | covs16 | normals4 | pose (row-major 3x4 T_world_frame).  Frames [0, keyframes + window) seed the keyframes and the window;
// the rest arrive one by one, forwards then backwards along the trajectory (consecutive arrivals are always neighbours).
// Output: one JSON line -- microseconds per frame (p50 / p99 / mean over the timed frames) and the p50 of every stage.
// Build: g++ -O2 -std=c++17 -Iinclude tools/odometry_frame_loop.cpp -Lglim_amd -lglim_amd -Wl,-rpath,$PWD/glim_amd -o odometry_frame_loop
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "glim_amd.h"
#include "glim_amd_diag.h"  // set_diag A/B switches, frame-stage and plan counters: measurement hooks

#define CHECK(call)                                                                                          \
  do {                                                                                                       \
    const int rc_ = (call);                                                                                  \
    if (rc_ != GLIM_AMD_OK) {                                                                                \
      fprintf(stderr, "%s failed: %s (%s)\n", #call, glim_amd_error_string(rc_), glim_amd_last_hip_error()); \
      return 1;                                                                                              \
    }                                                                                                        \
  } while (0)

namespace {
struct HostFrame {
  std::vector<double> points4, covs16, normals4;
  double pose[12];
};
struct DeviceFrame {
  glim_amd_cloud* cloud = nullptr;
  glim_amd_voxelmap* maps[2] = {nullptr, nullptr};
  int host = -1;
};
// inv(A) * B for row-major 3x4 rigid transforms
void relative(const double* A, const double* B, double* out) {
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) out[4 * r + c] = A[0 + r] * B[0 + c] + A[4 + r] * B[4 + c] + A[8 + r] * B[8 + c];
    out[4 * r + 3] = A[0 + r] * (B[3] - A[3]) + A[4 + r] * (B[7] - A[7]) + A[8 + r] * (B[11] - A[11]);
  }
}
double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
double pct(std::vector<double> v, double p) {
  std::sort(v.begin(), v.end());
  return v[(size_t)std::min<double>(v.size() - 1, p * (v.size() - 1) + 0.5)];
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s scene.bin [frames=300] [iters=3] [one_submission=0|1] [diag switches] [tidy=0|1|2|3|4]\n", argv[0]);
    return 2;
  }
  const int timed = argc > 2 ? atoi(argv[2]) : 300, iters = argc > 3 ? atoi(argv[3]) : 3, fused = argc > 4 ? atoi(argv[4]) : 0;
  // tidy (measurement): what the loop does about the runtime's bookkeeping at the END of a frame, inside the timed unit -- 0 nothing (GLIM's loop), 1 glim_amd_ctx_synchronize
  // every frame, 2 every 16th frame, 3 a stream query on every stream of the context every frame, 4 glim_amd_ctx_synchronize right BEHIND create_frame
  const int tidy = argc > 6 ? atoi(argv[6]) : 0;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  int32_t hdr[4];
  double res0 = 0.0;
  if (fread(hdr, sizeof(int32_t), 4, f) != 4 || fread(&res0, sizeof(double), 1, f) != 1) return 2;
  const int F = hdr[0], n = hdr[1], K = hdr[2], WIN = hdr[3];
  std::vector<HostFrame> host((size_t)F);
  for (auto& h : host) {
    h.points4.resize((size_t)n * 4);
    h.covs16.resize((size_t)n * 16);
    h.normals4.resize((size_t)n * 4);
    if (fread(h.points4.data(), sizeof(double), h.points4.size(), f) != h.points4.size() || fread(h.covs16.data(), sizeof(double), h.covs16.size(), f) != h.covs16.size() ||
        fread(h.normals4.data(), sizeof(double), h.normals4.size(), f) != h.normals4.size() || fread(h.pose, sizeof(double), 12, f) != 12)
      return 2;
  }
  fclose(f);
  if (glim_amd_device_count() <= 0) {
    printf("{\"skipped\": \"no HIP device\"}\n");
    return 0;
  }
  glim_amd_ctx* ctx = nullptr;
  CHECK(glim_amd_ctx_create_ex(0, 4, nullptr, /*priority=*/1, &ctx));  // as adapters/glim/odometry_estimation_hip_create.cpp creates the odometry's pool
  if (argc > 5 && argv[5][0]) CHECK(glim_amd_ctx_set_diag(ctx, argv[5]));  // (A/B of a diagnostic switch on the same box, e.g. "pull_gated=0,plan_recycle=0")
  const double levels[2] = {res0, 2.0 * res0};

  auto make_frame = [&](int h, DeviceFrame* out) -> int {
    out->host = h;
    if (fused) {
      CHECK(glim_amd_frame_create(ctx, n, host[h].points4.data(), host[h].covs16.data(), host[h].normals4.data(), 2, levels, &out->cloud, out->maps));
      return 0;
    }
    CHECK(glim_amd_cloud_create(ctx, n, host[h].points4.data(), host[h].covs16.data(), host[h].normals4.data(), &out->cloud));
    for (int lv = 0; lv < 2; lv++) {
      CHECK(glim_amd_voxelmap_create(ctx, levels[lv], 8192 * 2, 10, 1e-3, &out->maps[lv]));
      CHECK(glim_amd_voxelmap_insert(out->maps[lv], out->cloud));
    }
    return 0;
  };
  auto drop_frame = [&](DeviceFrame* d) {
    for (int lv = 0; lv < 2; lv++) (void)glim_amd_voxelmap_destroy(d->maps[lv]);
    (void)glim_amd_cloud_destroy(d->cloud);
    *d = DeviceFrame();
  };

  std::vector<DeviceFrame> keyframes((size_t)K), window((size_t)WIN);
  for (int k = 0; k < K; k++)
    if (make_frame(k, &keyframes[(size_t)k])) return 1;
  for (int w = 0; w < WIN; w++)
    if (make_frame(K + w, &window[(size_t)w])) return 1;

  const int first = K + WIN, arrivals = F - first;
  if (arrivals < 2) return 2;
  const int NF = (WIN + K) * 2;
  std::vector<double> T((size_t)NF * 12), Tov((size_t)K * 12);
  std::vector<glim_amd_linearized6> lin((size_t)NF);
  std::vector<const glim_amd_voxelmap*> ov_maps((size_t)K);
  enum { S_CLONE_MAPS = 0, S_LIN_FIRST, S_LIN_REST, S_OVERLAP, S_RETIRE, S_COUNT };
  std::vector<double> total, stage[S_COUNT], inside[7], when_ms;  // when_ms: start of the frame, milliseconds since the first timed frame
  std::vector<int> real_allocs;  // hipMalloc + hipHostMalloc calls the library made during the frame (glim_amd_debug_pool_stats): a steady loop should make none
  double t_first_timed = 0.0;
  double checksum = 0.0;
  const int warm = 20;
  for (int it = 0; it < warm + timed; it++) {
    // forwards then backwards along the arriving part of the trajectory
    const int period = 2 * (arrivals - 1), ph = it % period, h = first + (ph < arrivals ? ph : period - ph);
    uint64_t dm0 = 0, pm0 = 0, dm1 = 0, pm1 = 0;
    (void)glim_amd_debug_pool_stats(&dm0, nullptr, &pm0, nullptr);
    const double t0 = now_us();
    DeviceFrame cur;
    if (make_frame(h, &cur)) return 1;
    if (tidy == 4) (void)glim_amd_ctx_synchronize(ctx);  // (right behind create_frame: only the maps' records kernel is still on its way)
    const double t1 = now_us();
    double fs[7] = {0, 0, 0, 0, 0, 0, 0};
    if (fused) (void)glim_amd_debug_frame_stages(fs, 7);
    // the 34 factors of this frame: (window + keyframes) x levels, the new frame as source
    int nf = 0;
    struct Item { const glim_amd_voxelmap* map; uint32_t flags; } items[64];
    for (int w = 0; w < WIN; w++)
      for (int lv = 0; lv < 2; lv++) {
        items[nf] = {window[(size_t)w].maps[lv], GLIM_AMD_FACTOR_BINARY | GLIM_AMD_FACTOR_SURFACE_VALIDATION};
        relative(host[(size_t)window[(size_t)w].host].pose, host[(size_t)h].pose, &T[(size_t)nf * 12]);
        nf++;
      }
    for (int k = 0; k < K; k++)
      for (int lv = 0; lv < 2; lv++) {
        items[nf] = {keyframes[(size_t)k].maps[lv], GLIM_AMD_FACTOR_SURFACE_VALIDATION};
        relative(host[(size_t)k].pose, host[(size_t)h].pose, &T[(size_t)nf * 12]);
        nf++;
      }
    double t_first = 0.0;
    for (int i = 0; i < iters; i++) {  // the optimiser: a FRESH set per iteration (odometry_estimation_gpu.cpp:383-385)
      glim_amd_factor_set* set = nullptr;
      CHECK(glim_amd_factor_set_create(ctx, &set));
      for (int j = 0; j < nf; j++) CHECK(glim_amd_factor_set_add(set, items[j].map, cur.cloud, items[j].flags, nullptr));
      T[3] += 1e-4 * (i + 1);  // (the optimiser moves the pose between iterations)
      CHECK(glim_amd_factor_set_linearize(set, T.data(), lin.data()));
      CHECK(glim_amd_factor_set_destroy(set));
      checksum += lin[0].error;
      if (i == 0) t_first = now_us();
    }
    const double t2 = now_us();
    for (int k = 0; k < K; k++) {
      ov_maps[(size_t)k] = keyframes[(size_t)k].maps[1];
      relative(host[(size_t)k].pose, host[(size_t)h].pose, &Tov[(size_t)k * 12]);
    }
    double ov = 0.0;
    CHECK(glim_amd_overlap(ctx, K, ov_maps.data(), Tov.data(), cur.cloud, &ov));
    checksum += ov;
    const double t3 = now_us();
    drop_frame(&window[0]);  // the oldest window frame is marginalised
    for (int w = 0; w + 1 < WIN; w++) window[(size_t)w] = window[(size_t)w + 1];
    window[(size_t)WIN - 1] = cur;
    if (tidy == 1 || (tidy == 2 && it % 16 == 15)) (void)glim_amd_ctx_synchronize(ctx);
    if (tidy == 3) (void)glim_amd_debug_ctx_query_streams(ctx, nullptr);
    const double t4 = now_us();
    (void)glim_amd_debug_pool_stats(&dm1, nullptr, &pm1, nullptr);
    if (it >= warm) {
      real_allocs.push_back((int)((dm1 - dm0) + (pm1 - pm0)));
      if (total.empty()) t_first_timed = t0;
      when_ms.push_back((t0 - t_first_timed) * 1e-3);
      total.push_back(t4 - t0);
      stage[S_CLONE_MAPS].push_back(t1 - t0);
      stage[S_LIN_FIRST].push_back(t_first - t1);
      stage[S_LIN_REST].push_back(iters > 1 ? (t2 - t_first) / (iters - 1) : 0.0);
      stage[S_OVERLAP].push_back(t3 - t2);
      stage[S_RETIRE].push_back(t4 - t3);
      for (int k = 0; k < 7; k++) inside[k].push_back(fs[k]);
    }
  }
  uint64_t plans_built = 0, plans_recycled = 0;
  (void)glim_amd_debug_plan_stats(ctx, &plans_built, &plans_recycled, nullptr);
  double mean = 0.0;
  for (double v : total) mean += v / (double)total.size();
  printf("{\"frames\": %d, \"points_per_frame\": %d, \"factors_per_frame\": %d, \"optimiser_iterations\": %d, \"one_submission_create_frame\": %d, "
         "\"frame_us\": {\"p50\": %.2f, \"p99\": %.2f, \"mean\": %.2f, \"min\": %.2f}, "
         "\"stage_p50_us\": {\"clone_and_two_voxelmaps\": %.2f, \"first_linearisation_new_factor_list\": %.2f, \"each_further_linearisation\": %.2f, "
         "\"overlap_15_targets\": %.2f, \"retire_oldest_window_frame\": %.2f}, \"factor_plans_built\": %llu, \"of_them_in_the_buffers_of_an_evicted_plan\": %llu, "
         "\"inside_frame_create_p50_us_since_entry\": {\"cloud_allocated\": %.2f, \"staging_and_stream_allocations\": %.2f, \"pull_kernel_launched\": %.2f, "
         "\"host_conversion_done\": %.2f, \"map_kernels_enqueued\": %.2f, \"completion_word_seen\": %.2f, \"return\": %.2f}, \"checksum\": %.6g, ",
         (int)total.size(), n, NF, iters, fused, pct(total, 0.5), pct(total, 0.99), mean, pct(total, 0.0), pct(stage[S_CLONE_MAPS], 0.5), pct(stage[S_LIN_FIRST], 0.5),
         pct(stage[S_LIN_REST], 0.5), pct(stage[S_OVERLAP], 0.5), pct(stage[S_RETIRE], 0.5), (unsigned long long)plans_built, (unsigned long long)plans_recycled, pct(inside[0], 0.5), pct(inside[1], 0.5), pct(inside[2], 0.5), pct(inside[3], 0.5),
         pct(inside[4], 0.5), pct(inside[5], 0.5), pct(inside[6], 0.5), checksum);
  // the tail, named: every stage's p99, and the five slowest frames with the stage that carried their excess over the median
  printf("\"stage_p99_us\": {\"clone_and_two_voxelmaps\": %.2f, \"first_linearisation_new_factor_list\": %.2f, \"each_further_linearisation\": %.2f, "
         "\"overlap_15_targets\": %.2f, \"retire_oldest_window_frame\": %.2f}, \"slowest_frames\": [",
         pct(stage[S_CLONE_MAPS], 0.99), pct(stage[S_LIN_FIRST], 0.99), pct(stage[S_LIN_REST], 0.99), pct(stage[S_OVERLAP], 0.99), pct(stage[S_RETIRE], 0.99));
  {
    std::vector<size_t> order(total.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return total[a] > total[b]; });
    const double med[S_COUNT] = {pct(stage[S_CLONE_MAPS], 0.5), pct(stage[S_LIN_FIRST], 0.5), pct(stage[S_LIN_REST], 0.5) * (iters - 1), pct(stage[S_OVERLAP], 0.5), pct(stage[S_RETIRE], 0.5)};
    const char* names[S_COUNT] = {"clone_and_two_voxelmaps", "first_linearisation", "further_linearisations", "overlap", "retire"};
    for (size_t r = 0; r < 5 && r < order.size(); r++) {
      const size_t i = order[r];
      const double v[S_COUNT] = {stage[S_CLONE_MAPS][i], stage[S_LIN_FIRST][i], stage[S_LIN_REST][i] * (iters - 1), stage[S_OVERLAP][i], stage[S_RETIRE][i]};
      int worst = 0;
      for (int k = 1; k < S_COUNT; k++)
        if (v[k] - med[k] > v[worst] - med[worst]) worst = k;
      printf("%s{\"frame\": %zu, \"at_ms\": %.2f, \"frame_us\": %.1f, \"excess_in\": \"%s\", \"excess_us\": %.1f, \"stages_us\": [%.1f, %.1f, %.1f, %.1f, %.1f], \"runtime_allocations\": %d}",
             r ? ", " : "", i, when_ms[i], total[i], names[worst], v[worst] - med[worst], v[0], v[1], v[2], v[3], v[4], real_allocs[i]);
    }
    // slow frames (> 1.4 x the median) come in BURSTS of consecutive frames or alone?  [first frame, last frame, start ms] of every run of them
    const double slow = 1.4 * pct(total, 0.5);
    int n_slow = 0;
    printf("], \"frames_above_1p4x_median\": {\"runs\": [");
    bool first_run = true;
    for (size_t i = 0; i < total.size();) {
      if (total[i] <= slow) {
        i++;
        continue;
      }
      size_t j = i;
      while (j + 1 < total.size() && (total[j + 1] > slow || (j + 2 < total.size() && total[j + 2] > slow))) j++;  // (one fast frame inside a run does not end it)
      int run_allocs = 0;
      for (size_t k = i; k <= j; k++) n_slow += total[k] > slow ? 1 : 0, run_allocs += real_allocs[k] + (k > 0 && k == i ? real_allocs[k - 1] : 0);  // (the frame before a run counts: its allocation may be what the run pays for)
      printf("%s[%zu, %zu, %.2f, %d]", first_run ? "" : ", ", i, j, when_ms[i], run_allocs);
      first_run = false;
      i = j + 1;
    }
    int all_allocs = 0;
    for (int a : real_allocs) all_allocs += a;
    printf("], \"runs_are\": \"[first frame, last frame, start ms, runtime allocations in the run and the frame before it]\", \"count\": %d, \"of\": %zu, \"runtime_allocations_in_all_timed_frames\": %d}", n_slow, total.size(), all_allocs);
  }
  printf("}\n");
  for (auto& d : keyframes) drop_frame(&d);
  for (auto& d : window) drop_frame(&d);
  CHECK(glim_amd_ctx_destroy(ctx));
  return 0;
}
