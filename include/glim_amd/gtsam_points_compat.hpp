// gtsam_points_compat.hpp -- header-only C++17 mirror, over the C ABI (glim_amd.h), of the gtsam_points GPU classes GLIM's
// VGICP path uses.  Same class names, constructor arguments and method meaning as the reference-side
// types, so that GLIM's call sites translate one to one:
//
//   auto frame    = glim_amd::PointCloudGPU::clone(points, covs, n);                       // odometry_estimation_gpu.cpp:96
//   auto voxelmap = std::make_shared<glim_amd::GaussianVoxelMapGPU>(resolution);            // :103
//   voxelmap->insert(*frame);                                                              // :104
//   auto factor   = std::make_shared<glim_amd::IntegratedVGICPFactorGPU>(target_key, source_key, voxelmap, frame);  // :144
//   factor->set_enable_surface_validation(true);                                           // :145
//   glim_amd::NonlinearFactorSetGPU set; set.add(factor); set.linearize(values);           // :383-386
//   double ov = glim_amd::overlap_gpu(voxelmap, frame, delta);                             // :248
//
// Eigen / GTSAM are not available in this image, so poses are `Isometry3d` = 12 doubles (row-major 3x4) and `Values` is a map
// key -> Isometry3d; INTEGRATION.md shows the thin gtsam::NonlinearFactor adapter that wraps these where GTSAM exists.
// Error behaviour: like the reference, constructors and methods throw std::runtime_error on device failure; a factor with no
// inliers returns zero information (H = 0, b = 0, error = 0), never NaN.
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../glim_amd.h"

namespace glim_amd {

using Key = std::uint64_t;

// Eigen::Isometry3d stand-in: row-major 3x4 [R | t].
struct Isometry3d {
  std::array<double, 12> m{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}};
  static Isometry3d Identity() { return Isometry3d(); }
  Isometry3d inverse() const {
    Isometry3d o;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) o.m[4 * r + c] = m[4 * c + r];
    for (int r = 0; r < 3; r++) o.m[4 * r + 3] = -(o.m[4 * r + 0] * m[3] + o.m[4 * r + 1] * m[7] + o.m[4 * r + 2] * m[11]);
    return o;
  }
  Isometry3d operator*(const Isometry3d& b) const {
    Isometry3d o;
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) o.m[4 * r + c] = m[4 * r + 0] * b.m[c] + m[4 * r + 1] * b.m[4 + c] + m[4 * r + 2] * b.m[8 + c];
      o.m[4 * r + 3] = m[4 * r + 0] * b.m[3] + m[4 * r + 1] * b.m[7] + m[4 * r + 2] * b.m[11] + m[4 * r + 3];
    }
    return o;
  }
};
using Values = std::map<Key, Isometry3d>;  // gtsam::Values of Pose3

inline void check(int rc, const char* what) {
  if (rc != GLIM_AMD_OK) {
    std::string msg = std::string(what) + ": " + glim_amd_error_string(rc);
    if (rc == GLIM_AMD_ERR_HIP) msg += std::string(" (") + glim_amd_last_hip_error() + ")";
    throw std::runtime_error(msg);
  }
}

// Stream priority of the contexts THIS THREAD creates from now on (0 default, 1 = the device's greatest priority, -1 its least).  GLIM builds
// each module -- and with it the module's CUDAStream / StreamTempBufferRoundRobin members -- in one constructor call
// (odometry_estimation_gpu.cpp:76-77, sub_mapping.cpp:86-87, global_mapping.cpp:109-110), so the module's plugin entry point brackets that
// call: adapters/glim/odometry_estimation_hip_create.cpp raises the priority for the odometry, whose 25 us linearisations must not queue
// behind the mapping threads' millisecond kernels.
inline int& default_stream_priority() {
  static thread_local int p = 0;
  return p;
}
struct ScopedStreamPriority {
  explicit ScopedStreamPriority(int p) : saved(default_stream_priority()) { default_stream_priority() = p; }
  ~ScopedStreamPriority() { default_stream_priority() = saved; }
  int saved;
};

// gtsam_points::CUDAStream + StreamTempBufferRoundRobin (odometry_estimation_gpu.cpp:76-77): a context = a pool of HIP streams with its own
// mutex.  Every module owns its pools, as in the reference; clouds / voxel maps may be used across the contexts of a device.
class StreamTempBufferRoundRobin : public std::enable_shared_from_this<StreamTempBufferRoundRobin> {
public:
  explicit StreamTempBufferRoundRobin(int num_streams = 8, int device = 0, int priority = default_stream_priority()) {
    check(glim_amd_ctx_create_ex(device, num_streams, nullptr, priority, &ctx_), "ctx_create");
  }
  ~StreamTempBufferRoundRobin() { glim_amd_ctx_destroy(ctx_); }
  StreamTempBufferRoundRobin(const StreamTempBufferRoundRobin&) = delete;
  StreamTempBufferRoundRobin& operator=(const StreamTempBufferRoundRobin&) = delete;
  glim_amd_ctx* context() const { return ctx_; }
  static std::shared_ptr<StreamTempBufferRoundRobin> default_instance() {
    static std::shared_ptr<StreamTempBufferRoundRobin> inst = std::make_shared<StreamTempBufferRoundRobin>(8, 0, 0);
    return inst;
  }

private:
  glim_amd_ctx* ctx_ = nullptr;
};
using Context = std::shared_ptr<StreamTempBufferRoundRobin>;

// gtsam_points::PointCloudGPU
class PointCloudGPU {
public:
  using Ptr = std::shared_ptr<PointCloudGPU>;
  using ConstPtr = std::shared_ptr<const PointCloudGPU>;
  // points: n x Vector4d; covs: n x Matrix4d (column-major) or nullptr; normals: n x Vector4d or nullptr
  static Ptr clone(const double* points4, const double* covs16, const double* normals4, std::int64_t n, Context ctx = nullptr) {
    auto c = Ptr(new PointCloudGPU(ctx ? ctx : StreamTempBufferRoundRobin::default_instance()));
    check(glim_amd_cloud_create(c->ctx_->context(), n, points4, covs16, normals4, &c->h_), "PointCloudGPU::clone");
    return c;
  }
  static Ptr clone(const float* xyz, const float* cov33, const float* normals3, std::int64_t n, Context ctx = nullptr) {
    auto c = Ptr(new PointCloudGPU(ctx ? ctx : StreamTempBufferRoundRobin::default_instance()));
    check(glim_amd_cloud_create_f32(c->ctx_->context(), n, xyz, cov33, normals3, &c->h_), "PointCloudGPU::clone");
    return c;
  }
  // takes ownership of a cloud created by another C-ABI call (glim_amd_preprocess, glim_amd_cloud_deskew)
  static Ptr adopt(glim_amd_cloud* h, Context ctx) {
    auto c = Ptr(new PointCloudGPU(ctx ? ctx : StreamTempBufferRoundRobin::default_instance()));
    c->h_ = h;
    return c;
  }
  ~PointCloudGPU() { glim_amd_cloud_destroy(h_); }
  PointCloudGPU(const PointCloudGPU&) = delete;  // owns a device handle
  PointCloudGPU& operator=(const PointCloudGPU&) = delete;
  // parity / debug: FP32 coordinates back on the host (n x 3)
  std::vector<float> download_points() const {
    std::vector<float> xyz(size() * 3);
    check(glim_amd_cloud_download(h_, xyz.data(), nullptr, nullptr, nullptr), "PointCloudGPU::download_points");
    return xyz;
  }
  std::size_t size() const {
    std::int64_t n = 0;
    glim_amd_cloud_size(h_, &n);
    return (std::size_t)n;
  }
  std::size_t memory_usage_gpu() const {
    std::size_t b = 0;
    glim_amd_cloud_memory_usage(h_, &b);
    return b;
  }
  // CloudPreprocessor::find_neighbors + CloudCovarianceEstimation::estimate on the device
  std::vector<int> find_neighbors(int k) {
    std::vector<int> out(size() * (std::size_t)k);
    check(glim_amd_cloud_find_neighbors(h_, k, out.data()), "find_neighbors");
    return out;
  }
  void estimate_covariances(int k_neighbors) { check(glim_amd_cloud_estimate_covariances(h_, k_neighbors), "estimate_covariances"); }
  glim_amd_cloud* handle() const { return h_; }
  const Context& context() const { return ctx_; }

private:
  explicit PointCloudGPU(Context ctx) : ctx_(std::move(ctx)) {}
  Context ctx_;
  glim_amd_cloud* h_ = nullptr;
};

// gtsam_points::GaussianVoxelMapGPU
class GaussianVoxelMapGPU {
public:
  using Ptr = std::shared_ptr<GaussianVoxelMapGPU>;
  using ConstPtr = std::shared_ptr<const GaussianVoxelMapGPU>;
  explicit GaussianVoxelMapGPU(float resolution, int init_num_buckets = 8192 * 2, int max_bucket_scan_count = 10,
                               double target_points_drop_rate = 1e-3, Context ctx = nullptr)
      : ctx_(ctx ? ctx : StreamTempBufferRoundRobin::default_instance()) {
    check(glim_amd_voxelmap_create(ctx_->context(), resolution, init_num_buckets, max_bucket_scan_count, target_points_drop_rate, &h_),
          "GaussianVoxelMapGPU");
  }
  ~GaussianVoxelMapGPU() { glim_amd_voxelmap_destroy(h_); }
  GaussianVoxelMapGPU(const GaussianVoxelMapGPU&) = delete;
  GaussianVoxelMapGPU& operator=(const GaussianVoxelMapGPU&) = delete;
  double voxel_resolution() const {
    double r = 0;
    glim_amd_voxelmap_info(h_, nullptr, nullptr, &r, nullptr);
    return r;
  }
  // A second insert MERGES into the voxels already there (GaussianVoxelMapCPU semantics); upstream's GPU map is believed to rebuild from the new
  // frame only (unverified).  GLIM's GPU call sites insert once per map, where both agree; see glim_amd_voxelmap_insert in glim_amd.h.
  void insert(const PointCloudGPU& frame) { check(glim_amd_voxelmap_insert(h_, frame.handle()), "GaussianVoxelMapGPU::insert"); }
  struct VoxelMapInfo {
    int num_voxels, num_buckets;
    double voxel_resolution;
    std::size_t bytes;
  };
  VoxelMapInfo voxelmap_info() const {
    VoxelMapInfo i{};
    glim_amd_voxelmap_info(h_, &i.num_voxels, &i.num_buckets, &i.voxel_resolution, &i.bytes);
    return i;
  }
  glim_amd_voxelmap* handle() const { return h_; }
  const Context& context() const { return ctx_; }
  // GaussianVoxelMapCPU::set_lru_horizon (odometry_estimation_cpu.cpp:67) for an incrementally built map; <= 0: no eviction (default)
  void set_lru_horizon(int lru_horizon, int lru_clear_cycle = 10) { check(glim_amd_voxelmap_set_lru_horizon(h_, lru_horizon, lru_clear_cycle), "set_lru_horizon"); }
  // takes ownership of a map created by another C-ABI call (glim_amd_frame_create)
  static Ptr adopt(glim_amd_voxelmap* h, Context ctx) {
    Ptr m(new GaussianVoxelMapGPU(std::move(ctx), h));
    return m;
  }

private:
  GaussianVoxelMapGPU(Context ctx, glim_amd_voxelmap* h) : ctx_(ctx ? ctx : StreamTempBufferRoundRobin::default_instance()), h_(h) {}
  Context ctx_;
  glim_amd_voxelmap* h_ = nullptr;
};

// OdometryEstimationGPU::create_frame (odometry_estimation_gpu.cpp:86-107) as ONE submission: PointCloudGPU::clone of a frame that arrives with CPU
// covariances (+ normals) and one GaussianVoxelMapGPU per level, enqueued back to back with one completion (glim_amd_frame_create) -- what the three
// statements of create_frame (:96, :103-104) cost as separate calls minus two synchronises.  points4 / covs16 / normals4: the raw arrays of
// gtsam_points::PointCloud (Vector4d / column-major Matrix4d / Vector4d).  GLIM's unmodified source calls clone and insert one at a time.
struct FrameGPU {
  PointCloudGPU::Ptr frame;
  std::vector<GaussianVoxelMapGPU::Ptr> voxelmaps;
};
inline FrameGPU create_frame(std::int64_t n, const double* points4, const double* covs16, const double* normals4, const std::vector<double>& resolutions,
                             Context ctx = nullptr) {
  ctx = ctx ? ctx : StreamTempBufferRoundRobin::default_instance();
  glim_amd_cloud* c = nullptr;
  std::vector<glim_amd_voxelmap*> m(resolutions.size(), nullptr);
  check(glim_amd_frame_create(ctx->context(), n, points4, covs16, normals4, (std::int32_t)resolutions.size(), resolutions.data(), &c, m.data()), "create_frame");
  FrameGPU out;
  out.frame = PointCloudGPU::adopt(c, ctx);
  for (glim_amd_voxelmap* h : m) out.voxelmaps.push_back(GaussianVoxelMapGPU::adopt(h, ctx));
  return out;
}

// Result of linearize(): the ingredients of gtsam::HessianFactor(k_t, k_s, H_tt, H_ts, -b_t, H_ss, -b_s, error).
using LinearizedSystem6 = glim_amd_linearized6;

class NonlinearFactorSetGPU;

// gtsam_points::IntegratedVGICPFactorGPU
class IntegratedVGICPFactorGPU {
public:
  using shared_ptr = std::shared_ptr<IntegratedVGICPFactorGPU>;
  // binary: (target_key, source_key, target voxel map, source frame)          odometry_estimation_gpu.cpp:144
  // ctx: the (stream, buffer) pair the reference's constructors take from the module's StreamTempBufferRoundRobin, i.e. the pool a set of
  // such factors runs on; null = the context of the target map
  IntegratedVGICPFactorGPU(Key target_key, Key source_key, GaussianVoxelMapGPU::ConstPtr target, PointCloudGPU::ConstPtr source, Context ctx = nullptr)
      : is_binary_(true), target_key_(target_key), source_key_(source_key), target_(std::move(target)), source_(std::move(source)),
        ctx_(ctx ? std::move(ctx) : target_->context()) {}
  // unary: (fixed_target_pose, source_key, target voxel map, source frame)    odometry_estimation_gpu.cpp:161
  IntegratedVGICPFactorGPU(const Isometry3d& fixed_target_pose, Key source_key, GaussianVoxelMapGPU::ConstPtr target, PointCloudGPU::ConstPtr source,
                           Context ctx = nullptr)
      : is_binary_(false), target_key_(0), source_key_(source_key), fixed_target_pose_(fixed_target_pose), target_(std::move(target)),
        source_(std::move(source)), ctx_(ctx ? std::move(ctx) : target_->context()) {}
  const Context& context() const { return ctx_; }

  void set_enable_surface_validation(bool enable) { surface_validation_ = enable; }
  std::vector<Key> keys() const { return is_binary_ ? std::vector<Key>{target_key_, source_key_} : std::vector<Key>{source_key_}; }
  std::size_t dim() const { return 6; }
  const Isometry3d& get_fixed_target_pose() const { return fixed_target_pose_; }
  std::size_t memory_usage() const { return sizeof(LinearizedSystem6); }
  std::size_t memory_usage_gpu() const { return source_->memory_usage_gpu() + target_->voxelmap_info().bytes; }
  shared_ptr clone() const { return std::make_shared<IntegratedVGICPFactorGPU>(*this); }
  bool is_binary() const { return is_binary_; }
  std::uint32_t flags() const {
    return (is_binary_ ? GLIM_AMD_FACTOR_BINARY : 0u) | (surface_validation_ ? GLIM_AMD_FACTOR_SURFACE_VALIDATION : 0u);
  }
  // T_target_source at `values`
  Isometry3d calc_delta(const Values& values) const {
    const Isometry3d& Ts = values.at(source_key_);
    return (is_binary_ ? values.at(target_key_) : fixed_target_pose_).inverse() * Ts;
  }
  double inlier_fraction() const { return linearized_valid_ ? (double)linearized_.num_inliers / (double)std::max<std::size_t>(1, source_->size()) : 0.0; }

  // slow path (own upload / launch / download), used when no NonlinearFactorSetGPU pre-linearised this factor
  inline const LinearizedSystem6& linearize(const Values& values);
  // `linearization_values` != nullptr: correspondences and Mahalanobis matrices frozen at that point (the GPU factor's behaviour after a
  // linearisation); nullptr: recomputed at `values` (the CPU factor's behaviour)
  inline double error(const Values& values, const Values* linearization_values = nullptr);

  const GaussianVoxelMapGPU::ConstPtr& target() const { return target_; }
  const PointCloudGPU::ConstPtr& source() const { return source_; }
  void store_linearized(const LinearizedSystem6& l) {
    linearized_ = l;
    linearized_valid_ = true;
  }
  const LinearizedSystem6& linearized() const { return linearized_; }

private:
  bool is_binary_;
  Key target_key_, source_key_;
  Isometry3d fixed_target_pose_;
  GaussianVoxelMapGPU::ConstPtr target_;
  PointCloudGPU::ConstPtr source_;
  Context ctx_;
  bool surface_validation_ = false;
  bool linearized_valid_ = false;
  LinearizedSystem6 linearized_{};
};

// gtsam_points::NonlinearFactorSetGPU: one fused launch for every added factor.
// A set made without a context (the optimisers' linearisation hook makes it that way: offline_viewer.cpp:29) runs on the context of the first
// factor it is given -- i.e. on the stream pool of the module that built the factor.
class NonlinearFactorSetGPU {
public:
  explicit NonlinearFactorSetGPU(Context ctx = nullptr) : ctx_(std::move(ctx)) {
    if (ctx_) check(glim_amd_factor_set_create(ctx_->context(), &h_), "NonlinearFactorSetGPU");
  }
  ~NonlinearFactorSetGPU() { glim_amd_factor_set_destroy(h_); }
  NonlinearFactorSetGPU(const NonlinearFactorSetGPU&) = delete;
  NonlinearFactorSetGPU& operator=(const NonlinearFactorSetGPU&) = delete;
  bool add(const IntegratedVGICPFactorGPU::shared_ptr& factor) {
    if (!factor) return false;
    if (!h_) {
      ctx_ = factor->context() ? factor->context() : StreamTempBufferRoundRobin::default_instance();
      check(glim_amd_factor_set_create(ctx_->context(), &h_), "NonlinearFactorSetGPU");
    }
    check(glim_amd_factor_set_add(h_, factor->target()->handle(), factor->source()->handle(), factor->flags(), nullptr), "NonlinearFactorSetGPU::add");
    factors_.push_back(factor);
    return true;
  }
  void clear() {
    if (h_) glim_amd_factor_set_clear(h_);
    factors_.clear();
  }
  std::size_t size() const { return factors_.size(); }
  void linearize(const Values& values) {
    if (factors_.empty()) return;
    std::vector<double> poses(12 * factors_.size());
    for (std::size_t i = 0; i < factors_.size(); i++) std::memcpy(&poses[12 * i], factors_[i]->calc_delta(values).m.data(), 12 * sizeof(double));
    std::vector<LinearizedSystem6> out(factors_.size());
    check(glim_amd_factor_set_linearize(h_, poses.data(), out.data()), "NonlinearFactorSetGPU::linearize");
    for (std::size_t i = 0; i < factors_.size(); i++) factors_[i]->store_linearized(out[i]);
  }
  std::vector<double> error(const Values& values, const Values* linearization_values = nullptr) {
    std::vector<double> err(factors_.size());
    if (factors_.empty()) return err;
    std::vector<double> poses(12 * factors_.size()), lin(linearization_values ? 12 * factors_.size() : 0);
    for (std::size_t i = 0; i < factors_.size(); i++) {
      std::memcpy(&poses[12 * i], factors_[i]->calc_delta(values).m.data(), 12 * sizeof(double));
      if (linearization_values) std::memcpy(&lin[12 * i], factors_[i]->calc_delta(*linearization_values).m.data(), 12 * sizeof(double));
    }
    check(glim_amd_factor_set_error(h_, linearization_values ? lin.data() : nullptr, poses.data(), err.data(), nullptr), "NonlinearFactorSetGPU::error");
    return err;
  }
  const std::vector<IntegratedVGICPFactorGPU::shared_ptr>& factors() const { return factors_; }

private:
  Context ctx_;
  glim_amd_factor_set* h_ = nullptr;
  std::vector<IntegratedVGICPFactorGPU::shared_ptr> factors_;
};

inline const LinearizedSystem6& IntegratedVGICPFactorGPU::linearize(const Values& values) {
  glim_amd_factor_set* set = nullptr;
  check(glim_amd_factor_set_create(ctx_->context(), &set), "factor_set_create");
  int rc = glim_amd_factor_set_add(set, target_->handle(), source_->handle(), flags(), nullptr);
  const Isometry3d d = calc_delta(values);
  LinearizedSystem6 out{};
  if (rc == GLIM_AMD_OK) rc = glim_amd_factor_set_linearize(set, d.m.data(), &out);
  glim_amd_factor_set_destroy(set);
  check(rc, "IntegratedVGICPFactorGPU::linearize");
  store_linearized(out);
  return linearized_;
}

inline double IntegratedVGICPFactorGPU::error(const Values& values, const Values* linearization_values) {
  glim_amd_factor_set* set = nullptr;
  check(glim_amd_factor_set_create(ctx_->context(), &set), "factor_set_create");
  int rc = glim_amd_factor_set_add(set, target_->handle(), source_->handle(), flags(), nullptr);
  const Isometry3d d = calc_delta(values);
  Isometry3d dl;
  if (linearization_values) dl = calc_delta(*linearization_values);
  double e = 0.0;
  if (rc == GLIM_AMD_OK) rc = glim_amd_factor_set_error(set, linearization_values ? dl.m.data() : nullptr, d.m.data(), &e, nullptr);
  glim_amd_factor_set_destroy(set);
  check(rc, "IntegratedVGICPFactorGPU::error");
  return e;
}

// gtsam_points::KdTree of a target frame as GLIM's pose-graph module caches it (global_mapping_pose_graph.cpp:393,
// `candidate.target->tree`): here the device grid index of glim_amd_nn_index_create.
class NearestNeighborSearchGPU {
public:
  using Ptr = std::shared_ptr<NearestNeighborSearchGPU>;
  using ConstPtr = std::shared_ptr<const NearestNeighborSearchGPU>;
  explicit NearestNeighborSearchGPU(PointCloudGPU::ConstPtr target, double max_correspondence_distance_hint = 1.0) : target_(std::move(target)) {
    check(glim_amd_nn_index_create(target_->handle(), max_correspondence_distance_hint, &h_), "NearestNeighborSearchGPU");
  }
  ~NearestNeighborSearchGPU() { glim_amd_nn_index_destroy(h_); }
  NearestNeighborSearchGPU(const NearestNeighborSearchGPU&) = delete;
  NearestNeighborSearchGPU& operator=(const NearestNeighborSearchGPU&) = delete;
  glim_amd_nn_index* handle() const { return h_; }
  const PointCloudGPU::ConstPtr& target() const { return target_; }

private:
  PointCloudGPU::ConstPtr target_;  // kept alive: the index refers to it
  glim_amd_nn_index* h_ = nullptr;
};

// gtsam_points::IntegratedGICPFactor (sub_mapping.cpp:202, global_mapping.cpp:400-402, global_mapping_pose_graph.cpp:393-405)
class IntegratedGICPFactor {
public:
  using shared_ptr = std::shared_ptr<IntegratedGICPFactor>;
  // binary: (target_key, source_key, target frame, source frame[, target tree])
  IntegratedGICPFactor(Key target_key, Key source_key, PointCloudGPU::ConstPtr target, PointCloudGPU::ConstPtr source,
                       NearestNeighborSearchGPU::ConstPtr target_tree = nullptr)
      : is_binary_(true), target_key_(target_key), source_key_(source_key), source_(std::move(source)),
        tree_(target_tree ? std::move(target_tree) : std::make_shared<NearestNeighborSearchGPU>(std::move(target))) {}
  // unary: (fixed_target_pose, source_key, target frame, source frame[, target tree])
  IntegratedGICPFactor(const Isometry3d& fixed_target_pose, Key source_key, PointCloudGPU::ConstPtr target, PointCloudGPU::ConstPtr source,
                       NearestNeighborSearchGPU::ConstPtr target_tree = nullptr)
      : is_binary_(false), target_key_(0), source_key_(source_key), fixed_target_pose_(fixed_target_pose), source_(std::move(source)),
        tree_(target_tree ? std::move(target_tree) : std::make_shared<NearestNeighborSearchGPU>(std::move(target))) {}

  void set_max_correspondence_distance(double d) { max_correspondence_distance_ = d; }
  void set_num_threads(int) {}  // accepted for source compatibility (global_mapping.cpp:402); the device has no thread knob
  std::vector<Key> keys() const { return is_binary_ ? std::vector<Key>{target_key_, source_key_} : std::vector<Key>{source_key_}; }
  std::size_t dim() const { return 6; }
  bool is_binary() const { return is_binary_; }
  Isometry3d calc_delta(const Values& values) const {
    const Isometry3d& Ts = values.at(source_key_);
    return (is_binary_ ? values.at(target_key_) : fixed_target_pose_).inverse() * Ts;
  }
  const LinearizedSystem6& linearize(const Values& values) {
    const Isometry3d d = calc_delta(values);
    check(glim_amd_gicp_linearize(tree_->handle(), source_->handle(), d.m.data(), max_correspondence_distance_, is_binary_ ? GLIM_AMD_FACTOR_BINARY : 0u,
                                  &linearized_),
          "IntegratedGICPFactor::linearize");
    num_inliers_ = linearized_.num_inliers;
    return linearized_;
  }
  double error(const Values& values) {
    const Isometry3d d = calc_delta(values);
    double e = 0.0;
    check(glim_amd_gicp_error(tree_->handle(), source_->handle(), d.m.data(), max_correspondence_distance_, &e, &num_inliers_), "IntegratedGICPFactor::error");
    return e;
  }
  double inlier_fraction() const { return (double)num_inliers_ / (double)std::max<std::size_t>(1, source_->size()); }
  const LinearizedSystem6& linearized() const { return linearized_; }

private:
  bool is_binary_;
  Key target_key_, source_key_;
  Isometry3d fixed_target_pose_;
  PointCloudGPU::ConstPtr source_;
  NearestNeighborSearchGPU::ConstPtr tree_;
  double max_correspondence_distance_ = 1.0;  // gtsam_points default: max_correspondence_distance_sq = 1.0
  std::int64_t num_inliers_ = 0;
  LinearizedSystem6 linearized_{};
};

// gtsam_points::median_distance(frame, max_scan_count) and the resolution blend GLIM applies to it
// (odometry_estimation_gpu.cpp:90-93, global_mapping.cpp:238-241): host-side, as in the reference.
inline double median_distance(const double* points4, std::int64_t n, std::size_t max_scan_count = 256) {
  if (n <= 0) return 0.0;
  const std::size_t step = (std::size_t)n < max_scan_count ? 1 : (std::size_t)n / max_scan_count;
  std::vector<double> dists;
  dists.reserve((std::size_t)n / step + 1);
  for (std::size_t i = 0; i < (std::size_t)n; i += step) {
    const double* p = points4 + 4 * i;
    dists.push_back(std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]));
  }
  std::nth_element(dists.begin(), dists.begin() + dists.size() / 2, dists.end());
  return dists[dists.size() / 2];
}
inline double adaptive_voxel_resolution(double dist_median, double voxel_resolution, double voxel_resolution_max, double voxel_resolution_dmin,
                                        double voxel_resolution_dmax) {
  const double p = std::max(0.0, std::min(1.0, (dist_median - voxel_resolution_dmin) / (voxel_resolution_dmax - voxel_resolution_dmin)));
  return voxel_resolution + p * (voxel_resolution_max - voxel_resolution);
}

// gtsam_points::overlap_gpu (single and multi-target forms) / overlap_auto   (odometry_estimation_gpu.cpp:231-326)
inline double overlap_gpu(const GaussianVoxelMapGPU::ConstPtr& target, const PointCloudGPU::ConstPtr& source, const Isometry3d& delta) {
  const glim_amd_voxelmap* t = target->handle();
  double ov = 0.0;
  check(glim_amd_overlap(target->context()->context(), 1, &t, delta.m.data(), source->handle(), &ov), "overlap_gpu");
  return ov;
}
inline double overlap_gpu(const std::vector<GaussianVoxelMapGPU::ConstPtr>& targets, const PointCloudGPU::ConstPtr& source,
                          const std::vector<Isometry3d>& deltas) {
  if (targets.empty() || targets.size() != deltas.size()) throw std::runtime_error("overlap_gpu: targets/deltas mismatch");
  std::vector<const glim_amd_voxelmap*> t(targets.size());
  std::vector<double> T(12 * targets.size());
  for (std::size_t i = 0; i < targets.size(); i++) {
    t[i] = targets[i]->handle();
    std::memcpy(&T[12 * i], deltas[i].m.data(), 12 * sizeof(double));
  }
  double ov = 0.0;
  check(glim_amd_overlap(targets[0]->context()->context(), (int)targets.size(), t.data(), T.data(), source->handle(), &ov), "overlap_gpu");
  return ov;
}
inline double overlap_auto(const GaussianVoxelMapGPU::ConstPtr& target, const PointCloudGPU::ConstPtr& source, const Isometry3d& delta) {
  return overlap_gpu(target, source, delta);
}
// Extension (no upstream counterpart): many overlap_gpu calls answered by ONE launch -- the keyframe elimination loop of
// odometry_estimation_gpu.cpp:262-281 issues 43 of them back to back for 15 keyframes.  queries[q] = (targets, source, deltas).
struct OverlapQuery {
  std::vector<GaussianVoxelMapGPU::ConstPtr> targets;
  PointCloudGPU::ConstPtr source;
  std::vector<Isometry3d> deltas;
};
inline std::vector<double> overlap_gpu_batch(const std::vector<OverlapQuery>& queries) {
  std::vector<double> out(queries.size(), 0.0);
  if (queries.empty()) return out;
  std::vector<int32_t> num_targets;
  std::vector<const glim_amd_voxelmap*> maps;
  std::vector<const glim_amd_cloud*> sources;
  std::vector<double> T;
  for (const auto& q : queries) {
    if (q.targets.empty() || q.targets.size() != q.deltas.size() || !q.source) throw std::runtime_error("overlap_gpu_batch: targets/deltas mismatch");
    num_targets.push_back((int32_t)q.targets.size());
    sources.push_back(q.source->handle());
    for (std::size_t i = 0; i < q.targets.size(); i++) {
      maps.push_back(q.targets[i]->handle());
      T.insert(T.end(), q.deltas[i].m.begin(), q.deltas[i].m.end());
    }
  }
  check(glim_amd_overlap_batch(queries[0].source->context()->context(), (int32_t)queries.size(), num_targets.data(), maps.data(), T.data(), sources.data(), out.data()),
        "overlap_gpu_batch");
  return out;
}

}  // namespace glim_amd
