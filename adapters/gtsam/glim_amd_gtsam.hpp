// glim_amd_gtsam.hpp -- the GTSAM-facing adapter of the MI355X VGICP path: what a GLIM maintainer compiles inside GLIM (next to
// src/glim/odometry/odometry_estimation_gpu.cpp) so that the call sites keep their shape.  Layer (3) of INTEGRATION.md.
//
//   gtsam_points::IntegratedVGICPFactorGPU(target_key, source_key, voxelmap, frame, stream, buffer)   odometry_estimation_gpu.cpp:144
//   gtsam_points::IntegratedVGICPFactorGPU(fixed_target_pose, source_key, voxelmap, frame, ...)       odometry_estimation_gpu.cpp:161
//       -> glim_amd::IntegratedVGICPFactorHIP            (a gtsam::NonlinearFactor: error / linearize / dim / clone + the extras GLIM calls)
//   gtsam_points::IntegratedGICPFactor(...)                                                            sub_mapping.cpp:202, global_mapping.cpp:400
//       -> glim_amd::IntegratedGICPFactorHIP
//   gtsam_points::NonlinearFactorSetGPU / create_nonlinear_factor_set_gpu()                            odometry_estimation_gpu.cpp:383-385, offline_viewer.cpp:29
//       -> glim_amd::NonlinearFactorSetHIP / glim_amd::create_nonlinear_factor_set_hip()
//   gtsam_points::PointCloudGPU::clone(frame), overlap_gpu(voxelmap, frame, Isometry3d)                :96, :231-326
//       -> glim_amd::clone(frame), glim_amd::overlap_gpu(...)
//
// It needs GTSAM (>= 4.2), Eigen and gtsam_points' PointCloud / LinearizationHook headers, none of which exist in the image this
// repository is developed in: there it is compiled and exercised against the minimal stand-ins of tests/cpp/mock/
// (tests/test_adapter.py), which pin its logic -- keys, Hessian blocks, gradient signs, the batch protocol -- but not the exact
// upstream header paths and virtual signatures, which are written from the call sites and from memory.
// Only plain element access (`operator()(i, j)`, `.data()`) is used on Eigen / GTSAM matrix types, so no Eigen expression
// template can change the meaning of a line.
//
// Threading: like the reference's GPU factors, an object caches its last linearisation; GLIM evaluates a graph from one thread
// (it pins GTSAM's TBB arena to one thread), and the C ABI below serialises calls per context anyway.
// Errors: device failures surface as std::runtime_error from the constructors / linearize (as gtsam_points does); a factor without
// inliers yields zero information and zero error, never NaN.
#pragma once

#include <gtsam/geometry/Pose3.h>
#include <gtsam/linear/HessianFactor.h>
#include <gtsam/nonlinear/NonlinearFactor.h>
#include <gtsam/nonlinear/NonlinearFactorGraph.h>
#include <gtsam/nonlinear/Values.h>

#include <Eigen/Core>
#include <Eigen/Geometry>
#include <gtsam_points/optimizers/linearization_hook.hpp>
#include <gtsam_points/types/point_cloud.hpp>

#include <glim_amd/gtsam_points_compat.hpp>

// One of the three semantic choices that could not be checked against gtsam_points in the image this repository is developed in:
// whether IntegratedVGICPFactor{,GPU}::error() returns E = sum r^T M r or E / 2.  H, b -- and with them every Gauss-Newton / LM step -- do not depend
// on it; an LM accept / reject test compares error() of the SAME factor type at two points, so it does not either.  A maintainer who finds upstream
// returns E / 2 builds with -DGLIM_AMD_VGICP_ERROR_SCALE=0.5 (the oracle's twin is ORC_ERROR_SCALE in oracle/vgicp_oracle.h).
#ifndef GLIM_AMD_VGICP_ERROR_SCALE
#define GLIM_AMD_VGICP_ERROR_SCALE 1.0
#endif

namespace glim_amd {

// ---- type conversions ------------------------------------------------------------------------------------------------
template <class Matrix4>
inline Isometry3d to_iso_from_matrix(const Matrix4& m) {  // any 4x4 with (row, col) access -> row-major 3x4
  Isometry3d T;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) T.m[(std::size_t)(4 * r + c)] = m(r, c);
  return T;
}
inline Isometry3d to_iso(const gtsam::Pose3& p) { return to_iso_from_matrix(p.matrix()); }
inline Isometry3d to_iso(const Eigen::Isometry3d& T) { return to_iso_from_matrix(T.matrix()); }

// gtsam_points::PointCloudGPU::clone(frame): Vector4d points / Matrix4d covs / Vector4d normals (raw pointers) -> device frame
inline PointCloudGPU::Ptr clone(const gtsam_points::PointCloud& frame, Context ctx = nullptr) {
  return PointCloudGPU::clone(frame.size() ? frame.points[0].data() : nullptr, frame.has_covs() && frame.size() ? frame.covs[0].data() : nullptr,
                              frame.has_normals() && frame.size() ? frame.normals[0].data() : nullptr, (std::int64_t)frame.size(), std::move(ctx));
}

// gtsam::HessianFactor(keys..., H blocks, -b blocks, error) from a linearised system (row-major 6x6 blocks)
inline gtsam::GaussianFactor::shared_ptr make_hessian_factor(const gtsam::KeyVector& keys, bool binary, const LinearizedSystem6& l) {
  auto mat = [](const double* h) {
    gtsam::Matrix M(6, 6);
    for (int r = 0; r < 6; r++)
      for (int c = 0; c < 6; c++) M(r, c) = h[6 * r + c];
    return M;
  };
  auto neg = [](const double* b) {
    gtsam::Vector v(6);
    for (int r = 0; r < 6; r++) v(r) = -b[r];
    return v;
  };
  if (binary)
    return gtsam::GaussianFactor::shared_ptr(
      new gtsam::HessianFactor(keys[0], keys[1], mat(l.H_tt), mat(l.H_ts), neg(l.b_t), mat(l.H_ss), neg(l.b_s), GLIM_AMD_VGICP_ERROR_SCALE * l.error));
  return gtsam::GaussianFactor::shared_ptr(new gtsam::HessianFactor(keys[0], mat(l.H_ss), neg(l.b_s), GLIM_AMD_VGICP_ERROR_SCALE * l.error));
}

inline bool same_pose(const Isometry3d& a, const Isometry3d& b) { return a.m == b.m; }

// ---- gtsam_points::IntegratedVGICPFactorGPU -----------------------------------------------------------------------------
class IntegratedVGICPFactorHIP : public gtsam::NonlinearFactor {
public:
  using shared_ptr = std::shared_ptr<IntegratedVGICPFactorHIP>;

  // binary factor between two pose variables (odometry_estimation_gpu.cpp:144, sub_mapping.cpp:307, global_mapping.cpp:335,466,860).
  // ctx: the context (stream pool) a set of such factors runs on -- the reference's trailing (CUstream_st*, TempBufferManager) pair.
  IntegratedVGICPFactorHIP(gtsam::Key target_key, gtsam::Key source_key, GaussianVoxelMapGPU::ConstPtr target, PointCloudGPU::ConstPtr source, Context ctx = nullptr)
  : gtsam::NonlinearFactor(gtsam::KeyVector{target_key, source_key}),
    impl_(std::make_shared<IntegratedVGICPFactorGPU>((Key)target_key, (Key)source_key, std::move(target), std::move(source), std::move(ctx))) {}
  // unary factor against a fixed target pose (odometry_estimation_gpu.cpp:161)
  IntegratedVGICPFactorHIP(const gtsam::Pose3& fixed_target_pose, gtsam::Key source_key, GaussianVoxelMapGPU::ConstPtr target, PointCloudGPU::ConstPtr source,
                           Context ctx = nullptr)
  : gtsam::NonlinearFactor(gtsam::KeyVector{source_key}),
    impl_(std::make_shared<IntegratedVGICPFactorGPU>(to_iso(fixed_target_pose), (Key)source_key, std::move(target), std::move(source), std::move(ctx))) {}

  size_t dim() const override { return 6; }
  gtsam::NonlinearFactor::shared_ptr clone() const override {  // odometry_estimation_gpu.cpp:380: shares the device data, not the cache
    auto f = std::make_shared<IntegratedVGICPFactorHIP>(*this);
    f->impl_ = impl_->clone();
    return f;
  }

  // the extras GLIM calls on the GPU factor
  void set_enable_surface_validation(bool enable) { impl_->set_enable_surface_validation(enable); }   // odometry_estimation_gpu.cpp:145,162
  Eigen::Isometry3d get_fixed_target_pose() const {                                                    // standard_viewer_callbacks.cpp:283
    Eigen::Isometry3d T = Eigen::Isometry3d::Identity();
    const Isometry3d& f = impl_->get_fixed_target_pose();
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 4; c++) T.matrix()(r, c) = f.m[(std::size_t)(4 * r + c)];
    return T;
  }
  size_t memory_usage() const { return impl_->memory_usage(); }                                        // standard_viewer_mem.cpp:160
  size_t memory_usage_gpu() const { return impl_->memory_usage_gpu(); }                                // standard_viewer_mem.cpp:161
  double inlier_fraction() const { return impl_->inlier_fraction(); }
  bool is_binary() const { return impl_->is_binary(); }

  // gtsam::NonlinearFactor.  If a NonlinearFactorSetHIP linearised / evaluated this factor at the same poses, its result is used
  // (one fused launch for the whole graph); otherwise the factor does its own launch (slow path), as upstream.
  gtsam::GaussianFactor::shared_ptr linearize(const gtsam::Values& values) const override {
    const Isometry3d delta = impl_->calc_delta(to_values(values));
    if (!(lin_valid_ && same_pose(delta, lin_delta_))) {
      impl_->linearize(to_values(values));
      lin_delta_ = delta;
      lin_valid_ = true;
      lin_values_ = to_values(values);
      err_valid_ = false;  // a cached error was evaluated with the correspondences of the PREVIOUS linearisation point
    }
    return make_hessian_factor(keys(), impl_->is_binary(), impl_->linearized());
  }
  double error(const gtsam::Values& values) const override {
    const Isometry3d delta = impl_->calc_delta(to_values(values));
    if (err_valid_ && same_pose(delta, err_delta_)) return err_;
    // after a linearisation the GPU factor evaluates with the correspondences frozen at the linearisation point; the
    // CPU-named factor (gtsam_points/factors/integrated_vgicp_factor.hpp of the shim tree) finds them at `values` (row a6)
    err_ = GLIM_AMD_VGICP_ERROR_SCALE * impl_->error(to_values(values), (frozen_error_ && lin_valid_) ? &lin_values_ : nullptr);
    err_delta_ = delta;
    err_valid_ = true;
    return err_;
  }

  // ---- batch protocol (driven by NonlinearFactorSetHIP) ----
  const IntegratedVGICPFactorGPU::shared_ptr& impl() const { return impl_; }
  Values to_values(const gtsam::Values& values) const {
    Values out;
    for (const gtsam::Key k : keys()) out[(Key)k] = to_iso(values.at<gtsam::Pose3>(k));
    return out;
  }
  void store_linearized(const gtsam::Values& values) const {  // impl_ already holds the batch result
    lin_values_ = to_values(values);
    lin_delta_ = impl_->calc_delta(lin_values_);
    lin_valid_ = true;
    err_valid_ = false;  // see linearize()
  }
  void store_error(const gtsam::Values& values, double e) const {
    err_delta_ = impl_->calc_delta(to_values(values));
    err_ = e;
    err_valid_ = true;
  }
  bool has_linearization_point() const { return lin_valid_; }
  // false: error() finds its correspondences at the evaluation pose (CPU-factor semantics); true (default): frozen at the last linearisation point
  bool frozen_error() const { return frozen_error_; }
  const Values& linearization_values() const { return lin_values_; }

protected:
  void set_frozen_error(bool frozen) { frozen_error_ = frozen; }
  void reset_impl_clone() { impl_ = impl_->clone(); }  // (a copy-constructed factor shares impl_ with its source until this is called)

private:
  IntegratedVGICPFactorGPU::shared_ptr impl_;
  bool frozen_error_ = true;
  mutable bool lin_valid_ = false, err_valid_ = false;
  mutable Isometry3d lin_delta_, err_delta_;
  mutable Values lin_values_;
  mutable double err_ = 0.0;
};

// ---- gtsam_points::IntegratedGICPFactor (nearest-neighbour correspondences) ------------------------------------------
class IntegratedGICPFactorHIP : public gtsam::NonlinearFactor {
public:
  using shared_ptr = std::shared_ptr<IntegratedGICPFactorHIP>;
  IntegratedGICPFactorHIP(gtsam::Key target_key, gtsam::Key source_key, PointCloudGPU::ConstPtr target, PointCloudGPU::ConstPtr source,
                          NearestNeighborSearchGPU::ConstPtr target_tree = nullptr)   // sub_mapping.cpp:202, global_mapping_pose_graph.cpp:393
  : gtsam::NonlinearFactor(gtsam::KeyVector{target_key, source_key}),
    impl_(std::make_shared<IntegratedGICPFactor>((Key)target_key, (Key)source_key, std::move(target), std::move(source), std::move(target_tree))) {}
  IntegratedGICPFactorHIP(const gtsam::Pose3& fixed_target_pose, gtsam::Key source_key, PointCloudGPU::ConstPtr target, PointCloudGPU::ConstPtr source,
                          NearestNeighborSearchGPU::ConstPtr target_tree = nullptr)
  : gtsam::NonlinearFactor(gtsam::KeyVector{source_key}),
    impl_(std::make_shared<IntegratedGICPFactor>(to_iso(fixed_target_pose), (Key)source_key, std::move(target), std::move(source), std::move(target_tree))) {}

  size_t dim() const override { return 6; }
  gtsam::NonlinearFactor::shared_ptr clone() const override {
    auto f = std::make_shared<IntegratedGICPFactorHIP>(*this);
    f->impl_ = std::make_shared<IntegratedGICPFactor>(*impl_);
    return f;
  }
  void set_max_correspondence_distance(double d) { impl_->set_max_correspondence_distance(d); }   // global_mapping.cpp:401
  void set_num_threads(int n) { impl_->set_num_threads(n); }                                       // global_mapping.cpp:402 (no-op)
  double inlier_fraction() const { return impl_->inlier_fraction(); }                              // global_mapping_pose_graph.cpp:417

  gtsam::GaussianFactor::shared_ptr linearize(const gtsam::Values& values) const override {
    return make_hessian_factor(keys(), impl_->is_binary(), impl_->linearize(to_values(values)));
  }
  double error(const gtsam::Values& values) const override { return impl_->error(to_values(values)); }

private:
  Values to_values(const gtsam::Values& values) const {
    Values out;
    for (const gtsam::Key k : keys()) out[(Key)k] = to_iso(values.at<gtsam::Pose3>(k));
    return out;
  }
  std::shared_ptr<IntegratedGICPFactor> impl_;
};

// ---- gtsam_points::NonlinearFactorSetGPU ---------------------------------------------------------------------------------
// One fused launch for every HIP factor of a graph.  The Ext optimisers obtain it through LinearizationHook and call
// add(graph) / linearize(values) before graph.linearize(values); GLIM also drives it by hand (odometry_estimation_gpu.cpp:383-385).
class NonlinearFactorSetHIP : public gtsam_points::NonlinearFactorSet {
public:
  explicit NonlinearFactorSetHIP(Context ctx = nullptr) : set_(std::move(ctx)) {}

  int size() const override { return (int)factors_.size(); }
  void clear() override {
    set_.clear();
    factors_.clear();
  }
  void clear_counts() override {}
  // takes the factor if it is one of ours; false otherwise (the caller keeps linearising it the ordinary way)
  bool add(std::shared_ptr<gtsam::NonlinearFactor> factor) override {
    auto f = std::dynamic_pointer_cast<IntegratedVGICPFactorHIP>(factor);
    if (!f) return false;
    set_.add(f->impl());
    factors_.push_back(std::move(f));
    return true;
  }
  void add(const gtsam::NonlinearFactorGraph& graph) override {
    for (const auto& factor : graph) add(factor);
  }
  void linearize(const gtsam::Values& linearization_point) override {
    if (factors_.empty()) return;
    set_.linearize(gather(linearization_point));
    for (const auto& f : factors_) f->store_linearized(linearization_point);
  }
  void error(const gtsam::Values& values) override {
    if (factors_.empty()) return;
    // GPU-factor semantics: frozen correspondences, possible only when every factor has a linearisation point and they are mutually consistent
    bool frozen = true;
    Values lin;
    for (const auto& f : factors_) {
      frozen = frozen && f->frozen_error() && f->has_linearization_point();  // (a CPU-named factor in the set: correspondences at `values` for all)
      if (!frozen) break;
      for (const auto& kv : f->linearization_values()) {
        auto it = lin.find(kv.first);
        if (it == lin.end()) lin[kv.first] = kv.second;
        else frozen = frozen && same_pose(it->second, kv.second);
      }
    }
    const std::vector<double> e = set_.error(gather(values), frozen ? &lin : nullptr);
    for (std::size_t i = 0; i < factors_.size(); i++) factors_[i]->store_error(values, GLIM_AMD_VGICP_ERROR_SCALE * e[i]);
  }
  std::vector<gtsam::GaussianFactor::shared_ptr> calc_linear_factors(const gtsam::Values& linearization_point) override {
    linearize(linearization_point);
    std::vector<gtsam::GaussianFactor::shared_ptr> out;
    for (const auto& f : factors_) out.push_back(f->linearize(linearization_point));
    return out;
  }

private:
  Values gather(const gtsam::Values& values) const {
    Values out;
    for (const auto& f : factors_)
      for (const gtsam::Key k : f->keys())
        if (!out.count((Key)k)) out[(Key)k] = to_iso(values.at<gtsam::Pose3>(k));
    return out;
  }
  NonlinearFactorSetGPU set_;
  std::vector<IntegratedVGICPFactorHIP::shared_ptr> factors_;
};

// gtsam_points::create_nonlinear_factor_set_gpu() and its registration (offline_viewer.cpp:29; live runs: glim_ros).  Call once, e.g.
// from the constructor of the module built as libodometry_estimation_hip.so.
inline std::shared_ptr<gtsam_points::NonlinearFactorSet> create_nonlinear_factor_set_hip() { return std::make_shared<NonlinearFactorSetHIP>(); }
inline void register_linearization_hook() { gtsam_points::LinearizationHook::register_hook([] { return create_nonlinear_factor_set_hip(); }); }

// ---- gtsam_points::overlap_gpu / overlap_auto with Eigen poses (odometry_estimation_gpu.cpp:231,248,265,279,326; global_mapping.cpp:322,448) ----
inline double overlap_gpu(const GaussianVoxelMapGPU::ConstPtr& target, const PointCloudGPU::ConstPtr& source, const Eigen::Isometry3d& delta) {
  return overlap_gpu(target, source, to_iso(delta));
}
inline double overlap_gpu(const std::vector<GaussianVoxelMapGPU::ConstPtr>& targets, const PointCloudGPU::ConstPtr& source,
                          const std::vector<Eigen::Isometry3d>& deltas) {
  std::vector<Isometry3d> d;
  for (const auto& T : deltas) d.push_back(to_iso(T));
  return overlap_gpu(targets, source, d);
}
inline double overlap_auto(const GaussianVoxelMapGPU::ConstPtr& target, const PointCloudGPU::ConstPtr& source, const Eigen::Isometry3d& delta) {
  return overlap_gpu(target, source, delta);
}

}  // namespace glim_amd
