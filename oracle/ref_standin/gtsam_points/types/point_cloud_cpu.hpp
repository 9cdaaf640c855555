// Stand-in for gtsam_points/types/point_cloud_cpu.hpp (+ the free functions of point_cloud_cpu_funcs) for oracle/_ref: what the reference's
// cloud_preprocessor.cpp touches.  TEST INFRASTRUCTURE.  PointCloudCPU, sample() and filter() are plain containers / selections; the three
// LIBRARY algorithms the reference calls -- randomgrid_sampling, voxelgrid_sampling, remove_outliers -- are NOT available (koide3/gtsam_points is
// not vendored) and are answered by the oracle's restatements (oracle/preprocess_oracle.c).  What compiling cloud_preprocessor.cpp against this
// pins is therefore the reference's IN-TREE logic: the order of the stages, the distance / cropbox predicates, the time sort, the global-shutter
// rule, the assembly of the PreprocessedFrame and the neighbour layout -- not the sampling algorithms.
#pragma once
#include <Eigen/Core>
#include <cstdint>
#include <memory>
#include <random>
#include <vector>

#include "../../../vgicp_oracle.h"

namespace gtsam_points {

struct RefSamplingContext {  // what the oracle's counter-based sampler needs in place of the std::mt19937 the reference hands over
  std::uint64_t seed = 0;
  int voxelgrid_block_size = 1024;
  static RefSamplingContext& get() {
    static RefSamplingContext c;
    return c;
  }
};

struct PointCloud {
  using Ptr = std::shared_ptr<PointCloud>;
  using ConstPtr = std::shared_ptr<const PointCloud>;
  virtual ~PointCloud() {}
  size_t num_points = 0;
  double* times = nullptr;
  Eigen::Vector4d* points = nullptr;
  double* intensities = nullptr;
  size_t size() const { return num_points; }
};

struct PointCloudCPU : public PointCloud {
  using Ptr = std::shared_ptr<PointCloudCPU>;
  using ConstPtr = std::shared_ptr<const PointCloudCPU>;
  std::vector<double> times_storage, intensities_storage;
  std::vector<Eigen::Vector4d> points_storage;
  void add_times(const std::vector<double>& t) {
    times_storage = t;
    times = times_storage.data();
    num_points = t.size();
  }
  void add_points(const std::vector<Eigen::Vector4d>& p) {
    points_storage = p;
    points = points_storage.data();
    num_points = p.size();
  }
  void add_intensities(const std::vector<double>& v) {
    intensities_storage = v;
    intensities = intensities_storage.data();
  }
};

// gtsam_points::sample(frame, indices): the selected points, in the order of `indices`
inline PointCloudCPU::Ptr sample(const PointCloud::ConstPtr& frame, const std::vector<int>& indices) {
  auto out = std::make_shared<PointCloudCPU>();
  std::vector<double> t(indices.size()), in(indices.size());
  std::vector<Eigen::Vector4d> p(indices.size());
  for (size_t i = 0; i < indices.size(); i++) {
    p[i] = frame->points[indices[i]];
    if (frame->times) t[i] = frame->times[indices[i]];
    if (frame->intensities) in[i] = frame->intensities[indices[i]];
  }
  out->add_points(p);
  if (frame->times) out->add_times(t);
  if (frame->intensities) out->add_intensities(in);
  out->num_points = indices.size();
  return out;
}

// gtsam_points::filter(frame, pred): the points for which pred(point) holds, order kept
template <class Pred>
PointCloudCPU::Ptr filter(const PointCloud::ConstPtr& frame, const Pred& pred) {
  std::vector<int> idx;
  for (size_t i = 0; i < frame->size(); i++)
    if (pred(frame->points[i])) idx.push_back((int)i);
  return sample(frame, idx);
}

inline std::vector<double> flat4(const PointCloud& f) {
  std::vector<double> p(4 * f.size() + 4);
  for (size_t i = 0; i < f.size(); i++)
    for (int k = 0; k < 4; k++) p[4 * i + k] = f.points[i][k];
  return p;
}

// ---- library algorithms, answered by the oracle's restatements (see the header comment) ----
inline PointCloudCPU::Ptr randomgrid_sampling(const PointCloud::ConstPtr& frame, double resolution, double rate, std::mt19937&, int) {
  const int n = (int)frame->size();
  std::vector<int32_t> idx((size_t)(n > 0 ? n : 1));
  const std::vector<double> p = flat4(*frame);
  const int m = n > 0 ? orc_randomgrid_sampling(p.data(), n, resolution, rate, RefSamplingContext::get().seed, idx.data()) : 0;
  return sample(frame, std::vector<int>(idx.begin(), idx.begin() + m));
}

inline PointCloudCPU::Ptr voxelgrid_sampling(const PointCloud::ConstPtr& frame, double resolution, int) {
  const int n = (int)frame->size();
  const std::vector<double> p = flat4(*frame);
  std::vector<double> op(4 * (size_t)(n > 0 ? n : 1)), ot((size_t)(n > 0 ? n : 1)), oi((size_t)(n > 0 ? n : 1));
  const int m = orc_voxelgrid_sampling(p.data(), frame->times, frame->intensities, n, resolution, RefSamplingContext::get().voxelgrid_block_size, op.data(),
                                       ot.data(), oi.data());
  auto out = std::make_shared<PointCloudCPU>();
  std::vector<Eigen::Vector4d> pts((size_t)m);
  for (int i = 0; i < m; i++) pts[(size_t)i] = Eigen::Vector4d(op[4 * (size_t)i], op[4 * (size_t)i + 1], op[4 * (size_t)i + 2], op[4 * (size_t)i + 3]);
  out->add_points(pts);
  out->add_times(std::vector<double>(ot.begin(), ot.begin() + m));
  if (frame->intensities) out->add_intensities(std::vector<double>(oi.begin(), oi.begin() + m));
  out->num_points = (size_t)m;
  return out;
}

inline PointCloudCPU::Ptr remove_outliers(const PointCloud::ConstPtr& frame, int k, double std_mul, int num_threads) {
  const int n = (int)frame->size();
  std::vector<int32_t> keep((size_t)(n > 0 ? n : 1));
  const std::vector<double> p = flat4(*frame);
  const int m = n > 0 ? orc_find_inliers(p.data(), n, k, std_mul, keep.data(), num_threads) : 0;
  return sample(frame, std::vector<int>(keep.begin(), keep.begin() + m));
}

}  // namespace gtsam_points
