// cloud_preprocessor_hip.cpp -- the HIP-backed twin of GLIM's src/glim/preprocess/cloud_preprocessor.cpp.
//
// Same header (include/glim/preprocess/cloud_preprocessor.hpp), same classes and members: a HIP build of libglim compiles THIS file in place of the
// reference's, and glim_ros / the offline tools that construct a glim::CloudPreprocessor get the device pipeline (preprocess.hip K9 + knn.hip K2)
// without an edit: downsampling, range filter, time order, global shutter, cropbox, outlier removal and the k nearest neighbours
// (cloud_preprocessor.cpp:92-188, :190-221) run back to back on the device with one synchronise, and the PreprocessedFrame the callers read is filled
// from one download.  What differs, on purpose:
//   * gtsam_points' random-grid sampler draws from the class's std::mt19937 member; the device sampler is counter-based and takes ONE 64-bit seed per
//     frame, drawn from that same member -- same scan + same generator state = same sample, on every run and every device;
//   * PreprocessCallbacks::on_raw_points_received is raised as in the reference; the three callbacks that hand out intermediate HOST clouds
//     (on_preprocessing_begin / on_downsampling_finished / on_filtering_finished) have no host cloud to hand out and are not raised.
#include <glim/preprocess/cloud_preprocessor.hpp>
#include <glim/preprocess/callbacks.hpp>

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include <glim/util/config.hpp>

#include <glim_amd/glim_preprocess_compat.hpp>

namespace glim {

// config/config_preprocess.json + config_sensors.json: the keys and defaults of cloud_preprocessor.cpp:20-61
CloudPreprocessorParams::CloudPreprocessorParams() {
  const Config pre(GlobalConfig::get_config_path("config_preprocess"));
  const Config sensors(GlobalConfig::get_config_path("config_sensors"));
  const std::string m = "preprocess";
  global_shutter = sensors.param<bool>("sensors", "global_shutter_lidar", false);
  struct { double* field; const char* key; double fallback; } const reals[] = {
    {&distance_near_thresh, "distance_near_thresh", 1.0}, {&distance_far_thresh, "distance_far_thresh", 100.0},
    {&downsample_resolution, "downsample_resolution", 0.15}, {&downsample_rate, "random_downsample_rate", 0.3},
    {&outlier_std_mul_factor, "outlier_std_mul_factor", 2.0}};
  for (const auto& r : reals) *r.field = pre.param<double>(m, r.key, r.fallback);
  struct { bool* field; const char* key; bool fallback; } const flags[] = {
    {&use_random_grid_downsampling, "use_random_grid_downsampling", false}, {&enable_outlier_removal, "enable_outlier_removal", false},
    {&enable_cropbox_filter, "enable_cropbox_filter", false}};
  for (const auto& f : flags) *f.field = pre.param<bool>(m, f.key, f.fallback);
  struct { int* field; const char* key; int fallback; } const ints[] = {
    {&downsample_target, "random_downsample_target", 0}, {&outlier_removal_k, "outlier_removal_k", 10}, {&k_correspondences, "k_correspondences", 8},
    {&num_threads, "num_threads", 2}};
  for (const auto& i : ints) *i.field = pre.param<int>(m, i.key, i.fallback);
  crop_bbox_frame = "lidar";
  crop_bbox_min.setZero();
  crop_bbox_max.setZero();
  if (enable_cropbox_filter) {
    T_imu_lidar = sensors.param<Eigen::Isometry3d>("sensors", "T_lidar_imu", Eigen::Isometry3d::Identity()).inverse();
    crop_bbox_frame = pre.param<std::string>(m, "crop_bbox_frame", "lidar");
    crop_bbox_min = pre.param<Eigen::Vector3d>(m, "crop_bbox_min", Eigen::Vector3d(0.0, 0.0, 0.0));
    crop_bbox_max = pre.param<Eigen::Vector3d>(m, "crop_bbox_max", Eigen::Vector3d(0.0, 0.0, 0.0));
    if (crop_bbox_frame != "lidar" && crop_bbox_frame != "imu") throw std::runtime_error("Unsupported crop bbox frame: " + crop_bbox_frame);
    for (int a = 0; a < 3; a++)
      if (crop_bbox_min[a] > crop_bbox_max[a]) throw std::runtime_error("Misconfigured bbox: min > max on axis " + std::to_string(a));
  }
}

CloudPreprocessorParams::~CloudPreprocessorParams() {}

CloudPreprocessor::CloudPreprocessor(const CloudPreprocessorParams& params) : params(params) {}  // (no task arena: the device has no thread knob)

CloudPreprocessor::~CloudPreprocessor() {}

PreprocessedFrame::Ptr CloudPreprocessor::preprocess(const RawPoints::ConstPtr& raw_points) {
  PreprocessCallbacks::on_raw_points_received(raw_points);
  return preprocess_impl(raw_points);
}

namespace {
glim_amd_preprocess_params device_params(const CloudPreprocessorParams& p, std::uint64_t seed) {
  if (p.crop_bbox_frame != "lidar" && p.crop_bbox_frame != "imu") throw std::runtime_error("Unsupported crop bbox frame: " + p.crop_bbox_frame);  // (:159)
  glim_amd_preprocess_params d;
  glim_amd::check(glim_amd_preprocess_default_params(&d), "preprocess_default_params");
  d.distance_near_thresh = p.distance_near_thresh;
  d.distance_far_thresh = p.distance_far_thresh;
  d.use_random_grid_downsampling = p.use_random_grid_downsampling;
  d.downsample_target = p.downsample_target;
  d.downsample_resolution = p.downsample_resolution;
  d.downsample_rate = p.downsample_rate;
  d.global_shutter = p.global_shutter;
  d.enable_outlier_removal = p.enable_outlier_removal;
  d.outlier_removal_k = p.outlier_removal_k;
  d.outlier_std_mul_factor = p.outlier_std_mul_factor;
  d.enable_cropbox_filter = p.enable_cropbox_filter;
  d.crop_bbox_frame_imu = p.crop_bbox_frame == "imu";
  for (int a = 0; a < 3; a++) {
    d.crop_bbox_min[a] = p.crop_bbox_min[a];
    d.crop_bbox_max[a] = p.crop_bbox_max[a];
  }
  if (p.enable_cropbox_filter) {
    const auto& T = p.T_imu_lidar.matrix();
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 4; c++) d.T_imu_lidar[4 * r + c] = T(r, c);
  }
  d.k_correspondences = p.k_correspondences;
  d.seed = seed;
  return d;
}
}  // namespace

PreprocessedFrame::Ptr CloudPreprocessor::preprocess_impl(const RawPoints::ConstPtr& raw_points) {
  static_assert(sizeof(Eigen::Vector4d) == 4 * sizeof(double), "dense fixed-size Eigen storage");
  static_assert(sizeof(int) == sizeof(std::int32_t), "neighbour indices are int32 on the device");
  const std::int64_t n = raw_points->size();
  if ((std::int64_t)raw_points->times.size() != n) throw std::runtime_error("CloudPreprocessor: times / points size mismatch");
  const bool has_intensities = !raw_points->intensities.empty();
  // one 64-bit seed per frame from the generator the reference's sampler would have drawn from (cloud_preprocessor.hpp:74)
  std::uint64_t seed = ((std::uint64_t)mt() << 32) | (std::uint64_t)mt();
  glim_amd_preprocess_params d = device_params(params, seed);
#ifdef GLIM_AMD_TEST_SAMPLER_HOOK
  GLIM_AMD_TEST_SAMPLER_HOOK(d);  // tests only: the side-by-side run with the compiled reference hands both samplers the same seed / block size
#endif
  glim_amd::Context ctx = glim_amd::StreamTempBufferRoundRobin::default_instance();
  glim_amd_cloud* cloud = nullptr;
  glim_amd::check(glim_amd_preprocess(ctx->context(), n, n ? reinterpret_cast<const double*>(raw_points->points.data()) : nullptr, raw_points->times.data(),
                                      has_intensities ? raw_points->intensities.data() : nullptr, &d, &cloud),
                  "CloudPreprocessor::preprocess");
  std::int64_t m = 0;
  (void)glim_amd_cloud_size(cloud, &m);
  auto frame = std::make_shared<PreprocessedFrame>();
  frame->stamp = raw_points->stamp;
  frame->times.resize((std::size_t)m);
  frame->points.resize((std::size_t)m);
  if (has_intensities) frame->intensities.resize((std::size_t)m);
  frame->k_neighbors = params.k_correspondences;
  frame->neighbors.resize((std::size_t)m * (std::size_t)params.k_correspondences);
  const int rc = glim_amd_cloud_download_frame(cloud, m ? reinterpret_cast<double*>(frame->points.data()) : nullptr, frame->times.data(),
                                               has_intensities ? frame->intensities.data() : nullptr,
                                               params.k_correspondences > 0 ? reinterpret_cast<std::int32_t*>(frame->neighbors.data()) : nullptr);
  (void)glim_amd_cloud_destroy(cloud);
  glim_amd::check(rc, "CloudPreprocessor::download");
  frame->scan_end_time = m ? raw_points->stamp + frame->times.back() : raw_points->stamp;  // (:171)
  frame->raw_points = raw_points;
  return frame;
}

// exact k nearest neighbours of every point, the point itself included, row-major per point (:190-221)
std::vector<int> CloudPreprocessor::find_neighbors(const Eigen::Vector4d* points, const int num_points, const int k) const {
  std::vector<int> neighbors((std::size_t)num_points * (std::size_t)k);
  if (num_points <= 0 || k <= 0) return neighbors;
  glim_amd::Context ctx = glim_amd::StreamTempBufferRoundRobin::default_instance();
  glim_amd_cloud* cloud = nullptr;
  glim_amd::check(glim_amd_cloud_create(ctx->context(), num_points, reinterpret_cast<const double*>(points), nullptr, nullptr, &cloud), "find_neighbors: upload");
  const int rc = glim_amd_cloud_find_neighbors(cloud, k, reinterpret_cast<std::int32_t*>(neighbors.data()));
  (void)glim_amd_cloud_destroy(cloud);
  glim_amd::check(rc, "CloudPreprocessor::find_neighbors");
  return neighbors;
}

}  // namespace glim
