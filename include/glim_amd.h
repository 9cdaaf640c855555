/*
 * glim_amd.h -- C ABI of the MI355X-native (gfx950) VGICP scan-matching hot path for GLIM.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch/Eigen types.  Each entry
 * point names the reference interface it replaces.  The reference (koide3/glim v1.2.2) reaches this path through
 * koide3/gtsam_points (un-vendored, CMakeLists.txt:28); the file:line citations are GLIM's call sites of those
 * gtsam_points symbols, relative to /root/reference.
 *
 * Conventions
 *   - return value: 0 = GLIM_AMD_OK, < 0 = error code (glim_amd_error_string).  Never throws, never aborts.
 *   - poses: 12 doubles, row-major 3x4 [R | t]  (Eigen::Isometry3d::matrix().topRows<3>()).
 *     T_target_source = T_target^-1 * T_source for binary factors, fixed_target_pose^-1 * T_source for unary ones.
 *   - tangent order [omega(3); v(3)] (gtsam::Pose3), right perturbation T (+) xi = T * Exp(xi).
 *   - host point layouts are the reference's: Eigen::Vector4d points / normals (stride 4 doubles), Eigen::Matrix4d
 *     covariances (16 doubles, column-major, zero last row/col)   include/glim/preprocess/preprocessed_frame.hpp:31,
 *     src/glim/common/cloud_covariance_estimation.cpp:96.  Host arrays are borrowed only for the duration of a call.
 *   - all device work of a context runs on its HIP stream(s); calls are synchronous unless named *_async.
 *   - handles are owned by the caller; destroy children (clouds, voxel maps, factor sets, search indices) before their context:
 *     glim_amd_ctx_destroy refuses (GLIM_AMD_ERR_STATE) while any child is alive.
 */
#ifndef GLIM_AMD_H
#define GLIM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLIM_AMD_VERSION 100 /* 0.1.0 */

enum {
  GLIM_AMD_OK = 0,
  GLIM_AMD_ERR_INVALID = -1,     /* bad argument */
  GLIM_AMD_ERR_HIP = -2,         /* HIP runtime failure (glim_amd_last_hip_error) */
  GLIM_AMD_ERR_NO_DEVICE = -3,   /* no gfx950 device visible */
  GLIM_AMD_ERR_RANGE = -4,       /* voxel coordinate outside the +-2^20 key range */
  GLIM_AMD_ERR_STATE = -5,       /* call not valid in this state (e.g. covariances requested before kNN) */
  GLIM_AMD_ERR_UNSUPPORTED = -6, /* not implemented in this build */
  GLIM_AMD_ERR_NOMEM = -7
};

typedef struct glim_amd_ctx glim_amd_ctx;                 /* device + stream pool (CUDAStream / StreamTempBufferRoundRobin) */
typedef struct glim_amd_cloud glim_amd_cloud;             /* gtsam_points::PointCloudGPU */
typedef struct glim_amd_voxelmap glim_amd_voxelmap;       /* gtsam_points::GaussianVoxelMapGPU */
typedef struct glim_amd_factor_set glim_amd_factor_set;   /* gtsam_points::NonlinearFactorSetGPU of IntegratedVGICPFactorGPU */

/* Result of one factor linearisation -- the FP64 image of gtsam_points' LinearizedSystem6.
 * H_* row-major 6x6.  gtsam::HessianFactor(k_t, k_s, H_tt, H_ts, -b_t, H_ss, -b_s, error)  (unary: k_s, H_ss, -b_s, error). */
typedef struct {
  int64_t num_inliers;
  double error;
  double H_tt[36];
  double H_ss[36];
  double H_ts[36];
  double b_t[6];
  double b_s[6];
} glim_amd_linearized6;

/* factor flags */
#define GLIM_AMD_FACTOR_BINARY 0x1             /* fill H_tt, H_ts, b_t (otherwise they are zero: unary factor) */
#define GLIM_AMD_FACTOR_SURFACE_VALIDATION 0x2 /* IntegratedVGICPFactorGPU::set_enable_surface_validation(true) */

/* ---- library / context --------------------------------------------------------------------------------------- */
int glim_amd_version(void);
const char* glim_amd_error_string(int code);
/* last HIP error text seen by this thread's failing call ("" if none). */
const char* glim_amd_last_hip_error(void);
/* number of visible HIP devices (0 on a CPU-only box; never fails). */
int glim_amd_device_count(void);

/* Replaces gtsam_points::CUDAStream + StreamTempBufferRoundRobin(num_streams)
 * (src/glim/odometry/odometry_estimation_gpu.cpp:76-77, src/glim/mapping/sub_mapping.cpp:86-87, global_mapping.cpp:110).
 * external_stream: a hipStream_t to run on (e.g. torch's current stream) or NULL to create `num_streams` streams.  The null stream is
 * named explicitly: pass hipStreamLegacy ((hipStream_t)1) or hipStreamPerThread ((hipStream_t)2), never 0. */
int glim_amd_ctx_create(int device, int num_streams, void* external_stream, glim_amd_ctx** out);
/* The same with a scheduling priority for the context's own streams: 0 = default, 1 = the device's greatest stream priority, -1 = its least.
 * GLIM runs three modules in three threads on one device, each with its own stream pool (async_odometry_estimation.cpp:15 /
 * odometry_estimation_gpu.cpp:76-77, async_sub_mapping.cpp:8 / sub_mapping.cpp:86-87, async_global_mapping.cpp:24 / global_mapping.cpp:110):
 * one context per module, the odometry's with priority 1, keeps a 25 us odometry linearisation from queueing behind a sub-mapping merge on
 * the host (a context's calls serialise on ITS mutex only) and behind a 10 ms global-mapping kernel on the device.
 * Clouds, voxel maps and search indices may be used by calls and factor sets of ANY context of the same device (the modules hand frames and
 * maps to one another); an object is destroyed through, and counted by, the context that created it. */
int glim_amd_ctx_create_ex(int device, int num_streams, void* external_stream, int priority, glim_amd_ctx** out);
/* GLIM_AMD_ERR_STATE (and the context stays valid) while clouds, voxel maps, factor sets or search indices created from it are alive. */
int glim_amd_ctx_destroy(glim_amd_ctx* ctx);
int glim_amd_ctx_synchronize(glim_amd_ctx* ctx);
/* gtsam_points::cuda_device_names / cuda_mem_get_info (src/glim/util/debug.cpp:84, viewer/memory_monitor.cpp:39). */
int glim_amd_device_info(glim_amd_ctx* ctx, char* name, size_t name_len, size_t* free_bytes, size_t* total_bytes, int* num_cus);

/* ---- point clouds: PointCloudGPU::clone (odometry_estimation_gpu.cpp:96, sub_mapping.cpp:168,393, global_mapping.cpp:253,260,743) */
/* points4: n x 4 doubles (required).  covs16: n x 16 doubles or NULL.  normals4: n x 4 doubles or NULL.
 * Device layout: FP32 SoA -- float4 xyz1, symmetric covariance as float4 (c00 c01 c02 c11) + float2 (c12 c22), float4 normals. */
int glim_amd_cloud_create(glim_amd_ctx* ctx, int64_t n, const double* points4, const double* covs16, const double* normals4,
                          glim_amd_cloud** out);
/* same from compact FP32 arrays: xyz n x 3, cov33 n x 9 (row-major, symmetric) or NULL, normals3 n x 3 or NULL. */
int glim_amd_cloud_create_f32(glim_amd_ctx* ctx, int64_t n, const float* xyz, const float* cov33, const float* normals3,
                              glim_amd_cloud** out);
int glim_amd_cloud_destroy(glim_amd_cloud* cloud);
int glim_amd_cloud_size(const glim_amd_cloud* cloud, int64_t* n);
/* device bytes held (IntegratedVGICPFactorGPU::memory_usage_gpu accounting, viewer/standard_viewer_mem.cpp:52-56). */
int glim_amd_cloud_memory_usage(const glim_amd_cloud* cloud, size_t* bytes);
/* copy back (parity / debug): any pointer may be NULL.  xyz n x 3, cov33 n x 9, normals3 n x 3, neighbors n x k. */
int glim_amd_cloud_download(const glim_amd_cloud* cloud, float* xyz, float* cov33, float* normals3, int32_t* neighbors);

/* gtsam_points::PointCloud::save_compact(dir) / PointCloudCPU::load(dir) + clone, as SubMap::save / SubMap::load use them for a
 * submap's merged cloud (src/glim/mapping/sub_map.cpp:62, :142): FP32 files points_compact.bin (n x xyz), covs_compact.bin
 * (n x c00 c01 c02 c11 c12 c22), normals_compact.bin, times_compact.bin, intensities_compact.bin inside `dir` (which must exist).
 * load also accepts the full-precision points.bin / covs.bin / normals.bin (Vector4d / Matrix4d) when no compact files exist. */
int glim_amd_cloud_save_compact(const glim_amd_cloud* cloud, const char* dir);
int glim_amd_cloud_load_compact(glim_amd_ctx* ctx, const char* dir, glim_amd_cloud** out);

/* CloudDeskewing::deskew fused with the upload: src/glim/common/cloud_deskewing.cpp:11-53 (constant
 * velocity: n_imu == 0, linear_vel3 / angular_vel3, NULL = zero) and :55-133 (IMU poses: imu_times[n_imu], imu_poses12[n_imu x 12]
 * = T_world_imu row-major 3x4, `stamp` = scan start time), as called at src/glim/odometry/odometry_estimation_imu.cpp:313-316.
 * points4: n x Vector4d, times: n per-point offsets from the scan start (as the preprocessor leaves them: ascending), T_imu_lidar12:
 * extrinsic.
 * to_imu_frame != 0 fuses the step BOTH reference callers take next: every deskewed point is moved into the IMU frame,
 * `pt = T_imu_lidar * pt` (odometry_estimation_imu.cpp:314-316; sub_mapping.cpp:368-370 with T_lidar_imu.inverse()), as a second FP64
 * product with its own roundings, BEFORE the covariances are estimated (:320 / :374) -- so normals face the IMU-frame origin
 * (cloud_covariance_estimation.cpp:98-101) and every voxel map / factor built from the cloud lives in the IMU frame, as in GLIM.
 * to_imu_frame == 0 returns exactly CloudDeskewing::deskew's value (LiDAR frame).
 * The result is a device cloud of the deskewed points -- the exact FP64 values (what glim_amd_cloud_estimate_covariances reads and
 * glim_amd_cloud_download_frame returns) and their FP32 image for the factor path; no covariances yet. */
/* PointCloudGPU::clone of points only that KEEPS the exact FP64 values beside their FP32 image (what a preprocessed / deskewed cloud does): for
 * points that are not FP32-representable -- deskewed, IMU-frame points handed to CloudCovarianceEstimation::estimate by a caller that holds them on
 * the host (src/glim/odometry/odometry_estimation_imu.cpp:320; adapters/glim/cloud_covariance_estimation_hip.cpp) -- so that
 * glim_amd_cloud_estimate_covariances reads what the reference reads (the 1e-5 covariance gate is then met on every point). */
int glim_amd_cloud_create_exact(glim_amd_ctx* ctx, int64_t n, const double* points4, glim_amd_cloud** out);
int glim_amd_cloud_create_deskewed(glim_amd_ctx* ctx, int64_t n, const double* points4, const double* times, const double* T_imu_lidar12,
                                   int32_t n_imu, const double* imu_times, const double* imu_poses12, double stamp, const double* linear_vel3,
                                   const double* angular_vel3, int32_t to_imu_frame, glim_amd_cloud** out);

/* ---- scan preprocessing on device: CloudPreprocessor::preprocess_impl,
 * src/glim/preprocess/cloud_preprocessor.cpp:92-188 -- downsampling (gtsam_points::randomgrid_sampling / voxelgrid_sampling,
 * :104-109), range filter (:117-128), sort by time (:134-136), global shutter (:138-140), cropbox (:143-160), statistical outlier
 * removal (:162-164) and the kNN for the covariances (:183-184), without a host round trip between the stages. */
typedef struct glim_amd_preprocess_params { /* CloudPreprocessorParams, cloud_preprocessor.cpp:20-61; config/config_preprocess.json */
  double distance_near_thresh, distance_far_thresh;
  int32_t use_random_grid_downsampling;
  int32_t downsample_target; /* random_downsample_target: > 0 -> rate = target / n (:105) */
  double downsample_resolution, downsample_rate;
  int32_t global_shutter; /* config_sensors global_shutter_lidar (:24) */
  int32_t enable_outlier_removal, outlier_removal_k;
  double outlier_std_mul_factor;
  int32_t enable_cropbox_filter, crop_bbox_frame_imu; /* crop_bbox_frame == "imu" */
  double crop_bbox_min[3], crop_bbox_max[3];
  double T_imu_lidar[12]; /* row-major 3x4, used by the "imu" cropbox only */
  int32_t k_correspondences;
  int32_t voxelgrid_block_size; /* gtsam_points averages voxels inside blocks of 1024 sorted points; 0 = never split a voxel */
  uint64_t seed;                /* the reference draws from a std::mt19937 (:67); here: seed of the counter-based sampler */
} glim_amd_preprocess_params;
/* shipped defaults (config/config_preprocess.json) */
int glim_amd_preprocess_default_params(glim_amd_preprocess_params* params);
/* points4: n x Vector4d (RawPoints::points), times: n (RawPoints::times), intensities: n or NULL.  The result is a device
 * cloud holding the surviving points (FP32 for the factor path + the exact FP64 values), their times, intensities and -- when
 * k_correspondences > 0 -- their k nearest neighbours: everything PreprocessedFrame carries
 * (include/glim/preprocess/preprocessed_frame.hpp:26-39). */
int glim_amd_preprocess(glim_amd_ctx* ctx, int64_t n, const double* points4, const double* times, const double* intensities,
                        const glim_amd_preprocess_params* params, glim_amd_cloud** out);
/* PreprocessedFrame fields of a preprocessed cloud back on the host; any pointer may be NULL.  points4 n x 4 (w = 1), times n,
 * intensities n (GLIM_AMD_ERR_STATE if the scan had none), neighbors n x k. */
int glim_amd_cloud_download_frame(const glim_amd_cloud* cloud, double* points4, double* times, double* intensities, int32_t* neighbors);
/* CloudDeskewing::deskew (+ the IMU-frame step when to_imu_frame != 0) applied to a preprocessed cloud that is already on the device
 * (same arguments as glim_amd_cloud_create_deskewed).  The new cloud shares nothing with `pre`; the neighbour lists found on the raw scan
 * are carried over, as the reference does (odometry_estimation_imu.cpp:313-320: deskew, IMU frame, then covariances from raw_frame->neighbors). */
int glim_amd_cloud_deskew(const glim_amd_cloud* pre, const double* T_imu_lidar12, int32_t n_imu, const double* imu_times,
                          const double* imu_poses12, double stamp, const double* linear_vel3, const double* angular_vel3, int32_t to_imu_frame,
                          glim_amd_cloud** out);
/* ---- GICP factor on device: gtsam_points::IntegratedGICPFactor -- nearest-neighbour correspondences instead
 * of a voxel lookup -- as constructed at src/glim/mapping/sub_mapping.cpp:202 (between factors, one linearize, :203),
 * src/glim/mapping/global_mapping.cpp:400-402 (set_max_correspondence_distance(0.5), 10 LM iterations) and
 * src/glim/mapping/global_mapping_pose_graph.cpp:393-394 (loop validation; passes the target's pre-built tree). */
typedef struct glim_amd_nn_index glim_amd_nn_index; /* the target's search structure (gtsam_points::KdTree at global_mapping_pose_graph.cpp:393) */
/* Built once per target cloud (which needs covariances for the factor calls and must outlive the index), reused by every
 * linearisation.  max_correspondence_distance_hint sizes the grid cells (hint/3 .. hint wide); the calls below accept any distance up
 * to 21 x hint and return GLIM_AMD_ERR_UNSUPPORTED beyond that (the ring walk is bounded at 64 cells: rebuild with a larger hint). */
int glim_amd_nn_index_create(const glim_amd_cloud* target, double max_correspondence_distance_hint, glim_amd_nn_index** out);
int glim_amd_nn_index_destroy(glim_amd_nn_index* index);
/* IntegratedGICPFactor::linearize at T_target_source (flags: GLIM_AMD_FACTOR_BINARY fills the target-side blocks). */
int glim_amd_gicp_linearize(const glim_amd_nn_index* target, const glim_amd_cloud* source, const double* T_target_source12,
                            double max_correspondence_distance, uint32_t flags, glim_amd_linearized6* out);
/* IntegratedGICPFactor::error and ::inlier_fraction (num_inliers / source size) -- global_mapping_pose_graph.cpp:404-405. */
int glim_amd_gicp_error(const glim_amd_nn_index* target, const glim_amd_cloud* source, const double* T_target_source12,
                        double max_correspondence_distance, double* error, int64_t* num_inliers);
/* parity / debug: matched target index per source point, or -1. */
int glim_amd_gicp_correspondences(const glim_amd_nn_index* target, const glim_amd_cloud* source, const double* T_target_source12,
                                  double max_correspondence_distance, int32_t* correspondences);

/* ---- submap merge on device: gtsam_points::merge_frames(poses, frames, downsample_resolution, target_num_points)
 * as called at src/glim/mapping/sub_mapping.cpp:480-497 (the reference's own GPU variant, merge_frames_gpu, is commented out at :491).
 * Frame f (sizes[f] points: points4[f] n x Vector4d, covs16[f] n x column-major Matrix4d) is moved by poses12[f] (row-major 3x4
 * T_origin_frame): p' = T p, C' = R C R^T; the concatenation is voxel-grid averaged (points and covariances) at
 * `downsample_resolution`; when target_num_points > 0 and more points remain, a uniform random sample of that size is kept
 * (same meaning as sub_mapping_passthrough.cpp:149-151).  The result is a device cloud with covariances, ready for
 * GaussianVoxelMapGPU::insert / the factors (global_mapping.cpp:253-266); voxelgrid_block_size: see glim_amd_preprocess_params. */
int glim_amd_merge_frames(glim_amd_ctx* ctx, int32_t num_frames, const double* poses12, const double* const* points4, const double* const* covs16,
                          const int64_t* sizes, double downsample_resolution, int32_t target_num_points, int32_t voxelgrid_block_size, uint64_t seed,
                          glim_amd_cloud** out);
/* the merged submap back on the host as gtsam_points::PointCloudCPU holds it: points4 n x Vector4d, covs16 n x Matrix4d (exact FP64). */
int glim_amd_cloud_download_merged(const glim_amd_cloud* cloud, double* points4, double* covs16);

/* kNN on device: CloudPreprocessor::find_neighbors (src/glim/preprocess/cloud_preprocessor.cpp:190-221).
 * k nearest among all points including the query itself, ascending (distance, index); fewer than k points -> the tail is 0 (what the
 * reference's zero-initialised result vector holds there, cloud_preprocessor.cpp:193, :200).
 * Result stays on the device inside the cloud (and is copied to neighbors_out, n x k, when not NULL). */
int glim_amd_cloud_find_neighbors(glim_amd_cloud* cloud, int k, int32_t* neighbors_out);
/* upload caller-provided neighbours (n x k) instead; GLIM_AMD_ERR_INVALID if any index is outside [0, n). */
int glim_amd_cloud_set_neighbors(glim_amd_cloud* cloud, int k, const int32_t* neighbors);
/* CloudCovarianceEstimation::estimate (src/glim/common/cloud_covariance_estimation.cpp:43-122, PLANE regularisation :181-196):
 * fills the cloud's covariances and sensor-facing normals from the first k_neighbors stored neighbours. */
int glim_amd_cloud_estimate_covariances(glim_amd_cloud* cloud, int k_neighbors);

/* ---- Gaussian voxel maps: GaussianVoxelMapGPU(resolution, ...) + insert(frame)
 *      (odometry_estimation_gpu.cpp:103-104, sub_mapping.cpp:398-399, global_mapping.cpp:265-266,747-748) ------------ */
/* The reference's init_num_buckets / max_bucket_scan_count / target_points_drop_rate are accepted for signature
 * compatibility and ignored: this build is lossless (every point is inserted). */
int glim_amd_voxelmap_create(glim_amd_ctx* ctx, double resolution, int init_num_buckets, int max_bucket_scan_count,
                             double target_points_drop_rate, glim_amd_voxelmap** out);
/* build from a cloud that has covariances.  Voxel = mean of member means, mean of member covariances.  A second insert into the same map
 * adds its points to the voxels already there (GaussianVoxelMapCPU semantics: odometry_estimation_cpu.cpp:66-67,189); the map is rebuilt, so
 * GLIM's GPU callers, which insert once per map, pay nothing for it.
 * DEVIATION (unverified, gtsam_points is not under the reference tree): upstream's GaussianVoxelMapGPU::insert is believed to REBUILD the table
 * from the new frame only.  Every GPU call site of the reference inserts exactly once per map (odometry_estimation_gpu.cpp:103-104,
 * sub_mapping.cpp:398-399, global_mapping.cpp:265-266,747-748), so both semantics give the same map there; a port that re-inserts into a GPU map
 * and wants upstream's behaviour creates a new map per frame instead.  The re-opened voxels pass through their FP32 records (mean * n,
 * cov * n), i.e. each further insert adds a rounding of 1 FP32 ulp to the old voxels' statistics
 * (tests/test_gpu_parity.py::test_voxelmap_incremental_insert_matches_oracle bounds it against the FP64 oracle's merged map). */
int glim_amd_voxelmap_insert(glim_amd_voxelmap* vmap, const glim_amd_cloud* cloud);
/* create_frame of the GPU odometry as ONE submission (src/glim/odometry/odometry_estimation_gpu.cpp:86-107): PointCloudGPU::clone of a frame that
 * arrives with CPU covariances (+ normals) and GaussianVoxelMapGPU(resolutions[lv]).insert(frame) for lv < num_levels (<= 8; voxelmap_levels, 2 in the
 * shipped configuration) are enqueued back to back and the host synchronises once.  Same arguments as glim_amd_cloud_create; the results are what the
 * separate calls return (same bits), *cloud and maps[lv] are owned by the caller.  On failure nothing is created. */
int glim_amd_frame_create(glim_amd_ctx* ctx, int64_t n, const double* points4, const double* covs16, const double* normals4, int32_t num_levels,
                          const double* resolutions, glim_amd_cloud** cloud, glim_amd_voxelmap** maps);
int glim_amd_voxelmap_destroy(glim_amd_voxelmap* vmap);
/* GaussianVoxelMapCPU::set_lru_horizon of the CPU odometry's incremental target map (src/glim/odometry/odometry_estimation_cpu.cpp:63-68,
 * update_target :177-191; config_odometry_cpu.json "lru_thresh": 100): every insert carries a counter; every `lru_clear_cycle` inserts (<= 0: 10,
 * gtsam_points' default) the voxels that no insert has touched for more than `lru_horizon` inserts are dropped.  lru_horizon <= 0 (the default):
 * no eviction -- the GPU callers of the reference never set one.  Applies to the inserts that follow. */
int glim_amd_voxelmap_set_lru_horizon(glim_amd_voxelmap* vmap, int32_t lru_horizon, int32_t lru_clear_cycle);
/* VoxelMapInfo (standard_viewer_mem.cpp:76-77): num_voxels, num_buckets, resolution, device bytes. */
int glim_amd_voxelmap_info(const glim_amd_voxelmap* vmap, int32_t* num_voxels, int32_t* num_buckets, double* resolution,
                           size_t* bytes);
/* copy back all voxels (unspecified order): coords V x 3, counts V, means V x 3, cov33 V x 9.  Any may be NULL. */
int glim_amd_voxelmap_download(const glim_amd_voxelmap* vmap, int32_t* coords, int32_t* counts, float* means, float* cov33);

/* ---- factor sets: NonlinearFactorSetGPU of IntegratedVGICPFactorGPU
 *      (odometry_estimation_gpu.cpp:144,161,383-386; sub_mapping.cpp:307; global_mapping.cpp:335,466,860) ------------- */
int glim_amd_factor_set_create(glim_amd_ctx* ctx, glim_amd_factor_set** out);
int glim_amd_factor_set_destroy(glim_amd_factor_set* set);
/* add IntegratedVGICPFactorGPU(target voxel map, source cloud).  The set borrows both handles; keep them alive. */
int glim_amd_factor_set_add(glim_amd_factor_set* set, const glim_amd_voxelmap* target, const glim_amd_cloud* source, uint32_t flags,
                            int32_t* factor_index);
int glim_amd_factor_set_clear(glim_amd_factor_set* set);
int glim_amd_factor_set_size(const glim_amd_factor_set* set, int32_t* n);
/* NonlinearFactorSetGPU::linearize: one fused launch over every factor (+ a tiny FP64 finalise), one upload of the
 * poses, one download of the results.  T_target_source: n x 12.  out: n records. */
int glim_amd_factor_set_linearize(glim_amd_factor_set* set, const double* T_target_source, glim_amd_linearized6* out);
/* NonlinearFactorSetGPU::error.  T_lin == NULL: correspondences recomputed at T_eval (CPU-factor semantics, the parity
 * default); otherwise correspondences and Mahalanobis matrices frozen at T_lin (GPU-factor semantics).  errors: n; inliers: n or NULL. */
int glim_amd_factor_set_error(glim_amd_factor_set* set, const double* T_lin, const double* T_eval, double* errors, int64_t* inliers);
/* correspondences of one factor at a pose (parity/debug): corr n_points x 4 int32 = {cx, cy, cz, hit ? 1 : -1}. */
int glim_amd_factor_set_correspondences(glim_amd_factor_set* set, int32_t factor_index, const double* T_target_source, int32_t* corr);

/* Device-resident variant for multi-GPU cost evaluation and benchmarking: results are left on the device in
 * `out_device` (n x GLIM_AMD_COMPACT_DOUBLES doubles, caller-owned device memory, e.g. a torch tensor) without a host
 * round trip; the launch is asynchronous on the context stream.  Compact record: [num_inliers, error, 21 upper-triangular
 * entries of H_ss (row-major), 6 of b_s]; glim_amd_expand_compact turns records into glim_amd_linearized6 on the host. */
#define GLIM_AMD_COMPACT_DOUBLES 29
int glim_amd_factor_set_linearize_device_async(glim_amd_factor_set* set, const double* T_target_source_host, double* out_device,
                                               int64_t out_row_offset);
int glim_amd_expand_compact(const double* compact, const double* T_target_source, uint32_t flags, glim_amd_linearized6* out);

/* ---- multi-device cost evaluation (BASELINE.json configs[3]; no counterpart in the reference, which is single-device:
 *      src/glim/mapping/global_mapping.cpp:110 one StreamTempBufferRoundRobin(64), :430-484 create_matching_cost_factors) -------------
 * One process, N devices: a context + a host thread + an RCCL communicator (ncclCommInitAll; librccl is dlopen'ed on first use) per
 * device (device 0 is driven by the CALLING thread).  Clouds and voxel maps are replicated on every device, the factor list is sharded into
 * contiguous cost-balanced chunks, every device linearises its chunk -- in a few pieces, so that the ncclAllGather of one piece's 29-double
 * compact records travels over xGMI (and this device's own rows to the host over its own PCIe link) while the next piece's kernels run --
 * and the records are expanded on the host in the original factor order.  One hand-over to the devices' threads per evaluation.  All
 * calls are synchronous and must come from one host thread at a time.
 * Who reads what: the HOST optimiser (GLIM's ISAM2 / LM, global_mapping.cpp:501) gets every record through a second store of the finalising
 * kernels into one pinned host array, so a call returns when the kernels are done; the all-gather completes the DEVICE-resident copy of the
 * record array on every device for device-side consumers (glim_amd_multi_gathered_device) and finishes behind the call
 * (glim_amd_multi_set_gather_mode). */
typedef struct glim_amd_multi glim_amd_multi;
/* devices: distinct HIP device ordinals.  A multi-device handle without a working RCCL is refused (GLIM_AMD_ERR_HIP) rather than
 * silently gathering over PCIe; a single device works either way (GLIM_AMD_DIAG="multi_rccl=0" skips the collective there).  (Test boxes with one
 * GPU run the N > 1 path over "virtual devices" -- one ordinal listed several times -- only when GLIM_AMD_DIAG holds multi_virtual=1:
 * glim_amd_diag.h.) */
int glim_amd_multi_create(const int32_t* devices, int32_t num_devices, glim_amd_multi** out);
int glim_amd_multi_destroy(glim_amd_multi* multi);
int glim_amd_multi_info(const glim_amd_multi* multi, int32_t* num_devices, int32_t* uses_rccl, int64_t* num_factors);
/* replicated PointCloudGPU::clone (same arguments as glim_amd_cloud_create / _f32); cloud_id indexes the replicas */
int glim_amd_multi_add_cloud(glim_amd_multi* multi, int64_t n, const double* points4, const double* covs16, const double* normals4, int32_t* cloud_id);
int glim_amd_multi_add_cloud_f32(glim_amd_multi* multi, int64_t n, const float* xyz, const float* cov33, const float* normals3, int32_t* cloud_id);
/* kNN (k) + CloudCovarianceEstimation on every replica (deterministic kernels: the replicas stay bit-identical) */
int glim_amd_multi_cloud_estimate_covariances(glim_amd_multi* multi, int32_t cloud_id, int k);
/* replicated GaussianVoxelMapGPU(resolution).insert(cloud) */
int glim_amd_multi_add_voxelmap(glim_amd_multi* multi, int32_t cloud_id, double resolution, int32_t* map_id);
/* the factor list: factor f = IntegratedVGICPFactorGPU(maps[target_map_ids[f]], clouds[source_cloud_ids[f]]), flags[f] (NULL = unary);
 * replaces the previous list and shards it over the devices (cost of a factor = its source points). */
int glim_amd_multi_set_factors(glim_amd_multi* multi, int64_t num_factors, const int32_t* target_map_ids, const int32_t* source_cloud_ids,
                               const uint32_t* flags);
/* device d owns factors [bounds[d], bounds[d + 1]); bounds has num_devices + 1 entries */
int glim_amd_multi_shard(const glim_amd_multi* multi, int64_t* bounds);
/* H / b / error of every factor at T_target_source (n x 12); out (n records) and total_error may be NULL.  With out == NULL the compact
 * records stay in the handle's pinned host array (glim_amd_multi_records) and total_error is summed by the devices. */
int glim_amd_multi_linearize(glim_amd_multi* multi, const double* T_target_source, glim_amd_linearized6* out, double* total_error);
/* the last evaluation's compact 29-double records [num_inliers, error, 21 upper-triangular H_ss entries, 6 b_s] of factors
 * [first, first + count), in factor order (expand one with glim_amd_expand_compact) */
int glim_amd_multi_records(const glim_amd_multi* multi, int64_t first, int64_t count, double* compact29);
/* The exchange of an evaluation: 1 (default) = enqueued behind every piece, NOT waited for by glim_amd_multi_linearize (the next evaluation's
 * kernels wait for it on the device before they overwrite their send slots; glim_amd_multi_gathered_device / _wait_gather wait on the host);
 * 2 = the call returns only when every device holds every record (the form of versions <= 0.1.0 r5); 0 = no exchange at all (host records only). */
int glim_amd_multi_set_gather_mode(glim_amd_multi* multi, int32_t mode);
/* host wait for the last evaluation's exchange on every device (no-op when none is pending) */
int glim_amd_multi_wait_gather(glim_amd_multi* multi);
/* The consumer side of the all-gather: the device-resident record array of device `device` (an index into the handle's device list), complete on
 * return -- *gathered points at num_devices x max_rows records of GLIM_AMD_COMPACT_DOUBLES doubles in DEVICE memory (valid until the next
 * glim_amd_multi_set_factors / destroy; overwritten by the next evaluation), *rows (may be NULL) = num_devices x max_rows; factor f of the list sits at
 * the row glim_amd_shard_layout gives for it.  An on-device consumer (cost, gradient or a whole optimiser step built from every factor's blocks)
 * reads it from any stream of that device without a host trip. */
int glim_amd_multi_gathered_device(glim_amd_multi* multi, int32_t device, const double** gathered, int64_t* rows);
/* the sharding rule as a pure host function (no device needed): contiguous chunks whose cumulative cost is nearest to r / world of the total */
int glim_amd_shard_bounds(const double* costs, int64_t n, int32_t world, int64_t* bounds);
/* where every factor's 29-double record sits in the gathered [world x max_rows] array of an evaluation, as a pure host function (the rule
 * glim_amd_multi_linearize, its collectives and glim_amd_multi_records use): bounds from glim_amd_shard_bounds (world + 1 entries),
 * split_mode as glim_amd_multi_set_split; rows (bounds[world] entries, may be NULL) receives each factor's row, the other outputs (may be
 * NULL) the padded shard length, the number of pieces a shard is cut into and the rows of a full piece.  Piece p of every shard forms one
 * contiguous region of world equal slots -- one in-place ncclAllGather each. */
int glim_amd_shard_layout(const int64_t* bounds, int32_t world, int32_t split_mode, int64_t* rows, int64_t* max_rows, int32_t* pieces, int64_t* piece_rows);

/* ---- overlap: overlap_gpu / overlap_auto (odometry_estimation_gpu.cpp:231,248,265,279,326; sub_mapping.cpp:252-253;
 *      global_mapping.cpp:322,448).  Fraction of source points that hit an occupied voxel of ANY target under its delta. */
int glim_amd_overlap(glim_amd_ctx* ctx, int32_t num_targets, const glim_amd_voxelmap* const* targets, const double* T_target_source,
                     const glim_amd_cloud* source, double* overlap);
/* num_queries overlap_gpu calls answered by ONE launch -- the keyframe loops of odometry_estimation_gpu.cpp:262-281 issue up to
 * max_num_keyframes of them back to back.  Query q: num_targets[q] (map, delta) pairs, stored consecutively in `targets` /
 * `T_target_source` (12 doubles each) in query order, against sources[q]; overlaps[q] receives the fraction.  At most 1024 queries. */
int glim_amd_overlap_batch(glim_amd_ctx* ctx, int32_t num_queries, const int32_t* num_targets, const glim_amd_voxelmap* const* targets,
                           const double* T_target_source, const glim_amd_cloud* const* sources, double* overlaps);

#ifdef __cplusplus
}
#endif
#endif /* GLIM_AMD_H */
