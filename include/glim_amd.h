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
/* Diagnostic / tuning switches of a context (no counterpart in the reference; none is needed in production).  key_values:
 * "key=value,key=value"; NULL or "" restores the process defaults, which come from the ONE environment variable the library reads,
 * GLIM_AMD_DIAG (same syntax, parsed once per process).  Keys: knn_path=auto|grid|chunks|brute, knn_kernel=auto|wave64|pair|qgroup,
 * knn_select=0|1, plane=0|1, curve_order=0|1, ppt=<n>, poll=0|1, inline_pose=0|1, bucket_factor=<n>, plan_cache=0|1, plan_recycle=0|1, host_poses=0|1, host_pack=0|1, pull_gated=0|1, frame_fused=0|1,
 * view_fused=0|1 (a voxel map built from a plane-form cloud gets its plane view -- the (C_B + I)^-1 records the plane-form factor kernel reads --
 * from the map's own finalise kernel; 0: on the first factor that needs it),
 * fuse=0|1 (small synchronous sets in ONE dispatch), resident=0|1|auto + resident_idle_us=<n> (repeated synchronous linearisations of a small set
 * served by a resident kernel that leaves after <n> us without a request; auto, the default: only in a context created with priority 1 -- the
 * session costs whatever else runs on the device 1.3-1.4x while it is alive, so it is opt-in), pp_fast=0|1 (random-grid preprocessing without sorts),
 * knn_debug=<file>; and, in GLIM_AMD_DIAG ONLY (they are process-wide: set_diag refuses them), pool=0|1, multi_rccl=0|1,
 * multi_host_gather=0|1.  Unknown keys / bad values: GLIM_AMD_ERR_INVALID and nothing changes.  get_diag prints the current state in the
 * same syntax. */
int glim_amd_ctx_set_diag(glim_amd_ctx* ctx, const char* key_values);
int glim_amd_ctx_get_diag(glim_amd_ctx* ctx, char* buf, size_t len);
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
/* parity / debug only (host arithmetic, no device needed): the time table CloudDeskewing::deskew builds (cloud_deskewing.cpp:22-45 / :70-124) --
 * entry_out[i] = table entry of point i (n, may be NULL), table12_out = one row-major 3x4 T_lidar0_lidar1 per entry (table_cap entries, may
 * be NULL), *table_size = number of entries.  The deskewing kernels gather from exactly this table. */
int glim_amd_debug_deskew_table(int64_t n, const double* times, const double* T_imu_lidar12, int32_t n_imu, const double* imu_times, const double* imu_poses12,
                                double stamp, const double* linear_vel3, const double* angular_vel3, int32_t* entry_out, double* table12_out, int32_t table_cap,
                                int32_t* table_size);
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
/* parity / debug only: the stable device radix sort behind the preprocessing (sorts by the low `bits` key bits; vals_in NULL = 0..n-1). */
int glim_amd_debug_sort_pairs(glim_amd_ctx* ctx, int64_t n, int32_t bits, const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out,
                              uint32_t* vals_out);

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

/* Timing aid used by bench.py: runs `iters` back-to-back launches bracketed by HIP events on the set's stream.
 * ms_vgicp_kernel: average duration of the fused lookup+residual+Jacobian+reduce kernel alone;
 * ms_linearize: average duration of the whole device-resident linearise (kernel + finalise). */
int glim_amd_factor_set_profile(glim_amd_factor_set* set, const double* T_target_source, int iters, float* ms_vgicp_kernel,
                                float* ms_linearize);

/* wall-clock milliseconds per synchronous glim_amd_factor_set_linearize call (pose upload, launches, result in host memory),
 * measured inside the library so that no binding overhead is included. */
int glim_amd_factor_set_profile_sync(glim_amd_factor_set* set, const double* T_target_source, int iters, float* ms_per_call);
/* timing aid: EXACTLY `iters` synchronous glim_amd_factor_set_linearize calls from C, no warm-up, no clock -- the caller times it.  Call i
 * linearises at pose set i % num_pose_sets of T_target_source (num_pose_sets x n x 12: an optimiser moves the poses between its
 * relinearisations); out_last (n records, may be NULL) receives the last call's result. */
int glim_amd_factor_set_linearize_repeat(glim_amd_factor_set* set, const double* T_target_source, int num_pose_sets, int iters,
                                         glim_amd_linearized6* out_last);

/* GLIM's live call pattern (odometry_estimation_gpu.cpp:383-385; the optimisers' linearisation hook does clear -> add(graph) -> linearize per
 * iteration): a FRESH factor set per linearisation -- create, add the n factors, synchronous linearize (poses T: n x 12), destroy -- `iters`
 * times; microseconds per iteration, measured inside the library. */
int glim_amd_factor_set_profile_fresh(glim_amd_ctx* ctx, int32_t n, const glim_amd_voxelmap* const* targets, const glim_amd_cloud* const* sources,
                                      const uint32_t* flags, const double* T_target_source, int iters, float* us_per_iteration);

/* Measurement aid: wavefront trips of the general (36 B/pt) factor kernel that found no correspondence in any lane and skipped the record gather
 * and the algebra, summed over every evaluation of the set's current plan since the last reset, and the trips ONE evaluation of the plan makes
 * (blocks x 4 wavefronts x points per thread).  bench.py prices the kernel's instruction floor with the measured share instead of a constant. */
int glim_amd_factor_set_trip_stats(glim_amd_factor_set* set, uint64_t* skipped_trips, uint64_t* total_trips_per_evaluation, int reset);

/* The same pattern with every iteration timed on its own (samples_us: `iters` entries) and `gap_us` of host busy-waiting between iterations -- the
 * optimiser's own work between two linearisations --, for latency percentiles while other threads load the device (bench.py
 * --workload odometry_under_load). */
int glim_amd_factor_set_profile_fresh_samples(glim_amd_ctx* ctx, int32_t n, const glim_amd_voxelmap* const* targets, const glim_amd_cloud* const* sources,
                                              const uint32_t* flags, const double* T_target_source, int iters, double gap_us, float* samples_us);

/* One Levenberg-Marquardt iteration as the optimisers drive it (sub_mapping.cpp:435-443, odometry_estimation_cpu.cpp:116-149): a synchronous
 * linearize() of the whole set (records expanded on the host) and a synchronous error() at the trial values, each timed over `iters` calls. */
int glim_amd_factor_set_profile_lm(glim_amd_factor_set* set, const double* T_target_source, int iters, float* ms_linearize, float* ms_error);

/* parity / debug only: the device's resident session (repeated synchronous linearisations of a small factor list are served by a kernel that stays
 * on the device and takes its requests through host-mapped memory; it leaves by itself after `resident_idle_us` without a request): kernel
 * launches and requests served so far, whether one is alive right now. */
int glim_amd_debug_resident_stats(int device, uint64_t* launches, uint64_t* requests, int32_t* alive);
/* ends the device's resident session now instead of letting it idle out (GLIM_AMD_ERR_STATE while a request is in flight). */
int glim_amd_debug_resident_stop(int device);
/* parity / debug only: factor plans this context has built for new factor lists, how many of them took over the buffers of the plan its full
 * cache was about to evict (a new list of the same shape: GLIM's odometry brings one per frame), idle plans cached right now. */
int glim_amd_debug_plan_stats(glim_amd_ctx* ctx, uint64_t* built, uint64_t* recycled, int32_t* cached);
/* parity / debug only: host-side account of the calling thread's LAST one-submission glim_amd_frame_create, microseconds since its entry:
 * [0] cloud allocated, [1] staging block + stream allocations, [2] pull kernel launched, [3] host conversion done, [4] voxel-map kernels
 * enqueued, [5] completion word seen, [6] return (tools/odometry_frame_loop.cpp prints the medians). */
int glim_amd_debug_frame_stages(double* microseconds, int32_t num_fields);

/* ---- multi-device cost evaluation (BASELINE.json configs[3]; no counterpart in the reference, which is single-device:
 *      src/glim/mapping/global_mapping.cpp:110 one StreamTempBufferRoundRobin(64), :430-484 create_matching_cost_factors) -------------
 * One process, N devices: a context + a host thread + an RCCL communicator (ncclCommInitAll; librccl is dlopen'ed on first use) per
 * device (device 0 is driven by the CALLING thread).  Clouds and voxel maps are replicated on every device, the factor list is sharded into
 * contiguous cost-balanced chunks, every device linearises its chunk -- in a few pieces, so that the ncclAllGather of one piece's 29-double
 * compact records travels over xGMI (and this device's own rows to the host over its own PCIe link) while the next piece's kernels run --
 * and the records are expanded on the host in the original factor order.  One hand-over to the devices' threads per evaluation.  All
 * calls are synchronous and must come from one host thread at a time. */
typedef struct glim_amd_multi glim_amd_multi;
/* devices: distinct HIP device ordinals.  A multi-device handle without a working RCCL is refused (GLIM_AMD_ERR_HIP) rather than
 * silently gathering over PCIe; a single device works either way (GLIM_AMD_DIAG="multi_rccl=0" skips the collective there). */
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
/* wall-clock milliseconds per whole-cost evaluation (all devices + collective + host expansion skipped), over `iters` evaluations */
int glim_amd_multi_profile(glim_amd_multi* multi, const double* T_target_source, int iters, float* ms_per_evaluation);
/* per device, HIP-event milliseconds of the LAST evaluation: its factor kernels + finalise (kernel_ms[d]) and the collective + copy-out behind
 * them (gather_ms[d]); num_devices entries each, either may be NULL */
int glim_amd_multi_last_timing(const glim_amd_multi* multi, float* kernel_ms, float* gather_ms);
/* host-side account of the LAST evaluation on one device's thread, microseconds, GLIM_AMD_MULTI_BREAKDOWN_FIELDS values:
 *   [0] post        caller: handing the evaluation to the other devices' threads          (device 0 only)
 *   [1] wake        from the caller's entry to the start of this device's task            (0 for device 0: the caller's own thread)
 *   [2] pose_stage  this shard's poses copied into the pinned ring
 *   [3] enqueue     plan check + H2D pose copy + kernel launches
 *   [4] barrier     waiting until every device has enqueued (no collective starts before)
 *   [5] collective  ncclAllGather calls + copy-out enqueue
 *   [6] wait        hipStreamSynchronize: the device working
 *   [7] join        caller: waiting for the other devices' threads                        (device 0 only)
 *   [8] scan        caller: total error + expansion of the records in factor order        (device 0 only)
 *   [9] total       the whole glim_amd_multi_linearize call                                (device 0 only)
 *   [10] library_calls   inside the ncclAllGather calls (part of [5])
 *   [11] device_gather   HIP events: from this device's last kernel to the end of its last all-gather
 *   [12] device_copy_out HIP events: from there to the end of the copy-out and the error sum  ([11] + [12] = gather_ms of last_timing) */
#define GLIM_AMD_MULTI_BREAKDOWN_FIELDS 13
int glim_amd_multi_last_breakdown(const glim_amd_multi* multi, int32_t device, double* microseconds, int32_t num_fields);
/* how a device's shard is evaluated: n >= 2 = as n pieces (at most 8), the all-gather and copy-out of one piece overlapping the kernels of
 * the next; 0 or 1 = as one set and one all-gather; -1 (default) = pieces of at least 2048 factors, at most 4.  Takes effect with the next
 * glim_amd_multi_set_factors. */
int glim_amd_multi_set_split(glim_amd_multi* multi, int32_t mode);
/* ONE device has nothing to gather, so its evaluations make no library call; the binding is exercised when the handle is created (an in-place
 * one-rank all-gather that must come back unchanged; glim_amd_multi_info uses_rccl says whether it did).  on != 0: make the no-op
 * ncclAllGather in every evaluation as well (measurement aid: bench.py prices it).  No effect on several devices. */
int glim_amd_multi_set_one_rank_collective(glim_amd_multi* multi, int32_t on);
/* how every device's own records reach the host array (glim_amd_multi_records, the `out` of glim_amd_multi_linearize): 1 (default) = its
 * finalising kernels store them there as well (host-mapped memory, 232 B per factor over the device's PCIe link while the launch runs: nothing
 * is copied behind the kernels); 0 = device-to-host copies on the collective's stream behind each piece.  Takes effect with the next
 * glim_amd_multi_set_factors. */
int glim_amd_multi_set_host_records(glim_amd_multi* multi, int32_t mode);
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
/* timing aid: `iters` back-to-back glim_amd_overlap_batch calls with these arguments; microseconds per call, measured inside the library. */
int glim_amd_overlap_profile(glim_amd_ctx* ctx, int32_t num_queries, const int32_t* num_targets, const glim_amd_voxelmap* const* targets,
                             const double* T_target_source, const glim_amd_cloud* const* sources, int iters, float* us_per_call);

#ifdef __cplusplus
}
#endif
#endif /* GLIM_AMD_H */
