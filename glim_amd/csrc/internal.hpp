// internal.hpp -- host-side object model behind the C ABI (include/glim_amd.h) and shared device helpers.
// gfx950 (MI355X / CDNA4) only: 64-lane wavefronts, DPP row_bcast reductions, no other target is supported.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/glim_amd.h"
#include "../../include/glim_amd_diag.h"  // measurement / test hooks: implemented by the same library, not part of the drop-in boundary

namespace glim_amd {

// ---------------------------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------------------------
void set_hip_error(hipError_t e, const char* what);

// device memory pool (context.hip): drop-in for hipMalloc / hipFree on the current device
hipError_t pool_malloc_impl(void** p, size_t bytes);
hipError_t pool_free(void* p);
void pool_trim(int device);
// pinned + device-mapped host memory, cached by size class (context.hip)
hipError_t pinned_malloc_impl(void** p, size_t bytes);
hipError_t pinned_free(void* p);
template <class T>
inline hipError_t pinned_malloc(T** p, size_t bytes) {
  return pinned_malloc_impl(reinterpret_cast<void**>(p), bytes);
}
template <class T>
inline hipError_t pool_malloc(T** p, size_t bytes) {
  return pool_malloc_impl(reinterpret_cast<void**>(p), bytes);
}

// RAII holder of a pool allocation used as kernel scratch
struct DeviceTemp {
  void* p = nullptr;
  DeviceTemp() = default;
  DeviceTemp(const DeviceTemp&) = delete;
  DeviceTemp& operator=(const DeviceTemp&) = delete;
  ~DeviceTemp() {
    if (p) (void)pool_free(p);
  }
  template <class T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

// Hilbert rank of a cloud's points (knn.hip); no-op when present, tiny, or disabled
int cloud_curve_rank(::glim_amd_cloud* c, ::glim_amd_ctx* held, hipStream_t st);
// factor streams of a cloud that has covariances (cloud.hip); no-op when present.  `held`: the context whose mutex the caller holds -- its
// switches and its read-back scratch are the ones used (a cloud may be reached from a factor set of ANOTHER context than its owner's, whose
// mutex the caller does not hold: ADVICE r4).  Same for cloud_curve_rank.
int ensure_factor_streams(::glim_amd_cloud* c, ::glim_amd_ctx* held, hipStream_t st);
// the host-packed upload of a small cloud (<= HOST_PACK_MAX_POINTS) in two halves around ONE synchronise of ctx->stream() (cloud.hip)
// what the pull kernels take (pull.hpp): the staging block's sections (device view of pinned host memory; s_covA / s_nrm may be null), the cloud's
// arrays and factor streams, the tail words of the staging block ([0] violations of the plane-form test, [1] PULL_GAVE_UP) and the gate
struct PullArgs {
  int n = 0;
  const float4* s_pts = nullptr;
  const float4* s_covA = nullptr;
  const float2* s_covB = nullptr;
  const float4* s_nrm = nullptr;
  float4* pts = nullptr;
  float4* covA = nullptr;
  float2* covB = nullptr;
  float4* nrm = nullptr;
  unsigned int* host_tail = nullptr;
  float4* pn4 = nullptr;
  float2* n2 = nullptr;
  float4* gs0 = nullptr;
  float4* gs1 = nullptr;
  float* gs2 = nullptr;
  float4* gsn = nullptr;
  const unsigned int* gate = nullptr;  // null: the staging block was complete before the launch
  unsigned int gate_seq = 0u;
  int piece_len = 256;
};
struct SmallUpload {
  float* stage = nullptr;
  volatile unsigned int* violations = nullptr;  // the tail words of the staging block, host side
  bool maybe_plane = false;
  bool gated = false, packed = false;
  PullArgs args;
  const double *points4 = nullptr, *covs16 = nullptr, *normals4 = nullptr;  // the caller's arrays (cloud_small_pack)
};
// debug account of the LAST glim_amd_frame_create of this thread, microseconds from its entry (glim_amd_debug_frame_stages): [0] cloud allocated,
// [1] staging block + stream allocations, [2] pull kernel launched, [3] host conversion done, [4] voxel-map kernels enqueued, [5] completion
// word seen, [6] return
constexpr int FRAME_STAGES = 7;
extern thread_local double g_frame_stage_us[FRAME_STAGES];
extern thread_local double g_frame_t0_us;
inline double frame_now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline void frame_stamp(int i) { g_frame_stage_us[i] = frame_now_us() - g_frame_t0_us; }
constexpr unsigned int PULL_GAVE_UP = 0xffffffffu;  // violations word: the gated pull kernel stopped waiting for the host's conversion (cloud.hip)
int cloud_small_enqueue(::glim_amd_ctx* ctx, ::glim_amd_cloud* c, const double* points4, const double* covs16, const double* normals4, SmallUpload* up);
// its three steps, for a caller that launches its own pull kernel (glim_amd_frame_create): allocations + kernel arguments; the host conversion
// into the staging block (gated form: AFTER the launch, publishing piece by piece); the plain pull kernel on `st`
int cloud_small_prepare(::glim_amd_ctx* ctx, ::glim_amd_cloud* c, const double* points4, const double* covs16, const double* normals4, SmallUpload* up);
void cloud_small_pack(SmallUpload* up);
int cloud_small_launch(SmallUpload* up, hipStream_t st);
int cloud_small_finish(::glim_amd_cloud* c, SmallUpload* up);  // GLIM_AMD_ERR_UNSUPPORTED: the gated pull gave up (the cloud holds nothing; gating is off from now on)
int alloc_cloud_for_frame(::glim_amd_ctx* ctx, int64_t n, bool covs, bool normals, ::glim_amd_cloud** out);  // cloud.hip alloc_cloud
constexpr int64_t HOST_PACK_MAX_POINTS_FRAME = 32768;
// plane-form test of a freshly uploaded cloud with covariances and normals (cloud.hip): sets c->plane_form
int detect_plane_form(::glim_amd_cloud* c, hipStream_t st);
// Diagnostic / tuning switches.  None is needed in production; they exist for A/B measurements and for the cross-checks of the parity
// tests.  ONE structure per context: initialised at glim_amd_ctx_create from the process defaults -- the single environment variable
// GLIM_AMD_DIAG="key=value,key=value", parsed once per process -- and changed per context with glim_amd_ctx_set_diag (same syntax).
// No other getenv exists in the library.
struct Diag {
  int knn_path = 0;       // knn_path=auto|grid|chunks|brute     which kNN implementation answers glim_amd_cloud_find_neighbors
  int knn_kernel = 0;     // knn_kernel=auto|wave64|pair         64-query or pair-lane chunk kernel
  int knn_select = 1;     // knn_select=0|1                      per-lane threshold selection of the chunk kernels (k <= 10)
  int plane = 1;          // plane=0|1                           plane-form (24 B/pt) factor kernel for plane-form clouds
  int curve_order = 1;    // curve_order=0|1                     factor streams in the Hilbert order of the cloud
  int ppt = 0;            // ppt=<n>                             points per thread of the factor kernel (0: one resident set of blocks)
  int poll = 1;           // poll=0|1                            host-mapped completion word for small synchronous calls
  int inline_pose = 1;    // inline_pose=0|1                     pose of a single-factor set in the kernel arguments
  int bucket_factor = 0;  // bucket_factor=<n>                   buckets per voxel of a map table (0: default 6)
  int plan_cache = 1;     // plan_cache=0|1                      factor-set plans cached per context, keyed on the (map, cloud, flags) list
  int plan_recycle = 1;   // plan_recycle=0|1                    a new list of the shape of the plan a full cache is about to evict takes over its buffers
  int host_poses = 1;     // host_poses=0|1                      small synchronous sets: kernels read the poses from host-mapped memory (no H2D copy)
  int resident = 2;       // resident=0|1|auto                  repeated synchronous linearisations of a small set go through a resident kernel (no launch per
                          //                                    call); auto (default): only in a context created with priority 1 (the odometry module's)
  int resident_idle_us = 1000;  // resident_idle_us=<n>         the resident kernel leaves after this long without a request
  int pp_fast = 1;        // pp_fast=0|1                        random-grid preprocessing: one sort + counting ranks, one synchronise (preprocess.hip)
  int fuse = 1;           // fuse=0|1                           small synchronous sets: ONE dispatch (factors finalised inside the factor kernel)
  int view_fused = 1;     // view_fused=0|1                      a map built from a plane-form cloud gets its plane view (A_B records) from the finalise kernel; 0: on first use
  int host_pack = 1;      // host_pack=0|1                       small clouds (<= 32 768 pts) are converted to the device layout on the host, one kernel pulls them over
  int frame_fused = 1;    // frame_fused=0|1                     glim_amd_frame_create: one launch pulls the cloud and builds every level, one writes every level's records
  int pull_gated = 1;     // pull_gated=0|1                      ... and that kernel is launched BEFORE the conversion: its blocks wait for their piece of the staging block
  int pool = 1;           // pool=0|1                            device / pinned memory caches (process-wide: GLIM_AMD_DIAG only)
  int small_rows = 0;     // small_rows=<n>                      partial rows ONE factor of a small synchronous set is planned into at most (0: one per compute unit)
  int cull = 0;           // cull=0|1|2 (default 0: measured on configs[3] the pre-pass costs more than the walk saves, profiles/r06/probe/precull_*.json)                          large general-form sets: a pre-pass marks the wavefront trips whose chunk box misses the target's occupancy mask (2: sets of any size, tests)
  int multi_rccl = 1;     // multi_rccl=0|1                      glim_amd_multi: skip the collective on a single device
  int multi_host_gather = 0;  // multi_host_gather=0|1           glim_amd_multi: allow a host gather when librccl cannot be loaded (tests)
  int multi_virtual = 0;  // multi_virtual=0|1                   glim_amd_multi_create accepts one physical device several times ("virtual devices": the N > 1
                          //                                    code path on a one-GPU box; the exchange is a same-device stand-in for ncclAllGather)
  char knn_debug[256] = "";   // knn_debug=<file>                dump per-wavefront work counters of the 64-query chunk kernel
};
enum { RESIDENT_OFF = 0, RESIDENT_ON = 1, RESIDENT_AUTO = 2 };
enum { KNN_PATH_AUTO = 0, KNN_PATH_GRID = 1, KNN_PATH_CHUNKS = 2, KNN_PATH_BRUTE = 3 };
enum { KNN_KERNEL_AUTO = 0, KNN_KERNEL_WAVE64 = 1, KNN_KERNEL_PAIR = 2 };
const Diag& process_diag();                        // GLIM_AMD_DIAG, parsed once
int diag_parse(Diag& d, const char* key_values);   // GLIM_AMD_OK or GLIM_AMD_ERR_INVALID (unknown key / bad value); d untouched on error
int diag_print(const Diag& d, char* buf, size_t len);

// Small device -> host read-back (counters, bounding boxes: <= 1 KiB) followed by a synchronise of `st`.  A device-to-host copy into pageable
// memory (a stack variable) is staged by the runtime and blocks the caller for ~20 us; through the context's pinned scratch it is one DMA
// packet.  Caller holds ctx->mu.
hipError_t read_back_sync(::glim_amd_ctx* ctx, hipStream_t st, void* dst_host, const void* src_device, size_t bytes);
bool pinned_scratch_views(::glim_amd_ctx* ctx, void** host, void** device);  // the same block as kernels address it (mapped); caller holds ctx->mu

// stable LSD radix sort of (u64 key, u32 value) pairs (sort.hip)
size_t radix_sort_scratch_bytes(int n);
hipError_t radix_sort_pairs(hipStream_t st, int n, int bits, unsigned long long* keys_a, unsigned int* vals_a, unsigned long long* keys_b,
                            unsigned int* vals_b, bool vals_a_is_iota, int* scratch, unsigned long long** keys_sorted, unsigned int** vals_sorted);

#define GA_HIP(call)                                   \
  do {                                                 \
    hipError_t _e = (call);                            \
    if (_e != hipSuccess) {                            \
      ::glim_amd::set_hip_error(_e, #call);            \
      return GLIM_AMD_ERR_HIP;                         \
    }                                                  \
  } while (0)

#define GA_TRY(call)                 \
  do {                               \
    int _rc = (call);                \
    if (_rc != GLIM_AMD_OK) return _rc; \
  } while (0)

// ---------------------------------------------------------------------------------------------------------------
// device data layouts
// ---------------------------------------------------------------------------------------------------------------

// One hash BUCKET of a Gaussian voxel map: 128 bytes, 128-byte aligned (one L2 line), two ways.  A lookup is one 16-byte
// load of both keys followed by a 48-byte read of the matching record out of the line that load has just brought in, so the
// dependent read is an on-chip hit and there is no way/collision divergence inside a wavefront; a third key hashing to a
// full bucket spills to the next bucket (rare at the default 1/3 keys per bucket; exact compare keeps lookups lossless).
// (SURVEY.md 8a row a5 stores a 16-B bucket + a 52-B record in two arrays: two random lines per lookup.)
//   key[w]    : packed voxel coordinate (21 bits per axis, offset 2^20), EMPTY_KEY when free
//   rec[w]    : mx my mz c00 | c01 c02 c11 c12 | c22 count - -
//               mean of the member means stored RELATIVE TO THE VOXEL CENTRE (coord + 0.5) * resolution (|m| <= res/2: the FP32
//               image is ~1e-8 m accurate anywhere in the map and mu - q is formed without cancellation); mean of the member
//               covariances, symmetric storage; number of member points (int bits)
struct alignas(128) VoxelBucket {
  unsigned long long key[2];  // 16 B
  float rec[2][12];           // 2 x 48 B
  int pad[4];                 // 16 B
};
static_assert(sizeof(VoxelBucket) == 128, "VoxelBucket must be one 128-byte line");

constexpr unsigned long long EMPTY_KEY = ~0ull;
constexpr int KEY_BITS = 21;
constexpr int KEY_OFFSET = 1 << 20;

// Per-factor descriptor consumed by the fused VGICP kernel (device array, rebuilt when the set changes).
struct FactorDesc {
  const float4* pts;        // source xyz1, arrival order (correspondence / debug kernels)
  const float4* normals;    // arrival order, may be null
  // factor stream of the source cloud, in the Hilbert order of the cloud (glim_amd_cloud, "factor streams"):
  //   plane-form (plane != 0): s0 = float4 (x y z nx), s1 = float2 (ny nz)                                    24 B per point
  //   general                : s0 = float4 (x y z c00), s1 = float4 (c01 c02 c11 c12), s2 = float c22,        36 B per point
  //                            sn = float4 normals in stream order (null without normals; surface validation only)
  const void* s0;
  const void* s1;
  const void* s2;
  const float4* sn;
  int plane;
  int pad0;
  const VoxelBucket* buckets;  // target table
  unsigned int num_buckets;    // any size >= 1 (range reduction by multiply-shift)
  int n;                    // source points
  double inv_res;           // 1 / target resolution
  double res;               // target resolution
  unsigned int flags;
  int first_block;          // index of this factor's first partial row
  int num_blocks;           // partial rows (= chunks) of this factor
  int ppt;                  // points per thread: chunk = 256 * ppt consecutive points per block
};

// Pose AND descriptor of a single-factor set passed by value in the kernel arguments: the synchronous per-factor call of the odometry
// then needs no host-to-device copy, and its blocks no dependent blockmap -> descriptor -> stream load chain (two memory round trips
// of a kernel that lasts five).  valid == 0: poses, descriptors and the block map are read from device memory as usual.
struct InlineArgs {
  double m[12];
  int valid;
  int pad;
  FactorDesc d;
};

// Identity of a cloud / voxel map as a factor plan sees it: a fresh value at creation and after every change that invalidates descriptors
// built from the object (covariances re-estimated, map re-inserted).  Values never recur, so a cached plan can never match a new object
// that happens to live at a recycled address.
inline uint64_t next_uid() {
  static std::atomic<uint64_t> n{1};
  return n.fetch_add(1);
}

// Clouds and voxel maps may be used across contexts of one device (GLIM's three modules own a stream pool each and hand frames / maps to one
// another: sub_mapping.cpp:168, global_mapping.cpp:253-266), so what invalidates plans built from them is counted process-wide, not per context.
std::atomic<uint64_t>& global_mutation_epoch();
// waits for the asynchronous launches of EVERY context of `device` that may still be reading an object which is about to change or die
// uid: the cloud / voxel map whose memory is about to be recycled (0: unknown -- any object).  The device's resident session ends only when ITS plan
// uses that object: a mapping thread that drops an unrelated cloud must not take the odometry's session down with it.
void quiesce_device(int device, uint64_t uid = 0);
void resident_stop_device(int device, uint64_t uid);  // vgicp.hip (waits for a request in flight)

constexpr int PARTIAL_STRIDE = 32;  // floats per block partial: 6 Hww + 9 Hwv + 6 Hvv + 3 (u x p) + 3 u + 1 err + 1 count(int) + pad
constexpr int COMPACT = GLIM_AMD_COMPACT_DOUBLES;

}  // namespace glim_amd

// ---------------------------------------------------------------------------------------------------------------
// C-ABI object definitions (opaque to callers)
// ---------------------------------------------------------------------------------------------------------------
struct FactorPlan;
struct glim_amd_ctx {
  std::atomic<int> live_children{0};  // clouds, voxel maps, factor sets and search indices created from this context and not yet destroyed
  int device = 0;
  int num_cus = 0;
  bool owns_streams = true;
  std::vector<hipStream_t> streams;
  int next_stream = 0;
  std::mutex mu;
  // set by the *_async entry points: device work may still be reading buffers when the call returns, so the next call that recycles
  // device memory of this context (destroy / re-plan) synchronises the streams first (quiesce)
  std::atomic<bool> async_pending{false};
  int priority = 0;  // 1: its streams were created with the device's greatest priority (glim_amd_ctx_create_ex)
  std::vector<FactorPlan*> plan_cache;  // idle factor plans, most recently released first (vgicp.hip; guarded by mu)
  uint64_t plans_built = 0, plans_recycled = 0;  // plans built for new factor lists; how many of them in the buffers of an evicted plan (guarded by mu)
  // overlap scratch (vgicp.hip), allocated on first use: per-query arrival counters on the device, results + completion word in host-mapped memory
  static constexpr int OV_MAX_QUERIES = 1024;
  unsigned long long* ov_counters = nullptr;  // [OV_MAX_QUERIES] packed (arrived blocks << 32 | hits), then the queries-done word
  unsigned int* ov_host = nullptr;            // pinned: [0] completion word, [1 + q] hits of query q
  unsigned int* ov_host_dev = nullptr;
  unsigned int ov_seq = 0;
  unsigned int map_seq = 0;  // sequence number of the polled voxel-map builds of this context (voxelmap.hip; guarded by mu)
  std::vector<std::pair<int, double>> voxel_ratio_hints;  // (resolution class, voxels per point of the last map built there): voxelmap.hip
  void* pinned_scratch = nullptr;  // 1 KiB of pinned host memory for small read-backs (read_back_sync; guarded by mu)
  void* pinned_scratch_dev = nullptr;  // its device view (kernels that hand a few words to the host themselves)
  void quiesce() {
    if (async_pending.exchange(false))
      for (auto s : streams) (void)hipStreamSynchronize(s);
  }
  glim_amd::Diag diag;  // diagnostic switches of this context (internal.hpp "Diag")
  hipStream_t stream() const { return streams[0]; }
  hipStream_t round_robin() {
    hipStream_t s = streams[next_stream];
    next_stream = (next_stream + 1) % (int)streams.size();
    return s;
  }
};
// The resident session (vgicp.hip) holds wave slots and registers on every SIMD while it is alive and costs whatever else runs on the device
// 1.3-1.4x: it is OPT-IN -- on for a context created with priority 1 (GLIM's odometry thread, adapters/glim/odometry_estimation_hip_create.cpp)
// or with the switch resident=1, off for every other context (the mapping threads' contexts never start one).
inline bool resident_enabled(const glim_amd_ctx* c) {
  return c->diag.resident == glim_amd::RESIDENT_ON || (c->diag.resident == glim_amd::RESIDENT_AUTO && c->priority > 0);
}

// Back-pointer of a child object to its context that also keeps the context's live-children count: destroying a context that still
// has children is refused (GLIM_AMD_ERR_STATE) instead of leaving them dangling (SURVEY.md 8b "Ownership").
struct CtxRef {
  glim_amd_ctx* c = nullptr;
  CtxRef() = default;
  CtxRef(const CtxRef&) = delete;
  CtxRef& operator=(const CtxRef&) = delete;
  CtxRef& operator=(glim_amd_ctx* p) {
    if (c) c->live_children--;
    c = p;
    if (c) c->live_children++;
    return *this;
  }
  ~CtxRef() {
    if (c) c->live_children--;
  }
  operator glim_amd_ctx*() const { return c; }
  glim_amd_ctx* operator->() const { return c; }
};

struct glim_amd_cloud {
  CtxRef ctx;
  std::mutex build_mu;  // lazily built members (factor streams, Hilbert rank) are built by one caller at a time, whatever its context
  uint64_t uid = glim_amd::next_uid();
  int64_t n = 0;
  float4* pts = nullptr;
  float4* covA = nullptr;
  float2* covB = nullptr;
  float4* normals = nullptr;
  // Factor streams: what the VGICP kernel reads per source point, written in the Hilbert order of the cloud so that the 64 lanes of a
  // wavefront look up a handful of neighbouring voxels (the factor sums over all points: their order is free).
  //   plane-form clouds (C = I - (1 - 1e-3) n n^T, the only form GLIM's covariance estimator emits): pn4 + n2, 24 B per point;
  //   written by the covariance kernel, or lazily by ensure_factor_streams for clouds uploaded with matching covariances + normals
  //   any other cloud: gs0 + gs1 + gs2 (+ gsn), 36 B per point, built lazily by ensure_factor_streams (cloud.hip)
  float4* pn4 = nullptr;  // x y z nx
  float2* n2 = nullptr;   // ny nz
  float4* gs0 = nullptr;  // x y z c00
  float4* gs1 = nullptr;  // c01 c02 c11 c12
  float* gs2 = nullptr;   // c22
  float4* gsn = nullptr;  // normals in stream order
  // chunk boxes of the general stream (vgicp.hip "pre-cull"): min xyz | max xyz of every 64 consecutive stream points -- what one wavefront trip of
  // the factor kernel covers --, 6 floats per chunk; built on first use by a large factor set (ensure_chunk_boxes), dropped with the stream
  float* gbox = nullptr;
  unsigned int* curve_rank = nullptr;  // position of point i on the Hilbert curve through the cloud (knn.hip); orders the factor streams
  bool plane_form = false;
  int32_t* neighbors = nullptr;
  int k = 0;
  bool has_covs = false;
  bool has_normals = false;
  // scan-preprocessing outputs (preprocess.hip): the exact FP64 points `pts` was rounded from, and per-point time / intensity
  double4* pts64 = nullptr;
  double* times = nullptr;
  double* intensities = nullptr;
  double* cov64 = nullptr;  // merged submaps (glim_amd_merge_frames): exact FP64 covariances, 6 per point (c00 c01 c02 c11 c12 c22)
  std::vector<double> h_times;  // host copy of `times` (the deskewing time table is built from it)
  size_t bytes() const {
    size_t b = (size_t)n * sizeof(float4);
    if (pts64) b += (size_t)n * sizeof(double4);
    if (times) b += (size_t)n * sizeof(double);
    if (intensities) b += (size_t)n * sizeof(double);
    if (cov64) b += (size_t)n * 6 * sizeof(double);
    if (covA) b += (size_t)n * (sizeof(float4) + sizeof(float2));
    if (normals) b += (size_t)n * sizeof(float4);
    if (neighbors) b += (size_t)n * k * sizeof(int32_t);
    if (pn4) b += (size_t)n * (sizeof(float4) + sizeof(float2));
    if (gs0) b += (size_t)n * (2 * sizeof(float4) + sizeof(float));
    if (gsn) b += (size_t)n * sizeof(float4);
    return b;
  }
};

struct glim_amd_voxelmap {
  CtxRef ctx;
  uint64_t uid = glim_amd::next_uid();
  double resolution = 0.0;
  double inv_resolution = 0.0;
  int32_t num_voxels = 0;
  uint32_t num_buckets = 0;  // 0 until insert()
  glim_amd::VoxelBucket* buckets = nullptr;
  // "Plane view" of the same table for factors whose source cloud is plane-form (vgicp.hip, GLIM_AMD_PLANE_SM): same keys, same means, but the
  // six covariance slots of a record hold A_B = (C_B + I)^-1 (inverted in FP64 from the FP32 record), and one extra, all-zero bucket at index
  // num_buckets that lanes without a correspondence read.  Built on first use (ensure_plane_view), dropped whenever the table is rebuilt.
  glim_amd::VoxelBucket* buckets_sm = nullptr;
  std::mutex view_mu;
  // least-recently-used eviction of an incrementally built map (glim_amd_voxelmap_set_lru_horizon; GaussianVoxelMapCPU::set_lru_horizon of the
  // CPU odometry, odometry_estimation_cpu.cpp:63-68): slot 10 of a record holds the insert counter of the last insert that touched the voxel
  int32_t lru_horizon = 0, lru_clear_cycle = 10, lru_counter = 0;
  // A direct build returns to its caller as soon as the voxel count is on the host (a polled, host-mapped word the LAST kernel of the build writes
  // when it starts), i.e. while that kernel is still writing records: `ready_event` (recorded behind it) is what every later reader of the table
  // on another stream waits for (voxelmap_wait_ready), and the build's scratch stays with the map until the event has been seen complete.
  hipEvent_t ready_event = nullptr;
  std::atomic<bool> ready_pending{false};
  void *pending_acc = nullptr, *pending_stats = nullptr;
  // Occupancy mask (vgicp.hip "pre-cull"): one bit per cell of (voxel << occ_shift) inside the box of the occupied voxels, x along the bits of a
  // row of occ_row_words 32-bit words, rows ordered (z, y).  A transformed chunk box that touches no set bit cannot hold a correspondence.  Built on
  // first use by a large factor set (ensure_occupancy, under view_mu), dropped whenever the table is rebuilt.  occ_state: 0 not built, 1 built,
  // -1 not available (empty map / box too large for the 64 KiB cap even at the coarsest cell).
  unsigned int* occ = nullptr;
  int occ_org[3] = {0, 0, 0}, occ_dim[3] = {0, 0, 0}, occ_shift = 0, occ_row_words = 0, occ_state = 0;
};
namespace glim_amd {
int ensure_plane_view(::glim_amd_voxelmap* m, hipStream_t st);  // voxelmap.hip; complete (synchronised) before it returns
int ensure_occupancy(::glim_amd_voxelmap* m, hipStream_t st);   // voxelmap.hip; likewise (GLIM_AMD_OK also when no mask can be had: occ stays null)
int ensure_chunk_boxes(::glim_amd_cloud* c, hipStream_t st);    // cloud.hip; likewise (the general stream must exist)
// Before anything reads m->buckets / buckets_sm: no-op for a finished map; otherwise `consumer` (a stream) is made to wait for the build, or -- consumer
// == nullptr -- the host waits.  Any thread, any context.
int voxelmap_wait_ready(const ::glim_amd_voxelmap* m, hipStream_t consumer);
}

// Device plan of a factor list: descriptors, the block -> (factor, chunk) map, partial rows, pose / result staging.  Building one costs
// six pool allocations, three uploads and a stream synchronise, and GLIM builds a FRESH NonlinearFactorSetGPU for every linearisation
// (odometry_estimation_gpu.cpp:383-385; the optimisers' hook does clear -> add(graph) -> linearize per iteration), so plans are cached
// per context, keyed on the (voxel map, cloud, flags) list: a set that is cleared or destroyed parks its plan in ctx->plan_cache and the
// next set with the same list adopts it with no device work at all.
struct PlanKey {
  uint64_t target_uid, source_uid;
  uint32_t flags, pad;
  bool operator==(const PlanKey& o) const { return target_uid == o.target_uid && source_uid == o.source_uid && flags == o.flags; }
};
struct FactorPlan {
  std::vector<PlanKey> key;
  int built_plane = 1, built_ppt = 0, built_cull = 1, built_small_rows = 0;  // plan-time diagnostic switches the plan was built with
  // rows [0, plane_rows) = blocks of factors whose source cloud is plane-form (24 B/pt kernel), rows [plane_rows, total_rows) = blocks of
  // the other factors (36 B/pt kernel); each segment is its own launch
  int points_per_thread = 1;
  int max_rows_per_factor = 0;    // most blocks (partial rows) any factor of the plan owns: picks the finalise kernel's width
  int plane_rows = 0, total_rows = 0;
  glim_amd::FactorDesc* d_descs = nullptr;
  int2* d_blockmap = nullptr;     // total_rows x int2
  float* d_partials = nullptr;
  unsigned long long* d_trip_stats = nullptr;  // 64 counters of skipped wavefront trips (general kernel; glim_amd_factor_set_trip_stats)
  // pre-cull of the general segment (vgicp.hip cull_kernel): per (plan row, wavefront) one 64-bit word, bit t = trip t of that wavefront cannot
  // find a correspondence (its chunk box, moved by this evaluation's pose, touches no occupied cell of the target's mask) or holds no point
  void* d_cull_descs = nullptr;                 // CullDesc per factor
  unsigned long long* d_cull_words = nullptr;   // (total_rows - plane_rows) x 4
  unsigned long long* d_cull_stats = nullptr;   // 64 x {trips culled by the box test, trips that hold points}, summed over the counted evaluations since the last reset
  bool cull = false;
  bool cull_count = false;                      // the pre-pass also counts (glim_amd_factor_set_cull_stats arms it: atomics, measurement only)
  int cull_log2p = 0;                           // log2 of the pre-pass's lanes per (row, wavefront) pair (the next power of two >= points per thread)
  char* d_rows16 = nullptr;       // tagged partial rows of the single-dispatch synchronous form (vgicp.hip TAG_ROW_BYTES per row), or null
  int* d_finmap = nullptr;        // factor ids the trailing blocks of each segment's single-dispatch launch finalise (plane-form segment first)
  int fin_count[2] = {0, 0};
  char* h_rec16 = nullptr;        // single-dispatch form: the records as host-mapped 16-byte granules {value, sequence number}, COMPACT per factor
  char* h_rec16_dev = nullptr;
  std::vector<int> h_finmap;
  double* d_poses = nullptr;      // 2 x n x 12 (lin, eval)
  double* d_compact = nullptr;    // n x COMPACT
  // pinned pose staging: a ring, because an asynchronous call returns while its host-to-device copy may still be reading the slot
  static constexpr int POSE_RING = 4;
  double* h_poses = nullptr;      // POSE_RING x (2 x n x 12)
  double* h_poses_dev = nullptr;  // device view of h_poses
  hipEvent_t pose_events[POSE_RING] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t poses_free_event = nullptr;  // upload_stream form: "the kernels that read d_poses are done" (the next copy into d_poses waits for it)
  bool poses_free_pending = false;
  bool pose_pending[POSE_RING] = {false, false, false, false};
  int pose_slot = 0;
  double* h_compact = nullptr;       // pinned, host-mapped
  double* h_compact_dev = nullptr;   // device view of h_compact (small sets: results land in host memory, no D2H copy)
  int* d_done = nullptr;             // finished factors of a launch (polling fast path)
  unsigned int* h_flag = nullptr;    // host-mapped completion word, written by the last finalise block
  unsigned int* h_flag_dev = nullptr;
  unsigned int poll_seq = 0;
  size_t cap_factors = 0, cap_blocks = 0;
  long long alloc_rows = -1;         // partial rows the buffers were sized for (plan_build re-uses the buffers of an evicted plan of the same shape)
  bool fused_alloc = false, recycled = false;
  int sync_linearize_calls = 0;      // synchronous linearisations this plan has served (a resident session starts after a few)
  std::vector<glim_amd::FactorDesc> h_descs;
  std::vector<int2> h_blockmap;
  bool uploaded = false;              // d_descs / d_blockmap hold h_descs / h_blockmap (single-factor plans upload on first non-inline launch)
  // d_descs | d_blockmap | d_finmap are ONE device block filled by ONE copy from the pinned block h_upload (three pageable copies cost a new
  // plan ~15 us, and GLIM's odometry builds a new plan with every frame); d_done | d_trip_stats | d_rows16 are ONE block cleared by one memset
  char* d_upload = nullptr;
  char* h_upload = nullptr;
  size_t upload_bytes = 0;
  char* d_zeroed = nullptr;
  hipStream_t last_stream = nullptr;  // stream of the last enqueue
  bool maybe_busy = false;            // an asynchronous enqueue may still be running on last_stream
};

struct glim_amd_factor_set {
  CtxRef ctx;
  hipStream_t stream = nullptr;
  struct Entry {
    const glim_amd_voxelmap* target;
    const glim_amd_cloud* source;
    uint32_t flags;
  };
  std::vector<Entry> entries;
  bool dirty = true;             // the entry list changed since the plan was built / adopted
  uint64_t seen_epoch = 0;       // ctx->mutation_epoch when the plan was last validated
  FactorPlan* plan = nullptr;    // owned while set; parked in ctx->plan_cache on clear / destroy
  glim_amd::InlineArgs inline_args{};  // single-factor sets: pose + descriptor ride in the kernel arguments
  const double* poses_dev = nullptr;   // where this call's kernels read their poses (the plan's device array, or a host-mapped pinned slot)
  double last_pose_stage_us = 0.0;     // host time the last call spent copying poses into the pinned ring (glim_amd_multi_last_breakdown)
  // asynchronous calls only (glim_amd_multi: one per device): the pose copy goes to THIS stream and the set's stream waits for it, so that the
  // upload of one piece of a shard runs beside the kernels of the piece before it instead of queueing behind them
  hipStream_t upload_stream = nullptr;
  // asynchronous calls with caller-owned device records (glim_amd_factor_set_linearize_device_async): when set, the finalising blocks store every
  // record here as well, at the same row offset (device view of host-mapped memory)
  double* record_mirror = nullptr;
};

namespace glim_amd {
int factor_set_prepare(glim_amd_factor_set* set);
void factor_set_park_plan(glim_amd_factor_set* set);   // plan -> ctx->plan_cache (caller holds ctx->mu)
void ctx_release_factor_resources(glim_amd_ctx* ctx);  // frees every cached plan and the overlap scratch (context being destroyed)
}  // namespace glim_amd
