/*
 * glim_amd_diag.h -- measurement, test and tuning hooks of libglim_amd.so.  NOT part of the drop-in boundary.
 *
 * include/glim_amd.h is the stable C ABI a GLIM maintainer binds (every entry point there names the reference interface it replaces,
 * INTEGRATION.md 1).  Everything in THIS header has no counterpart in the reference: timing loops measured inside the library
 * (bench.py, tools/), counters, the diagnostic switches, debug views for the parity tests, failure injection.  Nothing under adapters/ or
 * examples/ includes it (tests/test_abi_cpu.py checks that), and a production build of GLIM never needs it.  Entry points may change
 * between versions without notice.
 */
#ifndef GLIM_AMD_DIAG_H
#define GLIM_AMD_DIAG_H

#include "glim_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- diagnostic / tuning switches ------------------------------------------------------------------------------ */
/* Diagnostic / tuning switches of a context (no counterpart in the reference; none is needed in production).  key_values:
 * "key=value,key=value"; NULL or "" restores the process defaults, which come from the ONE environment variable the library reads,
 * GLIM_AMD_DIAG (same syntax, parsed once per process).  Keys: knn_path=auto|grid|chunks|brute, knn_kernel=auto|wave64|pair|qgroup,
 * knn_select=0|1, plane=0|1, curve_order=0|1, ppt=<n>, poll=0|1, inline_pose=0|1, bucket_factor=<n>, plan_cache=0|1, plan_recycle=0|1, host_poses=0|1, host_pack=0|1, pull_gated=0|1, frame_fused=0|1,
 * view_fused=0|1 (a voxel map built from a plane-form cloud gets its plane view -- the (C_B + I)^-1 records the plane-form factor kernel reads --
 * from the map's own finalise kernel; 0: on the first factor that needs it),
 * fuse=0|1 (small synchronous sets in ONE dispatch), resident=0|1|auto + resident_idle_us=<n> (repeated synchronous linearisations of a small set
 * served by a resident kernel that leaves after <n> us without a request; auto, the default: only in a context created with priority 1 -- the
 * session costs whatever else runs on the device 1.3-1.4x while it is alive, so it is opt-in), pp_fast=0|1 (random-grid preprocessing without sorts),
 * small_rows=<n> (partial rows ONE factor of a small synchronous set is planned into at most; 0, the default: one per compute unit -- round 5: two),
 * cull=0|1|2 (general-form sets of >= 16 384 plan rows -- 2: of any size --: a pre-pass marks the wavefront trips whose chunk box misses the target's occupancy mask and the
 * factor kernel walks the live trips only; same bits either way),
 * knn_debug=<file>; and, in GLIM_AMD_DIAG ONLY (they are process-wide: set_diag refuses them), pool=0|1, multi_rccl=0|1,
 * multi_host_gather=0|1, multi_virtual=0|1 (glim_amd_multi_create accepts one physical device several times, see
 * glim_amd_debug_multi_create_virtual below).  Unknown keys / bad values: GLIM_AMD_ERR_INVALID and nothing changes.  get_diag prints the current state in the
 * same syntax. */
int glim_amd_ctx_set_diag(glim_amd_ctx* ctx, const char* key_values);
int glim_amd_ctx_get_diag(glim_amd_ctx* ctx, char* buf, size_t len);

/* ---- parity / debug views (host arithmetic or a single device pass; tests only) ------------------------------------- */
/* parity / debug only (host arithmetic, no device needed): the time table CloudDeskewing::deskew builds (cloud_deskewing.cpp:22-45 / :70-124) --
 * entry_out[i] = table entry of point i (n, may be NULL), table12_out = one row-major 3x4 T_lidar0_lidar1 per entry (table_cap entries, may
 * be NULL), *table_size = number of entries.  The deskewing kernels gather from exactly this table. */
int glim_amd_debug_deskew_table(int64_t n, const double* times, const double* T_imu_lidar12, int32_t n_imu, const double* imu_times, const double* imu_poses12,
                                double stamp, const double* linear_vel3, const double* angular_vel3, int32_t* entry_out, double* table12_out, int32_t table_cap,
                                int32_t* table_size);
/* test hook: writes `value` into 32-bit word `word` (< 256) of the context's pinned scratch block -- what an earlier read-back (kNN counters, kept
 * points) may have left where the polled voxel-map builds keep their completion word (word 2; 4 * (levels - 1) + 2 for glim_amd_frame_create).
 * value = 0xffffffff stands for "the sequence number the context's NEXT polled build will wait for". */
int glim_amd_debug_scratch_poke(glim_amd_ctx* ctx, int32_t word, uint32_t value);
/* parity / debug only: the stable device radix sort behind the preprocessing (sorts by the low `bits` key bits; vals_in NULL = 0..n-1). */
int glim_amd_debug_sort_pairs(glim_amd_ctx* ctx, int64_t n, int32_t bits, const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out,
                              uint32_t* vals_out);

/* ---- timing loops measured inside the library (bench.py, tools/) -------------------------------------------------- */
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
/* Measurement aid: the pre-cull of a large general-form set: wavefront trips the pre-pass marked as unable to find a correspondence
 * (their 64-point chunk box, moved by the evaluation's pose, touches no occupied cell of the target's occupancy mask) and the trips that hold
 * points at all.  The pre-pass counts only while armed: reset != 0 zeroes the counters and arms it (the evaluations that follow are counted, at the price
 * of a few thousand atomics each); reset == 0 reads the sums and disarms it.  GLIM_AMD_ERR_UNSUPPORTED (and zeros) when the plan has no
 * pre-cull (small set, plane-form factors only, switch cull=0). */
int glim_amd_factor_set_cull_stats(glim_amd_factor_set* set, uint64_t* culled_trips, uint64_t* trips_with_points, int reset);
/* The same pattern with every iteration timed on its own (samples_us: `iters` entries) and `gap_us` of host busy-waiting between iterations -- the
 * optimiser's own work between two linearisations --, for latency percentiles while other threads load the device (bench.py
 * --workload odometry_under_load). */
int glim_amd_factor_set_profile_fresh_samples(glim_amd_ctx* ctx, int32_t n, const glim_amd_voxelmap* const* targets, const glim_amd_cloud* const* sources,
                                              const uint32_t* flags, const double* T_target_source, int iters, double gap_us, float* samples_us);
/* One Levenberg-Marquardt iteration as the optimisers drive it (sub_mapping.cpp:435-443, odometry_estimation_cpu.cpp:116-149): a synchronous
 * linearize() of the whole set (records expanded on the host) and a synchronous error() at the trial values, each timed over `iters` calls. */
int glim_amd_factor_set_profile_lm(glim_amd_factor_set* set, const double* T_target_source, int iters, float* ms_linearize, float* ms_error);
/* timing aid: `iters` calls of glim_amd_cloud_find_neighbors with this k; wall milliseconds per call, and the HIP-event duration (events on the call's own
 * stream) of the query-group kernel inside it -- the dominant kernel of the kNN-led workloads, whose roofline bench.py quotes (0 when another kernel
 * answered: clouds <= 2 048 points, forced knn_path / knn_kernel). */
int glim_amd_cloud_profile_neighbors(glim_amd_cloud* cloud, int k, int iters, float* ms_per_call, float* ms_qgroup_kernel);
/* timing aid: `iters` back-to-back glim_amd_overlap_batch calls with these arguments; microseconds per call, measured inside the library. */
int glim_amd_overlap_profile(glim_amd_ctx* ctx, int32_t num_queries, const int32_t* num_targets, const glim_amd_voxelmap* const* targets,
                             const double* T_target_source, const glim_amd_cloud* const* sources, int iters, float* us_per_call);

/* ---- state of the resident session, the plan cache and the last frame_create ---------------------------------------- */
/* parity / debug only: the device's resident session (repeated synchronous linearisations of a small factor list are served by a kernel that stays
 * on the device and takes its requests through host-mapped memory; it leaves by itself after `resident_idle_us` without a request): kernel
 * launches and requests served so far, whether one is alive right now. */
int glim_amd_debug_resident_stats(int device, uint64_t* launches, uint64_t* requests, int32_t* alive);
/* Device timeline of the resident session's LAST request: ends the session, reports (us / num_fields may be NULL / 0: switch only)
 * and leaves the stamps on (enable != 0) or off for the sessions that start afterwards.  Microseconds, device stamps (s_memrealtime, 10 ns) relative
 * to the moment the session's leader saw the request in host memory:
 *   [0] host clock: posting the request -> last record granule seen      [1] leader: poses re-published on the device
 *   [2..4] worker blocks: pose seen, min / median / max   [5..7] first row computed   [8..10] row granules published
 *   [11] finaliser of factor 0: pose seen   [12] every row of its factor summed   [13] record stored towards the host
 *   [14] worker blocks with a complete account   [15] = [13]: the device's share of [0]
 *   [16..18] worker blocks: point loop left (before the wave / block reduction), min / median / max
 *   [19] finaliser: its 32 group sums added   [20] its 3x3 blocks rotated (the record store follows)
 *   [21] NOT a time: the shader clock the session ran at between [11] and [13], MHz (s_memtime ticks per s_memrealtime microsecond)
 * GLIM_AMD_ERR_STATE: a request is in flight, or no session has run with the stamps on. */
#define GLIM_AMD_RESIDENT_TIMELINE_FIELDS 22
int glim_amd_debug_resident_timeline(int device, int enable, double* microseconds, int32_t num_fields);
/* ends the device's resident session now instead of letting it idle out (GLIM_AMD_ERR_STATE while a request is in flight). */
int glim_amd_debug_resident_stop(int device);
/* parity / debug only: factor plans this context has built for new factor lists, how many of them took over the buffers of the plan its full
 * cache was about to evict (a new list of the same shape: GLIM's odometry brings one per frame), idle plans cached right now. */
int glim_amd_debug_plan_stats(glim_amd_ctx* ctx, uint64_t* built, uint64_t* recycled, int32_t* cached);
/* parity / debug only: calls of this process that went PAST the library's block caches to the runtime so far -- hipMalloc, hipFree, hipHostMalloc --
 * and the bytes the calling thread's current device holds in its cache of freed blocks.  A steady-state loop should add none (a real allocation
 * stalls everything on the device, not only its caller: tools/odometry_frame_loop.cpp reports the count per slow frame).  Any pointer may be NULL. */
int glim_amd_debug_pool_stats(uint64_t* device_mallocs, uint64_t* device_frees, uint64_t* pinned_mallocs, uint64_t* cached_bytes);
/* measurement only: hipStreamQuery on every stream of the context (how many still have work: *busy).  tools/odometry_frame_loop.cpp uses it to
 * ask whether the runtime retires its finished commands when a loop that only ever POLLS completion words looks at its streams now and then. */
int glim_amd_debug_ctx_query_streams(glim_amd_ctx* ctx, int32_t* busy);
/* parity / debug only: host-side account of the calling thread's LAST one-submission glim_amd_frame_create, microseconds since its entry:
 * [0] cloud allocated, [1] staging block + stream allocations, [2] pull kernel launched, [3] host conversion done, [4] voxel-map kernels
 * enqueued, [5] completion word seen, [6] return (tools/odometry_frame_loop.cpp prints the medians). */
int glim_amd_debug_frame_stages(double* microseconds, int32_t num_fields);

/* ---- multi-device evaluation: measurement, knobs, virtual devices, failure injection ---------------------------------- */
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
/* The N > 1 path on a box with ONE GPU ("virtual devices"): like glim_amd_multi_create, but a device ordinal may be listed several times.  Every
 * entry gets its own context, host thread, shard, pieces, upload / collective streams and gathered array; RCCL refuses one device twice, so the
 * exchange of a piece is a same-device stand-in for the in-place ncclAllGather -- every entry copies its slot into the others' arrays on its
 * collective stream, behind the piece's event (distinct devices still go through ncclCommInitAll / ncclAllGather).  The same happens in
 * glim_amd_multi_create itself when GLIM_AMD_DIAG holds multi_virtual=1.  glim_amd_multi_info reports uses_rccl = 0 for such a handle. */
int glim_amd_debug_multi_create_virtual(const int32_t* devices, int32_t num_devices, glim_amd_multi** out);
/* failure injection, one shot: the NEXT glim_amd_multi_linearize fails on `device` (index into the handle's list) -- where = 1: before the host
 * barrier (its kernels are never enqueued; no device starts an exchange, every thread returns the error, the handle stays usable); where = 2:
 * inside its exchange of the last piece (the others are in theirs: the communicators are aborted and the handle is retired -- every later
 * evaluation returns GLIM_AMD_ERR_STATE).  where = 0 or device = -1 disarms. */
int glim_amd_debug_multi_inject_failure(glim_amd_multi* multi, int32_t device, int32_t where);
/* glim_amd_ctx_set_diag on every context of the handle (e.g. "ppt=16": the same points-per-thread rule -- hence the same per-factor block
 * partition and the same bits -- as an unsharded set evaluated with that switch) */
int glim_amd_debug_multi_set_diag(glim_amd_multi* multi, const char* key_values);
/* the DEVICE-resident gathered array of `device` after the exchange, copied back and put in factor order like glim_amd_multi_records: what a
 * device-side consumer on that device would read (tests: every device must hold every record, bit for bit) */
int glim_amd_debug_multi_gathered_download(glim_amd_multi* multi, int32_t device, int64_t first, int64_t count, double* compact29);

#ifdef __cplusplus
}
#endif
#endif /* GLIM_AMD_DIAG_H */
