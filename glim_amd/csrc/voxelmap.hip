// voxelmap.hip -- GaussianVoxelMapGPU equivalent (kernel group K3): lossless open-addressing voxel hash built on the device.
// Replaces gtsam_points::GaussianVoxelMapGPU(resolution, ...)::insert(frame) as called at
// src/glim/odometry/odometry_estimation_gpu.cpp:103-104, src/glim/mapping/sub_mapping.cpp:398-399 and
// src/glim/mapping/global_mapping.cpp:265-266,747-748.  Statistic per voxel = mean of member means and mean of member
// covariances, voxel identity = integer coordinate fast_floor(p * (1/resolution)).
//
// A second insert() into the same map adds to its voxels (the CPU voxel map's semantics; "incremental insert" below).
// Build (all on the context stream):
//   1. insert_keys   : every point CASes its packed 64-bit coordinate key into an over-sized scratch table (2N slots,
//                      never full) and the distinct keys are counted                        -> V
//   2. move_keys     : the V distinct keys are re-inserted into the final table of 3V two-way 128-byte buckets
//   3. accumulate    : every point adds its mean / covariance as 64-bit FIXED-POINT integers with atomics -- integer
//                      addition is associative, so the sums (and the map) are bit-reproducible whatever the atomic order
//   4. finalise      : sums / count in FP64, stored as FP32 in the bucket (keys + statistics share one 128-byte line)
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "device_math.hpp"
#include "internal.hpp"
#include "pull.hpp"
#include "scope_sync.hpp"

using namespace glim_amd;

namespace {

constexpr double MEAN_SCALE = 268435456.0;      // 2^28  (3.7e-9 m resolution, |sum| < 3.4e10 m)
constexpr double COV_SCALE = 68719476736.0;     // 2^36  (1.5e-11 resolution, |sum| < 1.3e8)
constexpr int ACC_STRIDE = 10;                  // 3 mean + 6 cov + count

__global__ __launch_bounds__(256) void fill_u64_kernel(unsigned long long* __restrict__ p, size_t n, unsigned long long v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ __launch_bounds__(256) void init_buckets_kernel(VoxelBucket* __restrict__ buckets, unsigned int n) {
  // one 16-byte store per lane: 8 lanes cover a 128-byte bucket
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n * 8) return;
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if ((i & 7) == 0) v = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);  // key[0] = key[1] = EMPTY_KEY
  reinterpret_cast<uint4*>(buckets)[i] = v;
}

// direct build: buckets initialised, accumulators and the two counters zeroed by ONE launch instead of a kernel and two memsets
// (acc16: the accumulator table as 16-byte words, 10 per bucket against the bucket's own 8)
__global__ __launch_bounds__(256) void init_tables_kernel(VoxelBucket* __restrict__ buckets, unsigned int n, uint4* __restrict__ acc16, size_t acc_words,
                                                          int* __restrict__ stats) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) stats[0] = stats[1] = stats[2] = stats[3] = 0;  // ([2]: arrival counter of frame_build_kernel's blocks; every stats block holds 4 ints)
  if (i < (size_t)n * 8) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if ((i & 7) == 0) v = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);  // key[0] = key[1] = EMPTY_KEY
    reinterpret_cast<uint4*>(buckets)[i] = v;
  }
  if (i < acc_words) acc16[i] = make_uint4(0u, 0u, 0u, 0u);
}

// The same clearing for a RECYCLED table on the side stream (recycle_table), with a completion word instead of an event: the block that arrives
// last publishes `done_seq` in host-mapped memory, and the host hands the table to a new map only after it has seen that word -- a load where the
// event cost a hipEventRecord per retired map and a hipEventQuery per look (a frame of the odometry made four records and four queries, ~10 us of
// runtime calls, and fed the runtime's bulk retirement of marker commands).  A FEW FAT blocks (grid-stride), ONE release per block: a release at
// agent scope writes the XCD's L2 back, and the first form of this kernel -- init_tables_kernel's one thread per 16 bytes, every thread fencing before
// the arrival -- put 2 048 such write-backs behind one another: 125 us per table, and the frame build beside it went from 62 to 187 us.  Here
// every wavefront waits for its own stores (vmcnt), the barrier collects the block, thread 0 releases once.  arrivals: device counter, left at zero.
constexpr int CLEAR_BLOCKS = 64;
__global__ __launch_bounds__(256) void clear_recycled_kernel(VoxelBucket* __restrict__ buckets, unsigned int n, uint4* __restrict__ acc16, size_t acc_words,
                                                             int* __restrict__ stats, int* __restrict__ arrivals, unsigned int* __restrict__ done_word,
                                                             unsigned int done_seq) {
  const size_t stride = (size_t)gridDim.x * blockDim.x, first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (first == 0) stats[0] = stats[1] = stats[2] = stats[3] = 0;
  for (size_t i = first; i < (size_t)n * 8; i += stride) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if ((i & 7) == 0) v = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);  // key[0] = key[1] = EMPTY_KEY
    reinterpret_cast<uint4*>(buckets)[i] = v;
  }
  for (size_t i = first; i < acc_words; i += stride) acc16[i] = make_uint4(0u, 0u, 0u, 0u);
  __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0): this wavefront's stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    const int ticket = __hip_atomic_fetch_add(arrivals, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);  // (release: the block's lines leave this XCD's L2)
    if (ticket == (int)gridDim.x - 1) {
      __hip_atomic_store(arrivals, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(done_word, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Wavefront-level grouping of equal keys: consecutive points of a scan usually fall into the same voxel, and 64 lanes hammering
// one table word with atomics serialise in the L2.  Every distinct key of the wavefront elects ONE leader lane; `group` is the
// ballot of the lanes sharing this lane's key.  The loop is wave-uniform (one trip per distinct key, ~10 on LiDAR scans).
// (Round 3 measured the alternative -- RUNS of equal keys reduced with a six-step segmented scan, one set of atomics per run -- slower:
// accumulate 27 -> 43 us at 131 072 points: a key that comes back later in the wavefront costs a second set of ten 64-bit atomics, and the
// scan's 114 cross-lane moves are paid whatever the run lengths.)
__device__ __forceinline__ bool wave_group_by_key(unsigned long long key, bool valid, unsigned long long& group) {
  const int lane = threadIdx.x & 63;
  unsigned long long remaining = __ballot(valid);
  bool leader = false;
  group = 0ull;
  while (remaining) {
    const int l = __ffsll((long long)remaining) - 1;
    const unsigned int klo = __shfl((unsigned int)key, l, 64), khi = __shfl((unsigned int)(key >> 32), l, 64);
    const unsigned long long kl = ((unsigned long long)khi << 32) | klo;
    const bool mine = valid && key == kl;
    const unsigned long long same = __ballot(mine);
    if (mine) {
      group = same;
      leader = (lane == l);
    }
    remaining &= ~same;
  }
  return leader;
}

// stats[0] = distinct keys, stats[1] = points whose coordinate does not fit the 21-bit key range
__global__ __launch_bounds__(256) void insert_keys_kernel(int n, const float4* __restrict__ pts, double inv_res,
                                                          unsigned long long* __restrict__ tkeys, unsigned int tmask,
                                                          unsigned long long* __restrict__ pkeys, int* __restrict__ stats) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long key = EMPTY_KEY;
  if (i < n) {
    const float4 p = pts[i];
    key = voxel_key((double)p.x, (double)p.y, (double)p.z, inv_res);
    pkeys[i] = key;
    if (key == EMPTY_KEY) atomicAdd(&stats[1], 1);
  }
  unsigned long long group;
  if (!wave_group_by_key(key, key != EMPTY_KEY, group)) return;  // one CAS chain per distinct key of the wavefront
  unsigned int s = hash_key(key) & tmask;
  for (;;) {
    const unsigned long long prev = atomicCAS(&tkeys[s], EMPTY_KEY, key);
    if (prev == EMPTY_KEY) {
      atomicAdd(&stats[0], 1);
      return;
    }
    if (prev == key) return;
    s = (s + 1) & tmask;
  }
}

// Re-insert the distinct keys into the final bucket table: way 0, then way 1, then the next bucket.
__global__ __launch_bounds__(256) void move_keys_kernel(const unsigned long long* __restrict__ tkeys, unsigned int tsize,
                                                        VoxelBucket* __restrict__ buckets, unsigned int num_buckets) {
  const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= tsize) return;
  const unsigned long long key = tkeys[i];
  if (key == EMPTY_KEY) return;
  unsigned int b = bucket_of(key, num_buckets);
  for (;;) {
    if (atomicCAS(&buckets[b].key[0], EMPTY_KEY, key) == EMPTY_KEY) return;  // scratch keys are distinct: no equal-key case
    if (atomicCAS(&buckets[b].key[1], EMPTY_KEY, key) == EMPTY_KEY) return;
    b = (b + 1 == num_buckets) ? 0u : b + 1;
  }
}

__device__ __forceinline__ long long shfl_ll(long long v, int src) {
  const unsigned int lo = __shfl((unsigned int)v, src, 64), hi = __shfl((unsigned int)((unsigned long long)v >> 32), src, 64);
  return (long long)(((unsigned long long)hi << 32) | lo);
}

// Every point contributes 9 fixed-point sums + a count to its voxel.  Lanes of a wavefront that share a voxel are summed in
// registers first (integer adds: order-free, so still bit-reproducible) and only the group leader touches memory: ~10 leaders x 10
// atomics per wavefront instead of 64 x 10.
__global__ __launch_bounds__(256) void accumulate_kernel(int n, const float4* __restrict__ pts, const float4* __restrict__ covA,
                                                         const float2* __restrict__ covB, const unsigned long long* __restrict__ pkeys,
                                                         const VoxelBucket* __restrict__ buckets, unsigned int num_buckets,
                                                         long long* __restrict__ acc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long key = EMPTY_KEY;
  long long v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (i < n) {
    key = pkeys[i];
    const float4 p = pts[i];
    const float4 a = covA[i];
    const float2 b = covB[i];
    v[0] = __double2ll_rn((double)p.x * MEAN_SCALE);
    v[1] = __double2ll_rn((double)p.y * MEAN_SCALE);
    v[2] = __double2ll_rn((double)p.z * MEAN_SCALE);
    v[3] = __double2ll_rn((double)a.x * COV_SCALE);
    v[4] = __double2ll_rn((double)a.y * COV_SCALE);
    v[5] = __double2ll_rn((double)a.z * COV_SCALE);
    v[6] = __double2ll_rn((double)a.w * COV_SCALE);
    v[7] = __double2ll_rn((double)b.x * COV_SCALE);
    v[8] = __double2ll_rn((double)b.y * COV_SCALE);
  }
  unsigned long long group;
  const bool leader = wave_group_by_key(key, key != EMPTY_KEY, group);
  // leaders gather their group's values member by member (every lane takes part in the shuffles; trips = largest group)
  long long sum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long rest = leader ? group : 0ull;
  while (__ballot(rest != 0ull)) {
    const int src = rest ? (__ffsll((long long)rest) - 1) : 0;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      const long long t = shfl_ll(v[j], src);
      if (rest) sum[j] += t;
    }
    rest &= rest - 1ull;
  }
  if (!leader) return;
  const int s = find_slot(buckets, num_buckets, key);
  if (s < 0) return;
  long long* dst = acc + (size_t)s * ACC_STRIDE;
#pragma unroll
  for (int j = 0; j < 9; j++) atomicAdd(reinterpret_cast<unsigned long long*>(dst + j), (unsigned long long)sum[j]);
  atomicAdd(reinterpret_cast<unsigned long long*>(dst + 9), (unsigned long long)__popcll(group));
}

// Direct build (fresh maps whose table size is known before the voxels are counted, glim_amd_voxelmap_insert): the keys go straight into the
// FINAL table and the sums to the accumulators of the slot a key has claimed or found -- a slot is usable from the moment its key is in place,
// no matter which block put it there.  No per-point key array, one host synchronise per map, three launches.
// Round 4: the 256 points of a block are first summed per voxel in an LDS hash table (512 slots; integer LDS atomics, order-free like the
// global ones), and only the block's DISTINCT voxels -- about 40 of a scan's 256 consecutive points -- go to the global table: one claim
// and ten atomics each.  Round 3 grouped equal keys per wavefront with ballots and gathered a group's values with shuffles: 9 values x 2 words
// x (largest group) permutes per wavefront, a 1 200-instruction chain on 2 048 wavefronts that were alone on their SIMDs (32-44 us at
// 131 072 points).
// stats[0] = distinct keys, stats[1] = points whose coordinate does not fit the 21-bit key range
constexpr int LDS_SLOTS = 512;
struct BuildLds {
  unsigned long long key[LDS_SLOTS];
  unsigned long long acc[ACC_STRIDE][LDS_SLOTS];  // value-major: the lanes of a wavefront that add value j to different slots hit different banks
  int claimed;                                    // voxels this block was the first to put into the global table: ONE atomic per block on the shared counter
};
// one block's (up to) 256 points into one map's table; all lanes of the block must call (valid = this lane holds a point)
__device__ __forceinline__ void build_block(BuildLds& L, bool valid, const float4& p, const float4& a, const float2& b, double inv_res,
                                            VoxelBucket* __restrict__ buckets, unsigned int num_buckets, long long* __restrict__ acc, int* __restrict__ stats) {
  if (threadIdx.x == 0) L.claimed = 0;
  for (int s = threadIdx.x; s < LDS_SLOTS; s += 256) {
    L.key[s] = EMPTY_KEY;
#pragma unroll
    for (int j = 0; j < ACC_STRIDE; j++) L.acc[j][s] = 0ull;
  }
  __syncthreads();
  if (valid) {
    const unsigned long long key = voxel_key((double)p.x, (double)p.y, (double)p.z, inv_res);
    if (key == EMPTY_KEY) {
      atomicAdd(&stats[1], 1);
    } else {
      unsigned int s = hash_key(key) & (LDS_SLOTS - 1);
      for (;;) {  // at most 256 of the 512 slots are ever taken
        const unsigned long long prev = atomicCAS(&L.key[s], EMPTY_KEY, key);
        if (prev == EMPTY_KEY || prev == key) break;
        s = (s + 1) & (LDS_SLOTS - 1);
      }
      atomicAdd(&L.acc[0][s], (unsigned long long)__double2ll_rn((double)p.x * MEAN_SCALE));
      atomicAdd(&L.acc[1][s], (unsigned long long)__double2ll_rn((double)p.y * MEAN_SCALE));
      atomicAdd(&L.acc[2][s], (unsigned long long)__double2ll_rn((double)p.z * MEAN_SCALE));
      atomicAdd(&L.acc[3][s], (unsigned long long)__double2ll_rn((double)a.x * COV_SCALE));
      atomicAdd(&L.acc[4][s], (unsigned long long)__double2ll_rn((double)a.y * COV_SCALE));
      atomicAdd(&L.acc[5][s], (unsigned long long)__double2ll_rn((double)a.z * COV_SCALE));
      atomicAdd(&L.acc[6][s], (unsigned long long)__double2ll_rn((double)a.w * COV_SCALE));
      atomicAdd(&L.acc[7][s], (unsigned long long)__double2ll_rn((double)b.x * COV_SCALE));
      atomicAdd(&L.acc[8][s], (unsigned long long)__double2ll_rn((double)b.y * COV_SCALE));
      atomicAdd(&L.acc[9][s], 1ull);
    }
  }
  __syncthreads();
  for (int s = threadIdx.x; s < LDS_SLOTS; s += 256) {
    const unsigned long long key = L.key[s];
    if (key == EMPTY_KEY) continue;
    unsigned int bk = bucket_of(key, num_buckets);
    int slot = -1;
    while (slot < 0) {
#pragma unroll
      for (int w = 0; w < 2; w++) {
        if (slot >= 0) continue;
        const unsigned long long prev = atomicCAS(&buckets[bk].key[w], EMPTY_KEY, key);
        if (prev == EMPTY_KEY) atomicAdd(&L.claimed, 1);
        if (prev == EMPTY_KEY || prev == key) slot = (int)(2u * bk + (unsigned int)w);
      }
      bk = (bk + 1 == num_buckets) ? 0u : bk + 1;
    }
    long long* dst = acc + (size_t)slot * ACC_STRIDE;
#pragma unroll
    for (int j = 0; j < ACC_STRIDE; j++) atomicAdd(reinterpret_cast<unsigned long long*>(dst + j), L.acc[j][s]);
  }
  __syncthreads();
  if (threadIdx.x == 0 && L.claimed) atomicAdd(&stats[0], L.claimed);
}

__global__ __launch_bounds__(256) void build_direct_kernel(int n, const float4* __restrict__ pts, const float4* __restrict__ covA,
                                                           const float2* __restrict__ covB, double inv_res, VoxelBucket* __restrict__ buckets,
                                                           unsigned int num_buckets, long long* __restrict__ acc, int* __restrict__ stats) {
  __shared__ BuildLds L;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float4 p = make_float4(0.f, 0.f, 0.f, 0.f), a = p;
  float2 b = make_float2(0.f, 0.f);
  if (i < n) {
    p = pts[i];
    a = covA[i];
    b = covB[i];
  }
  build_block(L, i < n, p, a, b, inv_res, buckets, num_buckets, acc, stats);
}

// The maps of a FRAME in the launch that pulls the frame's cloud over (glim_amd_frame_create): a block waits for its piece of the staging block
// (pull.hpp), pulls its 256 points into the cloud's arrays and streams, and -- the values are in its registers -- sums them into every level's
// table straight away.  Pull kernel + one build kernel per level were 1 + L dependent launches in a row; a dependent launch costs the stream
// ~5 us before its first wavefront runs, more than any of these kernels computes on a 10 000-pt frame.
constexpr int FRAME_MAX_LEVELS = 8;
struct FrameLevels {
  int count = 0;
  double inv_res[FRAME_MAX_LEVELS], res[FRAME_MAX_LEVELS];
  VoxelBucket* buckets[FRAME_MAX_LEVELS];
  VoxelBucket* view[FRAME_MAX_LEVELS];
  long long* acc[FRAME_MAX_LEVELS];
  int* stats[FRAME_MAX_LEVELS];
  unsigned int nb[FRAME_MAX_LEVELS];
};
__global__ __launch_bounds__(256) void frame_build_kernel(const PullArgs pa, const FrameLevels lv, int* __restrict__ host_view, unsigned int poll_seq) {
  __shared__ BuildLds L;
  __shared__ int s_ok;
  if (!pull_wait(pa, &s_ok)) return;  // (gave up: no arrival either -- the completion word is never written, the host falls back to a synchronise)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float4 p = make_float4(0.f, 0.f, 0.f, 0.f), a = p, v = p;
  float2 b = make_float2(0.f, 0.f);
  bool bad = false;
  if (i < pa.n) bad = pull_point(pa, i, p, a, b, v);
  pull_report(pa, bad);
  for (int k = 0; k < lv.count; k++) {
    build_block(L, i < pa.n, p, a, b, lv.inv_res[k], lv.buckets[k], lv.nb[k], lv.acc[k], lv.stats[k]);
    __syncthreads();  // (the next level re-initialises the LDS table)
  }
  // The block that arrives LAST hands the host what it waits for -- every level's voxel count and range flag, then the completion word
  // (system-scope release) -- instead of the first thread of the records kernel behind this one: the counts are final when every block has
  // added its share, and the host need not wait for a dependent launch to start (2 us of a 10 000-pt frame).  (Every block's counter updates, and its report of a
  // point off the plane form, are ordered before its arrival by the fence; the arrival counter is left at zero for the table's next life.)
  if (poll_seq) {  // EVERY wavefront's stores (cloud arrays, factor streams, accumulator atomics) are performed before the block arrives: the
    // barrier alone orders the waves, it does not wait for their outstanding stores, and the host frees one stream family and lets
    // readers of other streams in the moment it sees the completion word.  Each wavefront waits for ITS stores to be acknowledged (vmcnt: they are
    // in this XCD's L2 then); the ONE system-scope fence of thread 0 below writes that L2 back.  (Round 5 had every thread execute __threadfence():
    // an L2 write-back per wavefront, which is what made the first completion-word form of the table-clearing kernel take 125 us.)
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
  }
  if (poll_seq && threadIdx.x == 0) {
    __threadfence_system();
    int* counter = lv.stats[0] + 2;
    const int ticket = __hip_atomic_fetch_add(counter, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket == (int)gridDim.x - 1) {
      for (int j = 0; j < lv.count; j++) {
        host_view[4 * j] = __hip_atomic_load(lv.stats[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        host_view[4 * j + 1] = __hip_atomic_load(lv.stats[j] + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(reinterpret_cast<unsigned int*>(host_view) + 4 * (lv.count - 1) + 2, poll_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---- incremental insert (a second insert() into a map: GaussianVoxelMapCPU semantics, gtsam_points GaussianVoxel::add re-opens a finalised voxel
// with mean *= n, cov *= n before accumulating -- SURVEY.md App. B.4; GLIM's GPU callers insert once per map, the CPU odometry inserts per frame,
// odometry_estimation_cpu.cpp:66-67,189).  The map is rebuilt: the keys of the old voxels and of the new cloud go into one scratch table (counts
// the voxels of the union), the old voxels are re-opened into the new table's fixed-point accumulators, then the new points accumulate as usual.
// one thread per (old bucket, way): the key of every old voxel into the scratch table
// expire_before (LRU eviction, 0 = none): an old voxel whose last-touch stamp is below it is NOT re-inserted -- if this insert's points touch it, their
// own key insertion (insert_keys_kernel, earlier on the stream) has put the key there already and the voxel survives with its contents
__global__ __launch_bounds__(256) void reinsert_old_keys_kernel(const VoxelBucket* __restrict__ old, unsigned int old_buckets, unsigned long long* __restrict__ tkeys,
                                                                unsigned int tmask, int* __restrict__ stats, int expire_before) {
  const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * old_buckets) return;
  const unsigned long long key = old[i >> 1].key[i & 1];
  if (key == EMPTY_KEY) return;
  if (expire_before > 0 && __float_as_int(old[i >> 1].rec[i & 1][10]) < expire_before) return;
  unsigned int s = hash_key(key) & tmask;
  for (;;) {
    const unsigned long long prev = atomicCAS(&tkeys[s], EMPTY_KEY, key);
    if (prev == EMPTY_KEY) {
      atomicAdd(&stats[0], 1);
      return;
    }
    if (prev == key) return;
    s = (s + 1) & tmask;
  }
}

// one thread per (old bucket, way): count x (mean, covariance) of the old voxel, as fixed-point sums, into its slot of the new table
// lru (optional, 2 ints per slot of the new table, zeroed): the old voxel's count and last-touch stamp -- finalize_kernel tells a voxel that this
// insert's points touched (count grew) from one that was only carried over
__global__ __launch_bounds__(256) void reopen_old_voxels_kernel(const VoxelBucket* __restrict__ old, unsigned int old_buckets, double res,
                                                                const VoxelBucket* __restrict__ buckets, unsigned int num_buckets, long long* __restrict__ acc,
                                                                int2* __restrict__ lru) {
  const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * old_buckets) return;
  const unsigned long long key = old[i >> 1].key[i & 1];
  if (key == EMPTY_KEY) return;
  const float* r = old[i >> 1].rec[i & 1];
  const long long cnt = (long long)__float_as_int(r[9]);
  int cx, cy, cz;
  unpack_key(key, cx, cy, cz);
  const double c = (double)cnt;
  const int s = find_slot(buckets, num_buckets, key);
  if (s < 0) return;  // (an evicted voxel: its key was not carried over)
  if (lru) lru[s] = make_int2((int)cnt, __float_as_int(r[10]));
  long long* dst = acc + (size_t)s * ACC_STRIDE;
  const double m[3] = {(double)r[0] + ((double)cx + 0.5) * res, (double)r[1] + ((double)cy + 0.5) * res, (double)r[2] + ((double)cz + 0.5) * res};
#pragma unroll
  for (int j = 0; j < 3; j++) atomicAdd(reinterpret_cast<unsigned long long*>(dst + j), (unsigned long long)__double2ll_rn(m[j] * c * MEAN_SCALE));
#pragma unroll
  for (int j = 0; j < 6; j++) atomicAdd(reinterpret_cast<unsigned long long*>(dst + 3 + j), (unsigned long long)__double2ll_rn((double)r[3 + j] * c * COV_SCALE));
  atomicAdd(reinterpret_cast<unsigned long long*>(dst + 9), (unsigned long long)cnt);
}

// The plane view of a finished table (internal.hpp glim_amd_voxelmap::buckets_sm): one thread per (bucket, way) of num_buckets + 1 buckets.
// For a plane-form source C_A = I - 0.999 n n^T, so  C_B + R C_A R^T = (C_B + I) - 0.999 m m^T  (m = R n)  and, by Sherman-Morrison,
//   M = (C_B + R C_A R^T)^-1 = A_B + 0.999 (A_B m)(A_B m)^T / (1 - 0.999 m^T A_B m),   A_B = (C_B + I)^-1:
// the per-point 3x3 inverse of the factor kernel becomes one matrix-vector product, one dot product, one reciprocal and one rank-1 update.
// A_B depends on the voxel alone, so it is inverted HERE, once per voxel, in FP64 (C_B + I has eigenvalues in [1, 2]: perfectly conditioned).
// r: a finished record of the plain table (FP32 mean | C_B | count); o: the same voxel's record of the plane view.  Always computed from the
// FP32 record, so the view is the same bits whether finalize_kernel writes it with the map or plane_view_kernel adds it later.
__device__ __forceinline__ void plane_record(const float* __restrict__ r, float* __restrict__ o) {
  const double s00 = (double)r[3] + 1.0, s01 = (double)r[4], s02 = (double)r[5], s11 = (double)r[6] + 1.0, s12 = (double)r[7], s22 = (double)r[8] + 1.0;
  const double k00 = s11 * s22 - s12 * s12, k01 = s02 * s12 - s01 * s22, k02 = s01 * s12 - s02 * s11;
  const double idet = 1.0 / (s00 * k00 + s01 * k01 + s02 * k02);
  o[0] = r[0];
  o[1] = r[1];
  o[2] = r[2];
  o[3] = (float)(k00 * idet);
  o[4] = (float)(k01 * idet);
  o[5] = (float)(k02 * idet);
  o[6] = (float)((s00 * s22 - s02 * s02) * idet);
  o[7] = (float)((s01 * s02 - s00 * s12) * idet);
  o[8] = (float)((s00 * s11 - s01 * s01) * idet);
  o[9] = r[9];
  o[10] = 0.f;
  o[11] = 0.f;
}

__global__ __launch_bounds__(256) void plane_view_kernel(const VoxelBucket* __restrict__ src, VoxelBucket* __restrict__ dst, unsigned int num_buckets) {
  const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * (num_buckets + 1)) return;
  const unsigned int b = i >> 1, w = i & 1;
  float* o = dst[b].rec[w];
  if (w == 0) {
    dst[b].pad[0] = dst[b].pad[1] = dst[b].pad[2] = dst[b].pad[3] = 0;
  }
  const unsigned long long key = b < num_buckets ? src[b].key[w] : EMPTY_KEY;
  dst[b].key[w] = key;
  if (key == EMPTY_KEY) {
#pragma unroll
    for (int k = 0; k < 12; k++) o[k] = 0.f;
    return;
  }
  plane_record(src[b].rec[w], o);
}

// ---- occupancy mask (the factor kernel's pre-cull, vgicp.hip) --------------------------------------------------------------------------------
// bbox[0..2] = min, bbox[3..5] = max voxel coordinate over the occupied (bucket, way) slots (initialised to INT_MAX / INT_MIN by the host)
__global__ __launch_bounds__(256) void occ_bbox_kernel(const VoxelBucket* __restrict__ buckets, unsigned int num_buckets, int* __restrict__ bbox) {
  const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  if (i < 2 * num_buckets) {
    const unsigned long long key = buckets[i >> 1].key[i & 1];
    if (key != EMPTY_KEY) {
      unpack_key(key, lo[0], lo[1], lo[2]);
      hi[0] = lo[0];
      hi[1] = lo[1];
      hi[2] = lo[2];
    }
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1)
#pragma unroll
    for (int a = 0; a < 3; a++) {
      lo[a] = min(lo[a], __shfl_xor(lo[a], m, 64));
      hi[a] = max(hi[a], __shfl_xor(hi[a], m, 64));
    }
  if ((threadIdx.x & 63) == 0 && lo[0] <= hi[0]) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      atomicMin(&bbox[a], lo[a]);
      atomicMax(&bbox[3 + a], hi[a]);
    }
  }
}
// bit ((v - org) >> shift) of the mask for every occupied voxel v; x along the bits of a row, rows ordered (z, y)
__global__ __launch_bounds__(256) void occ_fill_kernel(const VoxelBucket* __restrict__ buckets, unsigned int num_buckets, int ox, int oy, int oz, int shift, int dy,
                                                       int row_words, unsigned int* __restrict__ occ) {
  const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * num_buckets) return;
  const unsigned long long key = buckets[i >> 1].key[i & 1];
  if (key == EMPTY_KEY) return;
  int cx, cy, cz;
  unpack_key(key, cx, cy, cz);
  const int x = (cx - ox) >> shift, y = (cy - oy) >> shift, z = (cz - oz) >> shift;
  atomicOr(&occ[((size_t)z * dy + y) * row_words + (x >> 5)], 1u << (x & 31));
}

// one thread per (bucket, way)
// stats / host_stats (direct build): the voxel count and the range flag of the key insertion, handed to the host through mapped pinned
// memory by this last launch -- the call then needs a stream synchronise only, no device-to-host copy
// view (optional; launch 2 * (num_buckets + 1) threads then): the plane view of the table (plane_record), written with the map -- every way
// of every bucket, so it needs no clearing, plus the all-zero bucket behind the table.
// slot i (= 2 * bucket + way) of a table: sums -> record (and the plane view's record)
__device__ __forceinline__ void finalize_slot(unsigned int i, VoxelBucket* __restrict__ buckets, unsigned int num_buckets, const long long* __restrict__ acc, double res,
                                              VoxelBucket* __restrict__ view, const int2* __restrict__ lru, int lru_stamp) {
  if (i >= 2 * num_buckets + (view ? 2u : 0u)) return;
  const unsigned int b = i >> 1, w = i & 1;
  const unsigned long long key = b < num_buckets ? buckets[b].key[w] : EMPTY_KEY;
  if (view) {
    view[b].key[w] = key;
    if (w == 0) view[b].pad[0] = view[b].pad[1] = view[b].pad[2] = view[b].pad[3] = 0;
    if (key == EMPTY_KEY) {
#pragma unroll
      for (int k = 0; k < 12; k++) view[b].rec[w][k] = 0.f;
    }
  }
  if (key == EMPTY_KEY) return;
  const long long* a = acc + (size_t)i * ACC_STRIDE;
  const long long cnt = a[9];
  const double inv_n = 1.0 / (double)cnt;
  const double im = inv_n / MEAN_SCALE, ic = inv_n / COV_SCALE;
  int cx, cy, cz;
  unpack_key(key, cx, cy, cz);
  float* r = buckets[b].rec[w];
  // mean relative to the voxel centre (see VoxelBucket)
  r[0] = (float)((double)a[0] * im - ((double)cx + 0.5) * res);
  r[1] = (float)((double)a[1] * im - ((double)cy + 0.5) * res);
  r[2] = (float)((double)a[2] * im - ((double)cz + 0.5) * res);
  r[3] = (float)((double)a[3] * ic);  // c00
  r[4] = (float)((double)a[4] * ic);  // c01
  r[5] = (float)((double)a[5] * ic);  // c02
  r[6] = (float)((double)a[6] * ic);  // c11
  r[7] = (float)((double)a[7] * ic);  // c12
  r[8] = (float)((double)a[8] * ic);  // c22
  r[9] = __int_as_float((int)cnt);
  // slot 10: the insert counter of the last insert that touched the voxel (LRU maps; 0 = the first insert, which is also what every other map holds)
  int stamp = lru_stamp;
  if (lru) {
    const int2 o = lru[i];
    if ((long long)o.x == cnt) stamp = o.y;  // carried over untouched: keeps its stamp
  }
  r[10] = __int_as_float(stamp);
  r[11] = 0.f;
  if (view) plane_record(r, view[b].rec[w]);
}

__global__ __launch_bounds__(256) void finalize_kernel(VoxelBucket* __restrict__ buckets, unsigned int num_buckets,
                                                       const long long* __restrict__ acc, double res, const int* __restrict__ stats,
                                                       int* __restrict__ host_stats, VoxelBucket* __restrict__ view, const int2* __restrict__ lru = nullptr,
                                                       int lru_stamp = 0, unsigned int poll_seq = 0u) {
  const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && host_stats) {
    host_stats[0] = stats[0];
    host_stats[1] = stats[1];
    // poll_seq: the host spins on word 2 and returns to its caller while this kernel is still writing records (system-scope release: the two
    // words above are visible with it); 0: the host synchronises the stream instead
    if (poll_seq) __hip_atomic_store(reinterpret_cast<unsigned int*>(host_stats) + 2, poll_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  finalize_slot(i, buckets, num_buckets, acc, res, view, lru, lru_stamp);
}

// every level of a frame in ONE launch (blockIdx.y = level).  host_view: 4 words per level -- voxels, points out of range, (last level only)
// the completion word.  The first thread of the grid writes EVERY level's counts and then the completion word (system-scope release): the
// build kernel is complete when this one starts (the stream runs in order), so the counts are final, and the host returns while the
// records are still being written (ready_event).
__global__ __launch_bounds__(256) void frame_finalize_kernel(const FrameLevels lv, int* __restrict__ host_view, unsigned int poll_seq) {
  const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = (int)blockIdx.y;
  if (i == 0 && k == 0 && host_view) {  // (null: the build kernel's last block has written them, and the completion word)
    for (int j = 0; j < lv.count; j++) {
      host_view[4 * j] = lv.stats[j][0];
      host_view[4 * j + 1] = lv.stats[j][1];
    }
    if (poll_seq) __hip_atomic_store(reinterpret_cast<unsigned int*>(host_view) + 4 * (lv.count - 1) + 2, poll_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  finalize_slot(i, lv.buckets[k], lv.nb[k], lv.acc[k], lv.res[k], lv.view[k], nullptr, 0);
}

// voxels-per-point ratio of the last map built at a resolution class (sizes the direct build of the next one)
void remember_voxel_ratio(glim_amd_ctx* ctx, int res_class, double ratio) {
  for (auto& h : ctx->voxel_ratio_hints)
    if (h.first == res_class) {
      h.second = ratio;
      return;
    }
  if (ctx->voxel_ratio_hints.size() >= 32) ctx->voxel_ratio_hints.erase(ctx->voxel_ratio_hints.begin());
  ctx->voxel_ratio_hints.emplace_back(res_class, ratio);
}

// table sizes in 8 steps per octave: the maps of consecutive frames (whose voxel counts differ by a few per cent) get tables of the SAME
// size, so that the memory pool hands the previous frame's blocks back instead of missing on a slightly larger request
unsigned long long round_buckets(unsigned long long nb) {
  if (nb <= 64) return nb;
  unsigned long long step = 1;
  while ((step << 4) <= nb) step <<= 1;  // step = 2^(floor(log2 nb) - 3)
  return std::min<unsigned long long>(((nb + step - 1) / step) * step, 1ull << 25);
}

unsigned int next_pow2(unsigned long long v) {
  unsigned long long p = 1;
  while (p < v) p <<= 1;
  return (unsigned int)p;
}

// ---- pre-cleared tables --------------------------------------------------------------------------------------------------------------------
// A direct build starts with a launch that only writes EMPTY keys and zeros (init_tables_kernel: 3.5 us at 10 000 points, 6.5 us at 131 072,
// plus a launch boundary) -- a seventh of a map's wall time.  Maps come and go at frame rate (the odometry builds two per frame and drops two
// when a frame leaves its window) in tables of a few recurring sizes (round_buckets), so a table that a destroyed map hands back is cleared on a
// side stream at once, together with a matching block of accumulators and counters, and kept for the next map of that size: its build then
// starts with the keys.  The host takes a table only after it has SEEN the clearing kernel's event complete.
// Events without timing, recycled per device: a frame of the odometry makes and drops four of them (two maps' ready events, two cleared tables'),
// and create + destroy are runtime calls of about a microsecond each.  An event goes back only when its last record has been seen complete or no
// longer matters (a later record replaces it; a wait already enqueued keeps the record it was given).
struct EventPool {
  std::mutex mu;
  std::vector<hipEvent_t> free_events[16];
};
EventPool& event_pool() {
  static EventPool* pool = new EventPool();  // leaked on purpose, like the memory pools
  return *pool;
}
// `dev` = the device the event belongs to (the owner's, passed explicitly: the CURRENT device of a destroying or worker thread may be another
// one, and an event parked in the wrong device's list later fails its hipEventRecord there -- ADVICE r5)
hipError_t event_get(int dev, hipEvent_t* e) {
  if (dev >= 0 && dev < 16) {
    EventPool& P = event_pool();
    std::lock_guard<std::mutex> lock(P.mu);
    if (!P.free_events[dev].empty()) {
      *e = P.free_events[dev].back();
      P.free_events[dev].pop_back();
      return hipSuccess;
    }
  }
  return hipEventCreateWithFlags(e, hipEventDisableTiming);
}
void event_put(int dev, hipEvent_t e) {
  if (!e) return;
  if (dev >= 0 && dev < 16) {
    EventPool& P = event_pool();
    std::lock_guard<std::mutex> lock(P.mu);
    if (P.free_events[dev].size() < 64) {
      P.free_events[dev].push_back(e);
      return;
    }
  }
  (void)hipEventDestroy(e);
}

struct ClearedTable {
  VoxelBucket* buckets = nullptr;
  long long* acc = nullptr;
  int* stats = nullptr;
  unsigned int nb = 0;
  int slot = -1;              // completion word of its clearing kernel (ClearedCache::h_done[slot]) ...
  unsigned int done_seq = 0;  // ... and the value that word takes when the kernel is complete
};
constexpr int CLEARED_SLOTS = 32;
struct ClearedCache {
  std::mutex mu;
  std::vector<ClearedTable> tables;
  hipStream_t stream = nullptr;
  size_t bytes = 0;
  // completion words of the clearing kernels (host-mapped, one per table in flight or cached) and the kernels' arrival counter (device)
  unsigned int *h_done = nullptr, *h_done_dev = nullptr;
  int* d_arrivals = nullptr;
  bool slot_used[CLEARED_SLOTS] = {};
  unsigned int seq = 0;
};
constexpr size_t CLEARED_MAX_TABLES = 8, CLEARED_MAX_BYTES = 256ull << 20, CLEARED_MAX_TABLE_BYTES = 48ull << 20;
ClearedCache& cleared_cache(int device) {
  static ClearedCache* caches = new ClearedCache[64];  // leaked on purpose, like the memory pools
  return caches[(device >= 0 && device < 64) ? device : 0];
}
size_t cleared_bytes(unsigned int nb) { return (size_t)nb * (sizeof(VoxelBucket) + 2 * ACC_STRIDE * sizeof(long long)); }

// true when the clearing kernel of `t` has published its completion word (acquire)
bool cleared_done(const ClearedCache& C, const ClearedTable& t) {
  if (t.slot < 0 || *reinterpret_cast<const volatile unsigned int*>(C.h_done + t.slot) != t.done_seq) return false;
  std::atomic_thread_fence(std::memory_order_acquire);
  return true;
}

// the table of a map that is going away (nobody reads it any more: the caller has quiesced the device): cleared for re-use, or back to the pool
void recycle_table(int device, VoxelBucket* buckets, unsigned int nb) {
  ClearedCache& C = cleared_cache(device);
  const size_t bytes = cleared_bytes(nb);
  ClearedTable t;
  bool keep = nb >= 16 && bytes <= CLEARED_MAX_TABLE_BYTES;
  if (keep) {
    // room is made by the OLDEST tables (sizes nobody has asked for since): a cache full of yesterday's sizes must not keep today's out
    std::vector<ClearedTable> evicted;
    {
      std::lock_guard<std::mutex> lock(C.mu);
      while (!C.tables.empty() && (C.tables.size() >= CLEARED_MAX_TABLES || C.bytes + bytes > CLEARED_MAX_BYTES)) {
        evicted.push_back(C.tables.front());
        C.bytes -= cleared_bytes(C.tables.front().nb);
        C.tables.erase(C.tables.begin());
      }
      if (!C.stream && hipStreamCreateWithFlags(&C.stream, hipStreamNonBlocking) != hipSuccess) keep = false;
    }
    // once per device: the completion words (host-mapped) and the clearing kernels' arrival counter.  Allocated OUTSIDE the cache's lock (a pool
    // that runs out of memory trims this very cache), installed under it.
    if (keep && !C.h_done) {
      unsigned int *h = nullptr, *hd = nullptr;
      int* d = nullptr;
      bool ok = pinned_malloc(&h, CLEARED_SLOTS * sizeof(unsigned int)) == hipSuccess && pool_malloc(&d, sizeof(int)) == hipSuccess &&
                hipMemsetAsync(d, 0, sizeof(int), C.stream) == hipSuccess && hipHostGetDevicePointer(reinterpret_cast<void**>(&hd), h, 0) == hipSuccess;
      if (ok) {
        memset(h, 0, CLEARED_SLOTS * sizeof(unsigned int));
        std::lock_guard<std::mutex> lock(C.mu);
        if (!C.h_done) {
          C.h_done_dev = hd;
          C.d_arrivals = d;
          C.h_done = h;
          h = nullptr;
          d = nullptr;
        }
      } else {
        (void)hipGetLastError();
        keep = false;
      }
      if (h) (void)pinned_free(h);  // (not installed: another thread was first, or a step failed)
      if (d) {
        (void)hipStreamSynchronize(C.stream);  // (its memset may still be on the way)
        (void)pool_free(d);
      }
    }
    if (!evicted.empty()) {
      // (their clearing kernels were enqueued at least a cache's worth of maps ago; one synchronise of the side stream covers the rare one that is not done)
      bool all_done = true;
      for (const ClearedTable& old : evicted) all_done = all_done && cleared_done(C, old);
      if (!all_done && C.stream) (void)hipStreamSynchronize(C.stream);
      std::lock_guard<std::mutex> lock(C.mu);
      for (ClearedTable& old : evicted) {
        if (old.slot >= 0) C.slot_used[old.slot] = false;
        (void)pool_free(old.buckets);
        (void)pool_free(old.acc);
        (void)pool_free(old.stats);
      }
    }
  }
  if (keep) keep = pool_malloc(&t.acc, (size_t)nb * 2 * ACC_STRIDE * sizeof(long long)) == hipSuccess && pool_malloc(&t.stats, 4 * sizeof(int)) == hipSuccess;
  if (keep) {
    std::lock_guard<std::mutex> lock(C.mu);
    for (int k = 0; k < CLEARED_SLOTS && t.slot < 0; k++)
      if (!C.slot_used[k]) t.slot = k;
    if (t.slot >= 0) {
      C.slot_used[t.slot] = true;
      t.done_seq = ++C.seq ? C.seq : ++C.seq;  // (never 0: a fresh word reads 0)
      *reinterpret_cast<volatile unsigned int*>(C.h_done + t.slot) = 0u;  // (a word cannot carry this sequence number from an earlier life: the counter only grows)
    } else {
      keep = false;
    }
  }
  if (keep) {
    const size_t acc_words = (size_t)nb * (2 * ACC_STRIDE * sizeof(long long) / sizeof(uint4));
    // (clearing kernels of one device run one after the other on the side stream: ONE arrival counter, left at zero by every launch)
    clear_recycled_kernel<<<CLEAR_BLOCKS, 256, 0, C.stream>>>(buckets, nb, (uint4*)t.acc, acc_words, t.stats, C.d_arrivals, C.h_done_dev + t.slot, t.done_seq);
    keep = hipGetLastError() == hipSuccess;
    if (!keep) (void)hipStreamSynchronize(C.stream);
  }
  if (!keep) {
    (void)hipGetLastError();
    if (t.acc) (void)pool_free(t.acc);
    if (t.stats) (void)pool_free(t.stats);
    if (t.slot >= 0) {
      std::lock_guard<std::mutex> lock(C.mu);
      C.slot_used[t.slot] = false;
    }
    (void)pool_free(buckets);
    return;
  }
  t.buckets = buckets;
  t.nb = nb;
  std::lock_guard<std::mutex> lock(C.mu);
  C.tables.push_back(t);
  C.bytes += bytes;
}

// a cleared table of exactly nb buckets whose clearing the host has seen finished; false: none (the caller clears one itself)
bool take_cleared_table(int device, unsigned int nb, ClearedTable* out) {
  ClearedCache& C = cleared_cache(device);
  std::lock_guard<std::mutex> lock(C.mu);
  for (size_t i = 0; i < C.tables.size(); i++) {
    if (C.tables[i].nb != nb || !cleared_done(C, C.tables[i])) continue;
    *out = C.tables[i];
    C.tables.erase(C.tables.begin() + (long)i);
    C.bytes -= cleared_bytes(nb);
    C.slot_used[out->slot] = false;
    out->slot = -1;
    return true;
  }
  return false;
}

}  // namespace

namespace glim_amd {
// every cached table back to the pool (pool_trim: the device is short of memory)
void voxelmap_drop_cleared_tables(int device) {
  ClearedCache& C = cleared_cache(device);
  std::vector<ClearedTable> all;
  {
    std::lock_guard<std::mutex> lock(C.mu);
    all.swap(C.tables);
    C.bytes = 0;
  }
  if (!all.empty() && C.stream) (void)hipStreamSynchronize(C.stream);
  for (ClearedTable& t : all) {
    (void)pool_free(t.buckets);
    (void)pool_free(t.acc);
    (void)pool_free(t.stats);
  }
  std::lock_guard<std::mutex> lock(C.mu);
  for (ClearedTable& t : all)
    if (t.slot >= 0) C.slot_used[t.slot] = false;
}
}  // namespace glim_amd

namespace {
bool spin_word(const volatile unsigned int* word, unsigned int value) {  // acquire; false after ~100 ms
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned long spins = 0;; spins++) {
    if (*word == value) {
      std::atomic_thread_fence(std::memory_order_acquire);
      return true;
    }
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
    if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(100)) return false;
  }
}
// the build of `m` has been seen complete: its scratch goes back to the pool
void reap_build(glim_amd_voxelmap* m) {
  if (m->pending_acc) (void)pool_free(m->pending_acc);
  if (m->pending_stats) (void)pool_free(m->pending_stats);
  m->pending_acc = m->pending_stats = nullptr;
  m->ready_pending.store(false, std::memory_order_release);
}
}  // namespace

namespace glim_amd {
int voxelmap_wait_ready(const glim_amd_voxelmap* cm, hipStream_t consumer) {
  glim_amd_voxelmap* m = const_cast<glim_amd_voxelmap*>(cm);  // (the bookkeeping of a finished build is not a change of the map)
  if (!m->ready_pending.load(std::memory_order_acquire)) return GLIM_AMD_OK;
  std::lock_guard<std::mutex> lock(m->view_mu);
  if (!m->ready_pending.load(std::memory_order_relaxed)) return GLIM_AMD_OK;
  const hipError_t q = hipEventQuery(m->ready_event);
  if (q == hipSuccess) {
    reap_build(m);
    return GLIM_AMD_OK;
  }
  (void)hipGetLastError();  // (hipErrorNotReady is not an error)
  if (q != hipErrorNotReady) {
    set_hip_error(q, "voxelmap_wait_ready");
    return GLIM_AMD_ERR_HIP;
  }
  if (consumer) {
    GA_HIP(hipStreamWaitEvent(consumer, m->ready_event, 0));  // (stays pending: a consumer on another stream waits as well)
  } else {
    GA_HIP(hipEventSynchronize(m->ready_event));
    reap_build(m);
  }
  return GLIM_AMD_OK;
}

int ensure_plane_view(glim_amd_voxelmap* m, hipStream_t st) {
  if (!m->buckets || m->num_buckets == 0) return GLIM_AMD_ERR_STATE;
  GA_TRY(voxelmap_wait_ready(m, st));
  // a map may be reached from factor sets of several contexts at once (GLIM's modules share them): one builder, and the view is COMPLETE
  // before anybody sees its pointer
  std::lock_guard<std::mutex> lock(m->view_mu);
  if (m->buckets_sm) return GLIM_AMD_OK;
  VoxelBucket* v = nullptr;
  GA_HIP(pool_malloc(&v, ((size_t)m->num_buckets + 1) * sizeof(VoxelBucket)));
  const unsigned int threads = 2 * (m->num_buckets + 1);
  plane_view_kernel<<<(threads + 255) / 256, 256, 0, st>>>(m->buckets, v, m->num_buckets);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    (void)pool_free(v);
    set_hip_error(e, "ensure_plane_view");
    return GLIM_AMD_ERR_HIP;
  }
  m->buckets_sm = v;
  return GLIM_AMD_OK;
}
// The occupancy mask of a finished map: at most OCC_MAX_WORDS 32-bit words; the cell is the voxel itself when the box of the occupied voxels fits
// (a 60 x 40 x 8 m room at 1 m voxels: 2.8 KB), otherwise the smallest power-of-two multiple that does.  Two short kernels and two
// synchronises, once per map and only for maps a LARGE factor set is evaluated against (plan_build).
int ensure_occupancy(glim_amd_voxelmap* m, hipStream_t st) {
  if (!m->buckets || m->num_buckets == 0) return GLIM_AMD_ERR_STATE;
  GA_TRY(voxelmap_wait_ready(m, st));
  std::lock_guard<std::mutex> lock(m->view_mu);
  if (m->occ_state != 0) return GLIM_AMD_OK;
  constexpr long long OCC_MAX_WORDS = 16384;  // 64 KiB
  int* d_bbox = nullptr;
  GA_HIP(pool_malloc(&d_bbox, 6 * sizeof(int)));
  int h_bbox[6] = {0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000};
  hipError_t e = hipMemcpyAsync(d_bbox, h_bbox, sizeof(h_bbox), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    occ_bbox_kernel<<<(2 * m->num_buckets + 255) / 256, 256, 0, st>>>(m->buckets, m->num_buckets, d_bbox);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(h_bbox, d_bbox, sizeof(h_bbox), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)pool_free(d_bbox);
  if (e != hipSuccess) {
    set_hip_error(e, "ensure_occupancy: bounding box");
    return GLIM_AMD_ERR_HIP;
  }
  if (h_bbox[0] > h_bbox[3]) {  // no voxel at all: nothing to cull against (the kernel finds no correspondence anyway)
    m->occ_state = -1;
    return GLIM_AMD_OK;
  }
  int shift = 0;
  long long dim[3], words = 0;
  for (; shift <= 21; shift++) {
    for (int a = 0; a < 3; a++) dim[a] = (((long long)h_bbox[3 + a] - h_bbox[a]) >> shift) + 1;
    words = ((dim[0] + 31) / 32) * dim[1] * dim[2];
    if (words <= OCC_MAX_WORDS) break;
  }
  if (words > OCC_MAX_WORDS) {
    m->occ_state = -1;
    return GLIM_AMD_OK;
  }
  unsigned int* occ = nullptr;
  GA_HIP(pool_malloc(&occ, (size_t)words * sizeof(unsigned int)));
  const int row_words = (int)((dim[0] + 31) / 32);
  e = hipMemsetAsync(occ, 0, (size_t)words * sizeof(unsigned int), st);
  if (e == hipSuccess) {
    occ_fill_kernel<<<(2 * m->num_buckets + 255) / 256, 256, 0, st>>>(m->buckets, m->num_buckets, h_bbox[0], h_bbox[1], h_bbox[2], shift, (int)dim[1], row_words, occ);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);  // complete before anybody sees the pointer
  if (e != hipSuccess) {
    (void)pool_free(occ);
    set_hip_error(e, "ensure_occupancy: fill");
    return GLIM_AMD_ERR_HIP;
  }
  for (int a = 0; a < 3; a++) {
    m->occ_org[a] = h_bbox[a];
    m->occ_dim[a] = (int)dim[a];
  }
  m->occ_shift = shift;
  m->occ_row_words = row_words;
  m->occ = occ;
  m->occ_state = 1;
  return GLIM_AMD_OK;
}
}  // namespace glim_amd

extern "C" {

int glim_amd_voxelmap_create(glim_amd_ctx* ctx, double resolution, int /*init_num_buckets*/, int /*max_bucket_scan_count*/,
                             double /*target_points_drop_rate*/, glim_amd_voxelmap** out) {
  if (!ctx || !out || !(resolution > 0.0)) return GLIM_AMD_ERR_INVALID;
  glim_amd_voxelmap* m = new glim_amd_voxelmap();
  m->ctx = ctx;
  m->resolution = resolution;
  m->inv_resolution = 1.0 / resolution;  // same FP64 expression as the CPU restatement (parity contract)
  *out = m;
  return GLIM_AMD_OK;
}

int glim_amd_voxelmap_destroy(glim_amd_voxelmap* m) {
  if (!m) return GLIM_AMD_OK;
  if (m->ctx) {
    (void)hipSetDevice(m->ctx->device);
    quiesce_device(m->ctx->device, m->uid);  // asynchronous factor launches (of any context) may still be reading this table
  }
  global_mutation_epoch()++;  // factor sets re-validate their plans
  (void)voxelmap_wait_ready(m, nullptr);  // (a build the host has not seen complete: its last kernel may still be writing the table)
  if (m->pending_acc) (void)pool_free(m->pending_acc);
  if (m->pending_stats) (void)pool_free(m->pending_stats);
  if (m->ready_event) event_put(m->ctx ? m->ctx->device : -1, m->ready_event);  // (voxelmap_wait_ready above has seen it complete; no owner: destroyed)
  if (m->buckets_sm) (void)pool_free(m->buckets_sm);
  if (m->occ) (void)pool_free(m->occ);
  if (m->buckets) {
    if (m->ctx && m->ctx->diag.bucket_factor == 0) recycle_table(m->ctx->device, m->buckets, m->num_buckets);
    else (void)pool_free(m->buckets);
  }
  delete m;
  return GLIM_AMD_OK;
}

int glim_amd_voxelmap_insert(glim_amd_voxelmap* m, const glim_amd_cloud* cloud) {
  if (!m || !cloud || cloud->ctx->device != m->ctx->device) return GLIM_AMD_ERR_INVALID;  // (any context of the map's device)
  if (!cloud->has_covs) return GLIM_AMD_ERR_STATE;
  if (cloud->n > (int64_t)(1u << 28)) return GLIM_AMD_ERR_INVALID;
  glim_amd_ctx* ctx = m->ctx;
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream();
  const int n = (int)cloud->n;

  DeviceTemp tkeys, pkeys, stats, acc, lru;
  SyncOnExit in_flight(st);  // every exit that has not synchronised itself waits for the stream before the scratch above goes back to the pool
  constexpr int DIRECT_MAX_POINTS = 32768;
  GA_TRY(voxelmap_wait_ready(m, nullptr));
  VoxelBucket* const old = m->buckets;  // a map that already holds voxels: incremental insert (rebuild with the old voxels re-opened)
  const unsigned int old_buckets = old ? m->num_buckets : 0u;
  if (old) quiesce_device(ctx->device, m->uid);  // asynchronous factor launches may still be reading the table that is about to be replaced
  // (the plane view follows the table, but only at the COMMIT below: every failing exit leaves the map -- uid, table AND view -- as it was, so a
  //  plan that holds the view's pointer keeps reading live memory)
  // Direct build (keys straight into the final table, ONE synchronise: build_direct_kernel) needs the table size before the voxels are
  // counted.  Small clouds: 2 buckets per point (4 ways per point: load factor below 1/2 whatever the cloud).  Larger clouds: 6 buckets per
  // EXPECTED voxel, from the voxels-per-point ratio of the last map this context built at (about) this resolution -- consecutive frames of a
  // stream have the same density -- and never fewer than N / 2 buckets, which hold one key per point: the table cannot overflow whatever the
  // estimate, a poor one only costs longer probe chains until the next map corrects it.  The first map at a resolution takes the counting path.
  const int res_class = (int)lround(8.0 * log2(m->resolution));  // resolutions within ~9 % share a class
  double expected_ratio = 0.0;
  for (const auto& h : ctx->voxel_ratio_hints)
    if (h.first == res_class) expected_ratio = h.second;
  const bool direct_small = n > 0 && n <= DIRECT_MAX_POINTS;
  const bool direct_large = n > DIRECT_MAX_POINTS && expected_ratio > 0.0;
  // (`do { ... } while (0)`: a direct build that turns out unsuitable -- table request beyond the 2^25-bucket addressing limit, or a table that
  //  came out too full because the density hint was stale -- `break`s out and the counting path below builds the map instead)
  if (!old && (direct_small || direct_large) && ctx->diag.bucket_factor == 0) do {
    const unsigned long long want = direct_small ? 2ull * (unsigned long long)n
                                                 : std::max<unsigned long long>((unsigned long long)n / 2 + 1, (unsigned long long)(6.0 * expected_ratio * (double)n));
    if (want > (1ull << 25)) break;  // the counting path sizes the table from the voxels actually present (few voxels of a huge cloud still fit)
    const unsigned int nb = (unsigned int)round_buckets(std::max<unsigned long long>(16, want));
    VoxelBucket* buckets = nullptr;
    ClearedTable cleared;
    const bool have_cleared = take_cleared_table(ctx->device, nb, &cleared);  // (a map of this size went away a moment ago: "pre-cleared tables")
    if (have_cleared) {
      buckets = cleared.buckets;
      acc.p = cleared.acc;
      stats.p = cleared.stats;
    } else {
      GA_HIP(pool_malloc(&stats.p, 4 * sizeof(int)));
      GA_HIP(pool_malloc(&acc.p, (size_t)nb * 2 * ACC_STRIDE * sizeof(long long)));
      GA_HIP(pool_malloc(&buckets, (size_t)nb * sizeof(VoxelBucket)));
    }
    // three launches and one synchronise: tables, keys + sums, records (the last one also hands the counters to the host)
    static_assert((2 * ACC_STRIDE * sizeof(long long)) % sizeof(uint4) == 0, "accumulators are cleared in 16-byte words");
    const size_t acc_words = (size_t)nb * (2 * ACC_STRIDE * sizeof(long long) / sizeof(uint4));
    int *h_view = nullptr, *d_view = nullptr;
    const bool mapped = pinned_scratch_views(ctx, reinterpret_cast<void**>(&h_view), reinterpret_cast<void**>(&d_view));
    int h_stats[2] = {0, 0};
    // a map built from a plane-form cloud (a frame with kNN covariances) will be matched against plane-form frames: its plane view is written
    // with the map (no launch, no synchronise on the first factor that uses it); any other map gets the view on first use (ensure_plane_view)
    VoxelBucket* view = nullptr;
    if (cloud->plane_form && ctx->diag.view_fused) GA_HIP(pool_malloc(&view, ((size_t)nb + 1) * sizeof(VoxelBucket)));
    if (!have_cleared)
      init_tables_kernel<<<(unsigned int)((std::max<size_t>((size_t)nb * 8, acc_words) + 255) / 256), 256, 0, st>>>(buckets, nb, (uint4*)acc.p, acc_words, (int*)stats.p);
    build_direct_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, cloud->pts, cloud->covA, cloud->covB, m->inv_resolution, buckets, nb, (long long*)acc.p, (int*)stats.p);
    // The host needs the voxel count (the reference's insert returns it through voxelmap_info) and nothing else from this build: the last kernel
    // hands it over when it STARTS (word 2 of the mapped scratch = this build's sequence number) and the call returns while that kernel writes
    // the records; readers on other streams wait for `ready_event` (voxelmap_wait_ready).  40-45 -> 3x us per 131 072-pt map.
    const bool polled = mapped && (m->ready_event || event_get(ctx->device, &m->ready_event) == hipSuccess);
    const unsigned int seq = polled ? (++ctx->map_seq ? ctx->map_seq : ++ctx->map_seq) : 0u;
    // the completion word lives in the context's shared pinned scratch, where read_back_sync() leaves arbitrary small integers (kNN counters,
    // kept points): a leftover equal to this build's sequence number would end the wait before the kernel has written anything
    if (polled) reinterpret_cast<volatile unsigned int*>(h_view)[2] = 0u;
    finalize_kernel<<<(2 * (nb + 1) + 255) / 256, 256, 0, st>>>(buckets, nb, (const long long*)acc.p, m->resolution, (const int*)stats.p, mapped ? d_view : nullptr, view,
                                                                 nullptr, 0, seq);
    hipError_t e = hipGetLastError();
    bool running = false;  // the last kernel may still be running when this call returns
    if (e == hipSuccess && polled) {
      e = hipEventRecord(m->ready_event, st);
      if (e == hipSuccess) {
        if (spin_word(reinterpret_cast<volatile unsigned int*>(h_view) + 2, seq)) running = true;
        else e = hipStreamSynchronize(st);
      }
      h_stats[0] = h_view[0];
      h_stats[1] = h_view[1];
    } else if (e == hipSuccess && mapped) {
      e = hipStreamSynchronize(st);
      h_stats[0] = h_view[0];
      h_stats[1] = h_view[1];
    } else if (e == hipSuccess) {
      e = read_back_sync(ctx, st, h_stats, stats.p, sizeof(h_stats));
    }
    if (e != hipSuccess) {
      set_hip_error(e, "voxelmap_insert");
      (void)hipStreamSynchronize(st);  // (whatever did get enqueued must not outlive the table)
      (void)pool_free(buckets);
      if (view) (void)pool_free(view);
      return GLIM_AMD_ERR_HIP;
    }
    if (running && (h_stats[1] != 0 || (direct_large && (double)h_stats[0] > 0.35 * 2.0 * (double)nb))) {
      (void)hipStreamSynchronize(st);  // the rare exits below hand memory back: the build has to be over first
      running = false;
    }
    in_flight.dismiss();  // synchronised, or the scratch stays with the map until its event has been seen complete (below)
    if (h_stats[1] != 0) {
      (void)pool_free(buckets);
      if (view) (void)pool_free(view);
      return GLIM_AMD_ERR_RANGE;
    }
    remember_voxel_ratio(ctx, res_class, (double)h_stats[0] / (double)n);
    // A stale hint (the previous map at this resolution was dense, this cloud is sparse -- e.g. already downsampled at about the voxel size)
    // leaves the floor of N / 2 buckets, one way per point, 80-100 % full: lookups stay correct but every miss walks hundreds of buckets, for
    // the whole life of the map.  Above 0.35 keys per way the table is rebuilt by the counting path (6 buckets per voxel actually present).
    if (direct_large && (double)h_stats[0] > 0.35 * 2.0 * (double)nb) {
      (void)pool_free(buckets);
      if (view) (void)pool_free(view);
      (void)pool_free(stats.p);
      stats.p = nullptr;
      (void)pool_free(acc.p);
      acc.p = nullptr;
      in_flight.armed = true;  // the counting path enqueues again
      break;
    }
    if (running) {
      m->pending_acc = acc.p;
      m->pending_stats = stats.p;
      acc.p = stats.p = nullptr;
      m->ready_pending.store(true, std::memory_order_release);
    }
    m->buckets = buckets;
    m->buckets_sm = view;
    m->num_buckets = nb;
    m->num_voxels = h_stats[0];
    m->lru_counter += 1;  // (a first insert: every voxel carries stamp 0 = this insert's counter)
    m->uid = next_uid();
    global_mutation_epoch()++;
    return GLIM_AMD_OK;
  } while (0);
  const unsigned int tsize0 = next_pow2((unsigned long long)std::max<long long>(32, (long long)n + (old ? (long long)m->num_voxels : 0ll)) * 2);
  GA_HIP(pool_malloc(&tkeys.p, (size_t)tsize0 * sizeof(unsigned long long)));
  GA_HIP(pool_malloc(&pkeys.p, (size_t)(n > 0 ? n : 1) * sizeof(unsigned long long)));
  GA_HIP(pool_malloc(&stats.p, 4 * sizeof(int)));
  GA_HIP(hipMemsetAsync(stats.p, 0, 4 * sizeof(int), st));
  fill_u64_kernel<<<1024, 256, 0, st>>>((unsigned long long*)tkeys.p, tsize0, EMPTY_KEY);
  GA_HIP(hipGetLastError());
  if (n > 0) {
    insert_keys_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, cloud->pts, m->inv_resolution, (unsigned long long*)tkeys.p, tsize0 - 1,
                                                         (unsigned long long*)pkeys.p, (int*)stats.p);
    GA_HIP(hipGetLastError());
  }
  if (old) {
    // LRU eviction (glim_amd_voxelmap_set_lru_horizon): this insert carries counter c = lru_counter; when (c + 1) % clear_cycle == 0 every voxel with
    // stamp + horizon < c + 1 that the new points do not touch is dropped, i.e. not carried over into the rebuilt table
    const int c_after = m->lru_counter + 1;
    const int expire_before = (m->lru_horizon > 0 && c_after % std::max(1, m->lru_clear_cycle) == 0) ? std::max(0, c_after - m->lru_horizon) : 0;
    reinsert_old_keys_kernel<<<(2 * old_buckets + 255) / 256, 256, 0, st>>>(old, old_buckets, (unsigned long long*)tkeys.p, tsize0 - 1, (int*)stats.p, expire_before);
    GA_HIP(hipGetLastError());
  }
  int h_stats[2] = {0, 0};
  GA_HIP(read_back_sync(ctx, st, h_stats, stats.p, sizeof(h_stats)));
  if (h_stats[1] != 0) return GLIM_AMD_ERR_RANGE;  // (the map is unchanged)

  const int num_voxels = h_stats[0];
  // buckets per voxel (default 6: x-adjacent voxel pairs share a bucket -- device_math.hpp GLIM_AMD_PAIR_SHIFT -- so about 0.1 pairs per
  // bucket, and a pair finds its home bucket taken by another pair about 1 % of the time)
  // The table is addressed with 32-bit byte offsets (<= 2^25 buckets = 4 GiB): beyond 5.6 M voxels the factor drops step by step to 2
  // (16.7 M voxels) before the insert is refused -- a sparser table is a speed choice, never a correctness one.
  unsigned long long bucket_factor = ctx->diag.bucket_factor > 0 ? (unsigned long long)ctx->diag.bucket_factor : 6ull;
  while (bucket_factor > 2 && (unsigned long long)num_voxels * bucket_factor > (1ull << 25)) bucket_factor--;
  unsigned long long nb64 = std::max<unsigned long long>(16, (unsigned long long)num_voxels * bucket_factor);
  if (nb64 > (1ull << 25)) return GLIM_AMD_ERR_NOMEM;
  if (ctx->diag.bucket_factor == 0) nb64 = round_buckets(nb64);
  const unsigned int nb = (unsigned int)nb64;
  VoxelBucket *buckets = nullptr, *view2 = nullptr;
  GA_HIP(pool_malloc(&buckets, (size_t)nb * sizeof(VoxelBucket)));
  hipError_t e = pool_malloc(&acc.p, (size_t)nb * 2 * ACC_STRIDE * sizeof(long long));
  if (e == hipSuccess) e = hipMemsetAsync(acc.p, 0, (size_t)nb * 2 * ACC_STRIDE * sizeof(long long), st);
  if (e == hipSuccess) {
    init_buckets_kernel<<<(unsigned int)(((size_t)nb * 8 + 255) / 256), 256, 0, st>>>(buckets, nb);
    move_keys_kernel<<<(tsize0 + 255) / 256, 256, 0, st>>>((const unsigned long long*)tkeys.p, tsize0, buckets, nb);
    if (old && m->lru_horizon > 0) {
      e = pool_malloc(&lru.p, (size_t)nb * 2 * sizeof(int2));
      if (e == hipSuccess) e = hipMemsetAsync(lru.p, 0, (size_t)nb * 2 * sizeof(int2), st);
    }
    if (old && e == hipSuccess)
      reopen_old_voxels_kernel<<<(2 * old_buckets + 255) / 256, 256, 0, st>>>(old, old_buckets, m->resolution, buckets, nb, (long long*)acc.p, lru.as<int2>());
    if (n > 0)
      accumulate_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, cloud->pts, cloud->covA, cloud->covB, (const unsigned long long*)pkeys.p, buckets,
                                                          nb, (long long*)acc.p);
    if (cloud->plane_form && ctx->diag.view_fused && pool_malloc(&view2, ((size_t)nb + 1) * sizeof(VoxelBucket)) != hipSuccess) {
      (void)hipGetLastError();
      view2 = nullptr;  // (the view is a cache: it is built on first use then)
    }
    if (e == hipSuccess)
      finalize_kernel<<<(2 * (nb + 1) + 255) / 256, 256, 0, st>>>(buckets, nb, (const long long*)acc.p, m->resolution, nullptr, nullptr, view2, lru.as<int2>(),
                                                                    m->lru_horizon > 0 ? m->lru_counter : 0);
    if (e == hipSuccess) e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    set_hip_error(e, "voxelmap_insert");
    (void)hipStreamSynchronize(st);
    (void)pool_free(buckets);
    if (view2) (void)pool_free(view2);
    return GLIM_AMD_ERR_HIP;
  }
  in_flight.dismiss();
  {  // commit: the old view goes with the old table (the uid changes below, so no plan keeps either pointer)
    std::lock_guard<std::mutex> vlock(m->view_mu);
    if (m->buckets_sm) (void)pool_free(m->buckets_sm);
    m->buckets_sm = view2;
    if (m->occ) (void)pool_free(m->occ);  // (the occupancy mask likewise: rebuilt on the next use by a large factor set)
    m->occ = nullptr;
    m->occ_state = 0;
  }
  if (old) (void)pool_free(old);
  else if (n > 0) remember_voxel_ratio(ctx, res_class, (double)num_voxels / (double)n);
  m->buckets = buckets;
  m->num_buckets = nb;
  m->num_voxels = num_voxels;
  m->lru_counter += 1;
  m->uid = next_uid();  // plans built from the previous table are rebuilt (factor_set_prepare)
  global_mutation_epoch()++;
  return GLIM_AMD_OK;
}

// create_frame of GLIM's GPU odometry (odometry_estimation_gpu.cpp:86-107) as ONE submission: PointCloudGPU::clone of a frame that arrives with CPU
// covariances + normals and the GaussianVoxelMapGPU of every level (voxelmap_levels = 2 in the shipped configuration) are enqueued back to back on
// the context's stream -- pull kernel, then per map: tables, keys + sums, records -- and the host synchronises ONCE; the plane-form verdict of the
// cloud and every map's voxel count come back with that one completion (host-mapped words).  As three separate calls the same work pays three
// synchronises (58 + 2 x 24-29 us per 10 000-pt frame).  Frames that do not fit the small-cloud paths take the separate calls.
int glim_amd_frame_create(glim_amd_ctx* ctx, int64_t n, const double* points4, const double* covs16, const double* normals4, int32_t num_levels,
                          const double* resolutions, glim_amd_cloud** cloud_out, glim_amd_voxelmap** maps_out) {
  constexpr int MAX_LEVELS = 8;
  if (!ctx || !cloud_out || !maps_out || n < 0 || (n > 0 && !points4) || num_levels < 0 || num_levels > MAX_LEVELS || (num_levels > 0 && !resolutions))
    return GLIM_AMD_ERR_INVALID;
  for (int lv = 0; lv < num_levels; lv++)
    if (!(resolutions[lv] > 0.0)) return GLIM_AMD_ERR_INVALID;
  *cloud_out = nullptr;
  for (int lv = 0; lv < num_levels; lv++) maps_out[lv] = nullptr;
  const bool one_submission = n > 0 && n <= HOST_PACK_MAX_POINTS_FRAME && covs16 && ctx->diag.host_pack && ctx->diag.bucket_factor == 0;
  if (!one_submission) {
    GA_TRY(glim_amd_cloud_create(ctx, n, points4, covs16, normals4, cloud_out));
    int rc = GLIM_AMD_OK;
    for (int lv = 0; lv < num_levels && rc == GLIM_AMD_OK; lv++) {
      rc = glim_amd_voxelmap_create(ctx, resolutions[lv], 8192 * 2, 10, 1e-3, &maps_out[lv]);
      if (rc == GLIM_AMD_OK) rc = glim_amd_voxelmap_insert(maps_out[lv], *cloud_out);
    }
    if (rc != GLIM_AMD_OK) {
      for (int lv = 0; lv < num_levels; lv++) {
        (void)glim_amd_voxelmap_destroy(maps_out[lv]);
        maps_out[lv] = nullptr;
      }
      (void)glim_amd_cloud_destroy(*cloud_out);
      *cloud_out = nullptr;
    }
    return rc;
  }
  struct Build {
    glim_amd_voxelmap* m = nullptr;
    VoxelBucket *buckets = nullptr, *view = nullptr;
    void *acc = nullptr, *stats = nullptr;
    unsigned int nb = 0;
  } b[MAX_LEVELS];
  glim_amd_cloud* c = nullptr;
  SmallUpload up;
  int rc = GLIM_AMD_OK;
  bool enqueued = false;
  int *h_view = nullptr, *d_view = nullptr;
  unsigned int poll_seq = 0u;
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    GA_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream();
    g_frame_t0_us = frame_now_us();
    rc = alloc_cloud_for_frame(ctx, n, true, normals4 != nullptr, &c);
    frame_stamp(0);
    if (rc == GLIM_AMD_OK && !pinned_scratch_views(ctx, reinterpret_cast<void**>(&h_view), reinterpret_cast<void**>(&d_view))) rc = GLIM_AMD_ERR_UNSUPPORTED;
    // (the completion word shares the context's pinned scratch with read_back_sync(): whatever a kNN or preprocessing call left there must not read
    //  as this submission's sequence number -- ADVICE r5)
    if (rc == GLIM_AMD_OK && num_levels > 0) reinterpret_cast<volatile unsigned int*>(h_view)[4 * (num_levels - 1) + 2] = 0u;
    // frame_fused (default): ONE launch pulls the cloud and builds every level (frame_build_kernel), one more writes every level's records
    // (frame_finalize_kernel) -- 2 dependent launches instead of 1 + 2 per level; with the gated pull both go out BEFORE the host converts
    const bool fused = ctx->diag.frame_fused && num_levels >= 1 && num_levels <= FRAME_MAX_LEVELS;
    if (rc == GLIM_AMD_OK) rc = fused ? cloud_small_prepare(ctx, c, points4, covs16, normals4, &up) : cloud_small_enqueue(ctx, c, points4, covs16, normals4, &up);
    enqueued = rc == GLIM_AMD_OK;
    if (enqueued && fused && !up.gated) cloud_small_pack(&up);
    FrameLevels FL;
    const unsigned int nb = (unsigned int)round_buckets(std::max<unsigned long long>(16, 2ull * (unsigned long long)n));  // (the small direct build of insert)
    for (int lv = 0; lv < num_levels && rc == GLIM_AMD_OK; lv++) {
      Build& B = b[lv];
      B.nb = nb;
      B.m = new glim_amd_voxelmap();
      B.m->ctx = ctx;
      B.m->resolution = resolutions[lv];
      B.m->inv_resolution = 1.0 / resolutions[lv];
      ClearedTable cleared;
      const bool have_cleared = take_cleared_table(ctx->device, nb, &cleared);
      hipError_t e = hipSuccess;
      if (have_cleared) {
        B.buckets = cleared.buckets;
        B.acc = cleared.acc;
        B.stats = cleared.stats;
      } else {
        e = pool_malloc(&B.stats, 4 * sizeof(int));
        if (e == hipSuccess) e = pool_malloc(&B.acc, (size_t)nb * 2 * ACC_STRIDE * sizeof(long long));
        if (e == hipSuccess) e = pool_malloc(&B.buckets, (size_t)nb * sizeof(VoxelBucket));
      }
      // the frame arrives with covariances AND normals of the kNN estimator: its maps will be matched against plane-form frames (view written here)
      if (e == hipSuccess && normals4 && ctx->diag.view_fused) e = pool_malloc(&B.view, ((size_t)nb + 1) * sizeof(VoxelBucket));
      if (e != hipSuccess) {
        set_hip_error(e, "glim_amd_frame_create");
        rc = e == hipErrorOutOfMemory ? GLIM_AMD_ERR_NOMEM : GLIM_AMD_ERR_HIP;
        break;
      }
      const size_t acc_words = (size_t)nb * (2 * ACC_STRIDE * sizeof(long long) / sizeof(uint4));
      if (!have_cleared)
        init_tables_kernel<<<(unsigned int)((std::max<size_t>((size_t)nb * 8, acc_words) + 255) / 256), 256, 0, st>>>(B.buckets, nb, (uint4*)B.acc, acc_words, (int*)B.stats);
      if (fused) {
        FL.count = lv + 1;
        FL.inv_res[lv] = B.m->inv_resolution;
        FL.res[lv] = B.m->resolution;
        FL.buckets[lv] = B.buckets;
        FL.view[lv] = B.view;
        FL.acc[lv] = (long long*)B.acc;
        FL.stats[lv] = (int*)B.stats;
        FL.nb[lv] = nb;
        const hipError_t ei = hipGetLastError();
        if (ei != hipSuccess) {
          set_hip_error(ei, "glim_amd_frame_create");
          rc = GLIM_AMD_ERR_HIP;
        }
        continue;
      }
      build_direct_kernel<<<((int)n + 255) / 256, 256, 0, st>>>((int)n, c->pts, c->covA, c->covB, B.m->inv_resolution, B.buckets, nb, (long long*)B.acc, (int*)B.stats);
      // the LAST level's records kernel hands the host the completion word when it starts (the stream runs in order: everything before it -- the
      // pull kernel's plane-form verdict, the earlier levels' counts -- is complete and visible by then); it may still be writing its own
      // records when this call returns, which is what ready_event is for (voxelmap_wait_ready)
      const bool last = lv == num_levels - 1;
      if (last && event_get(ctx->device, &B.m->ready_event) == hipSuccess) poll_seq = ++ctx->map_seq ? ctx->map_seq : ++ctx->map_seq;
      finalize_kernel<<<(2 * (nb + 1) + 255) / 256, 256, 0, st>>>(B.buckets, nb, (const long long*)B.acc, B.m->resolution, (const int*)B.stats, d_view + 4 * lv, B.view,
                                                                   nullptr, 0, last ? poll_seq : 0u);
      e = hipGetLastError();
      if (e == hipSuccess && last && poll_seq) e = hipEventRecord(B.m->ready_event, st);
      if (e != hipSuccess) {
        set_hip_error(e, "glim_amd_frame_create");
        rc = GLIM_AMD_ERR_HIP;
      }
    }
    if (fused && enqueued) {
      bool launched = false;
      if (rc == GLIM_AMD_OK) {
        // The completion word is written by the LAST block of the build kernel: the records kernel behind it only has to be enqueued, and the
        // host returns before it has started.  EVERY level's table is therefore still being written when this call returns: every map gets its
        // ready event (readers on other streams wait for it: voxelmap_wait_ready) and keeps its accumulators until that event has been seen.
        bool events = true;
        for (int lv = 0; lv < num_levels && events; lv++) events = event_get(ctx->device, &b[lv].m->ready_event) == hipSuccess;
        if (events) poll_seq = ++ctx->map_seq ? ctx->map_seq : ++ctx->map_seq;
        frame_build_kernel<<<((int)n + 255) / 256, 256, 0, st>>>(up.args, FL, d_view, poll_seq);
        hipError_t e = hipGetLastError();
        launched = e == hipSuccess;
        frame_stamp(2);
        if (e == hipSuccess) {
          frame_finalize_kernel<<<dim3((2 * (nb + 1) + 255) / 256, (unsigned int)num_levels), 256, 0, st>>>(FL, poll_seq ? nullptr : d_view, 0u);
          e = hipGetLastError();
        }
        for (int lv = 0; lv < num_levels && e == hipSuccess && poll_seq; lv++) e = hipEventRecord(b[lv].m->ready_event, st);
        if (e != hipSuccess) {
          set_hip_error(e, "glim_amd_frame_create");
          rc = GLIM_AMD_ERR_HIP;
        }
      }
      // the host conversion, piece by piece, behind the launches (gated form).  Also after a failed step IF the pull kernel is out: its blocks wait
      // for the gate words
      if (launched) cloud_small_pack(&up);
    }
    frame_stamp(4);
    // ---- the one wait: the polled word, or a synchronise ----
    bool running = false;
    if (rc == GLIM_AMD_OK && num_levels > 0 && poll_seq && spin_word(reinterpret_cast<volatile unsigned int*>(h_view) + 4 * (num_levels - 1) + 2, poll_seq)) {
      running = true;
      for (int lv = 0; lv < num_levels; lv++) running = running && h_view[4 * lv + 1] == 0;  // (a range error hands memory back: synchronise first)
    }
    if (!running) {
      const hipError_t es = hipStreamSynchronize(st);
      if (rc == GLIM_AMD_OK && es != hipSuccess) {
        set_hip_error(es, "glim_amd_frame_create: synchronise");
        rc = GLIM_AMD_ERR_HIP;
      }
    }
    frame_stamp(5);
    if (enqueued) {
      if (rc == GLIM_AMD_OK) {
        rc = cloud_small_finish(c, &up);  // GLIM_AMD_ERR_UNSUPPORTED: the gated pull gave up -- the maps were built from nothing; the separate calls below
        if (rc != GLIM_AMD_OK && running) {
          (void)hipStreamSynchronize(st);  // (the last level's records kernel may still be writing the table that is about to be handed back)
          running = false;
        }
      } else if (up.stage) {
        (void)pinned_free(up.stage);
      }
    }
    for (int lv = 0; lv < num_levels; lv++) {
      Build& B = b[lv];
      if (running && (fused || lv == num_levels - 1)) {  // its records kernel may still be reading the accumulators: they stay with the map until the event has been seen
        B.m->pending_acc = B.acc;
        B.m->pending_stats = B.stats;
        B.m->ready_pending.store(true, std::memory_order_release);
      } else {
        if (B.acc) (void)pool_free(B.acc);
        if (B.stats) (void)pool_free(B.stats);
      }
      B.acc = B.stats = nullptr;
      if (rc == GLIM_AMD_OK && h_view[4 * lv + 1] != 0) rc = GLIM_AMD_ERR_RANGE;
    }
    if (rc == GLIM_AMD_OK) {
      for (int lv = 0; lv < num_levels; lv++) {
        Build& B = b[lv];
        B.m->buckets = B.buckets;
        B.m->buckets_sm = B.view;
        B.m->num_buckets = B.nb;
        B.m->num_voxels = h_view[4 * lv];
        B.m->lru_counter = 1;
        B.m->uid = next_uid();
        const int res_class = (int)lround(8.0 * log2(B.m->resolution));
        remember_voxel_ratio(ctx, res_class, (double)B.m->num_voxels / (double)n);
      }
      global_mutation_epoch()++;
    }
  }
  if (rc != GLIM_AMD_OK) {
    for (int lv = 0; lv < num_levels; lv++) {
      Build& B = b[lv];
      if (B.buckets) (void)pool_free(B.buckets);
      if (B.view) (void)pool_free(B.view);
      if (B.m) {
        if (B.m->ready_event) event_put(ctx->device, B.m->ready_event);  // (every path that gets here has synchronised the stream)
        B.m->ctx = nullptr;
        delete B.m;
      }
    }
    if (c) (void)glim_amd_cloud_destroy(c);
    if (rc == GLIM_AMD_ERR_UNSUPPORTED) {  // no device view of pinned memory here: the separate calls
      GA_TRY(glim_amd_cloud_create(ctx, n, points4, covs16, normals4, cloud_out));
      for (int lv = 0; lv < num_levels; lv++) {
        GA_TRY(glim_amd_voxelmap_create(ctx, resolutions[lv], 8192 * 2, 10, 1e-3, &maps_out[lv]));
        GA_TRY(glim_amd_voxelmap_insert(maps_out[lv], *cloud_out));
      }
      return GLIM_AMD_OK;
    }
    return rc;
  }
  *cloud_out = c;
  for (int lv = 0; lv < num_levels; lv++) maps_out[lv] = b[lv].m;
  frame_stamp(6);
  return GLIM_AMD_OK;
}

int glim_amd_debug_frame_stages(double* microseconds, int32_t num_fields) {
  if (!microseconds || num_fields < 0) return GLIM_AMD_ERR_INVALID;
  for (int i = 0; i < num_fields; i++) microseconds[i] = i < FRAME_STAGES ? g_frame_stage_us[i] : 0.0;
  return GLIM_AMD_OK;
}

int glim_amd_voxelmap_set_lru_horizon(glim_amd_voxelmap* m, int32_t lru_horizon, int32_t lru_clear_cycle) {
  if (!m || lru_clear_cycle < 0) return GLIM_AMD_ERR_INVALID;
  std::lock_guard<std::mutex> lock(m->ctx->mu);
  m->lru_horizon = lru_horizon > 0 ? lru_horizon : 0;
  m->lru_clear_cycle = lru_clear_cycle > 0 ? lru_clear_cycle : 10;
  return GLIM_AMD_OK;
}

int glim_amd_voxelmap_info(const glim_amd_voxelmap* m, int32_t* num_voxels, int32_t* num_buckets, double* resolution, size_t* bytes) {
  if (!m) return GLIM_AMD_ERR_INVALID;
  if (num_voxels) *num_voxels = m->num_voxels;
  if (num_buckets) *num_buckets = (int32_t)m->num_buckets;
  if (resolution) *resolution = m->resolution;
  if (bytes) *bytes = (size_t)m->num_buckets * sizeof(VoxelBucket);
  return GLIM_AMD_OK;
}

int glim_amd_voxelmap_download(const glim_amd_voxelmap* m, int32_t* coords, int32_t* counts, float* means, float* cov33) {
  if (!m) return GLIM_AMD_ERR_INVALID;
  if (!m->buckets) return GLIM_AMD_ERR_STATE;
  GA_TRY(voxelmap_wait_ready(m, nullptr));
  glim_amd_ctx* ctx = m->ctx;
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  std::vector<VoxelBucket> host(m->num_buckets);
  GA_HIP(hipMemcpyAsync(host.data(), m->buckets, (size_t)m->num_buckets * sizeof(VoxelBucket), hipMemcpyDeviceToHost, ctx->stream()));
  GA_HIP(hipStreamSynchronize(ctx->stream()));
  const unsigned int msk = (1u << KEY_BITS) - 1u;
  int v = 0;
  for (const VoxelBucket& bk : host) {
    for (int w = 0; w < 2; w++) {
      const unsigned long long key = bk.key[w];
      if (key == EMPTY_KEY) continue;
      if (v >= m->num_voxels) return GLIM_AMD_ERR_STATE;
      const float* r = bk.rec[w];
      const int cx = (int)((key >> (2 * KEY_BITS)) & msk) - KEY_OFFSET;
      const int cy = (int)((key >> KEY_BITS) & msk) - KEY_OFFSET;
      const int cz = (int)(key & msk) - KEY_OFFSET;
      if (coords) {
        coords[3 * v + 0] = cx;
        coords[3 * v + 1] = cy;
        coords[3 * v + 2] = cz;
      }
      if (counts) memcpy(&counts[v], &r[9], sizeof(int));
      if (means) {  // stored relative to the voxel centre
        means[3 * v] = (float)((double)r[0] + ((double)cx + 0.5) * m->resolution);
        means[3 * v + 1] = (float)((double)r[1] + ((double)cy + 0.5) * m->resolution);
        means[3 * v + 2] = (float)((double)r[2] + ((double)cz + 0.5) * m->resolution);
      }
      if (cov33) {
        float* c = cov33 + 9 * v;
        c[0] = r[3]; c[1] = r[4]; c[2] = r[5];
        c[3] = r[4]; c[4] = r[6]; c[5] = r[7];
        c[6] = r[5]; c[7] = r[7]; c[8] = r[8];
      }
      v++;
    }
  }
  return v == m->num_voxels ? GLIM_AMD_OK : GLIM_AMD_ERR_STATE;
}

}  // extern "C"
