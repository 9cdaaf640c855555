// preprocess.hip -- scan preprocessing on the device.
//
// Replaces glim::CloudPreprocessor::preprocess_impl (src/glim/preprocess/cloud_preprocessor.cpp:92-188) together with the three
// gtsam_points routines it calls (voxelgrid_sampling / randomgrid_sampling :104-109, remove_outliers :162-164): one upload of the
// raw scan, then every stage runs on the device and hands its result to the next one in HBM -- downsampling, range + cropbox
// filter, sort by time, outlier removal, kNN (knn.hip).  The host only reads back a few counters that size the next stage.
//
// Exactness contract (oracle/preprocess_oracle.c states the same rules on the CPU; tests/test_preprocess.py checks them):
//   * voxel coordinates: fast_floor(p * (1/res)) in FP64 -> identical voxel membership;
//   * order inside a voxel / between equal time stamps: ascending original index.  The reference sorts with unstable sorts, so its
//     order is implementation defined; the device gets the index order from a STABLE radix sort (sort.hip);
//   * voxel-grid means: one thread walks its run of the sorted order and adds sequentially in FP64 -- the same additions in the
//     same order as the CPU loop, hence bit-identical means (a parallel tree sum would not be);
//   * random-grid sampling: counter-based generator h = splitmix64(seed, index) instead of the reference's sequential
//     std::mt19937 stream (which no data-parallel implementation can replay): inside a voxel the points with the smallest
//     (h >> 32, index) survive, the global 1.2x cap keeps the smallest (h & 0xffffffff, index);
//   * predicates (range, cropbox) are evaluated in FP64 in one fixed operation order without contraction.
#include <algorithm>
#include <cmath>
#include <memory>
#include <vector>

#include "internal.hpp"
#include "scope_sync.hpp"
#include "device_math.hpp"
#include "scan.hpp"

using namespace glim_amd;

namespace {

using u64 = unsigned long long;
using u32 = unsigned int;

constexpr u64 INVALID_VKEY = ~0ull;

// == orc_sample_hash (splitmix64 of seed + (index + 1) * golden)
__host__ __device__ inline u64 sample_hash(u64 seed, u64 index) {
  u64 z = seed + (index + 1ull) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// ---- K9a: voxel key (x lowest, 21 bits per axis, offset 2^20) + bounding box of the valid coordinates ----
__global__ __launch_bounds__(256) void pp_key_kernel(int n, const double4* __restrict__ p4, double inv_res, u64* __restrict__ vkey,
                                                     int* __restrict__ bb) {
  __shared__ int s_tmp[16];
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const double4 p = p4[i];
    const double t[3] = {p.x * inv_res, p.y * inv_res, p.z * inv_res};
    bool valid = isfinite(p.w);
#pragma unroll
    for (int a = 0; a < 3; a++) valid = valid && (t[a] >= -1048576.0 && t[a] < 1048576.0);  // false for NaN / inf
    u64 key = INVALID_VKEY;
    if (valid) {
      int c[3];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        c[a] = fast_floor_d(t[a]) + KEY_OFFSET;
        lo[a] = min(lo[a], c[a]);
        hi[a] = max(hi[a], c[a]);
      }
      key = (u64)c[0] | ((u64)c[1] << 21) | ((u64)c[2] << 42);
    }
    vkey[i] = key;
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
    lo[a] = block_reduce_i<0>(lo[a], s_tmp);
    hi[a] = block_reduce_i<1>(hi[a], s_tmp);
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      if (lo[a] != 0x7fffffff) atomicMin(&bb[a], lo[a]);
      if (hi[a] != (int)0x80000000) atomicMax(&bb[3 + a], hi[a]);
    }
  }
}

// ---- K9b: key compacted to the bounding box: ((z - zmin) << (bx + by)) | ((y - ymin) << bx) | (x - xmin); invalid -> 1 << vbits ----
__global__ __launch_bounds__(256) void pp_compact_key_kernel(int n, const u64* __restrict__ vkey, int xmin, int ymin, int zmin, int bx, int by,
                                                             int vbits, u64* __restrict__ ckey) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u64 k = vkey[i];
  if (k == INVALID_VKEY) {
    ckey[i] = 1ull << vbits;
    return;
  }
  const u64 x = (k & 0x1FFFFFull) - (u64)xmin, y = ((k >> 21) & 0x1FFFFFull) - (u64)ymin, z = ((k >> 42) & 0x1FFFFFull) - (u64)zmin;
  ckey[i] = (z << (bx + by)) | (y << bx) | x;
}

__global__ __launch_bounds__(256) void pp_hash_key_kernel(int n, u64 seed, u64* __restrict__ keys) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) keys[i] = sample_hash(seed, (u64)i) >> 32;
}

__global__ __launch_bounds__(256) void pp_gather_key_kernel(int n, const u32* __restrict__ vals, const u64* __restrict__ ckey, u64* __restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < n) out[j] = ckey[vals[j]];
}

// counters[0] = number of distinct valid keys in the sorted order, counters[1] = number of valid entries
__global__ __launch_bounds__(256) void pp_count_voxels_kernel(int n, const u64* __restrict__ k, u64 invalid, int* __restrict__ counters) {
  __shared__ int s_tmp[16];
  int heads = 0, valids = 0;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
    const u64 key = k[j];
    const bool valid = key != invalid;
    valids += valid;
    heads += valid && (j == 0 || k[j - 1] != key);
  }
  heads = block_reduce_i<2>(heads, s_tmp);
  valids = block_reduce_i<2>(valids, s_tmp);
  if (threadIdx.x == 0) {
    if (heads) atomicAdd(&counters[0], heads);
    if (valids) atomicAdd(&counters[1], valids);
  }
}

// randomgrid_sampling: rank inside the voxel < points_per_voxel  <=>  the entry ppv places earlier belongs to another voxel
__global__ __launch_bounds__(256) void pp_select_kernel(int n, double rate, const u64* __restrict__ k, const u32* __restrict__ v, u64 invalid,
                                                        int* __restrict__ counters, int* __restrict__ sel) {
  __shared__ int s_tmp[16];
  const int num_voxels = counters[0];
  int kept = 0;
  if (num_voxels > 0) {
    const long long ppv = (long long)ceil((rate * (double)n) / (double)num_voxels);
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
      const u64 key = k[j];
      if (key != invalid && ((long long)j < ppv || k[j - ppv] != key)) {
        sel[v[j]] = 1;
        kept++;
      }
    }
  }
  kept = block_reduce_i<2>(kept, s_tmp);
  if (threadIdx.x == 0 && kept) atomicAdd(&counters[2], kept);
}

__global__ __launch_bounds__(256) void pp_cap_key_kernel(int n, u64 seed, const int* __restrict__ sel, u64* __restrict__ keys) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) keys[i] = sel[i] ? (sample_hash(seed, (u64)i) & 0xFFFFFFFFull) : (1ull << 32);
}

__global__ __launch_bounds__(256) void pp_cap_apply_kernel(int n, int max_num, const u32* __restrict__ v, int* __restrict__ sel) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < n && j >= max_num) sel[v[j]] = 0;
}

// ---- voxelgrid_sampling: run heads of the sorted order (key change, or a multiple of block_size), then one thread per run ----
__global__ __launch_bounds__(256) void pp_head_flag_kernel(int n, const u64* __restrict__ k, u64 invalid, int block_size, int* __restrict__ heads,
                                                           int* __restrict__ n_valid) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j > n) return;
  if (j == n) {
    heads[n] = 0;
    return;
  }
  const bool valid = k[j] != invalid;
  heads[j] = valid && (j == 0 || k[j - 1] != k[j] || (block_size > 0 && j % block_size == 0));
  if (valid && (j == n - 1 || k[j + 1] == invalid)) *n_valid = j + 1;
}

__global__ __launch_bounds__(256) void pp_seg_start_kernel(int n, const int* __restrict__ heads, const int* __restrict__ seg, const int* __restrict__ n_valid,
                                                           int* __restrict__ seg_start) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j > n) return;
  if (j == n) seg_start[seg[n]] = *n_valid;
  else if (heads[j]) seg_start[seg[j]] = j;
}

// the raw arrays permuted into the sorted order, so that the sequential walk below streams consecutive memory
__global__ __launch_bounds__(256) void pp_gather_sorted_kernel(int n, const u32* __restrict__ v, const double4* __restrict__ p4,
                                                               const double* __restrict__ times, const double* __restrict__ inten,
                                                               double4* __restrict__ sP, double* __restrict__ sT, double* __restrict__ sI) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const u32 i = v[j];
  sP[j] = p4[i];
  sT[j] = times[i];
  if (inten) sI[j] = inten[i];
}

__global__ __launch_bounds__(256) void pp_segment_mean_kernel(int m, const int* __restrict__ seg_start, const double4* __restrict__ sP,
                                                              const double* __restrict__ sT, const double* __restrict__ sI, double4* __restrict__ outP,
                                                              double* __restrict__ outT, double* __restrict__ outI) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= m) return;
  const int begin = seg_start[s], end = seg_start[s + 1];
  double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0, st = 0.0, si = 0.0;
#pragma unroll 4
  for (int j = begin; j < end; j++) {  // sequential: the same additions in the same order as the CPU loop
    const double4 p = sP[j];
    sx += p.x;
    sy += p.y;
    sz += p.z;
    sw += p.w;
    st += sT[j];
    if (sI) si += sI[j];
  }
  outP[s] = make_double4(sx / sw, sy / sw, sz / sw, sw / sw);
  outT[s] = st / sw;
  if (sI) outI[s] = si / sw;
}

// ---- range / finite / cropbox predicate (cloud_preprocessor.cpp:122-128, :146-160) ----
struct FilterParams {
  double near2, far2;
  int crop, crop_imu;
  double bmin[3], bmax[3], T[12];
};
inline void fill_filter_params(FilterParams& fp, const glim_amd_preprocess_params* prm) {
  fp.near2 = prm->distance_near_thresh * prm->distance_near_thresh;
  fp.far2 = prm->distance_far_thresh * prm->distance_far_thresh;
  fp.crop = prm->enable_cropbox_filter;
  fp.crop_imu = prm->crop_bbox_frame_imu;
  for (int a = 0; a < 3; a++) {
    fp.bmin[a] = prm->crop_bbox_min[a];
    fp.bmax[a] = prm->crop_bbox_max[a];
  }
  memcpy(fp.T, prm->T_imu_lidar, sizeof(fp.T));
}

__global__ __launch_bounds__(256) void pp_filter_flag_kernel(int m, const double4* __restrict__ P, const int* __restrict__ sel, FilterParams fp,
                                                             int* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i > m) return;
  if (i == m) {
    flags[m] = 0;
    return;
  }
  bool keep = sel ? sel[i] != 0 : true;
  if (keep) {
    const double4 p = P[i];
    const bool finite = isfinite(p.x) && isfinite(p.y) && isfinite(p.z) && isfinite(p.w);
    const double d2 = dadd(dadd(dmul(p.x, p.x), dmul(p.z, p.z)), dmul(p.y, p.y));
    keep = d2 > fp.near2 && d2 < fp.far2 && finite;
    if (keep && fp.crop) {
      double q[3] = {p.x, p.y, p.z};
      if (fp.crop_imu) {
#pragma unroll
        for (int r = 0; r < 3; r++)
          q[r] = dadd(dadd(dadd(dmul(fp.T[4 * r], p.x), dmul(fp.T[4 * r + 1], p.y)), dmul(fp.T[4 * r + 2], p.z)),
                           fp.T[4 * r + 3]);
      }
      bool inside = true;
#pragma unroll
      for (int a = 0; a < 3; a++) inside = inside && q[a] >= fp.bmin[a] && q[a] <= fp.bmax[a];
      keep = !inside;
    }
  }
  flags[i] = keep ? 1 : 0;
}

// order-preserving image of a double for an unsigned radix sort; -0.0 counts as +0.0 (they compare equal on the CPU)
__device__ __forceinline__ u64 orderable(double t) {
  const u64 b = (u64)__double_as_longlong(t + 0.0);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__global__ __launch_bounds__(256) void pp_compact_time_kernel(int m, const int* __restrict__ flags, const int* __restrict__ pos,
                                                              const double* __restrict__ T, u64* __restrict__ tkeys, u32* __restrict__ tvals) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m || !flags[i]) return;
  tkeys[pos[i]] = orderable(T[i]);
  tvals[pos[i]] = (u32)i;
}

__global__ __launch_bounds__(256) void pp_gather_out_kernel(int f, const u32* __restrict__ idx, const double4* __restrict__ P, const double* __restrict__ T,
                                                            const double* __restrict__ I, int zero_times, float4* __restrict__ pts,
                                                            double4* __restrict__ pts64, double* __restrict__ times, double* __restrict__ inten) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= f) return;
  const u32 i = idx[j];
  const double4 p = P[i];
  pts64[j] = p;
  pts[j] = make_float4((float)p.x, (float)p.y, (float)p.z, 1.0f);
  times[j] = zero_times ? 0.0 : T[i];
  if (inten) inten[j] = I[i];
}


// ============================================================================================================================================
// Fast path of the random-grid branch (the shipped configuration: 131 072 raw points -> ~10 000).  Round 3's form sorted the raw scan three
// times (by sample hash: 4 passes, by voxel: 3, for the 1.2x cap: 5 -- every frame of the shipped configuration hits the cap), sorted the
// survivors by time with 8 more passes and went back to the host four times for counters: 0.80-0.90 ms per scan, 23 radix passes of it 0.28 ms.
// Here ONE sort remains (by voxel); everything behind it is a counting rank on the device, sized by bounds the host knows in advance, with the
// actual counts read from device memory:
//   select   a point survives iff fewer than ppv points of its voxel have a smaller (hash >> 32, index): a walk over the voxel's run of the
//            sorted order that stops as soon as ppv smaller ones have been seen (most points stop after a step or two);
//   cap      when more than max_num survive: rank of every survivor among the survivors by (hash & 0xffffffff, index), counted against LDS tiles
//            of the compacted candidate list; rank >= max_num drops out;
//   time     the same counting rank on (time, index) replaces the stable radix sort by time.
// After the cap at most max_num points are alive, so the output cloud is allocated for that many up front.  The counting ranks are quadratic in
// the number of survivors: beyond RANK_FAST_MAX a flag is raised instead, and the caller repeats the call on the general (sorting) path.
// Same results as the sorting path bit for bit (tests/test_preprocess.py runs both).
// ============================================================================================================================================
constexpr int RANK_FAST_MAX = 32768;  // survivors the counting ranks accept
constexpr int RANK_TILE = 2048;       // keys per LDS tile of the counting rank
constexpr int RANK_SPLIT = 8;         // blocks that share the tiles of one group of 256 candidates
enum { C_VOXELS = 0, C_VALID = 1, C_KEPT = 2, C_FALLBACK = 3, C_CAND = 4, C_FINAL = 5, C_UNSORTED = 6, C_NUM = 8 };

__global__ __launch_bounds__(256) void pp_hash32_kernel(int n, u64 seed, u32* __restrict__ hhi, u32* __restrict__ hlo) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u64 h = sample_hash(seed, (u64)i);
  hhi[i] = (u32)(h >> 32);
  hlo[i] = (u32)(h & 0xFFFFFFFFull);
}

// ---- selection without a sort: voxels in a hash table, the ppv smallest (hash >> 32, index) of every voxel through an atomicMin cascade ----
constexpr int PPV_MAX = 4;  // points per voxel the cascade keeps (shipped configuration: 1-3); a larger quota takes the sorting path
constexpr u64 TABLE_EMPTY = ~0ull;

// every valid point finds (or claims) the table slot of its voxel: slot_of[i] (-1: invalid point); counters[C_VOXELS] = occupied slots
// Neighbouring raw indices are neighbouring points (scan order) and mostly share a voxel.  Device-scope atomics are served at the memory side
// of the fabric, one after the other per address (~10 ns): 131 072 points probing, claiming and cascading one by one cost 50 + 108 us.  The
// table kernels therefore work per GROUP of equal keys inside a wavefront: the lanes that hold the same voxel elect a leader, which alone
// touches the table -- one probe per group, and of a group's values only the ppv smallest go into the cascade.
__device__ __forceinline__ u64 readlane_u64(u64 v, int lane) {
  const unsigned int lo = (unsigned int)__shfl((int)(unsigned int)(v & 0xffffffffull), lane, 64);
  const unsigned int hi = (unsigned int)__shfl((int)(unsigned int)(v >> 32), lane, 64);
  return ((u64)hi << 32) | (u64)lo;
}
__device__ __forceinline__ u64 wave_min_u64(u64 v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned int lo = (unsigned int)__shfl_xor((int)(unsigned int)(v & 0xffffffffull), off, 64);
    const unsigned int hi = (unsigned int)__shfl_xor((int)(unsigned int)(v >> 32), off, 64);
    const u64 o = ((u64)hi << 32) | (u64)lo;
    v = o < v ? o : v;
  }
  return v;
}

__device__ __forceinline__ u64 pp_voxel_key_of(const double4 p, double inv_res) {
  const double t[3] = {p.x * inv_res, p.y * inv_res, p.z * inv_res};
  bool valid = isfinite(p.w);
#pragma unroll
  for (int a = 0; a < 3; a++) valid = valid && (t[a] >= -1048576.0 && t[a] < 1048576.0);  // false for NaN / inf (pp_key_kernel's rule)
  if (!valid) return TABLE_EMPTY;
  return (u64)(fast_floor_d(t[0]) + KEY_OFFSET) | ((u64)(fast_floor_d(t[1]) + KEY_OFFSET) << 21) | ((u64)(fast_floor_d(t[2]) + KEY_OFFSET) << 42);
}

// every valid point learns the table slot of its voxel: slot_of[i] (-1: invalid point); counters[C_VOXELS] = occupied slots
__global__ __launch_bounds__(256) void pp_table_insert_kernel(int n, const double4* __restrict__ p4, double inv_res, u64* __restrict__ table, unsigned int mask,
                                                              int* __restrict__ slot_of, int* __restrict__ counters) {
  __shared__ int s_new;
  if (threadIdx.x == 0) s_new = 0;
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const u64 key = i < n ? pp_voxel_key_of(p4[i], inv_res) : TABLE_EMPTY;
  int slot = -1;
  unsigned long long todo = __ballot(key != TABLE_EMPTY);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const u64 k = readlane_u64(key, leader);
    const unsigned long long grp = __ballot(key == k) & todo;
    int found = -1;
    if (lane == leader) {
      unsigned int h = (unsigned int)((k * 0x9E3779B97F4A7C15ull) >> 40) & mask;
      for (;;) {
        u64 cur = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == TABLE_EMPTY) {
          cur = atomicCAS(&table[h], TABLE_EMPTY, k);
          if (cur == TABLE_EMPTY) {
            atomicAdd(&s_new, 1);  // (LDS)
            cur = k;
          }
        }
        if (cur == k) break;
        h = (h + 1) & mask;
      }
      found = (int)h;
    }
    found = __shfl(found, leader, 64);
    if ((grp >> lane) & 1ull) slot = found;
    todo &= ~grp;
  }
  if (i < n) slot_of[i] = slot;
  __syncthreads();
  if (threadIdx.x == 0 && s_new) atomicAdd(&counters[C_VOXELS], s_new);  // one global atomic per block
}

// levels[slot][0 .. ppv) end up holding the ppv smallest packed (hash >> 32, index) of the voxel, ascending: a value offers itself to level 0
// with atomicMin and the larger of (what was there, itself) is carried on to the next level -- whatever the interleaving, level t receives
// every value except the final contents of the levels above it exactly once, so it ends as their minimum.  Per wavefront and voxel only the
// ppv smallest values of the group can matter; its leader offers them, smallest first, and stops at the first one that is larger than the
// current last level (levels only fall: it, and everything behind it, can never enter).
__global__ __launch_bounds__(256) void pp_cascade_kernel(int n, double rate, const int* __restrict__ slot_of, const u32* __restrict__ hhi, u64* __restrict__ levels,
                                                         int* __restrict__ counters) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int num_voxels = counters[C_VOXELS];
  if (num_voxels <= 0) return;
  const long long ppv = (long long)ceil((rate * (double)n) / (double)num_voxels);
  if (ppv > PPV_MAX) {
    if (i == 0) counters[C_FALLBACK] = 1;
    return;
  }
  int slot = i < n ? slot_of[i] : -1;
  u64 x = slot >= 0 ? (((u64)hhi[i] << 32) | (u64)(u32)i) : TABLE_EMPTY;
  // ONE parallel look at every lane's last level first: a value above it can never enter (levels only fall).  Once a voxel's first few
  // arrivals have settled its levels -- i.e. for almost every lane of a dense scan -- this load is all the kernel does.
  if (slot >= 0 && x > __hip_atomic_load(&levels[(size_t)slot * PPV_MAX + (ppv - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) slot = -1;
  // rank of this lane's value among the wavefront's values of the SAME voxel: only the ppv smallest of a voxel's values in this wavefront can
  // be among the voxel's ppv smallest overall.  Those lanes run their cascades side by side (the cascade is correct under any interleaving);
  // a leader walking a group's values one after the other made the kernel a chain of ppv^2 dependent returning atomics per group (76-108 us).
  int smaller = 0;
#pragma unroll
  for (int j = 0; j < 64; j++) {
    const int sj = __shfl(slot, j, 64);
    const u64 xj = readlane_u64(x, j);
    smaller += (int)(sj == slot) & (int)(xj < x);
  }
  if (slot >= 0 && smaller < (int)ppv) {
    u64* L = levels + (size_t)slot * PPV_MAX;
    u64 v = x;
    for (int q = 0; q < (int)ppv; q++) {
      const u64 old = atomicMin(&L[q], v);
      v = old > v ? old : v;
      if (v == TABLE_EMPTY) break;
    }
  }
}

__global__ __launch_bounds__(256) void pp_select_table_kernel(int n, double rate, const int* __restrict__ slot_of, const u32* __restrict__ hhi,
                                                              const u64* __restrict__ levels, int* __restrict__ counters, int* __restrict__ sel) {
  __shared__ int s_tmp[16];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int num_voxels = counters[C_VOXELS];
  int kept = 0;
  if (i < n && num_voxels > 0) {
    const long long ppv = (long long)ceil((rate * (double)n) / (double)num_voxels);
    const int slot = slot_of[i];
    if (slot >= 0 && ppv <= PPV_MAX) {
      const u64 me = ((u64)hhi[i] << 32) | (u64)(u32)i;
      const u64* L = levels + (size_t)slot * PPV_MAX;
      for (int t = 0; t < (int)ppv; t++) kept |= (L[t] == me);
      if (kept) sel[i] = 1;
    }
  }
  kept = block_reduce_i<2>(kept, s_tmp);
  if (threadIdx.x == 0 && kept) atomicAdd(&counters[C_KEPT], kept);
}

// candidates of the cap, compacted in index order: key = hash & 0xffffffff, value = index
__global__ __launch_bounds__(256) void pp_cap_compact_kernel(int n, const int* __restrict__ sel, const int* __restrict__ pos, const u32* __restrict__ hlo,
                                                             u64* __restrict__ ckeys, u32* __restrict__ cvals, int* __restrict__ counters, int max_num) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) {
    const int m = pos[n];
    counters[C_CAND] = m;
    if (m > max_num && m > RANK_FAST_MAX) counters[C_FALLBACK] = 1;
  }
  if (i >= n || !sel[i]) return;
  const int p = pos[i];
  if (p < RANK_FAST_MAX) {
    ckeys[p] = ((u64)hlo[i] << 32) | (u64)(u32)i;  // (hash & 0xffffffff, index) as ONE comparable word
    cvals[p] = (u32)i;
  }
}

// Counting rank: rank[j] += number of pairs of tiles s, s + RANK_SPLIT, ... that are smaller than pair j.  Block (x, s) serves the 256
// candidates of group x against its share of the tiles; *count_ptr candidates exist (<= RANK_FAST_MAX, else nothing is done); only_above: the
// rank is needed only when the count exceeds it (the cap), otherwise always (the time order).  PACKED: the key word alone orders the pairs
// (index in its low bits); otherwise ties of the key are broken by the value.  Branch-free inner loop: every lane reads the same LDS word.
template <bool PACKED>
__global__ __launch_bounds__(256) void pp_count_rank_kernel(const u64* __restrict__ keys, const u32* __restrict__ vals, const int* __restrict__ count_ptr,
                                                            int only_above, const int* __restrict__ needed, int* __restrict__ rank) {
  __shared__ u64 s_k[RANK_TILE];
  __shared__ u32 s_v[PACKED ? 1 : RANK_TILE];
  const int m = *count_ptr;
  if (m > RANK_FAST_MAX || m <= only_above || (needed && !*needed)) return;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if ((int)blockIdx.x * 256 >= m) return;
  const bool live = j < m;
  const u64 kj = live ? keys[j] : 0ull;
  const u32 vj = (live && !PACKED) ? vals[j] : 0u;
  int smaller = 0;
  for (int t = blockIdx.y; t * RANK_TILE < m; t += RANK_SPLIT) {
    const int base = t * RANK_TILE, len = min(RANK_TILE, m - base);
    __syncthreads();
    for (int i = threadIdx.x; i < RANK_TILE; i += 256) {
      const bool in = i < len;
      s_k[i] = in ? keys[base + i] : ~0ull;  // padding compares as "not smaller" (a real pair can carry ~0ull only with a larger value... see below)
      if (!PACKED) s_v[i] = in ? vals[base + i] : 0xffffffffu;
    }
    __syncthreads();
#pragma unroll 16
    for (int i = 0; i < RANK_TILE; i++) {
      const u64 ko = s_k[i];
      if (PACKED) smaller += (int)(ko < kj);
      else smaller += (int)(ko < kj) | ((int)(ko == kj) & (int)(s_v[i] < vj));
    }
  }
  if (live && smaller) atomicAdd(&rank[j], smaller);
}

// The cap needs no ranks, only the max_num-th smallest packed (hash & 0xffffffff, index) among the candidates: ONE block finds it by radix
// selection -- eight 8-bit digits from the top, a 256-bin LDS histogram of the candidates that share the prefix chosen so far -- and stores it;
// candidates above it leave the selection.  (A counting rank of the 14 400 candidates of the shipped configuration cost 42 us.)
__global__ __launch_bounds__(1024) void pp_cap_select_kernel(const u64* __restrict__ ckeys, const int* __restrict__ counters, int max_num, u64* __restrict__ threshold) {
  __shared__ int s_hist[256];
  __shared__ u64 s_prefix;
  __shared__ int s_k;
  const int m = counters[C_CAND];
  if (m <= max_num || m > RANK_FAST_MAX) return;
  if (threadIdx.x == 0) {
    s_prefix = 0ull;
    s_k = max_num;  // the k-th smallest (1-based) of what still matches the prefix
  }
  for (int shift = 56; shift >= 0; shift -= 8) {
    if (threadIdx.x < 256) s_hist[threadIdx.x] = 0;
    __syncthreads();
    const u64 prefix = s_prefix;
    for (int j = threadIdx.x; j < m; j += 1024) {
      const u64 v = ckeys[j];
      const bool match = shift == 56 || (v >> (shift + 8)) == (prefix >> (shift + 8));
      if (match) atomicAdd(&s_hist[(int)((v >> shift) & 0xffull)], 1);
    }
    __syncthreads();
    // the digit whose cumulative count first reaches k: inclusive scan of the 256 bins by the first four wavefronts (a serial walk of the
    // bins by one thread is 256 dependent LDS reads per digit: 60 us for the eight digits)
    __shared__ int s_wave_tot[4];
    int incl = 0, mine = 0;
    if (threadIdx.x < 256) {
      mine = s_hist[threadIdx.x];
      incl = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if ((int)(threadIdx.x & 63) >= off) incl += o;
      }
      if ((threadIdx.x & 63) == 63) s_wave_tot[threadIdx.x >> 6] = incl;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
      const int w = threadIdx.x >> 6;
      for (int q = 0; q < w; q++) incl += s_wave_tot[q];
      const int k = s_k;
      if (incl >= k && incl - mine < k) {  // exactly one bin qualifies
        s_k = k - (incl - mine);
        s_prefix = prefix | ((u64)threadIdx.x << shift);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *threshold = s_prefix;
}

__global__ __launch_bounds__(256) void pp_cap_drop_kernel(const u64* __restrict__ ckeys, const u32* __restrict__ cvals, const u64* __restrict__ threshold,
                                                          const int* __restrict__ counters, int max_num, int* __restrict__ sel) {
  const int m = counters[C_CAND];
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (m <= max_num || m > RANK_FAST_MAX || j >= m) return;
  if (ckeys[j] > *threshold) sel[cvals[j]] = 0;
}

// filtered survivors -> (orderable time, raw index) pairs in index order; their number goes to counters[C_FINAL]
__global__ __launch_bounds__(256) void pp_compact_time_fast_kernel(int m, const int* __restrict__ flags, const int* __restrict__ pos, const double* __restrict__ T,
                                                                   u64* __restrict__ tkeys, u32* __restrict__ tvals, int* __restrict__ counters) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) {
    const int f = pos[m];
    counters[C_FINAL] = f;
    if (f > RANK_FAST_MAX) counters[C_FALLBACK] = 1;
  }
  if (i >= m || !flags[i]) return;
  const int p = pos[i];
  if (p < RANK_FAST_MAX) {
    tkeys[p] = orderable(T[i]);
    tvals[p] = (u32)i;
  }
}

// The survivors arrive in index order, which for most sensors already IS time order: one pass over neighbouring pairs sets counters[C_UNSORTED]
// when it is not; the counting rank below then runs, otherwise the rank of survivor j is j.
__global__ __launch_bounds__(256) void pp_check_sorted_kernel(const u64* __restrict__ tkeys, int* __restrict__ counters) {
  const int f = counters[C_FINAL];
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (f > RANK_FAST_MAX || j + 1 >= f) return;
  if (tkeys[j] > tkeys[j + 1]) counters[C_UNSORTED] = 1;  // (equal stamps keep their index order)
}

// output position = rank in the (time, index) order
__global__ __launch_bounds__(256) void pp_gather_out_ranked_kernel(const int* __restrict__ counters, const int* __restrict__ rank, const u32* __restrict__ tvals,
                                                                   const double4* __restrict__ P, const double* __restrict__ T, const double* __restrict__ I,
                                                                   int zero_times, float4* __restrict__ pts, double4* __restrict__ pts64,
                                                                   double* __restrict__ times, double* __restrict__ inten) {
  const int f = counters[C_FINAL];
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (f > RANK_FAST_MAX || j >= f) return;
  const u32 i = tvals[j];
  const int o = counters[C_UNSORTED] ? rank[j] : j;
  const double4 p = P[i];
  pts64[o] = p;
  pts[o] = make_float4((float)p.x, (float)p.y, (float)p.z, 1.0f);
  times[o] = zero_times ? 0.0 : T[i];
  if (inten) inten[o] = I[i];
}

// ---- statistical outlier removal (gtsam_points::remove_outliers): mean distance to the k nearest neighbours ----
__global__ __launch_bounds__(256) void pp_mean_dist_kernel(int f, const double4* __restrict__ P, const int* __restrict__ nb, int k,
                                                           double* __restrict__ d, double* __restrict__ partial) {
  __shared__ double s_sum[4], s_sq[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double di = 0.0;
  if (i < f) {
    const double4 p = P[i];
    double s = 0.0;
    for (int j = 0; j < k; j++) {
      const double4 q = P[nb[(size_t)i * k + j]];
      const double dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
      s += __dsqrt_rn(dadd(dadd(dmul(dx, dx), dmul(dy, dy)), dmul(dz, dz)));
    }
    di = s / (double)k;
    d[i] = di;
  }
  // deterministic block sums of d and d^2 (fixed butterfly order), one partial pair per block
  double a = di, b = dmul(di, di);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    a += __shfl_xor(a, off, 64);
    b += __shfl_xor(b, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    s_sum[threadIdx.x >> 6] = a;
    s_sq[threadIdx.x >> 6] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
    partial[2 * blockIdx.x + 1] = (s_sq[0] + s_sq[1]) + (s_sq[2] + s_sq[3]);
  }
}

__global__ void pp_outlier_thresh_kernel(int f, int blocks, const double* __restrict__ partial, double std_mul, double* __restrict__ thresh) {
  double sum = 0.0, sq = 0.0;
  for (int b = 0; b < blocks; b++) {  // block order: deterministic
    sum += partial[2 * b];
    sq += partial[2 * b + 1];
  }
  const double mean = sum / (double)f;
  const double var = sq / (double)f - dmul(mean, mean);
  *thresh = dadd(mean, dmul(std_mul, sqrt(var > 0.0 ? var : 0.0)));
}

__global__ __launch_bounds__(256) void pp_inlier_flag_kernel(int f, const double* __restrict__ d, const double* __restrict__ thresh,
                                                             int* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i > f) return;
  flags[i] = (i < f && d[i] < *thresh) ? 1 : 0;
}

__global__ __launch_bounds__(256) void pp_compact_index_kernel(int m, const int* __restrict__ flags, const int* __restrict__ pos, u32* __restrict__ idx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < m && flags[i]) idx[pos[i]] = (u32)i;
}

// ---- submap merge (gtsam_points::merge_frames, sub_mapping.cpp:480-497): p' = T p, C' = (R C) R^T upper triangle ----
struct Pose12 {
  double m[12];
};

__global__ __launch_bounds__(256) void mg_transform_kernel(int n, const double4* __restrict__ p4, const double* __restrict__ c16, Pose12 T,
                                                           double4* __restrict__ P, double* __restrict__ C6) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double4 p = p4[i];
  double q[3];
#pragma unroll
  for (int r = 0; r < 3; r++)
    q[r] = dadd(dadd(dadd(dmul(T.m[4 * r], p.x), dmul(T.m[4 * r + 1], p.y)), dmul(T.m[4 * r + 2], p.z)),
                     dmul(T.m[4 * r + 3], p.w));
  P[i] = make_double4(q[0], q[1], q[2], p.w);
  const double* c = c16 + 16 * (size_t)i;  // column-major Matrix4d: (row, col) = c[4 * col + row]
  double RC[3][3];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int k = 0; k < 3; k++)
      RC[r][k] = dadd(dadd(dmul(T.m[4 * r], c[4 * k]), dmul(T.m[4 * r + 1], c[4 * k + 1])), dmul(T.m[4 * r + 2], c[4 * k + 2]));
  int o = 0;
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int k = r; k < 3; k++)
      C6[6 * (size_t)i + o++] =
        dadd(dadd(dmul(RC[r][0], T.m[4 * k]), dmul(RC[r][1], T.m[4 * k + 1])), dmul(RC[r][2], T.m[4 * k + 2]));
}

__global__ __launch_bounds__(256) void mg_gather_sorted_kernel(int n, const u32* __restrict__ v, const double4* __restrict__ P, const double* __restrict__ C6,
                                                               double4* __restrict__ sP, double* __restrict__ sC) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const u32 i = v[j];
  sP[j] = P[i];
#pragma unroll
  for (int a = 0; a < 6; a++) sC[6 * (size_t)j + a] = C6[6 * (size_t)i + a];
}

__global__ __launch_bounds__(256) void mg_segment_mean_kernel(int m, const int* __restrict__ seg_start, const double4* __restrict__ sP,
                                                              const double* __restrict__ sC, double4* __restrict__ outP, double* __restrict__ outC) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= m) return;
  const int begin = seg_start[s], end = seg_start[s + 1];
  double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0, sc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int j = begin; j < end; j++) {  // sequential: the same additions in the same order as the CPU loop
    const double4 p = sP[j];
    sx += p.x;
    sy += p.y;
    sz += p.z;
    sw += p.w;
#pragma unroll
    for (int a = 0; a < 6; a++) sc[a] += sC[6 * (size_t)j + a];
  }
  outP[s] = make_double4(sx / sw, sy / sw, sz / sw, sw / sw);
#pragma unroll
  for (int a = 0; a < 6; a++) outC[6 * (size_t)s + a] = sc[a] / sw;
}

__global__ __launch_bounds__(256) void mg_mark_kernel(int m, int keep, const u32* __restrict__ v, int* __restrict__ flags) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j > m) return;
  if (j == m) flags[m] = 0;
  else flags[v[j]] = j < keep ? 1 : 0;
}

// final gather: exact FP64 values + the FP32 SoA image the factor path streams (idx == nullptr: identity)
__global__ __launch_bounds__(256) void mg_output_kernel(int n, const u32* __restrict__ idx, const double4* __restrict__ P, const double* __restrict__ C6,
                                                        double4* __restrict__ pts64, double* __restrict__ cov64, float4* __restrict__ pts,
                                                        float4* __restrict__ covA, float2* __restrict__ covB) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const u32 i = idx ? idx[j] : (u32)j;
  const double4 p = P[i];
  double c[6];
#pragma unroll
  for (int a = 0; a < 6; a++) c[a] = C6[6 * (size_t)i + a];
  pts64[j] = p;
#pragma unroll
  for (int a = 0; a < 6; a++) cov64[6 * (size_t)j + a] = c[a];
  pts[j] = make_float4((float)p.x, (float)p.y, (float)p.z, 1.0f);
  covA[j] = make_float4((float)c[0], (float)c[1], (float)c[2], (float)c[3]);  // c00 c01 c02 c11
  covB[j] = make_float2((float)c[4], (float)c[5]);                            // c12 c22
}

inline int grid_for(int n) { return (n + 255) / 256; }
// grid-stride kernels that end in one set of atomics per block: few, fat blocks
inline int reduce_grid_for(int n) { return std::max(1, std::min((n + 2047) / 2048, 128)); }
inline int bits_for(int range) {
  int b = 0;
  while (range > 0) {
    b++;
    range >>= 1;
  }
  return b;
}

struct CloudGuard {
  glim_amd_cloud* c = nullptr;
  ~CloudGuard() {
    if (c) glim_amd_cloud_destroy(c);
  }
  glim_amd_cloud* release() {
    glim_amd_cloud* r = c;
    c = nullptr;
    return r;
  }
};

// Allocates the output arrays of a preprocessed cloud of f points.
int alloc_frame_cloud(glim_amd_ctx* ctx, int f, bool with_intensities, glim_amd_cloud** out) {
  glim_amd_cloud* c = new glim_amd_cloud();
  c->ctx = ctx;
  c->n = f;
  const size_t nn = (size_t)(f > 0 ? f : 1);
  hipError_t e = pool_malloc(&c->pts, nn * sizeof(float4));
  if (e == hipSuccess) e = pool_malloc(&c->pts64, nn * sizeof(double4));
  if (e == hipSuccess) e = pool_malloc(&c->times, nn * sizeof(double));
  if (e == hipSuccess && with_intensities) e = pool_malloc(&c->intensities, nn * sizeof(double));
  if (e != hipSuccess) {
    set_hip_error(e, "pool_malloc(preprocessed cloud)");
    glim_amd_cloud_destroy(c);
    return e == hipErrorOutOfMemory ? GLIM_AMD_ERR_NOMEM : GLIM_AMD_ERR_HIP;
  }
  *out = c;
  return GLIM_AMD_OK;
}

// Sort scratch shared by the stages of one call (sized for the raw scan).
struct SortBuffers {
  DeviceTemp ka, kb, va, vb, hist;
  int alloc(int n) {
    const size_t nn = (size_t)(n > 0 ? n : 1);
    GA_HIP(pool_malloc(&ka.p, nn * sizeof(u64)));
    GA_HIP(pool_malloc(&kb.p, nn * sizeof(u64)));
    GA_HIP(pool_malloc(&va.p, nn * sizeof(u32)));
    GA_HIP(pool_malloc(&vb.p, nn * sizeof(u32)));
    GA_HIP(pool_malloc(&hist.p, radix_sort_scratch_bytes(n)));
    return GLIM_AMD_OK;
  }
};

// Stage A (downsampling) on the raw device arrays.  Random grid: fills `sel` (n ints, 1 = survives) and leaves P/T/I = the raw
// arrays; voxel grid: writes the averaged arrays (avgP/avgT/avgI, *m entries).
int downsample_random(glim_amd_ctx* ctx, hipStream_t st, int n, const double4* p4, const u64* ckey, int vbits, double rate, u64 seed, SortBuffers& sb, int* counters,
                      int* h_counters, int* sel) {
  u64* ks = nullptr;
  u32* vs = nullptr;
  // sort 1: by hash (ties by index through stability); sort 2: by voxel -> inside every voxel ascending (hash, index)
  pp_hash_key_kernel<<<grid_for(n), 256, 0, st>>>(n, seed, sb.ka.as<u64>());
  GA_HIP(radix_sort_pairs(st, n, 32, sb.ka.as<u64>(), sb.va.as<u32>(), sb.kb.as<u64>(), sb.vb.as<u32>(), true, sb.hist.as<int>(), &ks, &vs));
  // after 4 passes the pairs are back in (ka, va); gather the voxel keys into kb, keeping the permutation in va
  u64* k2 = (ks == sb.ka.as<u64>()) ? sb.kb.as<u64>() : sb.ka.as<u64>();
  u32* v_other = (vs == sb.va.as<u32>()) ? sb.vb.as<u32>() : sb.va.as<u32>();
  pp_gather_key_kernel<<<grid_for(n), 256, 0, st>>>(n, vs, ckey, k2);
  u64* k_other = (k2 == sb.ka.as<u64>()) ? sb.kb.as<u64>() : sb.ka.as<u64>();
  GA_HIP(radix_sort_pairs(st, n, vbits + 1, k2, vs, k_other, v_other, false, sb.hist.as<int>(), &ks, &vs));
  const u64 invalid = 1ull << vbits;
  GA_HIP(hipMemsetAsync(counters, 0, 4 * sizeof(int), st));
  GA_HIP(hipMemsetAsync(sel, 0, (size_t)n * sizeof(int), st));
  pp_count_voxels_kernel<<<reduce_grid_for(n), 256, 0, st>>>(n, ks, invalid, counters);
  pp_select_kernel<<<reduce_grid_for(n), 256, 0, st>>>(n, rate, ks, vs, invalid, counters, sel);
  GA_HIP(hipGetLastError());
  GA_HIP(read_back_sync(ctx, st, h_counters, counters, 4 * sizeof(int)));
  const long long max_num_points = (long long)((double)n * rate * 1.2);
  if ((long long)h_counters[2] > max_num_points) {
    pp_cap_key_kernel<<<grid_for(n), 256, 0, st>>>(n, seed, sel, sb.ka.as<u64>());
    GA_HIP(radix_sort_pairs(st, n, 33, sb.ka.as<u64>(), sb.va.as<u32>(), sb.kb.as<u64>(), sb.vb.as<u32>(), true, sb.hist.as<int>(), &ks, &vs));
    pp_cap_apply_kernel<<<grid_for(n), 256, 0, st>>>(n, (int)max_num_points, vs, sel);
    GA_HIP(hipGetLastError());
  }
  return GLIM_AMD_OK;
}

int downsample_voxelgrid(glim_amd_ctx* ctx, hipStream_t st, int n, const double4* p4, const double* times, const double* inten, const u64* ckey, int vbits, int block_size,
                         SortBuffers& sb, int* counters, int* h_counters, DeviceTemp& avgP, DeviceTemp& avgT, DeviceTemp& avgI, int* m_out) {
  u64* ks = nullptr;
  u32* vs = nullptr;
  GA_HIP(hipMemcpyAsync(sb.ka.p, ckey, (size_t)n * sizeof(u64), hipMemcpyDeviceToDevice, st));
  GA_HIP(radix_sort_pairs(st, n, vbits + 1, sb.ka.as<u64>(), sb.va.as<u32>(), sb.kb.as<u64>(), sb.vb.as<u32>(), true, sb.hist.as<int>(), &ks, &vs));
  DeviceTemp heads, seg, tiles, seg_start;
  GA_HIP(pool_malloc(&heads.p, (size_t)(n + 1) * sizeof(int)));
  GA_HIP(pool_malloc(&seg.p, (size_t)(n + 1) * sizeof(int)));
  GA_HIP(pool_malloc(&tiles.p, scan_scratch_ints((unsigned int)n + 1) * sizeof(int)));
  GA_HIP(hipMemsetAsync(counters, 0, 4 * sizeof(int), st));
  pp_head_flag_kernel<<<grid_for(n + 1), 256, 0, st>>>(n, ks, 1ull << vbits, block_size, heads.as<int>(), counters + 1);
  GA_HIP(exclusive_scan_int(st, heads.as<int>(), (unsigned int)n + 1, tiles.as<int>(), seg.as<int>()));
  GA_HIP(read_back_sync(ctx, st, h_counters, seg.as<int>() + n, sizeof(int)));
  const int m = h_counters[0];
  *m_out = m;
  const size_t mm = (size_t)(m > 0 ? m : 1);
  GA_HIP(pool_malloc(&avgP.p, mm * sizeof(double4)));
  GA_HIP(pool_malloc(&avgT.p, mm * sizeof(double)));
  if (inten) GA_HIP(pool_malloc(&avgI.p, mm * sizeof(double)));
  if (m == 0) return GLIM_AMD_OK;
  GA_HIP(pool_malloc(&seg_start.p, (size_t)(m + 1) * sizeof(int)));
  pp_seg_start_kernel<<<grid_for(n + 1), 256, 0, st>>>(n, heads.as<int>(), seg.as<int>(), counters + 1, seg_start.as<int>());
  DeviceTemp sP, sT, sI;
  GA_HIP(pool_malloc(&sP.p, (size_t)n * sizeof(double4)));
  GA_HIP(pool_malloc(&sT.p, (size_t)n * sizeof(double)));
  if (inten) GA_HIP(pool_malloc(&sI.p, (size_t)n * sizeof(double)));
  pp_gather_sorted_kernel<<<grid_for(n), 256, 0, st>>>(n, vs, p4, times, inten, sP.as<double4>(), sT.as<double>(), sI.as<double>());
  pp_segment_mean_kernel<<<grid_for(m), 256, 0, st>>>(m, seg_start.as<int>(), sP.as<double4>(), sT.as<double>(), sI.as<double>(), avgP.as<double4>(),
                                                      avgT.as<double>(), avgI.as<double>());
  GA_HIP(hipGetLastError());
  GA_HIP(hipStreamSynchronize(st));  // the scratch of this scope is released on return
  return GLIM_AMD_OK;
}


// The random-grid branch on its fast path (see "Fast path of the random-grid branch" above): everything from the uploaded raw scan to the
// output cloud with ONE synchronise and no sort.  *fallback = true: the quota per voxel or the number of survivors is beyond what the fast path
// accepts -- nothing was produced, the caller takes the sorting path.  Caller holds ctx->mu.
int preprocess_random_fast(glim_amd_ctx* ctx, hipStream_t st, int n, const double4* P, const double* T, const double* I, double rate,
                           const glim_amd_preprocess_params* prm, const FilterParams& fp, SortBuffers& sb, bool has_int, glim_amd_cloud** out, bool* fallback) {
  *fallback = false;
  *out = nullptr;
  const long long max_num = (long long)((double)n * rate * 1.2);
  if (max_num < 1 || max_num > RANK_FAST_MAX) {
    *fallback = true;
    return GLIM_AMD_OK;
  }
  unsigned int slots = 1024;
  while (slots < 2u * (unsigned int)n) slots <<= 1;
  // every piece of scratch exists before the first launch (an error return must not hand memory back to the pool under running kernels)
  // two blocks of scratch so that ONE fill each initialises them: [table | levels] = 0xff, [counters | rank x 2 | sel] = 0
  DeviceTemp hh, pos, tiles, flags, cvals, ones, zeros, slot_of;
  const size_t ones_bytes = (size_t)slots * (1 + PPV_MAX) * sizeof(u64);
  const size_t zeros_ints = (size_t)C_NUM + 2 * (size_t)RANK_FAST_MAX + (size_t)(n + 1);
  GA_HIP(pool_malloc(&hh.p, (size_t)n * 2 * sizeof(u32)));
  GA_HIP(pool_malloc(&pos.p, (size_t)(n + 1) * sizeof(int)));
  GA_HIP(pool_malloc(&tiles.p, scan_scratch_ints((unsigned int)n + 1) * sizeof(int)));
  GA_HIP(pool_malloc(&flags.p, (size_t)(n + 1) * sizeof(int)));
  GA_HIP(pool_malloc(&cvals.p, (size_t)RANK_FAST_MAX * sizeof(u32)));
  GA_HIP(pool_malloc(&ones.p, ones_bytes));
  GA_HIP(pool_malloc(&zeros.p, zeros_ints * sizeof(int)));
  GA_HIP(pool_malloc(&slot_of.p, (size_t)n * sizeof(int)));
  u64* table_p = ones.as<u64>();
  u64* levels_p = ones.as<u64>() + slots;
  int* sel_p = zeros.as<int>() + C_NUM + 2 * RANK_FAST_MAX;
  CloudGuard result;
  GA_TRY(alloc_frame_cloud(ctx, (int)max_num, has_int, &result.c));  // at most max_num points survive the cap
  glim_amd_cloud* c = result.c;
  double* h_times_stage = nullptr;
  if (pinned_malloc(&h_times_stage, (size_t)max_num * sizeof(double)) != hipSuccess) {
    (void)hipGetLastError();
    h_times_stage = nullptr;
  }
  u32 *hhi = hh.as<u32>(), *hlo = hh.as<u32>() + n;
  int* cnt = zeros.as<int>();
  int *rank_cap = zeros.as<int>() + C_NUM, *rank_time = zeros.as<int>() + C_NUM + RANK_FAST_MAX;
  u64* ckeys = sb.ka.as<u64>();  // (the sort scratch of the general path: >= n words)
  hipError_t e = hipSuccess;
  int rc = GLIM_AMD_OK;
  do {
    if ((e = hipMemsetAsync(zeros.p, 0, zeros_ints * sizeof(int), st)) != hipSuccess) break;
    if ((e = hipMemsetAsync(ones.p, 0xff, ones_bytes, st)) != hipSuccess) break;
    // ---- selection (randomgrid_sampling): voxel table, the ppv smallest (hash >> 32, index) per voxel ----
    pp_hash32_kernel<<<grid_for(n), 256, 0, st>>>(n, prm->seed, hhi, hlo);
    pp_table_insert_kernel<<<grid_for(n), 256, 0, st>>>(n, P, 1.0 / prm->downsample_resolution, table_p, slots - 1, slot_of.as<int>(), cnt);
    pp_cascade_kernel<<<grid_for(n), 256, 0, st>>>(n, rate, slot_of.as<int>(), hhi, levels_p, cnt);
    pp_select_table_kernel<<<grid_for(n), 256, 0, st>>>(n, rate, slot_of.as<int>(), hhi, levels_p, cnt, sel_p);
    // ---- the 1.2x cap: survivors compacted, ranked by (hash & 0xffffffff, index), the tail dropped ----
    if ((e = exclusive_scan_int(st, sel_p, (unsigned int)n + 1, tiles.as<int>(), pos.as<int>())) != hipSuccess) break;
    pp_cap_compact_kernel<<<grid_for(n), 256, 0, st>>>(n, sel_p, pos.as<int>(), hlo, ckeys, cvals.as<u32>(), cnt, (int)max_num);
    const dim3 rank_grid((unsigned int)(RANK_FAST_MAX / 256), RANK_SPLIT);
    u64* threshold = reinterpret_cast<u64*>(rank_cap);  // (8-byte aligned: C_NUM ints precede it)
    pp_cap_select_kernel<<<1, 1024, 0, st>>>(ckeys, cnt, (int)max_num, threshold);
    pp_cap_drop_kernel<<<RANK_FAST_MAX / 256, 256, 0, st>>>(ckeys, cvals.as<u32>(), threshold, cnt, (int)max_num, sel_p);
    // ---- range + cropbox filter, compaction, order by time (counting rank), output ----
    pp_filter_flag_kernel<<<grid_for(n + 1), 256, 0, st>>>(n, P, sel_p, fp, flags.as<int>());
    if ((e = exclusive_scan_int(st, flags.as<int>(), (unsigned int)n + 1, tiles.as<int>(), pos.as<int>())) != hipSuccess) break;
    u64* tkeys = ckeys;  // (the cap is done with its candidates)
    u32* tvals = cvals.as<u32>();
    pp_compact_time_fast_kernel<<<grid_for(n), 256, 0, st>>>(n, flags.as<int>(), pos.as<int>(), T, tkeys, tvals, cnt);
    pp_check_sorted_kernel<<<RANK_FAST_MAX / 256, 256, 0, st>>>(tkeys, cnt);
    pp_count_rank_kernel<false><<<rank_grid, 256, 0, st>>>(tkeys, tvals, cnt + C_FINAL, -1, cnt + C_UNSORTED, rank_time);
    pp_gather_out_ranked_kernel<<<(unsigned int)grid_for((int)max_num), 256, 0, st>>>(cnt, rank_time, tvals, P, T, I, prm->global_shutter, c->pts, c->pts64, c->times,
                                                                                    c->intensities);
    if ((e = hipGetLastError()) != hipSuccess) break;
    if (h_times_stage && (e = hipMemcpyAsync(h_times_stage, c->times, (size_t)max_num * sizeof(double), hipMemcpyDeviceToHost, st)) != hipSuccess) break;
    int h_cnt[C_NUM];
    if ((e = read_back_sync(ctx, st, h_cnt, cnt, sizeof(h_cnt))) != hipSuccess) break;
    if (h_cnt[C_FALLBACK]) {
      *fallback = true;
      break;
    }
    const int f = h_cnt[C_FINAL];
    c->n = f;
    if (h_times_stage) c->h_times.assign(h_times_stage, h_times_stage + f);
  } while (0);
  if (e != hipSuccess) {
    (void)hipStreamSynchronize(st);  // nothing enqueued may outlive the scratch released below
    set_hip_error(e, "preprocess (random-grid fast path)");
    rc = GLIM_AMD_ERR_HIP;
  }
  if (h_times_stage) (void)pinned_free(h_times_stage);
  if (rc == GLIM_AMD_OK && !*fallback) *out = result.release();
  return rc;
}

}  // namespace

extern "C" {

int glim_amd_preprocess_default_params(glim_amd_preprocess_params* p) {
  if (!p) return GLIM_AMD_ERR_INVALID;
  memset(p, 0, sizeof(*p));
  p->distance_near_thresh = 0.5;
  p->distance_far_thresh = 100.0;
  p->use_random_grid_downsampling = 1;
  p->downsample_target = 10000;
  p->downsample_resolution = 1.0;
  p->downsample_rate = 0.1;
  p->outlier_removal_k = 10;
  p->outlier_std_mul_factor = 1.0;
  for (int a = 0; a < 3; a++) {
    p->crop_bbox_min[a] = -1.0;
    p->crop_bbox_max[a] = 1.0;
    p->T_imu_lidar[5 * a] = 1.0;
  }
  p->k_correspondences = 10;
  p->voxelgrid_block_size = 1024;
  return GLIM_AMD_OK;
}

int glim_amd_preprocess(glim_amd_ctx* ctx, int64_t n64, const double* points4, const double* times, const double* intensities,
                        const glim_amd_preprocess_params* prm, glim_amd_cloud** out) {
  if (!ctx || !out || !prm || n64 < 0 || (n64 > 0 && (!points4 || !times))) return GLIM_AMD_ERR_INVALID;
  if (n64 > (int64_t)(1 << 28)) return GLIM_AMD_ERR_INVALID;
  if (!(prm->downsample_resolution > 0.0) || prm->k_correspondences < 0 || prm->k_correspondences > 32) return GLIM_AMD_ERR_INVALID;
  if (prm->enable_outlier_removal && (prm->outlier_removal_k <= 0 || prm->outlier_removal_k > 32)) return GLIM_AMD_ERR_INVALID;
  *out = nullptr;
  const int n = (int)n64;
  const bool has_int = intensities != nullptr;
  CloudGuard result;

  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    GA_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream();
    if (n == 0) {
      GA_TRY(alloc_frame_cloud(ctx, 0, has_int, &result.c));
    } else {
      // ---- upload the raw scan ----
      DeviceTemp d_p4, d_t, d_i, d_vkey, d_ckey, d_bb, d_counters, d_sel, avgP, avgT, avgI, flags, pos, tiles;
      SortBuffers sb;
      PinnedTemp times_stage;
      SyncOnExit in_flight(st);  // declared after every temporary of this scope: an early return waits for the stream before they are released
      GA_HIP(pool_malloc(&d_p4.p, (size_t)n * sizeof(double4)));
      GA_HIP(pool_malloc(&d_t.p, (size_t)n * sizeof(double)));
      if (has_int) GA_HIP(pool_malloc(&d_i.p, (size_t)n * sizeof(double)));
      GA_HIP(pool_malloc(&d_vkey.p, (size_t)n * sizeof(u64)));
      GA_HIP(pool_malloc(&d_ckey.p, (size_t)n * sizeof(u64)));
      GA_HIP(pool_malloc(&d_bb.p, 6 * sizeof(int)));
      GA_HIP(pool_malloc(&d_counters.p, 4 * sizeof(int)));
      GA_HIP(hipMemcpyAsync(d_p4.p, points4, (size_t)n * sizeof(double4), hipMemcpyHostToDevice, st));
      GA_HIP(hipMemcpyAsync(d_t.p, times, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
      if (has_int) GA_HIP(hipMemcpyAsync(d_i.p, intensities, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
      GA_TRY(sb.alloc(n));
      int h_counters[4] = {0, 0, 0, 0};

      // ---- downsampling (cloud_preprocessor.cpp:103-109) ----
      const double rate = prm->downsample_target > 0 ? (double)prm->downsample_target / (double)n : prm->downsample_rate;
      const bool random = prm->use_random_grid_downsampling != 0;
      const bool sample_all = random && rate >= 0.99;  // randomgrid_sampling returns the cloud unchanged
      const double4* P = d_p4.as<double4>();
      const double* T = d_t.as<double>();
      const double* I = has_int ? d_i.as<double>() : nullptr;
      const int* sel = nullptr;
      bool done = false;  // the random-grid fast path has produced the output cloud
      int m = n;
      if (!sample_all && random && ctx->diag.pp_fast) {
        // the shipped configuration: no sort, no bounding box, one synchronise (see "Fast path of the random-grid branch")
        FilterParams fp0;
        fill_filter_params(fp0, prm);
        bool fallback = false;
        glim_amd_cloud* fast = nullptr;
        GA_TRY(preprocess_random_fast(ctx, st, n, P, T, I, rate, prm, fp0, sb, has_int, &fast, &fallback));
        if (!fallback) {
          result.c = fast;
          done = true;
        }
      }
      if (!sample_all && !done) {
        int h_bb[6];
        init_bbox_kernel<<<1, 64, 0, st>>>(d_bb.as<int>());
        pp_key_kernel<<<reduce_grid_for(n), 256, 0, st>>>(n, d_p4.as<double4>(), 1.0 / prm->downsample_resolution, d_vkey.as<u64>(), d_bb.as<int>());
        GA_HIP(hipGetLastError());
        GA_HIP(read_back_sync(ctx, st, h_bb, d_bb.p, sizeof(h_bb)));
        int bx = 0, by = 0, bz = 0;
        if (h_bb[0] <= h_bb[3]) {
          bx = bits_for(h_bb[3] - h_bb[0]);
          by = bits_for(h_bb[4] - h_bb[1]);
          bz = bits_for(h_bb[5] - h_bb[2]);
        } else {
          h_bb[0] = h_bb[1] = h_bb[2] = 0;  // no valid point at all
        }
        const int vbits = bx + by + bz;
        pp_compact_key_kernel<<<grid_for(n), 256, 0, st>>>(n, d_vkey.as<u64>(), h_bb[0], h_bb[1], h_bb[2], bx, by, vbits, d_ckey.as<u64>());
        GA_HIP(hipGetLastError());
        if (random) {
          GA_HIP(pool_malloc(&d_sel.p, (size_t)n * sizeof(int)));
          GA_TRY(downsample_random(ctx, st, n, P, d_ckey.as<u64>(), vbits, rate, prm->seed, sb, d_counters.as<int>(), h_counters, d_sel.as<int>()));
          sel = d_sel.as<int>();
        } else {
          GA_TRY(downsample_voxelgrid(ctx, st, n, P, T, I, d_ckey.as<u64>(), vbits, prm->voxelgrid_block_size, sb, d_counters.as<int>(), h_counters, avgP,
                                      avgT, avgI, &m));
          P = avgP.as<double4>();
          T = avgT.as<double>();
          I = has_int ? avgI.as<double>() : nullptr;
        }
      }

      // ---- range + cropbox filter (:117-128, :143-160), compaction, sort by time (:134-136), global shutter (:138-140) ----
      int f = 0;
      u32* order = nullptr;
      if (!done && m > 0) {
        FilterParams fp;
        fill_filter_params(fp, prm);
        GA_HIP(pool_malloc(&flags.p, (size_t)(m + 1) * sizeof(int)));
        GA_HIP(pool_malloc(&pos.p, (size_t)(m + 1) * sizeof(int)));
        GA_HIP(pool_malloc(&tiles.p, scan_scratch_ints((unsigned int)m + 1) * sizeof(int)));
        pp_filter_flag_kernel<<<grid_for(m + 1), 256, 0, st>>>(m, P, sel, fp, flags.as<int>());
        GA_HIP(exclusive_scan_int(st, flags.as<int>(), (unsigned int)m + 1, tiles.as<int>(), pos.as<int>()));
        pp_compact_time_kernel<<<grid_for(m), 256, 0, st>>>(m, flags.as<int>(), pos.as<int>(), T, sb.ka.as<u64>(), sb.va.as<u32>());
        GA_HIP(hipGetLastError());
        GA_HIP(read_back_sync(ctx, st, h_counters, pos.as<int>() + m, sizeof(int)));
        f = h_counters[0];
        u64* ks = nullptr;
        GA_HIP(radix_sort_pairs(st, f, 64, sb.ka.as<u64>(), sb.va.as<u32>(), sb.kb.as<u64>(), sb.vb.as<u32>(), false, sb.hist.as<int>(), &ks, &order));
      }
      if (!done) GA_TRY(alloc_frame_cloud(ctx, f, has_int, &result.c));
      glim_amd_cloud* c = result.c;
      if (!done && f > 0) {
        pp_gather_out_kernel<<<grid_for(f), 256, 0, st>>>(f, order, P, T, I, prm->global_shutter, c->pts, c->pts64, c->times, c->intensities);
        GA_HIP(hipGetLastError());
        // no outlier removal behind this: the host copy of the time stamps rides on this scope's synchronise (pinned staging: a
        // device-to-host copy into pageable memory is staged by the runtime and costs a second round trip)
        if (!prm->enable_outlier_removal && pinned_malloc_impl(&times_stage.p, (size_t)f * sizeof(double)) == hipSuccess)
          GA_HIP(hipMemcpyAsync(times_stage.p, c->times, (size_t)f * sizeof(double), hipMemcpyDeviceToHost, st));
        else
          (void)hipGetLastError();
      }
      if (!done) GA_HIP(hipStreamSynchronize(st));  // scratch of this scope is released below
      in_flight.dismiss();                          // (the fast path synchronised with its counter read-back)
      if (times_stage.p) c->h_times.assign(times_stage.as<double>(), times_stage.as<double>() + f);
    }
  }

  // ---- statistical outlier removal (:162-164): kNN on the filtered cloud, mean neighbour distance, global threshold ----
  if (prm->enable_outlier_removal && result.c->n > 0) {
    glim_amd_cloud* c = result.c;
    const int f = (int)c->n, k = prm->outlier_removal_k;
    GA_TRY(glim_amd_cloud_find_neighbors(c, k, nullptr));
    std::lock_guard<std::mutex> lock(ctx->mu);
    GA_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream();
    const int blocks = grid_for(f);
    DeviceTemp d, partial, thresh, flags, pos, tiles, idx;
    GA_HIP(pool_malloc(&d.p, (size_t)f * sizeof(double)));
    GA_HIP(pool_malloc(&partial.p, (size_t)blocks * 2 * sizeof(double)));
    GA_HIP(pool_malloc(&thresh.p, sizeof(double)));
    GA_HIP(pool_malloc(&flags.p, (size_t)(f + 1) * sizeof(int)));
    GA_HIP(pool_malloc(&pos.p, (size_t)(f + 1) * sizeof(int)));
    GA_HIP(pool_malloc(&tiles.p, scan_scratch_ints((unsigned int)f + 1) * sizeof(int)));
    GA_HIP(pool_malloc(&idx.p, (size_t)f * sizeof(u32)));
    pp_mean_dist_kernel<<<blocks, 256, 0, st>>>(f, c->pts64, c->neighbors, k, d.as<double>(), partial.as<double>());
    pp_outlier_thresh_kernel<<<1, 1, 0, st>>>(f, blocks, partial.as<double>(), prm->outlier_std_mul_factor, thresh.as<double>());
    pp_inlier_flag_kernel<<<grid_for(f + 1), 256, 0, st>>>(f, d.as<double>(), thresh.as<double>(), flags.as<int>());
    GA_HIP(exclusive_scan_int(st, flags.as<int>(), (unsigned int)f + 1, tiles.as<int>(), pos.as<int>()));
    pp_compact_index_kernel<<<grid_for(f), 256, 0, st>>>(f, flags.as<int>(), pos.as<int>(), idx.as<u32>());
    GA_HIP(hipGetLastError());
    int kept = 0;
    GA_HIP(read_back_sync(ctx, st, &kept, pos.as<int>() + f, sizeof(int)));
    CloudGuard filtered;
    GA_TRY(alloc_frame_cloud(ctx, kept, has_int, &filtered.c));
    if (kept > 0) {
      pp_gather_out_kernel<<<grid_for(kept), 256, 0, st>>>(kept, idx.as<u32>(), c->pts64, c->times, c->intensities, 0, filtered.c->pts, filtered.c->pts64,
                                                          filtered.c->times, filtered.c->intensities);
      GA_HIP(hipGetLastError());
    }
    GA_HIP(hipStreamSynchronize(st));
    std::swap(result.c, filtered.c);  // the unfiltered cloud is destroyed with `filtered`
  }

  // ---- host copy of the time stamps (deskewing builds its time table from them) + kNN for the covariances (:183-184) ----
  {
    glim_amd_cloud* c = result.c;
    if (c->n > 0 && c->h_times.size() != (size_t)c->n) {
      std::lock_guard<std::mutex> lock(ctx->mu);
      GA_HIP(hipSetDevice(ctx->device));
      c->h_times.resize((size_t)c->n);
      GA_HIP(hipMemcpyAsync(c->h_times.data(), c->times, (size_t)c->n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream()));
      GA_HIP(hipStreamSynchronize(ctx->stream()));
    }
    if (prm->k_correspondences > 0) GA_TRY(glim_amd_cloud_find_neighbors(c, prm->k_correspondences, nullptr));
  }
  *out = result.release();
  return GLIM_AMD_OK;
}

int glim_amd_merge_frames(glim_amd_ctx* ctx, int32_t num_frames, const double* poses12, const double* const* points4, const double* const* covs16,
                          const int64_t* sizes, double resolution, int32_t target_num_points, int32_t block_size, uint64_t seed, glim_amd_cloud** out) {
  if (!ctx || !out || num_frames < 0 || !(resolution > 0.0)) return GLIM_AMD_ERR_INVALID;
  if (num_frames > 0 && (!poses12 || !points4 || !covs16 || !sizes)) return GLIM_AMD_ERR_INVALID;
  *out = nullptr;
  int64_t total64 = 0, max_frame = 0;
  for (int f = 0; f < num_frames; f++) {
    if (sizes[f] < 0 || (sizes[f] > 0 && (!points4[f] || !covs16[f]))) return GLIM_AMD_ERR_INVALID;
    total64 += sizes[f];
    max_frame = std::max(max_frame, sizes[f]);
  }
  if (total64 > (int64_t)(1 << 28)) return GLIM_AMD_ERR_INVALID;
  const int n = (int)total64;
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream();
  CloudGuard result;

  // ---- upload frame by frame, transform into the submap origin, concatenate ----
  DeviceTemp P, C6, stage_p, stage_c;
  GA_HIP(pool_malloc(&P.p, (size_t)std::max(n, 1) * sizeof(double4)));
  GA_HIP(pool_malloc(&C6.p, (size_t)std::max(n, 1) * 6 * sizeof(double)));
  GA_HIP(pool_malloc(&stage_p.p, (size_t)std::max<int64_t>(max_frame, 1) * sizeof(double4)));
  GA_HIP(pool_malloc(&stage_c.p, (size_t)std::max<int64_t>(max_frame, 1) * 16 * sizeof(double)));
  int64_t at = 0;
  for (int f = 0; f < num_frames; f++) {
    const int nf = (int)sizes[f];
    if (nf == 0) continue;
    GA_HIP(hipMemcpyAsync(stage_p.p, points4[f], (size_t)nf * sizeof(double4), hipMemcpyHostToDevice, st));
    GA_HIP(hipMemcpyAsync(stage_c.p, covs16[f], (size_t)nf * 16 * sizeof(double), hipMemcpyHostToDevice, st));
    Pose12 T;
    memcpy(T.m, poses12 + 12 * (size_t)f, sizeof(T.m));
    mg_transform_kernel<<<grid_for(nf), 256, 0, st>>>(nf, stage_p.as<double4>(), stage_c.as<double>(), T, P.as<double4>() + at, C6.as<double>() + 6 * at);
    GA_HIP(hipGetLastError());
    at += nf;
  }

  // ---- voxelgrid_sampling of the concatenation: keys, stable sort, runs, sequential means of points and covariances ----
  int m = 0;
  DeviceTemp avgP, avgC;
  if (n > 0) {
    DeviceTemp d_vkey, d_ckey, d_bb, d_nvalid, heads, seg, tiles, seg_start, sP, sC;
    SortBuffers sb;
    GA_TRY(sb.alloc(n));
    GA_HIP(pool_malloc(&d_vkey.p, (size_t)n * sizeof(u64)));
    GA_HIP(pool_malloc(&d_bb.p, 6 * sizeof(int)));
    GA_HIP(pool_malloc(&d_nvalid.p, sizeof(int)));
    int h_bb[6];
    init_bbox_kernel<<<1, 64, 0, st>>>(d_bb.as<int>());
    pp_key_kernel<<<reduce_grid_for(n), 256, 0, st>>>(n, P.as<double4>(), 1.0 / resolution, d_vkey.as<u64>(), d_bb.as<int>());
    GA_HIP(hipGetLastError());
    GA_HIP(read_back_sync(ctx, st, h_bb, d_bb.p, sizeof(h_bb)));
    int bx = 0, by = 0, bz = 0;
    if (h_bb[0] <= h_bb[3]) {
      bx = bits_for(h_bb[3] - h_bb[0]);
      by = bits_for(h_bb[4] - h_bb[1]);
      bz = bits_for(h_bb[5] - h_bb[2]);
    } else {
      h_bb[0] = h_bb[1] = h_bb[2] = 0;
    }
    const int vbits = bx + by + bz;
    pp_compact_key_kernel<<<grid_for(n), 256, 0, st>>>(n, d_vkey.as<u64>(), h_bb[0], h_bb[1], h_bb[2], bx, by, vbits, sb.ka.as<u64>());
    u64* ks = nullptr;
    u32* vs = nullptr;
    GA_HIP(radix_sort_pairs(st, n, vbits + 1, sb.ka.as<u64>(), sb.va.as<u32>(), sb.kb.as<u64>(), sb.vb.as<u32>(), true, sb.hist.as<int>(), &ks, &vs));
    GA_HIP(pool_malloc(&heads.p, (size_t)(n + 1) * sizeof(int)));
    GA_HIP(pool_malloc(&seg.p, (size_t)(n + 1) * sizeof(int)));
    GA_HIP(pool_malloc(&tiles.p, scan_scratch_ints((unsigned int)n + 1) * sizeof(int)));
    GA_HIP(hipMemsetAsync(d_nvalid.p, 0, sizeof(int), st));
    pp_head_flag_kernel<<<grid_for(n + 1), 256, 0, st>>>(n, ks, 1ull << vbits, block_size, heads.as<int>(), d_nvalid.as<int>());
    GA_HIP(exclusive_scan_int(st, heads.as<int>(), (unsigned int)n + 1, tiles.as<int>(), seg.as<int>()));
    GA_HIP(read_back_sync(ctx, st, &m, seg.as<int>() + n, sizeof(int)));
    GA_HIP(pool_malloc(&avgP.p, (size_t)std::max(m, 1) * sizeof(double4)));
    GA_HIP(pool_malloc(&avgC.p, (size_t)std::max(m, 1) * 6 * sizeof(double)));
    if (m > 0) {
      GA_HIP(pool_malloc(&seg_start.p, (size_t)(m + 1) * sizeof(int)));
      GA_HIP(pool_malloc(&sP.p, (size_t)n * sizeof(double4)));
      GA_HIP(pool_malloc(&sC.p, (size_t)n * 6 * sizeof(double)));
      pp_seg_start_kernel<<<grid_for(n + 1), 256, 0, st>>>(n, heads.as<int>(), seg.as<int>(), d_nvalid.as<int>(), seg_start.as<int>());
      mg_gather_sorted_kernel<<<grid_for(n), 256, 0, st>>>(n, vs, P.as<double4>(), C6.as<double>(), sP.as<double4>(), sC.as<double>());
      mg_segment_mean_kernel<<<grid_for(m), 256, 0, st>>>(m, seg_start.as<int>(), sP.as<double4>(), sC.as<double>(), avgP.as<double4>(), avgC.as<double>());
      GA_HIP(hipGetLastError());
    }
    GA_HIP(hipStreamSynchronize(st));  // scratch of this scope is released here
  }

  // ---- (target size) uniform random sample in the original order, then the output cloud ----
  int out_n = m;
  DeviceTemp idx;
  if (target_num_points > 0 && m > target_num_points) {
    const double rate = (double)target_num_points / (double)m;
    out_n = (int)((double)m * rate);
    SortBuffers sb;
    GA_TRY(sb.alloc(m));
    DeviceTemp flags, pos, tiles;
    GA_HIP(pool_malloc(&flags.p, (size_t)(m + 1) * sizeof(int)));
    GA_HIP(pool_malloc(&pos.p, (size_t)(m + 1) * sizeof(int)));
    GA_HIP(pool_malloc(&tiles.p, scan_scratch_ints((unsigned int)m + 1) * sizeof(int)));
    GA_HIP(pool_malloc(&idx.p, (size_t)std::max(out_n, 1) * sizeof(u32)));
    u64* ks = nullptr;
    u32* vs = nullptr;
    pp_hash_key_kernel<<<grid_for(m), 256, 0, st>>>(m, seed, sb.ka.as<u64>());
    GA_HIP(radix_sort_pairs(st, m, 32, sb.ka.as<u64>(), sb.va.as<u32>(), sb.kb.as<u64>(), sb.vb.as<u32>(), true, sb.hist.as<int>(), &ks, &vs));
    mg_mark_kernel<<<grid_for(m + 1), 256, 0, st>>>(m, out_n, vs, flags.as<int>());
    GA_HIP(exclusive_scan_int(st, flags.as<int>(), (unsigned int)m + 1, tiles.as<int>(), pos.as<int>()));
    pp_compact_index_kernel<<<grid_for(m), 256, 0, st>>>(m, flags.as<int>(), pos.as<int>(), idx.as<u32>());
    GA_HIP(hipGetLastError());
    GA_HIP(hipStreamSynchronize(st));
  }
  {
    glim_amd_cloud* c = new glim_amd_cloud();
    result.c = c;
    c->ctx = ctx;
    c->n = out_n;
    const size_t nn = (size_t)std::max(out_n, 1);
    hipError_t e = pool_malloc(&c->pts, nn * sizeof(float4));
    if (e == hipSuccess) e = pool_malloc(&c->covA, nn * sizeof(float4));
    if (e == hipSuccess) e = pool_malloc(&c->covB, nn * sizeof(float2));
    if (e == hipSuccess) e = pool_malloc(&c->pts64, nn * sizeof(double4));
    if (e == hipSuccess) e = pool_malloc(&c->cov64, nn * 6 * sizeof(double));
    if (e != hipSuccess) {
      set_hip_error(e, "pool_malloc(merged cloud)");
      return e == hipErrorOutOfMemory ? GLIM_AMD_ERR_NOMEM : GLIM_AMD_ERR_HIP;
    }
    c->has_covs = true;
    if (out_n > 0) {
      mg_output_kernel<<<grid_for(out_n), 256, 0, st>>>(out_n, idx.p ? idx.as<u32>() : nullptr, avgP.as<double4>(), avgC.as<double>(), c->pts64, c->cov64,
                                                        c->pts, c->covA, c->covB);
      GA_HIP(hipGetLastError());
    }
    GA_HIP(hipStreamSynchronize(st));
  }
  *out = result.release();
  return GLIM_AMD_OK;
}

int glim_amd_cloud_download_merged(const glim_amd_cloud* c, double* points4, double* covs16) {
  if (!c) return GLIM_AMD_ERR_INVALID;
  if ((points4 && !c->pts64) || (covs16 && !c->cov64)) return GLIM_AMD_ERR_STATE;
  if (c->n == 0) return GLIM_AMD_OK;
  glim_amd_ctx* ctx = c->ctx;
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream();
  const size_t n = (size_t)c->n;
  std::vector<double> c6;
  if (points4) GA_HIP(hipMemcpyAsync(points4, c->pts64, n * sizeof(double4), hipMemcpyDeviceToHost, s));
  if (covs16) {
    c6.resize(n * 6);
    GA_HIP(hipMemcpyAsync(c6.data(), c->cov64, n * 6 * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  GA_HIP(hipStreamSynchronize(s));
  if (covs16) {
    for (size_t i = 0; i < n; i++) {  // symmetric 3x3 -> column-major Matrix4d with a zero last row / column
      const double* a = &c6[6 * i];
      double* o = covs16 + 16 * i;
      memset(o, 0, 16 * sizeof(double));
      o[0] = a[0]; o[1] = o[4] = a[1]; o[2] = o[8] = a[2]; o[5] = a[3]; o[6] = o[9] = a[4]; o[10] = a[5];
    }
  }
  return GLIM_AMD_OK;
}

int glim_amd_cloud_download_frame(const glim_amd_cloud* c, double* points4, double* times, double* intensities, int32_t* neighbors) {
  if (!c) return GLIM_AMD_ERR_INVALID;
  if ((points4 && !c->pts64) || (times && !c->times) || (intensities && !c->intensities) || (neighbors && !c->neighbors)) return GLIM_AMD_ERR_STATE;
  if (c->n == 0) return GLIM_AMD_OK;
  glim_amd_ctx* ctx = c->ctx;
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream();
  const size_t n = (size_t)c->n;
  if (points4) GA_HIP(hipMemcpyAsync(points4, c->pts64, n * sizeof(double4), hipMemcpyDeviceToHost, s));
  if (times) GA_HIP(hipMemcpyAsync(times, c->times, n * sizeof(double), hipMemcpyDeviceToHost, s));
  if (intensities) GA_HIP(hipMemcpyAsync(intensities, c->intensities, n * sizeof(double), hipMemcpyDeviceToHost, s));
  if (neighbors) GA_HIP(hipMemcpyAsync(neighbors, c->neighbors, n * (size_t)c->k * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  GA_HIP(hipStreamSynchronize(s));
  return GLIM_AMD_OK;
}

}  // extern "C"
