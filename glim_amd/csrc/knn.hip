// knn.hip -- kernel group K2: exact k-nearest-neighbour search on the device.
// Replaces CloudPreprocessor::find_neighbors (src/glim/preprocess/cloud_preprocessor.cpp:190-221, a nanoflann kd-tree):
// for every point the k nearest points among ALL points including itself, ascending
// squared distance; ties are ordered by ascending index (the oracle's rule; the reference leaves exact ties to the kd-tree).
//
// Distances are evaluated in FP64 as (dx*dx + dy*dy) + dz*dz on the FP32-representable inputs -- the same expression, in
// the same order and without fma contraction, as oracle/vgicp_oracle.c:sqdist3 -- so neighbour lists are bit-identical.
//
// Three implementations, same result:
//   * curve-ordered chunks (default, knn_curve): Hilbert sort, 64-point chunks with boxes, one wavefront per chunk streaming the
//     candidate chunks through LDS (knn_chunks.hip, knn_pairs.hip); adapts to the local density by construction.  0.36 ms for a
//     131 072-point LiDAR scan, 0.60 ms for a 307 104-point depth frame on MI355X.
//   * hashed uniform grid (diag knn_path=grid, knn_grid): counting sort into cells, ring walk per query with exactness bound and coarser
//     retry levels: 0.85 / 1.11 ms at the two sizes above, up to 5 ms on small clouds of uneven density; the cross-check of the chunk path.
//   * exhaustive: LDS-tiled scan of every point (tiny clouds, and the grid path's last resort).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "knn_common.hpp"
#include "scan.hpp"

using namespace glim_amd;

// measurement hook (glim_amd_cloud_profile_neighbors): when set, the calling thread's next kNN records these events around its query-group kernel
static thread_local hipEvent_t g_knn_probe_events[2] = {nullptr, nullptr};


namespace {

constexpr int TILE = 1024;
constexpr int MAX_RING = 6;

// ---- exhaustive scan.  `queries` (optional): list of point indices to answer; otherwise every point. ----
template <int K>
__global__ __launch_bounds__(256) void knn_bruteforce_kernel(int n, const float4* __restrict__ pts, int k, int32_t* __restrict__ out,
                                                             const int* __restrict__ queries, int num_queries) {
  __shared__ double s_x[TILE], s_y[TILE], s_z[TILE];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = t < num_queries;
  const int i = live ? (queries ? queries[t] : t) : 0;
  double qx = 0, qy = 0, qz = 0;
  if (live) {
    const float4 p = pts[i];
    qx = p.x; qy = p.y; qz = p.z;
  }
  TopK<K> best;
  best.init(i);
  for (int t0 = 0; t0 < n; t0 += TILE) {
    const int tn = min(TILE, n - t0);
    __syncthreads();
    for (int j = threadIdx.x; j < tn; j += blockDim.x) {
      const float4 p = pts[t0 + j];
      s_x[j] = p.x; s_y[j] = p.y; s_z[j] = p.z;
    }
    __syncthreads();
    if (live)
      for (int j = 0; j < tn; j++) best.push(sqdist(qx, qy, qz, s_x[j], s_y[j], s_z[j]), t0 + j);
  }
  if (live) {
    const int valid = min(n, K);
#pragma unroll
    for (int j = 0; j < K; j++)
      if (j < k) out[(size_t)i * k + j] = (j < valid) ? best.idx[j] : 0;  // fewer than k points: the reference's result vector stays 0 there (:193, :200)
  }
}

// ---- grid build ----
// stats: [0] occupied cells, [1] out-of-range points, [2] unresolved queries
__global__ __launch_bounds__(256) void grid_insert_kernel(int n, const float4* __restrict__ pts, double inv_h, unsigned long long* __restrict__ keys,
                                                          unsigned int mask, int* __restrict__ counts, int* __restrict__ slot_of, int* __restrict__ stats) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  const unsigned long long key = voxel_key((double)p.x, (double)p.y, (double)p.z, inv_h);
  if (key == EMPTY_KEY) {
    atomicAdd(&stats[1], 1);
    slot_of[i] = -1;
    return;
  }
  unsigned int s = hash_key(key) & mask;
  for (;;) {
    const unsigned long long prev = atomicCAS(&keys[s], EMPTY_KEY, key);
    if (prev == EMPTY_KEY) atomicAdd(&stats[0], 1);
    if (prev == EMPTY_KEY || prev == key) break;
    s = (s + 1) & mask;
  }
  atomicAdd(&counts[s], 1);
  slot_of[i] = (int)s;
}

__global__ __launch_bounds__(256) void grid_scatter_kernel(int n, const float4* __restrict__ pts, const int* __restrict__ slot_of,
                                                           const int* __restrict__ starts, int* __restrict__ cursor, float4* __restrict__ sorted) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = slot_of[i];
  if (s < 0) return;
  const int pos = starts[s] + atomicAdd(&cursor[s], 1);
  const float4 p = pts[i];
  sorted[pos] = make_float4(p.x, p.y, p.z, __int_as_float(i));
}

// ---- grid query: one lane per point, in cell order ----
// What bounds this kernel on MI355X (per-query counters of a diagnostic build): a wavefront lasts as long as its slowest lane, and a
// scan is far from uniformly dense -- the mean query of a 131 072-point LiDAR scan sees 278 candidates, the cells next to the sensor
// 4000 (p99), so the few wavefronts that hold those queries set the kernel time (0.8 of the 0.95 ms).  Tried and measured slower:
// G = 2 / 4 / 8 lanes per query with a shuffle merge of their top-K lists (1.2 / 1.4-1.8 / 2.4 ms: the walk is not latency-bound,
// the extra lanes only add merge work); queries in arrival order instead of cell order (0.94 / 1.44 vs 0.97 / 1.18 ms on the LiDAR
// scan / depth frame); a directly addressed grid with contiguous row runs instead of hash probes (1.42 ms); starting on a 4-8x
// finer grid with 2 rings per level (2.1-2.8 ms: every extra level costs a grid rebuild and another pass); parking accepted
// candidates in 4 staging slots so that the K-step insertion runs once per 4 acceptances (0.92 vs 0.86 ms).  SQ counters: 34 000
// VALU + 22 000 SALU instructions and 810 vector loads per wavefront, i.e. ~250 cell steps of ~130 instructions each -- the cost
// is the per-cell bookkeeping of the union of the lanes' walks.  What helped: four candidate loads in flight per lane (1.3 ->
// 0.95 ms) and skipping cells whose box is farther than the current k-th best (0.95 -> 0.86 ms, candidates 278 -> 122 per query).
template <int K>
__global__ __launch_bounds__(256) void knn_grid_kernel(int n, const float4* __restrict__ sorted, double h, double inv_h,
                                                       const unsigned long long* __restrict__ keys, unsigned int mask, const int* __restrict__ starts,
                                                       const int* __restrict__ counts, int k, int32_t* __restrict__ out, int* __restrict__ unresolved,
                                                       int* __restrict__ stats, const float4* __restrict__ pts, const int* __restrict__ queries,
                                                       int num_queries, int max_ring) {
  // first pass: every point in cell order (queries == nullptr); retry passes on a coarser grid: the listed points only
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= (queries ? num_queries : n)) return;
  float4 q4;
  if (queries) {
    const int qi = queries[s];
    const float4 p = pts[qi];
    q4 = make_float4(p.x, p.y, p.z, __int_as_float(qi));
  } else {
    q4 = sorted[s];
  }
  const int self = __float_as_int(q4.w);
  const double qx = q4.x, qy = q4.y, qz = q4.z;
  const double tx = qx * inv_h, ty = qy * inv_h, tz = qz * inv_h;
  const int cx = fast_floor_d(tx), cy = fast_floor_d(ty), cz = fast_floor_d(tz);
  // distance from the query to the nearest wall of its own cell (>= 0; conservative under rounding)
  double margin = fmin(fmin(tx - (double)cx, (double)(cx + 1) - tx), fmin(fmin(ty - (double)cy, (double)(cy + 1) - ty), fmin(tz - (double)cz, (double)(cz + 1) - tz)));
  margin = fmax(0.0, margin * h * 0.999999);
  TopK<K> best;
  best.init(self);
  bool done = false;
  for (int ring = 0; ring <= max_ring; ring++) {
    if (ring >= 1) {
      const double reach = (double)(ring - 1) * h * 0.999999 + margin;  // every unscanned point is at least this far
      if (best.d[K - 1] < reach * reach) {                               // strict: an unscanned tie could carry a smaller index
        done = true;
        break;
      }
    }
    for (int dz = -ring; dz <= ring; dz++)
      for (int dy = -ring; dy <= ring; dy++) {
        const bool shell_yz = (abs(dz) == ring) || (abs(dy) == ring);
        for (int dx = -ring; dx <= ring; dx += (shell_yz || ring == 0) ? 1 : 2 * ring) {
          // prune: no point of this cell can be closer than the gap between the query and the cell's box (shrunk a little so that
          // rounding in the cell assignment cannot make it optimistic); strictly greater, so that exact ties are still seen
          if (ring >= 1) {
            const double gx = dx > 0 ? (double)(cx + dx) - tx : (dx < 0 ? tx - (double)(cx + dx + 1) : 0.0);
            const double gy = dy > 0 ? (double)(cy + dy) - ty : (dy < 0 ? ty - (double)(cy + dy + 1) : 0.0);
            const double gz = dz > 0 ? (double)(cz + dz) - tz : (dz < 0 ? tz - (double)(cz + dz + 1) : 0.0);
            const double gap = h * 0.999999;
            const double ex = fmax(0.0, gx) * gap, ey = fmax(0.0, gy) * gap, ez = fmax(0.0, gz) * gap;
            if (ex * ex + ey * ey + ez * ez > best.d[K - 1]) continue;
          }
          const unsigned long long key = pack_key(cx + dx, cy + dy, cz + dz);
          if (key == EMPTY_KEY) continue;
          unsigned int sl = hash_key(key) & mask;
          int found = -1;
          for (;;) {
            const unsigned long long kk = keys[sl];
            if (kk == key) {
              found = (int)sl;
              break;
            }
            if (kk == EMPTY_KEY) break;
            sl = (sl + 1) & mask;
          }
          if (found < 0) continue;
          const int b = starts[found], e = b + counts[found];
          // four candidates per trip, loads issued back to back
          int j = b;
          for (; j + 4 <= e; j += 4) {
            const float4 c0 = sorted[j], c1 = sorted[j + 1], c2 = sorted[j + 2], c3 = sorted[j + 3];
            best.push(sqdist(qx, qy, qz, (double)c0.x, (double)c0.y, (double)c0.z), __float_as_int(c0.w));
            best.push(sqdist(qx, qy, qz, (double)c1.x, (double)c1.y, (double)c1.z), __float_as_int(c1.w));
            best.push(sqdist(qx, qy, qz, (double)c2.x, (double)c2.y, (double)c2.z), __float_as_int(c2.w));
            best.push(sqdist(qx, qy, qz, (double)c3.x, (double)c3.y, (double)c3.z), __float_as_int(c3.w));
          }
          for (; j < e; j++) {
            const float4 c = sorted[j];
            best.push(sqdist(qx, qy, qz, (double)c.x, (double)c.y, (double)c.z), __float_as_int(c.w));
          }
        }
      }
  }
  if (!done) {
    // the (2 max_ring + 1)^3 cube was scanned: accept only if that already proves the result, else hand over to the exhaustive kernel
    const double reach = (double)max_ring * h * 0.999999 + margin;
    done = best.d[K - 1] < reach * reach;
  }
  if (!done) {
    unresolved[atomicAdd(&stats[2], 1)] = self;
    return;
  }
#pragma unroll
  for (int j = 0; j < K; j++)
    if (j < k) out[(size_t)self * k + j] = best.idx[j];  // fewer than k points in the whole cloud never reaches here (n > k enforced)
}

// bounding box by ordered-int atomics: bb[0..2] = min, bb[3..5] = max (as order-preserving ints)
__device__ __forceinline__ int ordered(float f) {
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__global__ __launch_bounds__(256) void bbox_kernel(int n, const float4* __restrict__ pts, int* __restrict__ bb) {
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 p = pts[i];
    const int v[3] = {ordered(p.x), ordered(p.y), ordered(p.z)};
    for (int a = 0; a < 3; a++) {
      lo[a] = min(lo[a], v[a]);
      hi[a] = max(hi[a], v[a]);
    }
  }
  // one set of atomics per BLOCK (same-address atomics serialise in L2)
  __shared__ int s_tmp[16];
  for (int a = 0; a < 3; a++) {
    lo[a] = block_reduce_i<0>(lo[a], s_tmp);
    hi[a] = block_reduce_i<1>(hi[a], s_tmp);
  }
  if (threadIdx.x == 0)
    for (int a = 0; a < 3; a++) {
      atomicMin(&bb[a], lo[a]);
      atomicMax(&bb[3 + a], hi[a]);
    }
}
__host__ __device__ inline float unordered(int i) {
  const int j = i >= 0 ? i : i ^ 0x7fffffff;
  float f;
  memcpy(&f, &j, sizeof(f));
  return f;
}

template <int K>
void launch_brute(hipStream_t st, int n, const float4* pts, int k, int32_t* out, const int* queries, int nq) {
  if (nq > 0) knn_bruteforce_kernel<K><<<(nq + 255) / 256, 256, 0, st>>>(n, pts, k, out, queries, nq);
}
template <int K>
void launch_grid(hipStream_t st, int n, const float4* sorted, double h, const unsigned long long* keys, unsigned int mask, const int* starts,
                 const int* counts, int k, int32_t* out, int* unresolved, int* stats, const float4* pts, const int* queries, int nq, int max_ring) {
  const int work = queries ? nq : n;
  if (work > 0)
    knn_grid_kernel<K><<<(work + 255) / 256, 256, 0, st>>>(n, sorted, h, 1.0 / h, keys, mask, starts, counts, k, out, unresolved, stats, pts, queries, nq,
                                                           max_ring);
}


unsigned int next_pow2(unsigned long long v) {
  unsigned long long p = 1;
  while (p < v) p <<= 1;
  return (unsigned int)p;
}

// counting sort of all points into the hashed grid of cell edge h (keys/counts/starts/sorted are rebuilt)
struct GridBuffers {
  DeviceTemp keys, counts, starts, cursor, slot_of, sorted, stats, tile_sums;
  unsigned int T = 0;
};

int build_grid(glim_amd_ctx* ctx, hipStream_t st, int n, const float4* pts, double h, GridBuffers& g, int* h_stats) {
  GA_HIP(hipMemsetAsync(g.keys.p, 0xff, (size_t)g.T * sizeof(unsigned long long), st));
  GA_HIP(hipMemsetAsync(g.counts.p, 0, (size_t)g.T * sizeof(int), st));
  GA_HIP(hipMemsetAsync(g.stats.p, 0, 4 * sizeof(int), st));
  grid_insert_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, pts, 1.0 / h, (unsigned long long*)g.keys.p, g.T - 1, (int*)g.counts.p, (int*)g.slot_of.p,
                                                      (int*)g.stats.p);
  GA_HIP(read_back_sync(ctx, st, h_stats, g.stats.p, 4 * sizeof(int)));
  return GLIM_AMD_OK;
}

int sort_into_grid(hipStream_t st, int n, const float4* pts, GridBuffers& g) {
  GA_HIP(hipMemsetAsync(g.cursor.p, 0, (size_t)g.T * sizeof(int), st));
  GA_HIP(exclusive_scan_int(st, (const int*)g.counts.p, g.T, (int*)g.tile_sums.p, (int*)g.starts.p));
  grid_scatter_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, pts, (const int*)g.slot_of.p, (const int*)g.starts.p, (int*)g.cursor.p, (float4*)g.sorted.p);
  GA_HIP(hipGetLastError());
  return GLIM_AMD_OK;
}

int knn_grid(glim_amd_ctx* ctx, hipStream_t st, int n, const float4* pts, int k, int32_t* out) {
  // ---- cell edge from the data: a surface-like cloud of n points in its bounding box, ~3 points per occupied cell ----
  DeviceTemp bb, unresolved_a, unresolved_b;
  GridBuffers g;
  SyncOnExit in_flight(st);  // (cross-check path: the extra wait on the exits that have synchronised already is not worth dismissing)
  GA_HIP(pool_malloc(&bb.p, 6 * sizeof(int)));
  init_bbox_kernel<<<1, 64, 0, st>>>(bb.as<int>());
  bbox_kernel<<<std::max(1, std::min((n + 2047) / 2048, 128)), 256, 0, st>>>(n, pts, (int*)bb.p);
  int h_bb[6];
  GA_HIP(read_back_sync(ctx, st, h_bb, bb.p, sizeof(h_bb)));
  double ext[3];
  for (int a = 0; a < 3; a++) ext[a] = std::max(1e-6, (double)unordered(h_bb[3 + a]) - (double)unordered(h_bb[a]));
  const double area = ext[0] * ext[1] + ext[1] * ext[2] + ext[0] * ext[2];
  const double ppc = 3.0;  // target points per occupied cell of the level-0 grid
  double h = std::sqrt(ppc * 2.0 * area / (double)n);
  const double max_abs = std::max({std::fabs((double)unordered(h_bb[0])), std::fabs((double)unordered(h_bb[1])), std::fabs((double)unordered(h_bb[2])),
                                   std::fabs((double)unordered(h_bb[3])), std::fabs((double)unordered(h_bb[4])), std::fabs((double)unordered(h_bb[5]))});
  const double h_min = max_abs / 1.0e6 + 1e-9;  // keep cell coordinates inside the 21-bit key range
  h = std::max(h, h_min);
  const double diag = std::sqrt(ext[0] * ext[0] + ext[1] * ext[1] + ext[2] * ext[2]);

  g.T = next_pow2((unsigned long long)n * 2);
  GA_HIP(pool_malloc(&g.keys.p, (size_t)g.T * sizeof(unsigned long long)));
  GA_HIP(pool_malloc(&g.counts.p, (size_t)g.T * sizeof(int)));
  GA_HIP(pool_malloc(&g.starts.p, (size_t)g.T * sizeof(int)));
  GA_HIP(pool_malloc(&g.cursor.p, (size_t)g.T * sizeof(int)));
  GA_HIP(pool_malloc(&g.slot_of.p, (size_t)n * sizeof(int)));
  GA_HIP(pool_malloc(&g.sorted.p, (size_t)n * sizeof(float4)));
  GA_HIP(pool_malloc(&g.stats.p, 4 * sizeof(int)));
  GA_HIP(pool_malloc(&g.tile_sums.p, scan_scratch_ints(g.T) * sizeof(int)));
  GA_HIP(pool_malloc(&unresolved_a.p, (size_t)n * sizeof(int)));
  GA_HIP(pool_malloc(&unresolved_b.p, (size_t)n * sizeof(int)));

  // Schedule: level 0 = the grid sized for the mean density (`fine` = 1) walked up to MAX_RING rings; the queries it cannot prove
  // (sparse regions) move to a 4x coarser grid, and so on; once the scanned cube would cover the bounding box the remainder is
  // finished exhaustively.  Measured alternatives on MI355X (131 072-pt LiDAR scan / 307 104-pt depth frame, ms): starting 8x finer
  // with 2 rings per level 2.8 / 2.2, 4x finer 2.1 / 2.0, this schedule 0.97 / 1.22 -- every extra level costs a grid rebuild and a
  // latency-bound pass (~0.5 ms), which outweighs the shorter candidate lists of the dense cells next to the sensor.
  const int level_ring = MAX_RING;
  int h_stats[4] = {0, 0, 0, 0};
  int* todo = (int*)unresolved_a.p;
  int* next = (int*)unresolved_b.p;
  int num_todo = 0;
  bool first = true;
  for (int level = 0; level < 16; level++) {
    GA_TRY(build_grid(ctx, st, n, pts, h, g, h_stats));
    if (h_stats[1] != 0) {  // a coordinate fell outside the key range at this cell size: coarsen
      if (h > 1e9) return GLIM_AMD_ERR_RANGE;
      h *= 4.0;
      continue;
    }
    GA_TRY(sort_into_grid(st, n, pts, g));
    GA_HIP(hipMemsetAsync((int*)g.stats.p + 2, 0, sizeof(int), st));
    const bool last = (double)level_ring * h * 4.0 > diag;  // the next level could not do better than the exhaustive kernel
    DISPATCH_K(launch_grid, st, n, (const float4*)g.sorted.p, h, (const unsigned long long*)g.keys.p, g.T - 1, (const int*)g.starts.p,
               (const int*)g.counts.p, k, out, next, (int*)g.stats.p, pts, first ? (const int*)nullptr : todo, num_todo, last ? MAX_RING : level_ring);
    GA_HIP(hipGetLastError());
    GA_HIP(read_back_sync(ctx, st, h_stats, g.stats.p, 4 * sizeof(int)));
    first = false;
    num_todo = h_stats[2];
    if (num_todo == 0) return GLIM_AMD_OK;
    std::swap(todo, next);
    if ((double)MAX_RING * h > diag) break;  // the next cube would cover everything anyway: finish exhaustively
    h *= 4.0;
  }
  DISPATCH_K(launch_brute, st, n, pts, k, out, (const int*)todo, num_todo);
  GA_HIP(hipGetLastError());
  GA_HIP(hipStreamSynchronize(st));
  return GLIM_AMD_OK;
}

// =================================================================================================================
// Curve-ordered chunks (default): no grid, no rings.
//   1. the points are sorted along a Hilbert curve through the bounding box (stable radix sort, sort.hip);
//   2. every 64 consecutive points form a chunk with an axis-aligned box;
//   3. one wavefront answers the 64 queries of a chunk: it streams candidate chunks through LDS -- first its own chunk and its two
//      curve neighbours, which already contain most true neighbours, then every other chunk whose box can still hold a point
//      closer than some lane's current k-th best.  A chunk is skipped only when the gap between the query and the chunk's box is
//      strictly larger than that lane's k-th best distance, so nothing that could enter (or tie into) a top-k list is missed:
//      the result is exact for any point distribution, with the same (distance, index) order as the other two implementations.
// Work per query adapts to the local density by construction (a chunk is 64 points wherever they are), which is what the
// uniform grid above cannot do: its dense cells next to the sensor hold thousands of candidates and set the kernel time.
// Every lane of a wavefront scans the same LDS tile, so there is no divergent walking and all global loads are coalesced.
// =================================================================================================================
__device__ __forceinline__ unsigned long long spread3(unsigned int v) {  // 21 bits -> every third bit
  unsigned long long x = v & 0x1fffffu;
  x = (x | (x << 32)) & 0x1f00000000ffffull;
  x = (x | (x << 16)) & 0x1f0000ff0000ffull;
  x = (x | (x << 8)) & 0x100f00f00f00f00full;
  x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}

// stats[1] counts points with a non-finite coordinate (an error, as in the grid path).  The bounding box (bbox_kernel's ordered ints) is read
// HERE, not by the host: one synchronise per kNN call instead of two.  stats[2] is the guard word of the chunk kernels: 1 = the box is not
// finite, 2 = the cloud spans 1e18 m or more (the FP32 mask passes need finite FP32 squared distances: the exhaustive FP64 kernel answers).
__global__ __launch_bounds__(256) void curve_key_kernel(int n, const float4* __restrict__ pts, const int* __restrict__ bb, unsigned int qmax, int bits,
                                                        unsigned long long* __restrict__ keys, int* __restrict__ stats) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float lox = unordered(bb[0]), loy = unordered(bb[1]), loz = unordered(bb[2]);
  const float ext = fmaxf(fmaxf(unordered(bb[3]) - lox, unordered(bb[4]) - loy), unordered(bb[5]) - loz);
  const bool finite_box = ext >= 0.f && isfinite(ext) && isfinite(lox) && isfinite(loy) && isfinite(loz);
  if (i == 0 && !(finite_box && ext < 1e18f)) stats[2] = finite_box ? 2 : 1;
  if (i >= n) return;
  const float scale = (finite_box && ext > 0.f) ? (float)qmax / ext : 0.f;
  const float4 p = pts[i];
  if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) {
    atomicAdd(&stats[1], 1);
    keys[i] = ~0ull;
    return;
  }
  const unsigned int x = min(qmax, (unsigned int)fmaxf(0.f, (p.x - lox) * scale));
  const unsigned int y = min(qmax, (unsigned int)fmaxf(0.f, (p.y - loy) * scale));
  const unsigned int z = min(qmax, (unsigned int)fmaxf(0.f, (p.z - loz) * scale));
  // Hilbert index (Skilling's axes-to-transpose): consecutive points of a Hilbert curve are always in adjacent cells, so 64
  // consecutive points form a compact chunk.  A Morton curve jumps at octant boundaries: it gave chunks 28 m across whose lanes
  // wanted different candidate sets (one wavefront 985 us against a mean of 170 us); the order only affects speed, never results.
  unsigned int X[3] = {x, y, z};
  const unsigned int M = 1u << (bits - 1);
  for (unsigned int Q = M; Q > 1u; Q >>= 1) {
    const unsigned int P = Q - 1u;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      if (X[a] & Q) {
        X[0] ^= P;
      } else {
        const unsigned int t = (X[0] ^ X[a]) & P;
        X[0] ^= t;
        X[a] ^= t;
      }
    }
  }
  X[1] ^= X[0];
  X[2] ^= X[1];
  unsigned int t = 0u;
  for (unsigned int Q = M; Q > 1u; Q >>= 1)
    if (X[2] & Q) t ^= Q - 1u;
  X[0] ^= t;
  X[1] ^= t;
  X[2] ^= t;
  keys[i] = spread3(X[2]) | (spread3(X[1]) << 1) | (spread3(X[0]) << 2);
}

// one wavefront per chunk: points in curve order (xyz + original index, index -1 and +inf coordinates past the end) and the chunk boxes
__global__ __launch_bounds__(256) void curve_gather_kernel(int n, int C, const float4* __restrict__ pts, const unsigned int* __restrict__ order,
                                                           float4* __restrict__ sorted, float* __restrict__ box /* [C][6] */,
                                                           float* __restrict__ box32 /* [2 C][6] or null */, unsigned int* __restrict__ rank) {
  const int c = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (c >= C) return;
  const int s = c * CHUNK + lane;
  const float inf = __int_as_float(0x7f800000);
  float4 q = make_float4(inf, inf, inf, __int_as_float(-1));
  if (s < n) {
    const unsigned int i = order[s];
    const float4 p = pts[i];
    q = make_float4(p.x, p.y, p.z, __int_as_float((int)i));
    if (rank) rank[i] = (unsigned int)s;
  }
  sorted[s] = q;
  float lo[3] = {q.x, q.y, q.z}, hi[3] = {s < n ? q.x : -inf, s < n ? q.y : -inf, s < n ? q.z : -inf};
  // xor offsets <= 16 stay inside a 32-lane half: the boxes of the two 32-point half chunks (the pair-lane kNN kernel works on those,
  // box32[2 c + half]) fall out of the same butterfly, one step before the chunk's own box
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, 64));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off, 64));
    }
  if (box32 && (lane & 31) == 0) {
    const int h = 2 * c + (lane >> 5);
#pragma unroll
    for (int a = 0; a < 3; a++) {
      box32[6 * h + a] = lo[a];
      box32[6 * h + 3 + a] = hi[a];
    }
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
    lo[a] = fminf(lo[a], __shfl_xor(lo[a], 32, 64));
    hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], 32, 64));
  }
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      box[6 * c + a] = lo[a];
      box[6 * c + 3 + a] = hi[a];
    }
  }
}

// bounding-box accumulator (init_bbox_kernel's values) and the four status words of a kNN call, in one launch
__global__ void knn_scratch_init_kernel(int* __restrict__ bb, int* __restrict__ stats) {
  if (threadIdx.x < 6) bb[threadIdx.x] = threadIdx.x < 3 ? 0x7fffffff : (int)0x80000000;
  if (threadIdx.x < 4) stats[threadIdx.x] = 0;
}

int knn_curve(glim_amd_ctx* ctx, hipStream_t st, int n, const float4* pts, int k, int32_t* out, unsigned int* rank) {
  DeviceTemp bb, ka, kb, va, vb, hist, sorted, box, stats, box32, dbg;  // (all returned to the pool after the synchronise at the end)
  SyncOnExit in_flight(st);  // an early return after the first launch waits for the stream before the scratch above goes back to the pool
  const int C = (n + CHUNK - 1) / CHUNK;
  GA_HIP(pool_malloc(&bb.p, 6 * sizeof(int)));
  GA_HIP(pool_malloc(&ka.p, (size_t)n * sizeof(unsigned long long)));
  GA_HIP(pool_malloc(&kb.p, (size_t)n * sizeof(unsigned long long)));
  GA_HIP(pool_malloc(&va.p, (size_t)n * sizeof(unsigned int)));
  GA_HIP(pool_malloc(&vb.p, (size_t)n * sizeof(unsigned int)));
  GA_HIP(pool_malloc(&hist.p, radix_sort_scratch_bytes(n)));
  GA_HIP(pool_malloc(&sorted.p, (size_t)C * CHUNK * sizeof(float4)));
  GA_HIP(pool_malloc(&box.p, ((size_t)C + (size_t)(C + CHUNK - 1) / CHUNK) * 6 * sizeof(float)));  // chunk boxes, then the boxes of the groups of 64 chunks
  GA_HIP(pool_malloc(&stats.p, 4 * sizeof(int)));
  knn_scratch_init_kernel<<<1, 64, 0, st>>>(bb.as<int>(), stats.as<int>());
  bbox_kernel<<<std::max(1, std::min((n + 2047) / 2048, 128)), 256, 0, st>>>(n, pts, bb.as<int>());
  const int bits = n < 32768 ? 8 : 13;  // per axis: 24-bit keys / 3 sort passes for small clouds, 39 bits / 5 passes otherwise; the order only affects speed
  const unsigned int qmax = (1u << bits) - 1u;
  curve_key_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, pts, bb.as<int>(), qmax, bits, ka.as<unsigned long long>(), stats.as<int>());
  unsigned long long* ks = nullptr;
  unsigned int* order = nullptr;
  GA_HIP(radix_sort_pairs(st, n, 3 * bits, ka.as<unsigned long long>(), va.as<unsigned int>(), kb.as<unsigned long long>(), vb.as<unsigned int>(), true,
                          hist.as<int>(), &ks, &order));
  const Diag& diag = ctx->diag;
  if (diag.knn_debug[0]) GA_HIP(pool_malloc(&dbg.p, (size_t)std::max(C * 4, 8) * sizeof(int)));  // (the query-group kernel writes 5 counters whatever C)
  // Which kernel answers (all three return identical lists; diag knn_kernel=wave64 / pair / qgroup forces one):
  //  * qgroup (knn_qgroup.hip; the default since round 4): lanes are CANDIDATES, a wavefront answers 1 (clouds below 49 152 points) or 2
  //    consecutive queries -- n or n / 2 short independent work items instead of n / 64 long lock-step chains, so there is no tail and small
  //    clouds fill the chip.  Per call, same box, against the faster of the kernels below: 10 000 points 0.294 -> 0.112 ms, 32 768 0.381 -> 0.167,
  //    65 536 0.267 -> 0.169, 131 072 0.305 -> 0.233, 307 104 0.551 -> 0.437 ms (profiles/r04/probe/knn_qgroup.txt).  Four queries per
  //    wavefront share more of the walk but need 82 registers (5 wavefronts per SIMD instead of 8): 0.252 / 0.466 ms at the two large sizes.
  //  * wave64 (knn_chunks.hip) / pair (knn_pairs.hip): a query per lane (64 or 32 queries per wavefront).  Both are tail-bound -- the launch
  //    lasts as long as its slowest wavefront (rocprofv3 + SQ counters, 131 072-pt scan: mean wavefront 157 us, kernel 266 us; 41 % VALU
  //    utilisation).  Kept as independent cross-checks: wave64 from 98 304 points up, pair below (k <= 16), as round 3 shipped them.
  const bool qgroup = k > 0 && (diag.knn_kernel == KNN_KERNEL_QGROUP || diag.knn_kernel == KNN_KERNEL_AUTO);
  if (qgroup && dbg.p) GA_HIP(hipMemsetAsync(dbg.p, 0, (size_t)std::max(C * 4, 8) * sizeof(int), st));
  const bool pair_lanes = !qgroup && !dbg.p && k > 0 && k <= 16 && diag.knn_kernel != KNN_KERNEL_WAVE64 && (n <= 98304 || diag.knn_kernel == KNN_KERNEL_PAIR);
  if (pair_lanes) GA_HIP(pool_malloc(&box32.p, (size_t)C * 2 * 6 * sizeof(float)));
  curve_gather_kernel<<<(C * CHUNK + 255) / 256, 256, 0, st>>>(n, C, pts, order, sorted.as<float4>(), box.as<float>(), box32.as<float>(), rank);
  // Measured and removed: 2 / 4 wavefronts per query chunk, each owning every 2nd / 4th candidate chunk with its own top-k list, the query's
  // bound shared through LDS (ds_min_u64) and the lists merged by rank at the end -- bit-identical lists, but 0.73 / 0.62 ms against 0.50 ms
  // at 131 072 points and 1.06 / 1.19 against 0.77 ms at 307 104: every list has to be filled and pruned on its own, so the total work grows
  // faster than the longest wavefront shrinks.
  // The FP32 mask passes of the lane-per-query kernels need finite FP32 squared distances (3 ext^2 < FLT_MAX): a cloud that spans more than
  // 1e18 m is answered by the exhaustive FP64 kernel instead.
  const bool select = diag.knn_select != 0;  // per-lane threshold selection of the chunk kernels (k <= 10); knn_select=0: the plain mask pass
  const int* guard = stats.as<int>() + 2;
  if (qgroup) {
    if (g_knn_probe_events[0]) (void)hipEventRecord(g_knn_probe_events[0], st);  // (glim_amd_cloud_profile_neighbors: HIP events around the dominant kernel)
    knn_launch_qgroup(st, n, C, sorted.as<float4>(), box.as<float>(), k, out, guard, n < 49152 ? 1 : 2, dbg.as<int>());
    if (g_knn_probe_events[1]) (void)hipEventRecord(g_knn_probe_events[1], st);
  } else if (pair_lanes) {
    knn_launch_pairs(st, n, 2 * C, sorted.as<float4>(), box32.as<float>(), k, out, select, guard);
    GA_HIP(hipGetLastError());
  } else if (k > 0) {
    knn_launch_chunks(st, n, C, sorted.as<float4>(), box.as<float>(), k, out, dbg.as<int>(), select, guard);  // k == 0: ordering only
  }
  GA_HIP(hipGetLastError());
  if (dbg.p) {
    std::vector<int> hd((size_t)C * 4);
    GA_HIP(hipStreamSynchronize(st));  // (the context's streams do not synchronise with the null stream the copy below runs on)
    GA_HIP(hipMemcpy(hd.data(), dbg.p, hd.size() * sizeof(int), hipMemcpyDeviceToHost));
    if (FILE* f = fopen(diag.knn_debug, "wb")) {
      fwrite(hd.data(), sizeof(int), hd.size(), f);
      fclose(f);
    }
  }
  int h_stats[4];
  GA_HIP(read_back_sync(ctx, st, h_stats, stats.p, sizeof(h_stats)));
  in_flight.dismiss();  // synchronised
  if (h_stats[1] != 0 || h_stats[2] == 1) return GLIM_AMD_ERR_RANGE;
  if (h_stats[2] == 2 && k > 0) {  // astronomic extent: the chunk kernels stood down
    DISPATCH_K(launch_brute, st, n, pts, k, out, (const int*)nullptr, n);
    GA_HIP(hipGetLastError());
    GA_HIP(hipStreamSynchronize(st));
  }
  return GLIM_AMD_OK;
}

}  // namespace

namespace glim_amd {

// Hilbert rank of every point of a cloud that has none yet (clouds whose neighbours came from the host or from the grid path).
// Caller holds held->mu (the cloud's owner may be another context: its mutex is NOT held, so nothing of c->ctx is touched here).  Clouds with non-finite points simply stay in arrival order, and so do clouds below 32 768 points: sorting a cloud
// ONLY for its rank costs 70 us at 10 000 points, which GLIM's odometry would pay for every frame (the frame arrives with CPU covariances,
// so no kNN runs here whose by-product the rank would be) to gain 0.5 us per 34-factor linearisation
// (`bench.py --workload odometry_frame`, create_frame_us.factor_streams_on_first_use: 115 -> 44 us).
int cloud_curve_rank(glim_amd_cloud* c, glim_amd_ctx* held, hipStream_t st) {
  if (c->curve_rank || c->n < 32768 || c->n > (int64_t)(1 << 28) || !held->diag.curve_order) return GLIM_AMD_OK;
  GA_HIP(pool_malloc(&c->curve_rank, (size_t)c->n * sizeof(unsigned int)));
  const int rc = knn_curve(held, st, (int)c->n, c->pts, 0, nullptr, c->curve_rank);
  if (rc != GLIM_AMD_OK) {
    (void)pool_free(c->curve_rank);
    c->curve_rank = nullptr;
  }
  return rc == GLIM_AMD_ERR_RANGE ? GLIM_AMD_OK : rc;
}

}  // namespace glim_amd

extern "C" {

// Timing aid (bench.py rooflines of the kNN-led workloads): `iters` find_neighbors calls; wall milliseconds per call and the HIP-event duration of
// the query-group kernel inside it (0 when another kernel answered: clouds <= 2 048 points, forced paths).
int glim_amd_cloud_profile_neighbors(glim_amd_cloud* c, int k, int iters, float* ms_per_call, float* ms_qgroup_kernel) {
  if (!c || iters <= 0) return GLIM_AMD_ERR_INVALID;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  GA_HIP(hipSetDevice(c->ctx->device));
  GA_HIP(hipEventCreate(&e0));
  GA_HIP(hipEventCreate(&e1));
  int rc = glim_amd_cloud_find_neighbors(c, k, nullptr);  // warm-up (allocations, first-use paths)
  double kernel_ms = 0.0;
  int timed = 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters && rc == GLIM_AMD_OK; i++) {
    g_knn_probe_events[0] = e0;
    g_knn_probe_events[1] = e1;
    rc = glim_amd_cloud_find_neighbors(c, k, nullptr);
    g_knn_probe_events[0] = g_knn_probe_events[1] = nullptr;
    float ms = 0.f;
    if (rc == GLIM_AMD_OK && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) {
      kernel_ms += ms;
      timed++;
    }
    (void)hipGetLastError();
  }
  const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc != GLIM_AMD_OK) return rc;
  if (ms_per_call) *ms_per_call = (float)(wall_ms / iters);
  if (ms_qgroup_kernel) *ms_qgroup_kernel = timed ? (float)(kernel_ms / timed) : 0.f;
  return GLIM_AMD_OK;
}

int glim_amd_cloud_find_neighbors(glim_amd_cloud* c, int k, int32_t* neighbors_out) {
  if (!c || k <= 0) return GLIM_AMD_ERR_INVALID;
  if (k > 32) return GLIM_AMD_ERR_UNSUPPORTED;
  glim_amd_ctx* ctx = c->ctx;
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  if (c->neighbors) {
    (void)pool_free(c->neighbors);
    c->neighbors = nullptr;
  }
  c->k = k;
  GA_HIP(pool_malloc(&c->neighbors, (size_t)(c->n > 0 ? c->n : 1) * k * sizeof(int32_t)));
  if (c->n == 0) return GLIM_AMD_OK;
  if (c->n > (int64_t)(1 << 28)) return GLIM_AMD_ERR_INVALID;
  const int n = (int)c->n;
  hipStream_t st = ctx->stream();
  const Diag& diag = ctx->diag;
  const bool brute = n <= 2048 || n <= 2 * k || diag.knn_path == KNN_PATH_BRUTE;
  if (brute) {
    DISPATCH_K(launch_brute, st, n, c->pts, k, c->neighbors, (const int*)nullptr, n);
    GA_HIP(hipGetLastError());
  } else {
    // The Hilbert-chunk path answers every cloud above the exhaustive kernel's range.  The hashed grid (first implementation) is kept as an
    // independent cross-check (diag knn_path=grid): on an evenly sampled 12 000-point scan it used to be 0.09 ms faster than the chunk path, but
    // on clouds of uneven density its coarser retry levels explode -- a random 10 000 ... 32 768-point subset of a 131 072-pt scan takes it
    // 0.5 ... 5.5 ms against 0.20 ... 0.48 ms for the chunks (tools/knn_small_time.py, profiles/r03/probe/knn_small_clouds.txt).
    const bool grid = diag.knn_path == KNN_PATH_GRID;
    if (grid) GA_TRY(knn_grid(ctx, st, n, c->pts, k, c->neighbors));
    else {
      // the Hilbert rank of every point is kept: estimate_covariances writes the factor's plane-form stream in that order
      if (!c->curve_rank && diag.curve_order) GA_HIP(pool_malloc(&c->curve_rank, (size_t)n * sizeof(unsigned int)));
      GA_TRY(knn_curve(ctx, st, n, c->pts, k, c->neighbors, c->curve_rank));
    }
  }
  if (neighbors_out) GA_HIP(hipMemcpyAsync(neighbors_out, c->neighbors, (size_t)n * k * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  GA_HIP(hipStreamSynchronize(st));
  return GLIM_AMD_OK;
}

}  // extern "C"
