// gicp.hip -- the GICP (nearest-neighbour correspondence) matching-cost factor on gfx950.
//
// Replaces gtsam_points::IntegratedGICPFactor::{linearize, error, inlier_fraction} as constructed at
// src/glim/mapping/sub_mapping.cpp:202 (between factors), src/glim/mapping/global_mapping.cpp:400 (10 LM iterations between
// consecutive submaps) and src/glim/mapping/global_mapping_pose_graph.cpp:393 (loop-candidate validation, with the target's
// pre-built search tree `candidate.target->tree`).  Same cost as the VGICP factor (vgicp.hip) except for the correspondence:
//
//   q = R p + t                                   FP64, the oracle's fma order (shared with the VGICP factor)
//   j = argmin_j |q - b_j|^2 over the target      exact: FP64 (dx^2 + dy^2) + dz^2 with separate roundings, ties to the smaller index
//   valid iff |q - b_j|^2 <= max_correspondence_distance^2
//   M = (C_B[j] + R C_A R^T)^-1,  r = b_j - q,  e = r^T M r,  H_ss += J_s^T M J_s, b_s += J_s^T M r      (as vgicp.hip)
//
// Search structure = the kd-tree's stand-in (glim_amd_nn_index, built once per target cloud and reused by every linearisation,
// like `candidate.target->tree`): the target points counting-sorted into a uniform grid.  The sort is the stable radix sort of
// sort.hip on cell keys compacted to the bounding box, so the sorted order -- (cell, original index) -- is deterministic; an
// open-addressing table maps a cell key to its [begin, end) run.  A query walks growing Chebyshev rings of cells around q's cell
// and stops as soon as the best distance is provably inside the scanned cube, or the cube already covers the correspondence
// radius (nothing farther can be accepted).  One lane per source point; the per-point algebra and the block reduction are the
// VGICP ones (source-frame form, 28 FP32 accumulators, DPP wave sums, fixed-order FP64 finalisation: bit-reproducible).
#include <algorithm>
#include <cmath>
#include <memory>

#include "device_math.hpp"
#include "internal.hpp"
#include "scope_sync.hpp"
#include "scan.hpp"

using namespace glim_amd;

struct glim_amd_nn_index {
  CtxRef ctx;
  const glim_amd_cloud* cloud = nullptr;  // not owned; must outlive the index
  int n = 0;
  double h = 0.0;                         // cell edge
  unsigned int mask = 0;                  // table size - 1 (power of two)
  unsigned long long* keys = nullptr;     // cell key per table slot (EMPTY_KEY when free)
  int2* runs = nullptr;                   // [begin, end) of the cell's run in the sorted order
  float4* sorted = nullptr;               // xyz + original index (int bits), cell order
  float4* covA = nullptr;                 // target covariances in the same order
  float2* covB = nullptr;
};

namespace {

using u64 = unsigned long long;
using u32 = unsigned int;
constexpr int BLOCK = 256;
constexpr int NACC = 28;
constexpr int GICP_MAX_RING = 64;
__constant__ int c_acc_of_upper_g[21] = {0, 1, 2, 6, 7, 8, 3, 4, 9, 10, 11, 5, 12, 13, 14, 15, 16, 17, 18, 19, 20};

// ---- index build ----
__global__ __launch_bounds__(256) void gi_key_kernel(int n, const float4* __restrict__ pts, double inv_h, u64* __restrict__ vkey, int* __restrict__ bb) {
  __shared__ int s_tmp[16];
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float4 p = pts[i];
    const double t[3] = {(double)p.x * inv_h, (double)p.y * inv_h, (double)p.z * inv_h};
    bool valid = true;
#pragma unroll
    for (int a = 0; a < 3; a++) valid = valid && (t[a] >= -1048576.0 && t[a] < 1048576.0);
    u64 key = EMPTY_KEY;
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

__global__ __launch_bounds__(256) void gi_compact_key_kernel(int n, const u64* __restrict__ vkey, int xmin, int ymin, int zmin, int bx, int by, int vbits,
                                                             u64* __restrict__ ckey) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u64 k = vkey[i];
  if (k == EMPTY_KEY) {
    ckey[i] = 1ull << vbits;  // points without a cell (non-finite / out of range) sort last and are never indexed
    return;
  }
  const u64 x = (k & 0x1FFFFFull) - (u64)xmin, y = ((k >> 21) & 0x1FFFFFull) - (u64)ymin, z = ((k >> 42) & 0x1FFFFFull) - (u64)zmin;
  ckey[i] = (z << (bx + by)) | (y << bx) | x;
}

__device__ __forceinline__ u32 cell_hash(u64 key) {
  u64 z = key * 0x9E3779B97F4A7C15ull;
  z ^= z >> 29;
  return (u32)(z * 0xBF58476D1CE4E5B9ull >> 32);
}

// sorted order -> gather the points / covariances, and register every cell run in the table
__global__ __launch_bounds__(256) void gi_gather_kernel(int n, const u64* __restrict__ ckey_sorted, const u32* __restrict__ order, u64 invalid,
                                                        const u64* __restrict__ vkey, const float4* __restrict__ pts, const float4* __restrict__ covA,
                                                        const float2* __restrict__ covB, float4* __restrict__ sorted, float4* __restrict__ sA,
                                                        float2* __restrict__ sB, u64* __restrict__ keys, int2* __restrict__ runs, u32 mask) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const u32 i = order[j];
  const float4 p = pts[i];
  sorted[j] = make_float4(p.x, p.y, p.z, __int_as_float((int)i));
  if (covA) {
    sA[j] = covA[i];
    sB[j] = covB[i];
  }
  const u64 ck = ckey_sorted[j];
  if (ck == invalid) return;
  const bool head = j == 0 || ckey_sorted[j - 1] != ck;
  const bool tail = j == n - 1 || ckey_sorted[j + 1] != ck;
  if (!head && !tail) return;
  // the full (uncompacted) cell key identifies the cell; the head claims the slot, head and tail fill begin / end
  const u64 key = vkey[i];
  u32 s = cell_hash(key) & mask;
  for (;;) {
    const u64 prev = atomicCAS(&keys[s], EMPTY_KEY, key);
    if (prev == EMPTY_KEY || prev == key) break;
    s = (s + 1) & mask;
  }
  if (head) runs[s].x = j;
  if (tail) runs[s].y = j + 1;
}

// ---- the factor ----
struct GicpArgs {
  const float4* sorted;
  const float4* tA;
  const float2* tB;
  const u64* keys;
  const int2* runs;
  u32 mask;
  int nt;
  double h, inv_h;
  const float4* pts;   // source xyz1
  const float4* covA;  // source covariances
  const float2* covB;
  int n;
  int ppt;             // points per thread (chunk = 256 * ppt consecutive points per block)
  double T[12];
  double max_sq;       // max_correspondence_distance^2
  int max_ring;        // rings after which the scanned cube covers the correspondence radius
};

__device__ __forceinline__ double sqdist_nc(double qx, double qy, double qz, double x, double y, double z) {
  const double dx = qx - x, dy = qy - y, dz = qz - z;
  return dadd(dadd(dmul(dx, dx), dmul(dy, dy)), dmul(dz, dz));
}

// exact nearest target point of q within sqrt(max_sq): position in the sorted order, or -1
__device__ __forceinline__ int nearest(const GicpArgs& a, double qx, double qy, double qz, double& best_d) {
  const double tx = qx * a.inv_h, ty = qy * a.inv_h, tz = qz * a.inv_h;
  if (!(tx >= -1048576.0 && tx < 1048576.0 && ty >= -1048576.0 && ty < 1048576.0 && tz >= -1048576.0 && tz < 1048576.0)) return -1;
  const int cx = fast_floor_d(tx), cy = fast_floor_d(ty), cz = fast_floor_d(tz);
  double margin = fmin(fmin(tx - (double)cx, (double)(cx + 1) - tx), fmin(fmin(ty - (double)cy, (double)(cy + 1) - ty), fmin(tz - (double)cz, (double)(cz + 1) - tz)));
  margin = fmax(0.0, margin * a.h * 0.999999);
  int best = -1, best_idx = 0x7fffffff;
  best_d = __longlong_as_double(0x7ff0000000000000ll);
  for (int ring = 0; ring <= a.max_ring; ring++) {
    if (ring >= 1) {
      const double reach = (double)(ring - 1) * a.h * 0.999999 + margin;  // every unscanned point is at least this far
      const double r2 = reach * reach;
      if (best_d < r2 || r2 > a.max_sq) break;  // proven nearest (strict: an unscanned tie could carry a smaller index) / beyond the radius
    }
    for (int dz = -ring; dz <= ring; dz++)
      for (int dy = -ring; dy <= ring; dy++) {
        const bool shell_yz = (abs(dz) == ring) || (abs(dy) == ring);
        for (int dx = -ring; dx <= ring; dx += (shell_yz || ring == 0) ? 1 : 2 * ring) {
          if (ring >= 1) {
            // prune: no point of this cell is closer than the gap between q and the cell's box (shrunk so that rounding in the cell
            // assignment cannot make it optimistic); strictly greater than the current best / the radius, so ties are still seen
            const double gx = dx > 0 ? (double)(cx + dx) - tx : (dx < 0 ? tx - (double)(cx + dx + 1) : 0.0);
            const double gy = dy > 0 ? (double)(cy + dy) - ty : (dy < 0 ? ty - (double)(cy + dy + 1) : 0.0);
            const double gz = dz > 0 ? (double)(cz + dz) - tz : (dz < 0 ? tz - (double)(cz + dz + 1) : 0.0);
            const double gap = a.h * 0.999999;
            const double ex = fmax(0.0, gx) * gap, ey = fmax(0.0, gy) * gap, ez = fmax(0.0, gz) * gap;
            const double g2 = ex * ex + ey * ey + ez * ez;
            if (g2 > best_d || g2 > a.max_sq) continue;
          }
          const u32 ux = (u32)(cx + dx + KEY_OFFSET), uy = (u32)(cy + dy + KEY_OFFSET), uz = (u32)(cz + dz + KEY_OFFSET);
          if ((ux | uy | uz) >> KEY_BITS) continue;
          const u64 key = (u64)ux | ((u64)uy << 21) | ((u64)uz << 42);
          u32 s = cell_hash(key) & a.mask;
          int2 run = make_int2(0, 0);
          for (;;) {
            const u64 kk = a.keys[s];
            if (kk == key) {
              run = a.runs[s];
              break;
            }
            if (kk == EMPTY_KEY) break;
            s = (s + 1) & a.mask;
          }
          for (int j = run.x; j < run.y; j++) {
            const float4 c = a.sorted[j];
            const double d = sqdist_nc(qx, qy, qz, (double)c.x, (double)c.y, (double)c.z);
            const int idx = __float_as_int(c.w);
            if (d < best_d || (d == best_d && idx < best_idx)) {
              best_d = d;
              best = j;
              best_idx = idx;
            }
          }
        }
      }
  }
  return (best >= 0 && best_d <= a.max_sq) ? best : -1;
}

template <bool LINEARIZE>
__global__ __launch_bounds__(BLOCK) void gicp_kernel(const GicpArgs a, float* __restrict__ partials, int32_t* __restrict__ corr) {
  __shared__ float s_red[4][PARTIAL_STRIDE];
  const double* T = a.T;
  const float R00 = (float)T[0], R01 = (float)T[1], R02 = (float)T[2];
  const float R10 = (float)T[4], R11 = (float)T[5], R12 = (float)T[6];
  const float R20 = (float)T[8], R21 = (float)T[9], R22 = (float)T[10];
  float acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; j++) acc[j] = 0.f;
  int inliers = 0;
  const int base = blockIdx.x * (BLOCK * a.ppt) + threadIdx.x;
  for (int it = 0; it < a.ppt; it++) {
    const int i = base + it * BLOCK;
    if (i >= a.n) break;
    const float4 p = a.pts[i];
    double qx, qy, qz;
    transform_point_d(T, (double)p.x, (double)p.y, (double)p.z, qx, qy, qz);
    double best_d;
    const int j = nearest(a, qx, qy, qz, best_d);
    if (corr) corr[i] = j >= 0 ? __float_as_int(a.sorted[j].w) : -1;
    if (j < 0) continue;
    inliers++;
    const float4 b = a.sorted[j];
    const float4 tA = a.tA[j];
    const float2 tB = a.tB[j];
    const float4 ca = a.covA[i];
    const float2 cb = a.covB[i];
    // residual b_j - q: formed in FP64 (|r| <= the correspondence radius), then FP32
    const float rx = (float)((double)b.x - qx), ry = (float)((double)b.y - qy), rz = (float)((double)b.z - qz);
    // S = R^T C_B R + C_A (source frame, symmetric)
    const float b00 = tA.x, b01 = tA.y, b02 = tA.z, b11 = tA.w, b12 = tB.x, b22 = tB.y;
    const float w00 = b00 * R00 + b01 * R10 + b02 * R20, w01 = b00 * R01 + b01 * R11 + b02 * R21, w02 = b00 * R02 + b01 * R12 + b02 * R22;
    const float w10 = b01 * R00 + b11 * R10 + b12 * R20, w11 = b01 * R01 + b11 * R11 + b12 * R21, w12 = b01 * R02 + b11 * R12 + b12 * R22;
    const float w20 = b02 * R00 + b12 * R10 + b22 * R20, w21 = b02 * R01 + b12 * R11 + b22 * R21, w22 = b02 * R02 + b12 * R12 + b22 * R22;
    const float S00 = ca.x + R00 * w00 + R10 * w10 + R20 * w20;
    const float S01 = ca.y + R00 * w01 + R10 * w11 + R20 * w21;
    const float S02 = ca.z + R00 * w02 + R10 * w12 + R20 * w22;
    const float S11 = ca.w + R01 * w01 + R11 * w11 + R21 * w21;
    const float S12 = cb.x + R01 * w02 + R11 * w12 + R21 * w22;
    const float S22 = cb.y + R02 * w02 + R12 * w12 + R22 * w22;
    const float k00 = S11 * S22 - S12 * S12;
    const float k01 = S02 * S12 - S01 * S22;
    const float k02 = S01 * S12 - S02 * S11;
    const float det = S00 * k00 + S01 * k01 + S02 * k02;
    float idet = __builtin_amdgcn_rcpf(det);
    idet = fmaf(fmaf(-det, idet, 1.0f), idet, idet);
    const float A00 = k00 * idet, A01 = k01 * idet, A02 = k02 * idet;
    const float A11 = (S00 * S22 - S02 * S02) * idet;
    const float A12 = (S01 * S02 - S00 * S12) * idet;
    const float A22 = (S00 * S11 - S01 * S01) * idet;
    const float rsx = R00 * rx + R10 * ry + R20 * rz;
    const float rsy = R01 * rx + R11 * ry + R21 * rz;
    const float rsz = R02 * rx + R12 * ry + R22 * rz;
    const float ux = A00 * rsx + A01 * rsy + A02 * rsz;
    const float uy = A01 * rsx + A11 * rsy + A12 * rsz;
    const float uz = A02 * rsx + A12 * rsy + A22 * rsz;
    acc[27] += rsx * ux + rsy * uy + rsz * uz;
    if (LINEARIZE) {
      const float x = p.x, y = p.y, z = p.z;
      const float g00 = y * A02 - z * A01, g01 = y * A12 - z * A11, g02 = y * A22 - z * A12;
      const float g10 = z * A00 - x * A02, g11 = z * A01 - x * A12, g12 = z * A02 - x * A22;
      const float g20 = x * A01 - y * A00, g21 = x * A11 - y * A01, g22 = x * A12 - y * A02;
      acc[0] += y * g02 - z * g01;
      acc[1] += z * g00 - x * g02;
      acc[2] += x * g01 - y * g00;
      acc[3] += z * g10 - x * g12;
      acc[4] += x * g11 - y * g10;
      acc[5] += x * g21 - y * g20;
      acc[6] += g00; acc[7] += g01; acc[8] += g02;
      acc[9] += g10; acc[10] += g11; acc[11] += g12;
      acc[12] += g20; acc[13] += g21; acc[14] += g22;
      acc[15] += A00; acc[16] += A01; acc[17] += A02; acc[18] += A11; acc[19] += A12; acc[20] += A22;
      acc[21] += uy * z - uz * y;
      acc[22] += uz * x - ux * z;
      acc[23] += ux * y - uy * x;
      acc[24] += ux; acc[25] += uy; acc[26] += uz;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  wave_sums_to_lane63<NACC>(acc);  // (step-major: device_math.hpp; the same six additions per value)
  if (lane == 63) {
#pragma unroll
    for (int j = 0; j < NACC; j++) s_red[wave][j] = acc[j];
  }
  {
    const float v = wave_sum_to_lane63((float)inliers);
    if (lane == 63) s_red[wave][28] = v;
  }
  __syncthreads();
  if (threadIdx.x < PARTIAL_STRIDE) {
    const int j = threadIdx.x;
    partials[(size_t)blockIdx.x * PARTIAL_STRIDE + j] = j <= 28 ? (s_red[0][j] + s_red[1][j]) + (s_red[2][j] + s_red[3][j]) : 0.f;
  }
}

// fixed-order FP64 sum of the block partials -> compact record (one block of 256 threads; same order as vgicp.hip's finalise)
__global__ __launch_bounds__(256) void gicp_finalize_kernel(const float* __restrict__ partials, int nb, int linearize, double* __restrict__ out) {
  __shared__ double s_part[8][PARTIAL_STRIDE];
  __shared__ double s_sum[PARTIAL_STRIDE];
  const int j = threadIdx.x & 31, g = threadIdx.x >> 5;
  double s = 0.0;
  constexpr int INFLIGHT = 16;  // loads of 16 trips in flight, additions in the same order (see vgicp.hip finalize_factor)
  for (int c = g; c < nb; c += 8 * INFLIGHT) {
    float v[INFLIGHT];
#pragma unroll
    for (int u = 0; u < INFLIGHT; u++) v[u] = partials[(size_t)min(c + 8 * u, nb - 1) * PARTIAL_STRIDE + j];
#pragma unroll
    for (int u = 0; u < INFLIGHT; u++)
      if (c + 8 * u < nb) s += (double)v[u];
  }
  s_part[g][j] = s;
  __syncthreads();
  if (threadIdx.x < PARTIAL_STRIDE) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 8; k++) t += s_part[k][threadIdx.x];
    s_sum[threadIdx.x] = t;
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t == 0) out[0] = s_sum[28];
  if (t == 1) out[1] = s_sum[27];
  if (linearize) {
    if (t < 21) out[2 + t] = s_sum[c_acc_of_upper_g[t]];
    if (t >= 21 && t < 24) out[2 + t] = s_sum[t];
    if (t >= 24 && t < 27) out[2 + t] = -s_sum[t];
  } else if (t >= 2 && t < COMPACT) {
    out[t] = 0.0;
  }
}

inline int grid_for(int n) { return (n + 255) / 256; }
inline int bits_for(int range) {
  int b = 0;
  while (range > 0) {
    b++;
    range >>= 1;
  }
  return b;
}
unsigned int next_pow2(unsigned long long v) {
  unsigned long long p = 1;
  while (p < v) p <<= 1;
  return (unsigned int)p;
}

int run_gicp(const glim_amd_nn_index* ix, const glim_amd_cloud* source, const double* T12, double max_dist, bool linearize, double* compact_host,
             int32_t* corr_host) {
  if (!ix || !source || !T12 || !(max_dist >= 0.0)) return GLIM_AMD_ERR_INVALID;
  if (source->ctx->device != ix->ctx->device) return GLIM_AMD_ERR_INVALID;
  if (!source->has_covs || !ix->covA) return GLIM_AMD_ERR_STATE;
  glim_amd_ctx* ctx = ix->ctx;
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream();
  const int n = (int)source->n;
  for (int i = 0; i < COMPACT; i++) compact_host[i] = 0.0;
  if (n == 0 || ix->n == 0) {
    if (corr_host)
      for (int i = 0; i < n; i++) corr_host[i] = -1;
    return GLIM_AMD_OK;
  }
  GicpArgs a;
  a.sorted = ix->sorted;
  a.tA = ix->covA;
  a.tB = ix->covB;
  a.keys = ix->keys;
  a.runs = ix->runs;
  a.mask = ix->mask;
  a.nt = ix->n;
  a.h = ix->h;
  a.inv_h = 1.0 / ix->h;
  a.pts = source->pts;
  a.covA = source->covA;
  a.covB = source->covB;
  a.n = n;
  memcpy(a.T, T12, sizeof(a.T));
  a.max_sq = max_dist * max_dist;
  // rings after which the scanned cube covers the correspondence radius; the walk is bounded, so a radius far beyond what the index was
  // sized for (cells are hint/3 .. hint wide) is refused rather than searched incompletely
  const double rings = std::ceil(max_dist / ix->h) + 1.0;
  if (!(rings <= (double)GICP_MAX_RING)) return GLIM_AMD_ERR_UNSUPPORTED;
  a.max_ring = (int)rings;
  // the search is latency-bound: spread the points over >= 4 blocks per CU when there are enough of them
  const int target_blocks = std::max(1, ctx->num_cus * 4);
  a.ppt = std::max(1, std::min(64, (n + BLOCK * target_blocks - 1) / (BLOCK * target_blocks)));
  const int nb = (n + BLOCK * a.ppt - 1) / (BLOCK * a.ppt);
  DeviceTemp partials, compact, corr;
  SyncOnExit in_flight(st);  // an error exit after the launches waits for the stream before the scratch goes back to the pool
  GA_HIP(pool_malloc(&partials.p, (size_t)nb * PARTIAL_STRIDE * sizeof(float)));
  GA_HIP(pool_malloc(&compact.p, COMPACT * sizeof(double)));
  if (corr_host) GA_HIP(pool_malloc(&corr.p, (size_t)n * sizeof(int32_t)));
  if (linearize) gicp_kernel<true><<<nb, BLOCK, 0, st>>>(a, partials.as<float>(), corr.as<int32_t>());
  else gicp_kernel<false><<<nb, BLOCK, 0, st>>>(a, partials.as<float>(), corr.as<int32_t>());
  gicp_finalize_kernel<<<1, 256, 0, st>>>(partials.as<float>(), nb, linearize ? 1 : 0, compact.as<double>());
  GA_HIP(hipGetLastError());
  GA_HIP(hipMemcpyAsync(compact_host, compact.p, COMPACT * sizeof(double), hipMemcpyDeviceToHost, st));
  if (corr_host) GA_HIP(hipMemcpyAsync(corr_host, corr.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  GA_HIP(hipStreamSynchronize(st));
  in_flight.dismiss();
  return GLIM_AMD_OK;
}

}  // namespace

extern "C" {

int glim_amd_nn_index_create(const glim_amd_cloud* target, double max_correspondence_distance_hint, glim_amd_nn_index** out) {
  if (!target || !out) return GLIM_AMD_ERR_INVALID;
  *out = nullptr;
  if (target->n > (int64_t)(1 << 28)) return GLIM_AMD_ERR_INVALID;
  glim_amd_ctx* ctx = target->ctx;
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream();
  const int n = (int)target->n;
  std::unique_ptr<glim_amd_nn_index, int (*)(glim_amd_nn_index*)> ix(new glim_amd_nn_index(), glim_amd_nn_index_destroy);
  ix->ctx = ctx;
  ix->cloud = target;
  ix->n = n;
  ix->h = 1.0;
  ix->mask = 0;
  const size_t nn = (size_t)std::max(n, 1);
  GA_HIP(pool_malloc(&ix->sorted, nn * sizeof(float4)));
  if (target->has_covs) {
    GA_HIP(pool_malloc(&ix->covA, nn * sizeof(float4)));
    GA_HIP(pool_malloc(&ix->covB, nn * sizeof(float2)));
  }
  if (n == 0) {
    GA_HIP(pool_malloc(&ix->keys, sizeof(u64)));
    GA_HIP(pool_malloc(&ix->runs, sizeof(int2)));
    GA_HIP(hipMemsetAsync(ix->keys, 0xff, sizeof(u64), st));
    GA_HIP(hipStreamSynchronize(st));
    *out = ix.release();
    return GLIM_AMD_OK;
  }
  DeviceTemp vkey, bb, ka, kb, va, vb, hist;
  // (declared after the index object too: an error exit waits for the stream before scratch AND the half-built index are released)
  SyncOnExit in_flight(st);
  GA_HIP(pool_malloc(&vkey.p, nn * sizeof(u64)));
  GA_HIP(pool_malloc(&bb.p, 6 * sizeof(int)));
  GA_HIP(pool_malloc(&ka.p, nn * sizeof(u64)));
  GA_HIP(pool_malloc(&kb.p, nn * sizeof(u64)));
  GA_HIP(pool_malloc(&va.p, nn * sizeof(u32)));
  GA_HIP(pool_malloc(&vb.p, nn * sizeof(u32)));
  GA_HIP(pool_malloc(&hist.p, radix_sort_scratch_bytes(n)));
  const int blocks = std::max(1, std::min((n + 2047) / 2048, 128));
  int h_bb[6];
  // pass 1 with a provisional cell edge to learn the extent, then the edge that gives ~3 points per occupied cell of a surface-like
  // cloud, clamped to [R / 3, R] (R = the correspondence radius hint): at most 4 rings are ever scanned
  const double R = max_correspondence_distance_hint > 0.0 ? max_correspondence_distance_hint : 1.0;
  double h = R;
  for (int pass = 0; pass < 2; pass++) {
    init_bbox_kernel<<<1, 64, 0, st>>>(bb.as<int>());
    gi_key_kernel<<<blocks, 256, 0, st>>>(n, target->pts, 1.0 / h, vkey.as<u64>(), bb.as<int>());
    GA_HIP(hipGetLastError());
    GA_HIP(read_back_sync(ctx, st, h_bb, bb.p, sizeof(h_bb)));
    if (pass == 1 || h_bb[0] > h_bb[3]) break;
    const double ex = (h_bb[3] - h_bb[0] + 1) * h, ey = (h_bb[4] - h_bb[1] + 1) * h, ez = (h_bb[5] - h_bb[2] + 1) * h;
    const double area = ex * ey + ey * ez + ex * ez;
    const double h_density = std::sqrt(3.0 * 2.0 * area / (double)n);
    const double h_new = std::min(R, std::max(R / 3.0, h_density));
    if (h_new == h) break;
    h = h_new;
  }
  ix->h = h;
  int bx = 0, by = 0, bz = 0;
  if (h_bb[0] <= h_bb[3]) {
    bx = bits_for(h_bb[3] - h_bb[0]);
    by = bits_for(h_bb[4] - h_bb[1]);
    bz = bits_for(h_bb[5] - h_bb[2]);
  } else {
    h_bb[0] = h_bb[1] = h_bb[2] = 0;
  }
  const int vbits = bx + by + bz;
  gi_compact_key_kernel<<<grid_for(n), 256, 0, st>>>(n, vkey.as<u64>(), h_bb[0], h_bb[1], h_bb[2], bx, by, vbits, ka.as<u64>());
  u64* ks = nullptr;
  u32* vs = nullptr;
  GA_HIP(radix_sort_pairs(st, n, vbits + 1, ka.as<u64>(), va.as<u32>(), kb.as<u64>(), vb.as<u32>(), true, hist.as<int>(), &ks, &vs));
  const unsigned int T = next_pow2((unsigned long long)n * 2);
  ix->mask = T - 1;
  GA_HIP(pool_malloc(&ix->keys, (size_t)T * sizeof(u64)));
  GA_HIP(pool_malloc(&ix->runs, (size_t)T * sizeof(int2)));
  GA_HIP(hipMemsetAsync(ix->keys, 0xff, (size_t)T * sizeof(u64), st));
  gi_gather_kernel<<<grid_for(n), 256, 0, st>>>(n, ks, vs, 1ull << vbits, vkey.as<u64>(), target->pts, target->has_covs ? target->covA : nullptr,
                                                target->has_covs ? target->covB : nullptr, ix->sorted, ix->covA, ix->covB, ix->keys, ix->runs, ix->mask);
  GA_HIP(hipGetLastError());
  GA_HIP(hipStreamSynchronize(st));
  in_flight.dismiss();
  *out = ix.release();
  return GLIM_AMD_OK;
}

int glim_amd_nn_index_destroy(glim_amd_nn_index* ix) {
  if (!ix) return GLIM_AMD_OK;
  if (ix->ctx) (void)hipSetDevice(ix->ctx->device);
  if (ix->keys) (void)pool_free(ix->keys);
  if (ix->runs) (void)pool_free(ix->runs);
  if (ix->sorted) (void)pool_free(ix->sorted);
  if (ix->covA) (void)pool_free(ix->covA);
  if (ix->covB) (void)pool_free(ix->covB);
  delete ix;
  return GLIM_AMD_OK;
}

int glim_amd_gicp_linearize(const glim_amd_nn_index* target, const glim_amd_cloud* source, const double* T_target_source12,
                            double max_correspondence_distance, uint32_t flags, glim_amd_linearized6* out) {
  if (!out) return GLIM_AMD_ERR_INVALID;
  double compact[COMPACT];
  GA_TRY(run_gicp(target, source, T_target_source12, max_correspondence_distance, true, compact, nullptr));
  return glim_amd_expand_compact(compact, T_target_source12, flags, out);
}

int glim_amd_gicp_error(const glim_amd_nn_index* target, const glim_amd_cloud* source, const double* T_target_source12,
                        double max_correspondence_distance, double* error, int64_t* num_inliers) {
  double compact[COMPACT];
  GA_TRY(run_gicp(target, source, T_target_source12, max_correspondence_distance, false, compact, nullptr));
  if (error) *error = compact[1];
  if (num_inliers) *num_inliers = (int64_t)llround(compact[0]);
  return GLIM_AMD_OK;
}

int glim_amd_gicp_correspondences(const glim_amd_nn_index* target, const glim_amd_cloud* source, const double* T_target_source12,
                                  double max_correspondence_distance, int32_t* correspondences) {
  if (!correspondences) return GLIM_AMD_ERR_INVALID;
  double compact[COMPACT];
  return run_gicp(target, source, T_target_source12, max_correspondence_distance, false, compact, correspondences);
}

}  // extern "C"
