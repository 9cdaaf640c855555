// knn_qgroup.hip -- kernel group K2, the query-group Hilbert-chunk kernel (see knn.hip for the method and the host side).
//
// The 64-query kernel (knn_chunks.hip) and the pair-lane kernel (knn_pairs.hip) give every LANE a query: a wavefront is a lock-step chain of
// (candidate chunks scanned) x 64 candidate steps + (insertion rounds of its slowest lane), all wavefronts are resident at once and the launch
// lasts as long as the slowest of them -- twice the mean at 131 072 points -- while a 10 000-point cloud fills 313 of the chip's 1 024 SIMDs with one
// such chain each.  Here the lanes are the CANDIDATES: a wavefront answers Q (1 or 2) consecutive queries of the curve order, a scanned chunk is ONE
// coalesced load (lane j holds candidate j), the Q distances per lane are the oracle's FP64 expression directly (no FP32 pre-pass to undo),
// v_cmp against the query's k-th best IS the ballot of candidates that enter, and each query's list lives in lanes 0..K-1 (lane j = j-th best),
// so an insertion is a vote, a population count and a one-lane shift.  A cloud of n points is n / Q short independent work items: no tail, and
// small clouds fill the chip.  The first chunk (the query's own) is sorted by a 64-lane bitonic network instead of 64 insertions.
// Lists are bit-identical to the other kernels and to the oracle: same FP64 distance expression, ties by original index, every point of every
// chunk that can hold a better candidate is offered (the chunk test below is conservative).
#include "knn_common.hpp"

using namespace glim_amd;

namespace {

__device__ __forceinline__ double uniform_d(double v) {  // a wave-uniform double, moved to scalar registers
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ double lane_d(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
// (d1, i1) orders strictly before (d2, i2); bitwise on purpose: the short-circuit form compiles to exec-mask branches
__device__ __forceinline__ bool before(double d1, int i1, double d2, int i2) { return (d1 < d2) | ((d1 == d2) & (i1 < i2)); }

// Cross-lane moves without the LDS permute path where the hardware has a cheaper one: DPP for lane ^ 1, 2 (quad_perm) and, as two chained
// mirrors, lane ^ 4 (row_half_mirror of the quad mirror) and lane ^ 8 (row_mirror of the half mirror); ds_swizzle for lane ^ 16; the one
// lane ^ 32 stage goes through ds_bpermute.
template <int CTRL>
__device__ __forceinline__ int dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int J>
__device__ __forceinline__ int lane_xor(int v) {
  if constexpr (J == 1) return dpp<0xB1>(v);                     // quad_perm [1,0,3,2]
  else if constexpr (J == 2) return dpp<0x4E>(v);                // quad_perm [2,3,0,1]
  else if constexpr (J == 4) return dpp<0x141>(dpp<0x1B>(v));    // row_half_mirror o quad_perm [3,2,1,0]
  else if constexpr (J == 8) return dpp<0x140>(dpp<0x141>(v));   // row_mirror o row_half_mirror
  else if constexpr (J == 16) return __builtin_amdgcn_ds_swizzle(v, 0x401F);  // bit mode: and 0x1f, or 0, xor 0x10
  else return __shfl_xor(v, 32, 64);
}
// value of lane - 1 (lane 0: unspecified): DPP wave_shr:1
__device__ __forceinline__ int from_below(int v) { return dpp<0x138>(v); }

template <int K2, int J>
__device__ __forceinline__ void sort_stage(double& d, int& idx, int lane) {
  const int plo = lane_xor<J>(__double2loint(d)), phi = lane_xor<J>(__double2hiint(d)), pi = lane_xor<J>(idx);
  const double pd = __hiloint2double(phi, plo);
  const bool keep_min = ((lane & J) == 0) == ((lane & K2) == 0);
  // (keys are distinct except among padding entries, which are all (+inf, INT_MAX): "not before" then means "after or identical")
  const bool take = before(pd, pi, d, idx) == keep_min;
  d = take ? pd : d;
  idx = take ? pi : idx;
}
template <int K2, int J>
__device__ __forceinline__ void sort_merge(double& d, int& idx, int lane) {
  sort_stage<K2, J>(d, idx, lane);
  if constexpr (J > 1) sort_merge<K2, J / 2>(d, idx, lane);
}
// 64-lane bitonic sort, ascending by (d, idx)
__device__ __forceinline__ void sort64(double& d, int& idx, int lane) {
  sort_merge<2, 1>(d, idx, lane);
  sort_merge<4, 2>(d, idx, lane);
  sort_merge<8, 4>(d, idx, lane);
  sort_merge<16, 8>(d, idx, lane);
  sort_merge<32, 16>(d, idx, lane);
  sort_merge<64, 32>(d, idx, lane);
}

template <int K, int Q, bool DBG>
__global__ __launch_bounds__(256) void knn_qgroup_kernel(int n, int C, const float4* __restrict__ sorted, const float* __restrict__ box, int k,
                                                         int32_t* __restrict__ out, const int* __restrict__ guard, int* __restrict__ dbg) {
  static_assert(CHUNK % Q == 0 && K <= 32, "query groups tile a chunk; a list fits the lower half of a wavefront");
  constexpr int GPC = CHUNK / Q;  // query groups per chunk
  const int lane = threadIdx.x & 63;
  // consecutive workgroups go to different XCDs (8, each with its own L2): XCD x takes the x-th eighth of the curve, so that an L2 holds one
  // eighth of the sorted cloud instead of all of it (a 307 104-pt frame is 4.9 MB, an L2 4 MB)
  const int per_xcd = gridDim.x >> 3;  // (the grid is a multiple of 8 workgroups)
  const int wg = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int gq = __builtin_amdgcn_readfirstlane(wg * 4 + (threadIdx.x >> 6));
  const int c = gq / GPC, sub = gq % GPC;
  if (c >= C) return;       // whole wavefront
  if (*guard != 0) return;  // see knn_chunk_kernel
  const double inf = __longlong_as_double(0x7ff0000000000000ll);

  // the Q queries (wave-uniform).  Padding queries at the end of the last chunk repeat the group's first query (they never widen the search)
  // and write nothing; a group that starts in the padding has nothing to do.
  float qxf[Q], qyf[Q], qzf[Q];
  double qx[Q], qy[Q], qz[Q];
  int self[Q];
#pragma unroll
  for (int i = 0; i < Q; i++) {
    const float4 q4 = sorted[c * CHUNK + sub * Q + i];
    self[i] = __builtin_amdgcn_readfirstlane(__float_as_int(q4.w));
    const bool livei = self[i] >= 0;
    qxf[i] = uniform_f(livei || i == 0 ? q4.x : qxf[0]);
    qyf[i] = uniform_f(livei || i == 0 ? q4.y : qyf[0]);
    qzf[i] = uniform_f(livei || i == 0 ? q4.z : qzf[0]);
    qx[i] = uniform_d((double)qxf[i]);
    qy[i] = uniform_d((double)qyf[i]);
    qz[i] = uniform_d((double)qzf[i]);
  }
  if (self[0] < 0) return;

  double ld[Q];   // lane j < K: distance of the query's j-th best
  int li[Q];      //             and its index
  double bd[Q];   // the k-th best of the moment (wave-uniform): the acceptance bound ...
  int bi[Q];
  float bd32[Q];  // ... and its FP32 image for the chunk test, inflated beyond that test's rounding
#pragma unroll
  for (int i = 0; i < Q; i++) {
    ld[i] = inf;
    li[i] = 0x7fffffff;
    bd[i] = inf;
    bi[i] = 0x7fffffff;
    bd32[i] = __int_as_float(0x7f800000);
  }
  auto refresh_bound = [&](int i) {
    bd[i] = lane_d(ld[i], K - 1);
    bi[i] = __builtin_amdgcn_readlane(li[i], K - 1);
    bd32[i] = (float)bd[i] * 1.00001f + 1e-37f;  // +inf stays +inf
  };

  int dbg_scans = 0, dbg_exact = 0, dbg_inserts = 0, dbg_tests = 0;
  // One coalesced load: lane j holds candidate j of chunk cc.  first: the query's own chunk, sorted instead of inserted.
  auto scan = [&](int cc, bool first) {
    const float4 p = sorted[cc * CHUNK + lane];
    const int raw = __float_as_int(p.w);
    const bool valid = raw >= 0;  // (< 0: padding of the last chunk)
    const int cidx = valid ? raw : 0x7fffffff;
    if constexpr (DBG) dbg_scans++;
#pragma unroll
    for (int i = 0; i < Q; i++) {
      if (!first) {
        // FP32 image of the distance (< 4 ulp off) against the inflated FP32 image of the bound: most chunks that pass the box test hold no
        // candidate for this query, and then the FP64 evaluation is skipped for the whole wavefront
        const float dx = qxf[i] - p.x, dy = qyf[i] - p.y, dz = qzf[i] - p.z;
        if (__ballot(valid & (fmaf(dz, dz, fmaf(dy, dy, dx * dx)) <= bd32[i])) == 0ull) continue;
      }
      if constexpr (DBG) dbg_exact++;
      double d = sqdist(qx[i], qy[i], qz[i], (double)p.x, (double)p.y, (double)p.z);
      d = valid ? d : inf;
      if (first) {
        int idx = cidx;
        sort64(d, idx, lane);
        ld[i] = d;
        li[i] = idx;
        refresh_bound(i);
        continue;
      }
      unsigned long long m = __ballot(before(d, cidx, bd[i], bi[i]));
      while (m) {
        const int j = (int)__builtin_ctzll(m);
        const double dc = lane_d(d, j);
        const int ic = __builtin_amdgcn_readlane(cidx, j);
        // entries that order before the candidate form a prefix of the sorted list: its length is the candidate's place
        const int pos = (int)__popcll(__ballot(lane < K && before(ld[i], li[i], dc, ic)));
        const int slo = from_below(__double2loint(ld[i])), shi = from_below(__double2hiint(ld[i])), si = from_below(li[i]);
        const double sd = __hiloint2double(shi, slo);
        ld[i] = lane > pos ? sd : (lane == pos ? dc : ld[i]);
        li[i] = lane > pos ? si : (lane == pos ? ic : li[i]);
        refresh_bound(i);
        if constexpr (DBG) dbg_inserts++;
        m &= m - 1ull;
        m &= __ballot(before(d, cidx, bd[i], bi[i]));  // the bound has tightened: candidates it now excludes are dropped without a visit
      }
    }
  };

  // can the box b[0..5] hold a point that enters some query's list?  FP32 gap, deflated beyond its own rounding (inputs are exact FP32 values,
  // three subtractions / squares / two sums: < 1e-6 relative) against the inflated FP32 image of the bound: it can only say yes too often.
  auto box_gaps = [&](const float* b, float* g2) {
#pragma unroll
    for (int i = 0; i < Q; i++) {
      const float gx = fmaxf(0.f, fmaxf(b[0] - qxf[i], qxf[i] - b[3]));
      const float gy = fmaxf(0.f, fmaxf(b[1] - qyf[i], qyf[i] - b[4]));
      const float gz = fmaxf(0.f, fmaxf(b[2] - qzf[i], qzf[i] - b[5]));
      g2[i] = (gx * gx + gy * gy + gz * gz) * 0.99999f;
    }
  };
  auto gaps_may_help = [&](const float* g2) -> bool {
    bool ok = false;
#pragma unroll
    for (int i = 0; i < Q; i++) ok = ok | (g2[i] <= bd32[i]);
    return ok;
  };

  scan(c, true);
  if (c > 0) scan(c - 1, false);
  if (c + 1 < C) scan(c + 1, false);

  // Groups of 64 chunks (their boxes sit behind the chunk boxes), visited from the query's own group outwards so that the bounds tighten early.
  const int G = (C + CHUNK - 1) / CHUNK, gc = c / CHUNK;
  unsigned long long gmask[4] = {~0ull, ~0ull, ~0ull, ~0ull};  // groups worth walking (all of them beyond 256 groups)
  if (G <= 256) {
#pragma unroll
    for (int wd = 0; wd < 4; wd++) {
      const int g = wd * 64 + lane;
      bool okg = g < G;
      if (okg) {
        const float* bp = box + 6 * (size_t)(C + g);
        const float b[6] = {bp[0], bp[1], bp[2], bp[3], bp[4], bp[5]};
        float g2[Q];
        box_gaps(b, g2);
        okg = gaps_may_help(g2);
      }
      gmask[wd] = __ballot(okg);
    }
  }
  auto visit = [&](int gi) {
    const int g0 = gi * CHUNK;
    const int cc_l = g0 + lane;  // lane l owns chunk g0 + l of this group
    const bool mine = cc_l < C && cc_l != c && cc_l != c - 1 && cc_l != c + 1;
    // gaps between the queries and this lane's chunk box: computed once, compared with the bounds -- which only shrink -- after every scan
    float g2[Q];
    {
      float b[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (mine) {
        const float* bp = box + 6 * (size_t)cc_l;
#pragma unroll
        for (int a = 0; a < 6; a++) b[a] = bp[a];
      }
      box_gaps(b, g2);
    }
    unsigned long long todo = __ballot(mine);
    while (todo) {
      if constexpr (DBG) dbg_tests++;
      todo &= __ballot(mine & gaps_may_help(g2));
      if (!todo) break;
      const int j = (int)__builtin_ctzll(todo);
      todo &= todo - 1ull;
      scan(g0 + j, false);
    }
  };
  if (G <= 256) {
    // the surviving groups (3 of 32 on a 131 072-pt scan), nearest to the query's own group first: two cursors over the set bits of gmask
    // (a loop over all 2 G positions costs more scalar instructions per wavefront than the scans it finds)
    auto next_up = [&](int p) -> int {  // smallest set bit >= p, or -1
      if (p >= 256) return -1;
#pragma unroll
      for (int w = 0; w < 4; w++) {
        if (w < (p >> 6)) continue;
        const unsigned long long m = gmask[w] & (w == (p >> 6) ? (~0ull << (p & 63)) : ~0ull);
        if (m) return w * 64 + (int)__builtin_ctzll(m);
      }
      return -1;
    };
    auto next_down = [&](int p) -> int {  // largest set bit <= p, or -1
      if (p < 0) return -1;
#pragma unroll
      for (int w = 3; w >= 0; w--) {
        if (w > (p >> 6)) continue;
        const unsigned long long m = gmask[w] & (w == (p >> 6) ? (~0ull >> (63 - (p & 63))) : ~0ull);
        if (m) return w * 64 + 63 - (int)__builtin_clzll(m);
      }
      return -1;
    };
    int up = next_up(gc), down = next_down(gc - 1);
    while (up >= 0 || down >= 0) {
      const bool take_up = down < 0 || (up >= 0 && up - gc <= gc - down);
      if (take_up) {
        visit(up);
        up = next_up(up + 1);
      } else {
        visit(down);
        down = next_down(down - 1);
      }
    }
  } else {
    for (int t = 0; t < 2 * G; t++) {
      const int gi = (t & 1) ? gc + ((t + 1) >> 1) : gc - (t >> 1);
      if (gi >= 0 && gi < G) visit(gi);
    }
  }
#pragma unroll
  for (int i = 0; i < Q; i++)
    if (self[i] >= 0 && lane < k && lane < K) out[(size_t)self[i] * k + lane] = li[i];
  if (DBG && dbg && lane == 0) {  // diag knn_debug: totals over the launch
    atomicAdd(dbg + 0, 1);
    atomicAdd(dbg + 1, dbg_scans);
    atomicAdd(dbg + 2, dbg_exact);
    atomicAdd(dbg + 3, dbg_inserts);
    atomicAdd(dbg + 4, dbg_tests);
  }
}

template <int K>
void launch_qgroup(hipStream_t st, int n, int C, const float4* sorted, const float* box, int k, int32_t* out, const int* guard, int q, int* dbg) {
  auto grid = [&](int per_chunk) { return (unsigned int)((((C * per_chunk + 3) / 4 + 7) / 8) * 8); };  // 4 query groups per workgroup, whole rounds of the 8 XCDs
  if (dbg) {  // the counting instantiation (diag knn_debug): two queries per wavefront
    knn_qgroup_kernel<K, 2, true><<<grid(CHUNK / 2), 256, 0, st>>>(n, C, sorted, box, k, out, guard, dbg);
    return;
  }
  if (q == 1) knn_qgroup_kernel<K, 1, false><<<grid(CHUNK), 256, 0, st>>>(n, C, sorted, box, k, out, guard, dbg);
  else knn_qgroup_kernel<K, 2, false><<<grid(CHUNK / 2), 256, 0, st>>>(n, C, sorted, box, k, out, guard, dbg);
}

}  // namespace

namespace glim_amd {

void knn_launch_qgroup(hipStream_t st, int n, int C, const float4* sorted, float* box, int k, int32_t* out, const int* guard, int queries_per_wave, int* dbg) {
  knn_launch_group_boxes(st, C, box);
  DISPATCH_K(launch_qgroup, st, n, C, sorted, box, k, out, guard, queries_per_wave, dbg);
}

}  // namespace glim_amd
