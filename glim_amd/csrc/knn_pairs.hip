// knn_pairs.hip -- kernel group K2, the pair-lane Hilbert-chunk kernel (see knn.hip for the method and the host side).
#include "knn_common.hpp"

using namespace glim_amd;

namespace {

// ---- pair-lane variant (clouds up to 98 304 points): 32 queries per wavefront, two lanes per query ----------------------------------------
// The 64-query kernel (knn_chunks.hip) leaves a 65 536-point keyframe with 1024 wavefronts -- one per SIMD -- each a serial chain of tens of
// thousands of instructions, i.e. latency-bound with nothing to overlap.  Here a wavefront answers the 32 queries of a 32-point chunk and lanes l and l + 32 serve the
// SAME query: every step streams TWO candidate chunks through LDS, the lower half of the wavefront scans the first, the upper half the second,
// so a (query, candidate) pair is still evaluated exactly once but there are twice as many wavefronts of half the length.  The two lanes of a
// query keep separate top-k lists over disjoint candidate sets and share their pruning threshold: the k-th best of the union is at most the
// smaller of the two lists' k-th bests, so min(d_k(l), d_k(l + 32)) is a valid -- and after the first step tight -- acceptance bound for both.
// At the end the two sorted lists are merged by rank (position of an entry = its own index + the number of entries of the partner list that
// order before it), exchanged with shuffles.
// The mask pass runs in FP32: d32 = fl((qx - x)^2 + ...) differs from the exact squared distance by < 4 ulp-relative (3e-7), so
// "d32 <= thr * (1 + 2e-6)" can only ADD false candidates; every accepted candidate is re-evaluated in FP64 with the oracle's expression
// before it is offered to the list, which applies the exact (distance, index) test.  Results are bit-identical to the other implementations.
template <int K, bool SELECT>
__global__ __launch_bounds__(256) void knn_pair_kernel(int n, int C /* 32-point chunks */, const float4* __restrict__ sorted, const float* __restrict__ box,
                                                       int k, int32_t* __restrict__ out, const int* __restrict__ guard) {
  __shared__ float4 s_pt[4][64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, ql = lane & 31;
  const int c = blockIdx.x * 4 + w;
  if (c >= C) return;  // whole wavefront
  if (*guard != 0) return;  // see knn_chunk_kernel
  const float4 q4 = sorted[c * QCH + ql];
  const int self = __float_as_int(q4.w);
  const bool live = self >= 0;
  const float4 q0 = sorted[c * QCH];  // padding lanes of the last chunk query the chunk's first point: they never widen the search
  const float qxf = live ? q4.x : q0.x, qyf = live ? q4.y : q0.y, qzf = live ? q4.z : q0.z;
  const double qx = (double)qxf, qy = (double)qyf, qz = (double)qzf;
  const float inf = __int_as_float(0x7f800000);
  TopK<K> best;
  best.init(self);

  // my_cc: the candidate chunk of this lane's half (-1: none); need: this lane wants it; seed: first step, lower half, own chunk
  auto scan_pair = [&](int my_cc, bool need, bool seed) {
    float4 p = make_float4(inf, inf, inf, __int_as_float(-1));
    if (my_cc >= 0) p = sorted[my_cc * QCH + ql];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    s_pt[w][lane] = p;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int base = half * QCH;
    unsigned int seeded = 0u;
    if (seed) {
      // own chunk: the lower lane of every query first inserts the points nearest ALONG THE CURVE (itself, +-1, +-2, ...), mostly its nearest in
      // space too, so that the first mask pass already has a tight threshold -- for both lanes of the query, through the shared bound below
      constexpr int NSEED = (K + 2 < QCH) ? K + 2 : QCH;
#pragma unroll
      for (int t = 0; t < NSEED; t++) {
        const int off = (t & 1) ? ((t + 1) >> 1) : -(t >> 1);
        const int j = (ql + off) & (QCH - 1);
        if (half == 0) {
          seeded |= 1u << j;
          const float4 sp = s_pt[w][j];
          const int idx = __float_as_int(sp.w);
          if (idx >= 0) best.push(sqdist(qx, qy, qz, (double)sp.x, (double)sp.y, (double)sp.z), idx);
        } else {
          // keep the wavefront converged through TopK::push (it uses wave-wide votes): offer nothing
          best.push(__longlong_as_double(0x7ff0000000000000ll), 0x7fffffff);
        }
      }
    }
    const double mine = best.d[K - 1];
    const double shared_thr = fmin(mine, __shfl_xor(mine, 32, 64));
    const double thr = need ? shared_thr : -1.0;  // lanes that do not need their chunk accept nothing
    // FP32 image of the bound, inflated beyond the FP32 evaluation error (relative 2e-6, plus an absolute 1e-37 that covers the denormal range);
    // -1 stays negative, +inf stays +inf
    const float thr32 = (float)(thr * 1.000002) + 1e-37f;
    unsigned int m = 0u;
    if constexpr (SELECT) {
    // the per-lane threshold selection of knn_chunk_kernel (see there), over this lane's 32 candidates and ITS OWN list: a candidate that cannot
    // enter the lane's list cannot be among the K best of the two lists merged at the end either
    float dv[QCH];
#pragma unroll
    for (int j = 0; j < QCH; j++) {
      const float4 cp = s_pt[w][base + j];
      const float dx = qxf - cp.x, dy = qyf - cp.y, dz = qzf - cp.z;
      dv[j] = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
      m |= (dv[j] <= thr32 ? 1u : 0u) << j;
    }
    m &= ~seeded;
    {
      constexpr int SELECT_MIN = 12;
      if (__any(__popc(m) > SELECT_MIN)) {
        const float inf32 = __int_as_float(0x7f800000);
#pragma unroll
        for (int j = 0; j < QCH; j++) dv[j] = ((m >> j) & 1u) ? dv[j] : inf32;
        auto count_le = [&](float t) -> int {
          int cnt = 0;
#pragma unroll
          for (int j = 0; j < QCH; j++) cnt += dv[j] <= t ? 1 : 0;
#pragma unroll
          for (int j = 0; j < K; j++) cnt += best.d[j] <= (double)t ? 1 : 0;
          return cnt;
        };
        unsigned int hi = __float_as_uint(fminf(thr32, 3.4028234e38f));
        const bool sel = need && count_le(__uint_as_float(hi)) >= K;
        unsigned int lo = hi > (16u << 23) ? hi - (16u << 23) : 0u;
        for (int step = 0; step < 8; step++) {
          const unsigned int mid = lo + ((hi - lo) >> 1);
          const bool ok = count_le(__uint_as_float(mid)) >= K;
          hi = ok ? mid : hi;
          lo = ok ? lo : mid + 1u;
        }
        const float keep = __uint_as_float(hi) * 1.000002f + 1e-37f;
        unsigned int km = 0u;
#pragma unroll
        for (int j = 0; j < QCH; j++) km |= (dv[j] <= keep ? 1u : 0u) << j;
        if (sel) m &= km;
      }
    }
    } else {
#pragma unroll
    for (int j = 0; j < QCH; j++) {
      const float4 cp = s_pt[w][base + j];
      const float dx = qxf - cp.x, dy = qyf - cp.y, dz = qzf - cp.z;
      const float d32 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
      m |= (d32 <= thr32 ? 1u : 0u) << j;
    }
    m &= ~seeded;
    }
    while (__any(m != 0u)) {
      double d = __longlong_as_double(0x7ff0000000000000ll);
      int idx = 0x7fffffff;
      if (m != 0u) {
        const int j = __builtin_ctz(m);
        m &= m - 1u;
        const float4 cp = s_pt[w][base + j];
        const int ci = __float_as_int(cp.w);
        if (ci >= 0) {
          d = sqdist(qx, qy, qz, (double)cp.x, (double)cp.y, (double)cp.z);
          idx = ci;
        }
      }
      best.push(d, idx);  // (+inf, INT_MAX) never enters a list: it loses the (distance, index) test against every entry incl. the placeholders
    }
  };

  // first the query chunk itself and its curve neighbours
  scan_pair(half == 0 ? c : (c + 1 < C ? c + 1 : -1), true, true);
  scan_pair(half == 0 ? (c > 0 ? c - 1 : -1) : (c + 2 < C ? c + 2 : -1), true, false);

  const float qlo[3] = {box[6 * c], box[6 * c + 1], box[6 * c + 2]}, qhi[3] = {box[6 * c + 3], box[6 * c + 4], box[6 * c + 5]};
  // groups of 64 chunks, from the query chunk's own group outwards (alternating sides): near on the curve is mostly near in space, so the
  // bounds tighten early and prune what comes later
  const int G = (C + 63) / 64, gc = c / 64;
  for (int t = 0; t < 2 * G; t++) {
    const int gi = (t & 1) ? gc + ((t + 1) >> 1) : gc - (t >> 1);
    if (gi < 0 || gi >= G) continue;
    const int g0 = gi * 64;
    // wave-wide search radius: the largest shared bound among the queries (+inf while some query has fewer than K candidates)
    const double mine = best.d[K - 1];
    const double pm = fmin(mine, __shfl_xor(mine, 32, 64));
    double r2 = pm;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) r2 = fmax(r2, __shfl_xor(r2, off, 64));
    const float R = (float)(sqrt(r2) * 1.000001) + 1e-30f;
    const int cc_l = g0 + lane;  // lane l tests chunk g0 + l against the query chunk's box grown by R
    bool cand = cc_l < C && (cc_l < c - 1 || cc_l > c + 2);
    if (cand) {
      const float* b = box + 6 * (size_t)cc_l;
#pragma unroll
      for (int a = 0; a < 3; a++) cand = cand && (b[a] <= qhi[a] + R) && (b[3 + a] >= qlo[a] - R);
    }
    unsigned long long mask = __ballot(cand);
    while (mask) {
      const int cc_a = g0 + (int)__builtin_ctzll(mask);
      mask &= mask - 1;
      int cc_b = -1;
      if (mask) {
        cc_b = g0 + (int)__builtin_ctzll(mask);
        mask &= mask - 1;
      }
      const int my_cc = half == 0 ? cc_a : cc_b;
      bool need = false;
      if (my_cc >= 0) {
        // per-lane test: gap between the query and the chunk's box against the query's shared bound (strictly farther => skip)
        const float* b = box + 6 * (size_t)my_cc;
        const double gx = fmax(0.0, fmax((double)b[0] - qx, qx - (double)b[3]));
        const double gy = fmax(0.0, fmax((double)b[1] - qy, qy - (double)b[4]));
        const double gz = fmax(0.0, fmax((double)b[2] - qz, qz - (double)b[5]));
        const double cur = fmin(best.d[K - 1], pm);  // own list's current bound and the query's shared bound at the start of this group
        need = (gx * gx + gy * gy + gz * gz) * (1.0 - 1e-12) <= cur;
      }
      if (__ballot(need) == 0ull) continue;
      scan_pair(my_cc, need, false);
    }
  }

  // merge the two lists of every query by rank; both lanes write their own entries
  int rank[K];
#pragma unroll
  for (int i = 0; i < K; i++) rank[i] = i;
#pragma unroll
  for (int j = 0; j < K; j++) {
    const double od = __shfl_xor(best.d[j], 32, 64);
    const int oi = __shfl_xor(best.idx[j], 32, 64);
#pragma unroll
    for (int i = 0; i < K; i++) rank[i] += (od < best.d[i] || (od == best.d[i] && oi < best.idx[i])) ? 1 : 0;
  }
  if (live) {
#pragma unroll
    for (int i = 0; i < K; i++)
      if (rank[i] < k) out[(size_t)self * k + rank[i]] = best.idx[i];
  }
}

template <int K>
void launch_pairs(hipStream_t st, int n, int C32, const float4* sorted, const float* box32, int k, int32_t* out, bool select, const int* guard) {
  if constexpr (K <= 10) {
    if (select) {
      knn_pair_kernel<K, true><<<(C32 + 3) / 4, 256, 0, st>>>(n, C32, sorted, box32, k, out, guard);
      return;
    }
  }
  knn_pair_kernel<K, false><<<(C32 + 3) / 4, 256, 0, st>>>(n, C32, sorted, box32, k, out, guard);
}

}  // namespace

namespace glim_amd {

void knn_launch_pairs(hipStream_t st, int n, int C32, const float4* sorted, const float* box32, int k, int32_t* out, bool select, const int* guard) {
  DISPATCH_K16(launch_pairs, st, n, C32, sorted, box32, k, out, select, guard);
}

}  // namespace glim_amd
