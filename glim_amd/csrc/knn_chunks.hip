// knn_chunks.hip -- kernel group K2, the 64-query Hilbert-chunk kernel (see knn.hip for the method and the host side).
#include "knn_common.hpp"

using namespace glim_amd;

namespace {

// Boxes of the groups of 64 chunks, stored behind the chunk boxes (box[6 * (C + g) ...]).  A query chunk tests the (<= 256) group boxes
// against its search radius after the first three scans and walks only the groups that pass -- 3.0 of 32 on average for a 131 072-pt scan --
// where the walk over all groups used to be most of the ~47 us every wavefront paid whatever it scanned.  A group box contains its chunk
// boxes and the radius only shrinks, so no chunk that the full walk would have scanned is lost.
__global__ __launch_bounds__(256) void group_box_kernel(int C, float* __restrict__ box /* [C + G][6] */) {
  const int g = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const int G = (C + CHUNK - 1) / CHUNK;
  if (g >= G) return;
  const int cc = g * CHUNK + lane;
  const float inf = __int_as_float(0x7f800000);
  float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
  if (cc < C) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      lo[a] = box[6 * (size_t)cc + a];
      hi[a] = box[6 * (size_t)cc + 3 + a];
    }
  }
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, 64));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off, 64));
    }
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      box[6 * (size_t)(C + g) + a] = lo[a];
      box[6 * (size_t)(C + g) + 3 + a] = hi[a];
    }
  }
}

// One wavefront answers the 64 queries of chunk c by streaming candidate chunks through LDS.
//
// Mask pass in packed FP32: the candidate coordinates lie x / y / z-planar in LDS, so two consecutive candidates load as one register pair
// and the 64 x 3 subtract / multiply / fma steps run as v_pk_* (two per instruction).  d32 = fl((qx - x)^2 + ...) differs from the exact
// squared distance by < 4 ulp-relative (3e-7), so "d32 <= thr * (1 + 2e-6)" can only ADD candidates, and every accepted candidate is
// re-evaluated with the oracle's FP64 expression before it is offered to the list: the lists stay bit-identical.  The host only takes this
// kernel when the cloud's extent keeps FP32 squared distances finite.
//
// SELECT (k <= 10): the lock-step insertion loop lasts as long as the lane with the most accepted candidates, and a lane whose bound is
// still loose accepts most of a chunk although only a handful end up in its list.  So the 64 FP32 distances stay in registers and every
// lane first finds, by bisection over FP32 bit patterns, a threshold t with
//     #(list entries with d <= t) + #(accepted candidates with d32 <= t) >= K;
// candidates with d32 > t * (1 + 2e-6) are dropped without being popped: once the others are inserted the list holds K entries whose exact
// distances are <= t * (1 + 3e-7), strictly below the exact distance of every dropped candidate (> t * (1 + 2e-6) * (1 - 3e-7)).
// Insertion rounds 116 -> 50 per wavefront, 395 -> 140 for the slowest (tools/knn_model.py emulates this code step by step on the CPU:
// tests/test_knn_model.py); 131 072-pt scan 0.46 -> 0.36 ms, 307 104-pt depth frame 0.72 -> 0.60 ms with the group boxes and the packed
// mask pass (profiles/r02/probe/knn_select_groupbox_ab.txt, BENCH_r02 `staged`).
//
// The launch lasts as long as its slowest wavefront (a sparse chunk next to a dense patch streams 2-3 x the average number of candidate
// chunks).  Measured in round 3 and NOT adopted: handing the chunks whose grown box meets the most chunk boxes to the pair-lane kernel (two
// wavefronts of half the queries each) -- exact at every threshold, but 0.54-0.64 ms against 0.34 ms for a 131 072-pt scan: the split
// machinery alone costs 0.05 ms and a pair-lane wavefront of a heavy chunk is not much shorter than the 64-query one, the lock-step steps
// last as long as their slowest lane either way (profiles/r03/probe/knn_heavy_split_*.txt); 10 instead of 13 Hilbert bits per axis (one sort
// pass less, worse chunks: 0.41 ms).
// Measured against this kernel and NOT adopted (profiles/r02/probe/knn_chunk_variants_ab.txt; all bit-identical): candidates read with
// v_readlane instead of LDS (-1 %); branch-free insertion (K compares + selects, or min / max on the distances) instead of the early-exit
// bubble, one or two candidates per round (+17 ... +28 %); loads issued one step ahead of their use (+4 %); an all-FP64 mask pass (+8 %).
template <int K, bool SELECT>
__global__ __launch_bounds__(256) void knn_chunk_kernel(int n, int C, const float4* __restrict__ sorted, const float* __restrict__ box, int k,
                                                        int32_t* __restrict__ out, int* __restrict__ dbg, const int* __restrict__ guard) {
  __shared__ __attribute__((aligned(8))) float s_px[4][CHUNK], s_py[4][CHUNK], s_pz[4][CHUNK];
  __shared__ int s_pi[4][CHUNK];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + w;
  if (c >= C) return;  // whole wavefront
  if (*guard != 0) return;  // the cloud's extent is not finite or beyond the FP32 mask's range (knn.hip: curve_key_kernel): the caller answers otherwise
  const float4 q4 = sorted[c * CHUNK + lane];
  const int self = __float_as_int(q4.w);
  const bool live = self >= 0;
  // padding lanes of the last chunk query the chunk's first point, so that they never widen the search
  const float4 q0 = sorted[c * CHUNK];
  const float qxf = live ? q4.x : q0.x, qyf = live ? q4.y : q0.y, qzf = live ? q4.z : q0.z;
  const double qx = (double)qxf, qy = (double)qyf, qz = (double)qzf;
  TopK<K> best;
  best.init(self);
  int dbg_tiles = 0, dbg_pops = 0;
  const long long dbg_t0 = dbg ? wall_clock64() : 0;

  // candidate j of the staged chunk: its index (< 0: padding) and its exact squared distance
  auto cand_idx = [&](int j) -> int { return s_pi[w][j]; };
  auto cand_dist = [&](int j) -> double { return sqdist(qx, qy, qz, (double)s_px[w][j], (double)s_py[w][j], (double)s_pz[w][j]); };
  // Stream chunk cc through LDS.  Two phases keep the hot loop free of branches: (1) every lane evaluates all 64 candidates and records
  // which ones pass its k-th best of the moment in a 64-bit mask (inclusive test, so ties are kept); (2) the lanes pop their masks
  // together -- the K-step insertion then runs max-popcount times per chunk instead of once per candidate, and for most chunks after the
  // first three nobody has anything to insert.
  auto scan_chunk = [&](int cc, bool need, bool seed) {
    const float4 p = sorted[cc * CHUNK + lane];
    dbg_tiles++;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    s_px[w][lane] = p.x;
    s_py[w][lane] = p.y;
    s_pz[w][lane] = p.z;
    s_pi[w][lane] = __float_as_int(p.w);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    unsigned long long seeded = 0ull;
    if (seed) {
      // own chunk: every lane first inserts its K + 2 nearest points ALONG THE CURVE (itself, then +-1, +-2, ...), which are mostly
      // its nearest in space too, so that the threshold of the mask pass below is already tight (otherwise all 64 candidates of
      // the first chunk pass an infinite threshold and cost one insertion round each)
#pragma unroll
      for (int t = 0; t < K + 2; t++) {
        const int off = (t & 1) ? ((t + 1) >> 1) : -(t >> 1);
        const int j = (lane + off) & (CHUNK - 1);
        seeded |= 1ull << j;
        const int idx = cand_idx(j);
        if (idx >= 0) best.push(cand_dist(j), idx);
      }
    }
    const double thr = need ? best.d[K - 1] : -1.0;  // lanes that do not need this chunk accept nothing
    // FP32 image of the bound, inflated beyond the FP32 evaluation error; -1 stays negative, +inf stays +inf
    const float thr32 = (float)(thr * 1.000002) + 1e-37f;
    unsigned int mlo = 0u, mhi = 0u;
    if constexpr (SELECT) {
      float dv[CHUNK];
#pragma unroll
      for (int j = 0; j < CHUNK; j += 2) {
        const v2f dx = v2f{qxf, qxf} - *reinterpret_cast<const v2f*>(&s_px[w][j]), dy = v2f{qyf, qyf} - *reinterpret_cast<const v2f*>(&s_py[w][j]),
                  dz = v2f{qzf, qzf} - *reinterpret_cast<const v2f*>(&s_pz[w][j]);
        const v2f d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
        dv[j] = d.x;
        dv[j + 1] = d.y;
      }
#pragma unroll
      for (int j = 0; j < 32; j++) mlo |= (dv[j] <= thr32 ? 1u : 0u) << j;
#pragma unroll
      for (int j = 0; j < 32; j++) mhi |= (dv[32 + j] <= thr32 ? 1u : 0u) << j;
      const unsigned long long m0 = (((unsigned long long)mhi << 32) | mlo) & ~seeded;
      constexpr int SELECT_MIN = 12;  // below this many accepted candidates in every lane the selection costs more than it saves (model)
      if (__any(__popcll(m0) > SELECT_MIN)) {
        const float inf32 = __int_as_float(0x7f800000);
        // only candidates that are up for insertion count (rejected at thr, seeded and padding candidates become +inf)
#pragma unroll
        for (int j = 0; j < CHUNK; j++) dv[j] = ((m0 >> j) & 1ull) ? dv[j] : inf32;
        auto count_le = [&](float t) -> int {
          int cnt = 0;
#pragma unroll
          for (int j = 0; j < CHUNK; j++) cnt += dv[j] <= t ? 1 : 0;
#pragma unroll
          for (int j = 0; j < K; j++) cnt += best.d[j] <= (double)t ? 1 : 0;
          return cnt;
        };
        unsigned int hi = __float_as_uint(fminf(thr32, 3.4028234e38f));  // a list that is not full yet searches below FLT_MAX
        const bool sel = need && count_le(__uint_as_float(hi)) >= K;
        unsigned int lo = hi > (16u << 23) ? hi - (16u << 23) : 0u;  // 16 octaves below the bound; invariant: count_le(hi) >= K
        for (int step = 0; step < 8; step++) {                        // 2^27 bit patterns -> 2^19: t within 6 % of the smallest valid one
          const unsigned int mid = lo + ((hi - lo) >> 1);
          const bool ok = count_le(__uint_as_float(mid)) >= K;
          hi = ok ? mid : hi;
          lo = ok ? lo : mid + 1u;
        }
        const float keep = __uint_as_float(hi) * 1.000002f + 1e-37f;
        unsigned int klo = 0u, khi = 0u;
#pragma unroll
        for (int j = 0; j < 32; j++) klo |= (dv[j] <= keep ? 1u : 0u) << j;
#pragma unroll
        for (int j = 0; j < 32; j++) khi |= (dv[32 + j] <= keep ? 1u : 0u) << j;
        if (sel) {  // seeded / rejected / padding candidates are +inf here; the AND keeps them out even if `keep` overflowed to +inf
          mlo = klo & (unsigned int)m0;
          mhi = khi & (unsigned int)(m0 >> 32);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < CHUNK; j += 2) {
        const v2f dx = v2f{qxf, qxf} - *reinterpret_cast<const v2f*>(&s_px[w][j]), dy = v2f{qyf, qyf} - *reinterpret_cast<const v2f*>(&s_py[w][j]),
                  dz = v2f{qzf, qzf} - *reinterpret_cast<const v2f*>(&s_pz[w][j]);
        const v2f d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
        const unsigned int two = (d.x <= thr32 ? 1u : 0u) | (d.y <= thr32 ? 2u : 0u);
        if (j < 32) mlo |= two << j;
        else mhi |= two << (j - 32);
      }
    }
    unsigned long long m = (((unsigned long long)mhi << 32) | mlo) & ~seeded;
    while (__any(m != 0ull)) {
      dbg_pops++;
      if (m != 0ull) {
        const int j = (int)__builtin_ctzll(m);
        m &= m - 1ull;
        const int idx = cand_idx(j);
        // a candidate the inflated FP32 bound let through is settled by the exact (distance, index) test of push
        if (idx >= 0) best.push(cand_dist(j), idx);
      }
    }
  };

  scan_chunk(c, true, true);
  if (c > 0) scan_chunk(c - 1, true, false);
  if (c + 1 < C) scan_chunk(c + 1, true, false);

  const float qlo[3] = {box[6 * c], box[6 * c + 1], box[6 * c + 2]}, qhi[3] = {box[6 * c + 3], box[6 * c + 4], box[6 * c + 5]};
  // Groups of 64 chunks are visited from the query chunk's own group outwards (alternating sides): chunks that are close on the
  // curve are mostly close in space, so the k-th best distances tighten early and prune what comes later.  (In plain index order a
  // sparse query next to a dense patch can meet ever closer tiles and insert all 64 candidates of each: one such wavefront took
  // 985 us against a mean of 170 us.)
  const int G = (C + CHUNK - 1) / CHUNK, gc = c / CHUNK;
  unsigned long long gmask[4] = {~0ull, ~0ull, ~0ull, ~0ull};  // groups worth walking (all of them beyond 256 groups)
  if (G <= 256) {
    double r20 = best.d[K - 1];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) r20 = fmax(r20, __shfl_xor(r20, off, 64));
    const float R0 = (float)(sqrt(r20) * 1.000001) + 1e-30f;
#pragma unroll
    for (int wd = 0; wd < 4; wd++) {
      const int g = wd * 64 + lane;
      bool okg = g < G;
      if (okg) {
        const float* b = box + 6 * (size_t)(C + g);
#pragma unroll
        for (int a = 0; a < 3; a++) okg = okg && (b[a] <= qhi[a] + R0) && (b[3 + a] >= qlo[a] - R0);
      }
      gmask[wd] = __ballot(okg);
    }
  }
  for (int t = 0; t < 2 * G; t++) {
    const int gi = (t & 1) ? gc + ((t + 1) >> 1) : gc - (t >> 1);
    if (gi < 0 || gi >= G) continue;
    if (G <= 256 && !((gmask[gi >> 6] >> (gi & 63)) & 1ull)) continue;
    const int g0 = gi * CHUNK;
    // wave-wide search radius: the largest k-th best distance among the lanes (+inf while some list is not full)
    double r2 = best.d[K - 1];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) r2 = fmax(r2, __shfl_xor(r2, off, 64));
    const float R = (float)(sqrt(r2) * 1.000001) + 1e-30f;
    // lane l tests chunk g0 + l against the query chunk's box grown by R
    const int cc_l = g0 + lane;
    bool cand = cc_l < C && cc_l != c && cc_l != c - 1 && cc_l != c + 1;
    if (cand) {
      const float* b = box + 6 * (size_t)cc_l;
#pragma unroll
      for (int a = 0; a < 3; a++) cand = cand && (b[a] <= qhi[a] + R) && (b[3 + a] >= qlo[a] - R);
    }
    unsigned long long mask = __ballot(cand);
    while (mask) {
      const int cc = g0 + (int)__builtin_ctzll(mask);
      mask &= mask - 1;
      // per-lane test: gap between the query and the chunk's box against the lane's own k-th best (strictly farther => skip)
      const float* b = box + 6 * (size_t)cc;
      const double gx = fmax(0.0, fmax((double)b[0] - qx, qx - (double)b[3]));
      const double gy = fmax(0.0, fmax((double)b[1] - qy, qy - (double)b[4]));
      const double gz = fmax(0.0, fmax((double)b[2] - qz, qz - (double)b[5]));
      const bool need = (gx * gx + gy * gy + gz * gz) * (1.0 - 1e-12) <= best.d[K - 1];
      if (__ballot(need) == 0ull) continue;
      scan_chunk(cc, need, false);
    }
  }
  if (dbg && lane == 0) {
    dbg[4 * c + 0] = dbg_tiles;
    dbg[4 * c + 1] = dbg_pops;
    dbg[4 * c + 2] = (int)(wall_clock64() - dbg_t0);
    dbg[4 * c + 3] = (int)(1000.f * fmaxf(fmaxf(qhi[0] - qlo[0], qhi[1] - qlo[1]), qhi[2] - qlo[2]));  // chunk box extent, mm
  }
  if (live) {
#pragma unroll
    for (int j = 0; j < K; j++)
      if (j < k) out[(size_t)self * k + j] = best.idx[j];
  }
}

template <int K>
void launch_chunks(hipStream_t st, int n, int C, const float4* sorted, const float* box, int k, int32_t* out, int* dbg, bool select, const int* guard) {
  if constexpr (K <= 10) {  // the selection keeps 64 distances in registers next to the list: beyond k = 10 it spills
    if (select) {
      knn_chunk_kernel<K, true><<<(C + 3) / 4, 256, 0, st>>>(n, C, sorted, box, k, out, dbg, guard);
      return;
    }
  }
  knn_chunk_kernel<K, false><<<(C + 3) / 4, 256, 0, st>>>(n, C, sorted, box, k, out, dbg, guard);
}

}  // namespace

namespace glim_amd {

void knn_launch_group_boxes(hipStream_t st, int C, float* box) { group_box_kernel<<<((C + CHUNK - 1) / CHUNK + 3) / 4, 256, 0, st>>>(C, box); }

void knn_launch_chunks(hipStream_t st, int n, int C, const float4* sorted, float* box, int k, int32_t* out, int* dbg, bool select, const int* guard) {
  knn_launch_group_boxes(st, C, box);
  DISPATCH_K(launch_chunks, st, n, C, sorted, box, k, out, dbg, select, guard);
}

}  // namespace glim_amd
