// vgicp.hip -- the VGICP matching-cost factor on gfx950 (kernels K4 linearise, K5 error, K6 overlap) and the factor-set
// (NonlinearFactorSetGPU) entry points of the C ABI.
//
// Replaces gtsam_points::IntegratedVGICPFactorGPU::{linearize,error} + NonlinearFactorSetGPU::linearize + overlap_gpu as
// called from src/glim/odometry/odometry_estimation_gpu.cpp:144,161,231,248,383-386, src/glim/mapping/sub_mapping.cpp:252,307
// and src/glim/mapping/global_mapping.cpp:322,335,448,466,860.  The arithmetic follows the CPU factor
// (gtsam_points::IntegratedVGICPFactor, the parity oracle: SURVEY.md App. B.5):
//
//   q = R p + t                      FP64, fixed fma order (bit-exact voxel coordinates / correspondences)
//   voxel = table[floor(q / res)]    exact 64-bit key compare, two-way 128-byte buckets: one line per lookup
//   M = (C_B + R C_A R^T)^-1,  r = mu_B - q,  e = r^T M r
//   H_ss += J_s^T M J_s,  b_s += J_s^T M r,   J_s = [R hat(p) | -R]
//
// Two algebraic restructurings keep the per-point work small (DESIGN.md "K4"):
//   (1) the rotation is factored out of the Jacobian: J_s = [R hat(p) | -R] = [hat(q') | -I] diag(R, R) with q' = R p = q - t, so the
//       kernel accumulates H' = sum J'^T M J' = [[-Q M Q, Q M], [(Q M)^T, M]] (Q = hat(q')) and b' = [u x q'; -u], u = M r, in the
//       target frame -- 21 + 6 + 1 sums per point -- and the finalise step applies diag(R, R) once per factor in FP64.  For
//       plane-form clouds R C_A R^T = I - (1 - 1e-3) m m^T with m = R n, so M needs no 3x3 sandwich at all.
//   (2) the target-side blocks of a BINARY factor are never accumulated per point: J_t = -J_s Ad(delta^-1) exactly, hence
//       H_tt = Ad^T H_ss Ad, H_ts = -Ad^T H_ss, b_t = -Ad^T b_s are recovered in FP64 from the 6x6 source block
//       (glim_amd_expand_compact).  A binary factor costs the same 28 accumulators as a unary one instead of 122.
//
// Reduction: per-thread FP32 accumulators over up to `points_per_thread` points -> 64-lane DPP wave sum -> LDS across the
// 4 waves of a block -> one 32-float partial row per block -> FP64 fixed-order sum per factor (finalise kernel).
// The result is bit-reproducible run to run (no floating-point atomics anywhere).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <thread>
#include <unordered_map>

#include "device_math.hpp"
#include "internal.hpp"

using namespace glim_amd;

// FP contraction as WRITTEN: a * b + c inside one expression becomes a fused multiply-add, nothing is fused across statements.  hipcc's default
// (`fast`) lets the backend fuse across statements wherever it sees fit, and what it sees depends on the code AROUND an inlined function: the same
// compute_row() inlined into the launch-per-call kernel and into the resident kernel then rounded a handful of products differently (8e-6
// relative on H of a general-form factor) -- and every form of the synchronous call has to return the same bits
// (tests/test_gpu_edge_cases.py).  The few cross-statement fusions the default found in the hot loop are written out as fmaf() below.
#pragma clang fp contract(on)

namespace {

constexpr int BLOCK = 256;
constexpr int NACC = 28;  // FP32 accumulators per thread (see layout below); slot 28 of a partial row = inlier count (int bits)

// accumulator layout:  0..5  Hww (00 01 02 11 12 22)   6..14 G = hat(p) A (row-major 3x3 = H_wv)   15..20 A (00 01 02 11 12 22)
//                      21..23 u x p (= b_w)            24..26 u (b_v = -u)                          27 e
// compact record:      [count, error, 21 upper-triangular H_ss entries row-major, 6 b_s]
// A/B knobs of round 6's work on the synchronous call (tools/ab_variant.sh <name> -DGLIM_AMD_ROT27=0 ...): one thread per rotated value in the
// finalisers (rotate_element) / the four-thread form; the resident workers skip the probe of a point no lane has (TAILSKIP) / run it
#ifndef GLIM_AMD_ROT27
#define GLIM_AMD_ROT27 1
#endif
// the 28 wavefront sums of a row step-major (device_math.hpp wave_sums_to_lane63) / value by value (rounds 1-5)
#ifndef GLIM_AMD_STEP_MAJOR_SUMS
#define GLIM_AMD_STEP_MAJOR_SUMS 1
#endif
#ifndef GLIM_AMD_RES_TAILSKIP
#define GLIM_AMD_RES_TAILSKIP 1
#endif
// the four 16-lane ROW sums of every wavefront go to LDS and whoever forms the row's values adds them as (r3 + r2) + (r1 + r0), what the two
// row_bcast steps left in lane 63 (same bits; 4 instead of 10 instructions per value and wavefront) / 0: the wavefront sums as before
#ifndef GLIM_AMD_ROW_SUMS_LDS
#define GLIM_AMD_ROW_SUMS_LDS 1
#endif
#if GLIM_AMD_ROW_SUMS_LDS
constexpr int RED_ROWS = 4;  // s_red[wavefront][16-lane row][value]
#else
constexpr int RED_ROWS = 1;  // s_red[wavefront][0][value]
#endif
typedef float RedRow[RED_ROWS][glim_amd::PARTIAL_STRIDE];
// accumulator slot of entry i of the upper triangle of H_ss -- {0, 1, 2, 6, 7, 8, 3, 4, 9, 10, 11, 5, 12, 13, 14, 15, 16, 17, 18, 19, 20} -- as two
// immediates (5 bits per entry, entries 0..11 and 12..20): the finalisers' last step looks its slot up with two shifts instead of a load from
// constant memory behind the last barrier of a call the host is waiting for
__device__ __forceinline__ int acc_of_upper(int i) {
  const unsigned long long lo = 0x2ad4920d0730820ull, hi = 0x149ca307b9acull;
  return (int)(((i < 12 ? lo >> (5 * i) : hi >> (5 * (i - 12)))) & 31ull);
}

__device__ __forceinline__ float4 ld16(const void* p) { return *reinterpret_cast<const float4*>(p); }

enum { MODE_LINEARIZE = 0, MODE_ERROR = 1 };

// Global-address-space views: the pointers come out of a descriptor loaded from memory, so without the explicit address
// space hipcc emits FLAT loads (which tick both vmcnt and lgkmcnt and force vmcnt(0) waits); with it, global_load + counted waits.
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef float v3f_t __attribute__((ext_vector_type(3)));
typedef const __attribute__((address_space(1))) v4f_t* gf4_t;
typedef const __attribute__((address_space(1))) v3f_t* gf3_t;
typedef const __attribute__((address_space(1))) v2f_t* gf2_t;
__device__ __forceinline__ float4 gld4(const void* p) {
  const v4f_t v = *reinterpret_cast<gf4_t>(reinterpret_cast<uintptr_t>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
// xyz of a float4 record as ONE 12-byte load: no dead 4th destination register (a dead register gets recycled by the
// allocator while the load is still in flight, and the write-after-write hazard then stalls the wave on that load)
__device__ __forceinline__ float4 gld3(const void* p) {
  const v3f_t v = *reinterpret_cast<gf3_t>(reinterpret_cast<uintptr_t>(p));
  return make_float4(v.x, v.y, v.z, 1.f);
}
__device__ __forceinline__ float2 gld2(const void* p) {
  const v2f_t v = *reinterpret_cast<gf2_t>(reinterpret_cast<uintptr_t>(p));
  return make_float2(v.x, v.y);
}
__device__ __forceinline__ float gld1(const void* p) {
  return *reinterpret_cast<const __attribute__((address_space(1))) float*>(reinterpret_cast<uintptr_t>(p));
}

struct PointIn {
  float4 p;   // PLANE: x y z nx        general: x y z c00
  float4 ca;  // PLANE: unused          general: c01 c02 c11 c12
  float2 cb;  // PLANE: ny nz           general: c22 -
};

// Factor streams (glim_amd_cloud, internal.hpp), both written in the Hilbert order of the cloud.
// PLANE = the source cloud's covariances are the PLANE-regularised form C = I - (1 - 1e-3) n n^T (the only form GLIM's
// CloudCovarianceEstimation emits, cloud_covariance_estimation.cpp:20, :181-196): the factor streams 24 B per point (xyz + unit normal)
// and rebuilds C in registers; any other cloud streams 36 B per point (xyz + the six covariance coefficients).
template <bool PLANE>
__device__ __forceinline__ PointIn load_point(const FactorDesc& d, unsigned int i) {
  PointIn r;
  r.p = gld4(reinterpret_cast<const char*>(d.s0) + i * 16u);  // uniform base + 32-bit lane offset (n <= 2^28)
  if (PLANE) {
    r.cb = gld2(reinterpret_cast<const char*>(d.s1) + i * 8u);
    r.ca = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    r.ca = gld4(reinterpret_cast<const char*>(d.s1) + i * 16u);
    r.cb = make_float2(gld1(reinterpret_cast<const char*>(d.s2) + i * 4u), 0.f);
  }
  return r;
}

// Arguments of the per-factor finalisation (finalize_kernel).  In-kernel finalisation by the last block of a factor was measured slower
// twice (per-block agent-scope release: 122 vs 80 us per 64 factors; write-through row stores: 149 vs 143 us per 128 factors, 22.9 vs
// 18.8 us per synchronous single-factor call) and is gone: the second dispatch costs less than any in-kernel hand-off.
struct FinalizeArgs {
  double* out;               // compact records
  double* out_mirror;        // the same records once more, same row offset: host-mapped memory of an asynchronous caller (glim_amd_multi), or null
  long long out_row_offset;
  int* done_counter;         // finished factors of this launch (polling fast path)
  unsigned int* host_flag;   // host-mapped completion word or null
  unsigned int seq;
  int num_factors;
  // single-dispatch form (vgicp_kernel<..., FUSED>): tagged partial rows and, per launch segment, the factors its trailing blocks finalise
  char* rows16;              // TAG_ROW_BYTES per plan row
  unsigned long long* trip_stats;  // 64 counters: skipped wavefront trips of the factors f with f % 64 == slot (null: not collected)
  char* rec16;               // host-mapped record granules (COMPACT x 16 B per factor) of the single-dispatch form, or null
  const int* finmap;         // factor ids, plane-form segment first (null for a single-factor set: factor 0)
  // pre-cull (vgicp_kernel<..., CULL>): 4 words per row of the general segment, written by cull_kernel on the same stream just before the launch
  const unsigned long long* cull_words;
};

// ---- tagged rows: the hand-off of the single-dispatch synchronous call --------------------------------------------------------------
// A block's partial row travels to the block that finalises its factor INSIDE the launch as ten self-validating 16-byte granules
// {v[3p], v[3p+1], v[3p+2], tag}, p = 0..9, tag = the call's sequence number: ONE write-through (sc1) dwordx4 store per granule -- it
// leaves the writing XCD's L2 at once and is performed as a unit -- and sc1 loads on the reading side, which are served past the reader's L1
// and its XCD's L2 copy (MI355X_MICROARCH.md "inter-workgroup visibility": the granule form needs no fence, no flag and no counter).  A
// granule whose tag is not this call's has not arrived yet; the reader just looks again.  No atomics: the three in-kernel finalisations of
// round 3 serialised ~500 arrivals of a chip-wide factor on one counter (~10 ns per same-address atomic = 5 us).
constexpr int TAG_PIECES = 10;
constexpr int TAG_ROW_BYTES = TAG_PIECES * 16;
constexpr unsigned int AUX_SYSTEM = 17u;                 // sc0 sc1: system scope (host-mapped memory: through to the host at once)
constexpr unsigned int AUX_SC1 = 16u;                    // cache policy of the raw buffer intrinsics: agent-coherent (write-through / L2-bypass)
constexpr unsigned int AUX_SC1_VOLATILE = 16u | (1u << 31);  // ... and not to be hoisted out of / merged across the polling loop
typedef int v4i_t __attribute__((ext_vector_type(4)));

// The last step of a linearisation, shared by the device finalise kernel and by the host-finalised single-factor call so that both give
// the same bits: the kernel accumulated H' = sum J'^T M J' and b' = sum J'^T M r for J' = [hat(R p) | -I] in the target frame; with
// J_s = J' diag(R, R):  H_ss = diag(R, R)^T H' diag(R, R), b_s = diag(R, R)^T b', i.e. every 3x3 block B' becomes R^T B' R and every
// 3-vector R^T v.  No fma contraction (host and device compilers would contract differently).
__host__ __device__ inline void rotate_block(const double* B, const double* R, double* O) {
#pragma clang fp contract(off)
  double BR[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) BR[3 * r + c] = B[3 * r] * R[c] + B[3 * r + 1] * R[3 + c] + B[3 * r + 2] * R[6 + c];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) O[3 * r + c] = R[r] * BR[c] + R[3 + r] * BR[3 + c] + R[6 + r] * BR[6 + c];  // (R^T BR)[r][c]
}
// part 0..2: rotate block Hww / Hwv / Hvv of the 32 summed accumulators `sum` into `rot` (accumulator layout); part 3: the two vectors
__host__ __device__ inline void rotate_part(int part, const double* sum, const double* T, double* rot) {
#pragma clang fp contract(off)
  double R[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) R[3 * r + c] = T[4 * r + c];
  if (part < 3) {
    double B[9], O[9];
    if (part == 0) {
      B[0] = sum[0]; B[1] = sum[1]; B[2] = sum[2]; B[3] = sum[1]; B[4] = sum[3]; B[5] = sum[4]; B[6] = sum[2]; B[7] = sum[4]; B[8] = sum[5];
    } else if (part == 1) {
      for (int i = 0; i < 9; i++) B[i] = sum[6 + i];
    } else {
      B[0] = sum[15]; B[1] = sum[16]; B[2] = sum[17]; B[3] = sum[16]; B[4] = sum[18]; B[5] = sum[19]; B[6] = sum[17]; B[7] = sum[19]; B[8] = sum[20];
    }
    rotate_block(B, R, O);
    if (part == 0) {
      rot[0] = O[0]; rot[1] = O[1]; rot[2] = O[2]; rot[3] = O[4]; rot[4] = O[5]; rot[5] = O[8];
    } else if (part == 1) {
      for (int i = 0; i < 9; i++) rot[6 + i] = O[i];
    } else {
      rot[15] = O[0]; rot[16] = O[1]; rot[17] = O[2]; rot[18] = O[4]; rot[19] = O[5]; rot[20] = O[8];
    }
  } else {
    for (int c = 0; c < 3; c++) {
      rot[21 + c] = R[c] * sum[21] + R[3 + c] * sum[22] + R[6 + c] * sum[23];   // R^T (sum u x q')
      rot[24 + c] = R[c] * sum[24] + R[3 + c] * sum[25] + R[6 + c] * sum[26];   // R^T (sum u)
    }
  }
}

// One element of rotate_part's result, for a thread of its own (the device finalisers: 27 threads instead of 4, a chain of 20 dependent FP64
// operations instead of 72): slot j of the accumulator layout -- 0..5 Hww (upper triangle), 6..14 Hwv, 15..20 Hvv (upper triangle), 21..23 and 24..26
// the two vectors.  The SAME expressions in the same order as rotate_block / rotate_part, no contraction: the same bits (the host-finalised call and
// finalize_short_kernel keep rotate_part; tests/test_gpu_edge_cases.py compares the forms bit for bit).
__device__ __forceinline__ double rotate_element(int j, const double* sum, const double* T) {
#pragma clang fp contract(off)
  // (sum and T are LDS arrays: every run-time index below is an address, not a register number; R[3 a + b] of rotate_part is T[4 a + b])
  if (j >= 21) {
    const int c = (j - 21) % 3, o = j < 24 ? 21 : 24;
    return T[c] * sum[o] + T[4 + c] * sum[o + 1] + T[8 + c] * sum[o + 2];
  }
  double B0, B1, B2, B3, B4, B5, B6, B7, B8;
  int r, c;
  if (j >= 6 && j < 15) {
    B0 = sum[6]; B1 = sum[7]; B2 = sum[8]; B3 = sum[9]; B4 = sum[10]; B5 = sum[11]; B6 = sum[12]; B7 = sum[13]; B8 = sum[14];
    r = (j - 6) / 3;
    c = (j - 6) % 3;
  } else {
    const int o = j < 6 ? 0 : 15, u = j - o;  // upper-triangle index 0..5 -> (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
    B0 = sum[o]; B1 = sum[o + 1]; B2 = sum[o + 2]; B3 = B1; B4 = sum[o + 3]; B5 = sum[o + 4]; B6 = B2; B7 = B5; B8 = sum[o + 5];
    r = u < 3 ? 0 : (u < 5 ? 1 : 2);
    c = u < 3 ? u : (u < 5 ? u - 2 : 2);
  }
  const double Rc0 = T[c], Rc1 = T[4 + c], Rc2 = T[8 + c];
  const double br0 = B0 * Rc0 + B1 * Rc1 + B2 * Rc2;
  const double br1 = B3 * Rc0 + B4 * Rc1 + B5 * Rc2;
  const double br2 = B6 * Rc0 + B7 * Rc1 + B8 * Rc2;
  return T[r] * br0 + T[4 + r] * br1 + T[8 + r] * br2;
}

// Fixed-order FP64 sum of factor f's partial rows -> compact record; executed by the 256 threads of ONE block of the finalise kernel.  A
// factor's rows are consecutive (row first_block + c belongs to its chunk c).  Thread t sums rows g, g + 32, g + 64, ... (g = t / 8) of the
// four values 4 q .. 4 q + 3 (q = t % 8), as ONE batch of up to 16 independent 16-byte loads per thread and per 512 rows; the 32 group sums
// are then added in group order.  The order depends only on the plan: results are bit-reproducible.
// Finalising INSIDE the fused kernel (the block that finds all rows of its factor written sums them) was measured three times and is gone: per-block
// agent-scope release + arrival count (round 2: 22.9 vs 18.8 us per synchronous single-factor call), and in round 3 with this very summation
// code 25.1 us (fences) / 20.6 us (write-through rows, counted waits) against 17.5 us for the two dispatches, 38 / 29 against 27 us for the
// odometry's 34-factor set (profiles/r03/probe/c2_*): several hundred blocks finishing together serialise on the arrival counter (~10 ns per
// same-address atomic) for longer than the second dispatch costs.  So did summing the rows on the host as they arrive (35.9 us: the host
// ping-pongs cache lines with the device's writes).
constexpr int FIN_GROUPS = 32;
__device__ __forceinline__ void finalize_tail(int f, const FinalizeArgs& fa, int mode, double (*s_part)[PARTIAL_STRIDE], double* s_sum, const double* Tl,
                                              unsigned long long* tail_stamps = nullptr);
__device__ __forceinline__ void finalize_factor(const FactorDesc& d, int f, const float* __restrict__ partials, const FinalizeArgs& fa, int mode,
                                                double (*s_part)[PARTIAL_STRIDE], double* s_sum, const double* __restrict__ T) {
  const int q = threadIdx.x & 7, g = threadIdx.x >> 3;
  const int first = d.first_block, nb = d.num_blocks;
  // the pose is needed at the very end (rotate_part): fetched now, so that its latency hides behind the row loads
  __shared__ double Tl[12];
  if (threadIdx.x < 12) Tl[threadIdx.x] = T[threadIdx.x];  // (finalize_tail's first barrier orders the stores before the rotation reads them)
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  constexpr int INFLIGHT = 16;
  for (int c = g; c < nb; c += FIN_GROUPS * INFLIGHT) {
    float4 v[INFLIGHT];
#pragma unroll
    for (int u = 0; u < INFLIGHT; u++) {
      const float* src = partials + (size_t)(first + min(c + FIN_GROUPS * u, nb - 1)) * PARTIAL_STRIDE + 4 * q;  // past the end: a valid row, value unused
      v[u] = *reinterpret_cast<const float4*>(src);
    }
#pragma unroll
    for (int u = 0; u < INFLIGHT; u++)
      if (c + FIN_GROUPS * u < nb) {
        s0 += (double)v[u].x;
        s1 += (double)v[u].y;
        s2 += (double)v[u].z;
        s3 += (double)v[u].w;
      }
  }
  s_part[g][4 * q + 0] = s0;
  s_part[g][4 * q + 1] = s1;
  s_part[g][4 * q + 2] = s2;
  s_part[g][4 * q + 3] = s3;
  finalize_tail(f, fa, mode, s_part, s_sum, Tl);
}

// Second half of a factor's finalisation, shared by the finalise kernel and by the finalising blocks of the single-dispatch kernel (same
// bits): the 32 group sums of s_part added in group order, the four R^T B R rotations, the record, the completion word.
// Tl: the factor's pose, 12 doubles in LDS written before this call
// tail_stamps (device timeline of a resident session, else null): [0] the group sums are added, [1] the blocks are rotated -- s_memrealtime, kept in
// registers and stored behind the record so that the stamps do not sit in front of a barrier
__device__ __forceinline__ void finalize_tail(int f, const FinalizeArgs& fa, int mode, double (*s_part)[PARTIAL_STRIDE], double* s_sum, const double* Tl,
                                              unsigned long long* tail_stamps) {
  __syncthreads();
  if (threadIdx.x < PARTIAL_STRIDE) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < FIN_GROUPS; k++) t += s_part[k][threadIdx.x];
    s_sum[threadIdx.x] = t;
  }
  __syncthreads();
  unsigned long long st0 = 0ull, st1 = 0ull;
  if (tail_stamps) st0 = __builtin_amdgcn_s_memrealtime();
  const int t = threadIdx.x;
  if (t == 29 && fa.trip_stats && s_sum[29] > 0.0) atomicAdd(&fa.trip_stats[f & 63], (unsigned long long)s_sum[29]);
  __shared__ double s_rot[32];
  if (mode == MODE_LINEARIZE) {
#if GLIM_AMD_ROT27
    if (t < 27) s_rot[t] = rotate_element(t, s_sum, Tl);  // one thread per rotated value (rotate_part's bits)
#else
    if (t < 4) rotate_part(t, s_sum, Tl, s_rot);  // thread k < 3 rotates block k (Hww, Hwv, Hvv), thread 3 the two vectors
#endif
    __syncthreads();
  }
  if (tail_stamps) st1 = __builtin_amdgcn_s_memrealtime();
  // slot t of the compact record: [count, error, 21 upper-triangular H_ss entries, b_w = R^T sum u x q', b_v = -R^T sum u]
  double value = 0.0;
  if (t == 0) value = s_sum[28];
  else if (t == 1) value = s_sum[27];
  else if (mode == MODE_LINEARIZE && t < COMPACT) value = t < 23 ? s_rot[acc_of_upper(t - 2)] : (t < 26 ? s_rot[t - 2] : -s_rot[t - 2]);
  if (fa.rec16) {
    // Single-dispatch form: the record goes to host-mapped memory as 29 self-validating 16-byte granules {value, sequence number}, one
    // store each -- a PCIe write lands as a unit, so a granule whose tag is this call's carries this call's value.  The host polls the
    // tags: no system fence (~1 us of write acknowledgements), no arrival counter, no completion word.
    if (t < COMPACT) {
      const long long bits = __double_as_longlong(value);
      const v4i_t g = {(int)(bits & 0xffffffffll), (int)(bits >> 32), (int)fa.seq, 0};
      // system-scope (sc0 sc1) store: written through to host memory NOW.  A plain store may sit in the L2 until the kernel ends -- which a
      // resident kernel does not do (measured: every request then took exactly one idle time-out).
      const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(fa.rec16 + (size_t)f * COMPACT * 16, 0, COMPACT * 16, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(g, rsrc, t * 16, 0, AUX_SYSTEM);
    }
    if (tail_stamps && t == 0) {
      tail_stamps[0] = st0;
      tail_stamps[1] = st1;
    }
    return;
  }
  double* o = fa.out + ((size_t)fa.out_row_offset + f) * COMPACT;
  if (t < COMPACT) o[t] = value;
  // (a caller that wants the records on the host as well gets them by a second store -- 232 B per factor over PCIe while the launch runs --
  // instead of a device-to-host copy behind it: 0.12 ms per evaluation of the 32 640-pair cost, glim_amd_multi)
  if (fa.out_mirror && t < COMPACT) fa.out_mirror[((size_t)fa.out_row_offset + f) * COMPACT + t] = value;
  if (fa.host_flag) {
    // completion word of the polling fast path: every finalising block waits until its record is visible system-wide (ONE fence), the one
    // that completes the launch publishes `seq` into host-mapped memory (the host spins on it instead of a stream synchronise).  A
    // single-factor launch has nobody to count.
    __threadfence_system();
    __syncthreads();
    if (t == 0) {
      bool last = true;
      if (fa.num_factors > 1) {
        last = __hip_atomic_fetch_add(fa.done_counter, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == fa.num_factors - 1;
        if (last) __hip_atomic_store(fa.done_counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (last) __hip_atomic_store(fa.host_flag, fa.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Finalising block of the single-dispatch kernel: the same fixed-order FP64 sum as finalize_factor -- value j of the factor's rows c = g, g + 32,
// g + 64, ... added in that order into group sum g, the 32 group sums then added in group order (finalize_tail) -- taken over the tagged
// rows as they arrive.  (group, piece) pair k = 10 g + p is handled by thread k, pairs 256..319 by threads 0..63 in a second round.  A
// thread re-reads the <= 16 granules of a round until every tag is this call's; then it adds them.  The spin is bounded (~1 s): a row that
// never arrives (cannot happen: the writing blocks wait for nothing) ends in a NaN record instead of a hung device.
// (Round 6: the 320 (group, piece) pairs are 256 + 64, so threads 0..63 own TWO pairs.  They used to poll them one after the other -- each pair in
//  nb / (32 INFLIGHT) all-or-nothing batches -- which put up to four dependent polling round trips (~1.2 us each: write-through granules read past the
//  L2) behind the last row of a 512-row factor: the device timeline of the resident call showed 5 us between the last row and the record.  Both
//  pairs now advance in LOCKSTEP, 2 x INFLIGHT loads in flight per batch; the order in which a pair's rows are added is unchanged: same bits.)
template <int INFLIGHT = 8>
__device__ __forceinline__ void fused_finalize(const FactorDesc& d, int f, const FinalizeArgs& fa, int mode, double (*s_part)[PARTIAL_STRIDE], double* s_sum,
                                               const double* __restrict__ T, unsigned long long* rows_seen_stamp = nullptr) {
  const int first = d.first_block, nb = d.num_blocks;
  // the pose is needed at the very end (rotate_part): fetched now so that its latency hides behind the row polls, and parked in LDS -- 24 registers
  // of every thread for the whole poll otherwise, which is what made the two-pair form spill under the resident kernel's 128-register cap
  __shared__ double s_T[12];
  if (threadIdx.x < 12) s_T[threadIdx.x] = T[threadIdx.x];
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(fa.rows16 + (size_t)first * TAG_ROW_BYTES, 0, nb * TAG_ROW_BYTES, 0x00020000);
  const int seq = (int)fa.seq;
  static_assert(FIN_GROUPS * TAG_PIECES <= 2 * BLOCK, "a thread owns at most two (group, piece) pairs");
  const int kA = (int)threadIdx.x, kB = (int)threadIdx.x + BLOCK;
  const bool hasB = kB < FIN_GROUPS * TAG_PIECES;
  const int gA = kA / TAG_PIECES, pA = kA - gA * TAG_PIECES;
  const int gB = hasB ? kB / TAG_PIECES : gA, pB = hasB ? kB - gB * TAG_PIECES : pA;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0;
  bool lost = false;
  for (int base = 0;; base += FIN_GROUPS * INFLIGHT) {
    const int cA = gA + base, cB = gB + base;
    const bool moreA = cA < nb, moreB = hasB && cB < nb;
    if (!moreA && !moreB) break;
    v4i_t va[INFLIGHT], vb[INFLIGHT];
    bool okA = !moreA, okB = !moreB;
    for (unsigned int spins = 0;; spins++) {
      if (!okA) {
        bool ok = true;
#pragma unroll
        for (int u = 0; u < INFLIGHT; u++) {
          const int row = min(cA + FIN_GROUPS * u, nb - 1);  // past the end: a valid row, value unused
          va[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, row * TAG_ROW_BYTES + pA * 16, 0, AUX_SC1_VOLATILE);
        }
        if (!okB) {  // (both pairs' loads are in flight before either is looked at)
#pragma unroll
          for (int u = 0; u < INFLIGHT; u++) {
            const int row = min(cB + FIN_GROUPS * u, nb - 1);
            vb[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, row * TAG_ROW_BYTES + pB * 16, 0, AUX_SC1_VOLATILE);
          }
        }
#pragma unroll
        for (int u = 0; u < INFLIGHT; u++) ok = ok && va[u].w == seq;
        okA = ok;
        if (!okB) {
          bool okb = true;
#pragma unroll
          for (int u = 0; u < INFLIGHT; u++) okb = okb && vb[u].w == seq;
          okB = okb;
        }
      } else if (!okB) {
        bool okb = true;
#pragma unroll
        for (int u = 0; u < INFLIGHT; u++) {
          const int row = min(cB + FIN_GROUPS * u, nb - 1);
          vb[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, row * TAG_ROW_BYTES + pB * 16, 0, AUX_SC1_VOLATILE);
          okb = okb && vb[u].w == seq;
        }
        okB = okb;
      }
      if (okA && okB) break;
      if (spins > (1u << 20)) {
        lost = true;
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
    if (moreA) {
#pragma unroll
      for (int u = 0; u < INFLIGHT; u++)
        if (cA + FIN_GROUPS * u < nb) {
          a0 += (double)__int_as_float(va[u].x);
          a1 += (double)__int_as_float(va[u].y);
          a2 += (double)__int_as_float(va[u].z);
        }
    }
    if (moreB) {
#pragma unroll
      for (int u = 0; u < INFLIGHT; u++)
        if (cB + FIN_GROUPS * u < nb) {
          b0 += (double)__int_as_float(vb[u].x);
          b1 += (double)__int_as_float(vb[u].y);
          b2 += (double)__int_as_float(vb[u].z);
        }
    }
    if (lost) break;
  }
  if (lost) a0 = a1 = a2 = b0 = b1 = b2 = __builtin_nan("");
  s_part[gA][3 * pA + 0] = a0;
  s_part[gA][3 * pA + 1] = a1;
  s_part[gA][3 * pA + 2] = a2;
  if (hasB) {
    s_part[gB][3 * pB + 0] = b0;
    s_part[gB][3 * pB + 1] = b1;
    s_part[gB][3 * pB + 2] = b2;
  }
  if (threadIdx.x < FIN_GROUPS) s_part[threadIdx.x][30] = s_part[threadIdx.x][31] = 0.0;
  if (rows_seen_stamp && threadIdx.x == 0) *rows_seen_stamp = __builtin_amdgcn_s_memrealtime();  // (thread 0 owns two pairs: the longer path; the barrier of finalize_tail follows)
  finalize_tail(f, fa, mode, s_part, s_sum, s_T, rows_seen_stamp ? rows_seen_stamp + 2 : nullptr);  // (its first barrier orders the s_T stores before the rotation reads them)
}

// resident waves per SIMD the register allocation aims for: the plane-form kernel fits 96 VGPRs (5 waves), the general one needs 99 (4 waves)
#ifndef GLIM_AMD_MINW_PLANE
#define GLIM_AMD_MINW_PLANE 5
#endif
#ifndef GLIM_AMD_MINW_GENERAL
#define GLIM_AMD_MINW_GENERAL 4
#endif

// Plane-form factors read the "plane view" of the target table (A_B = (C_B + I)^-1 in the covariance slots, voxelmap.hip plane_view_kernel)
// and form M by Sherman-Morrison instead of building S = C_B + R C_A R^T and inverting it by cofactors: 26 FP32
// operations instead of 39, same conditioning (the denominator 1 - 0.999 m^T A_B m >= 0.002 carries the cancellation the determinant of S
// carried).  0: the cofactor form on the plain table (A/B builds: tools/ab_variant.sh sm0 -DGLIM_AMD_PLANE_SM=0).
#ifndef GLIM_AMD_PLANE_SM
#define GLIM_AMD_PLANE_SM 1
#endif
// key compares of the probe resolution: 1 = every predicate formed once (four 64-bit compares per trip; adopted in round 6: configs[3] kernel
// 11.11 -> 10.96 ms same box, alternating, -1.4 %; plane-form kernel and synchronous call unchanged: profiles/r06/probe/keycmp4_ab.json), 0 = round 5's form (six)
#ifndef GLIM_AMD_KEYCMP4
#define GLIM_AMD_KEYCMP4 1
#endif

struct Rot32 {  // rotation of the linearisation pose in FP32 (wave-uniform: lives in SGPRs)
  float r00, r01, r02, r10, r11, r12, r20, r21, r22;
};

// What a point carries from the probe stage (trip t-1) to the algebra stage (trip t) of the pipelined loop.
template <bool PLANE>
struct Probe {
  unsigned long long key;  // packed voxel coordinate of q, EMPTY_KEY when the lane has no point / out of range / failed validation
  unsigned int bkt;        // home bucket
  float4 head;             // both keys of the home bucket (in flight until the next trip)
  float qr0, qr1, qr2;     // q relative to the centre of its voxel (|.| <= res/2): FP32 without cancellation
  float qp0, qp1, qp2;     // q' = R p = q - t: the rotated source point, the lever arm of the target-frame Jacobian
  float c0, c1, c2, c3, c4, c5;  // PLANE: m = R n in c0..c2; general: C_t = R C_A R^T (t00 t01 t02 t11 t12 t22)
};

template <bool FROZEN, bool PLANE>
__device__ __forceinline__ Probe<PLANE> probe_point(const FactorDesc& d, const PointIn& pt, int i, bool in_trip, const double* __restrict__ Tl,
                                                    const double* __restrict__ Te, const Rot32& R, bool validate, int last) {
  Probe<PLANE> o;
  const bool ok = in_trip && (i < d.n);
  // (An FP32 transform with an exactness guard -- FP32 q, floor accepted only when the point lies provably inside its voxel, FP64 fallback for
  // the other 3e-4 of the lanes -- removes 33 half-rate instructions here; measured twice, -1.5 % on the plane-form kernel and -2 % on the
  // VALU-bound 256-submap cost at the price of 1e-5 m in the residuals: not adopted, profiles/r03/probe/ab_fp32_transform_global256_and_m1.jsonl.)
  double qx, qy, qz;
  transform_point_d(Tl, (double)pt.p.x, (double)pt.p.y, (double)pt.p.z, qx, qy, qz);
  const double tx = qx * d.inv_res, ty = qy * d.inv_res, tz = qz * d.inv_res;
  o.qp0 = (float)(qx - Tl[3]);
  o.qp1 = (float)(qy - Tl[7]);
  o.qp2 = (float)(qz - Tl[11]);
  // floor(t) == fast_floor(t) for every in-range coordinate (same integer, bit-exact); v_floor_f64 + one subtraction also gives the
  // in-voxel fraction for free
  const double fx = floor(tx), fy = floor(ty), fz = floor(tz);
  const int cx = __double2int_rz(fx), cy = __double2int_rz(fy), cz = __double2int_rz(fz);
  // out-of-range coordinates saturate in v_cvt_i32_f64 and fail the unsigned 21-bit range check
  const unsigned int ux = (unsigned int)(cx + KEY_OFFSET), uy = (unsigned int)(cy + KEY_OFFSET), uz = (unsigned int)(cz + KEY_OFFSET);
  bool keep = ok && (((ux | uy | uz) >> KEY_BITS) == 0u);
  const unsigned int hsh = hash_fields(ux, uy, uz);
  if (FROZEN) {
    double ex, ey, ez;
    transform_point_d(Te, (double)pt.p.x, (double)pt.p.y, (double)pt.p.z, ex, ey, ez);
    o.qr0 = (float)(ex - ((double)cx + 0.5) * d.res);
    o.qr1 = (float)(ey - ((double)cy + 0.5) * d.res);
    o.qr2 = (float)(ez - ((double)cz + 0.5) * d.res);
  } else {
    const float resf = (float)d.res;
    o.qr0 = ((float)(tx - fx) - 0.5f) * resf;
    o.qr1 = ((float)(ty - fy) - 0.5f) * resf;
    o.qr2 = ((float)(tz - fz) - 0.5f) * resf;
  }
  if (PLANE) {
    // C_A = I - (1 - 1e-3) n n^T  =>  R C_A R^T = I - (1 - 1e-3) m m^T with m = R n
    const float nx = pt.p.w, ny = pt.cb.x, nz = pt.cb.y;
    o.c0 = R.r00 * nx + R.r01 * ny + R.r02 * nz;
    o.c1 = R.r10 * nx + R.r11 * ny + R.r12 * nz;
    o.c2 = R.r20 * nx + R.r21 * ny + R.r22 * nz;
    o.c3 = o.c4 = o.c5 = 0.f;
  } else {
    // C_t = R C_A R^T (symmetric: t00 t01 t02 t11 t12 t22), computed here rather than in the algebra stage: its 9 temporaries are dead
    // before the algebra's own peak of live values
    const float c00 = pt.p.w, c01 = pt.ca.x, c02 = pt.ca.y, c11 = pt.ca.z, c12 = pt.ca.w, c22 = pt.cb.x;
    const float a00 = R.r00 * c00 + R.r01 * c01 + R.r02 * c02, a01 = R.r00 * c01 + R.r01 * c11 + R.r02 * c12, a02 = R.r00 * c02 + R.r01 * c12 + R.r02 * c22;
    const float a10 = R.r10 * c00 + R.r11 * c01 + R.r12 * c02, a11 = R.r10 * c01 + R.r11 * c11 + R.r12 * c12, a12 = R.r10 * c02 + R.r11 * c12 + R.r12 * c22;
    const float a20 = R.r20 * c00 + R.r21 * c01 + R.r22 * c02, a21 = R.r20 * c01 + R.r21 * c11 + R.r22 * c12, a22 = R.r20 * c02 + R.r21 * c12 + R.r22 * c22;
    o.c0 = a00 * R.r00 + a01 * R.r01 + a02 * R.r02; o.c1 = a00 * R.r10 + a01 * R.r11 + a02 * R.r12; o.c2 = a00 * R.r20 + a01 * R.r21 + a02 * R.r22;
    o.c3 = a10 * R.r10 + a11 * R.r11 + a12 * R.r12; o.c4 = a10 * R.r20 + a11 * R.r21 + a12 * R.r22; o.c5 = a20 * R.r20 + a21 * R.r21 + a22 * R.r22;
  }
  if (validate) {
    // surface validation (upstream predicate unverified -- SURVEY.md App. B.5): the source normal faces the source sensor
    // (p . n <= 0, cloud_covariance_estimation.cpp:98-101); reject when the transformed surface faces away from the target origin.
    float rnx, rny, rnz;
    if (PLANE) {
      rnx = o.c0; rny = o.c1; rnz = o.c2;
    } else {
      const float4 nn = gld4(reinterpret_cast<const char*>(d.sn) + (unsigned int)min(i, last) * 16u);
      rnx = R.r00 * nn.x + R.r01 * nn.y + R.r02 * nn.z;
      rny = R.r10 * nn.x + R.r11 * nn.y + R.r12 * nn.z;
      rnz = R.r20 * nn.x + R.r21 * nn.y + R.r22 * nn.z;
    }
    keep = keep && !(rnx * (float)qx + rny * (float)qy + rnz * (float)qz > 0.f);
    asm volatile("" : "+v"(o.qp0));  // keeps this block a (wave-uniform) branch: flattened into selects it costs every factor 12 instructions
  }
  o.key = keep ? (((unsigned long long)ux << (2 * KEY_BITS)) | ((unsigned long long)uy << KEY_BITS) | (unsigned long long)uz) : EMPTY_KEY;
  o.bkt = __umulhi(hsh, d.num_buckets);
  o.head = gld4(reinterpret_cast<const char*>(d.buckets) + o.bkt * 128u);  // both keys of the home bucket (32-bit offset: <= 2^25 buckets)
  return o;
}

// Per-point algebra in the TARGET frame (all lanes run it; `hit` predicates the contributions through idet = 0).
template <int MODE, bool PLANE>
__device__ __forceinline__ void accumulate_point(float (&acc)[NACC], bool hit, const float4& r0, const float4& r1, float r2c22, const Probe<PLANE>& s,
                                                 const Rot32& R) {
  // C_t = R C_A R^T, the source covariance in the target frame (symmetric: t00 t01 t02 t11 t12 t22)
  float t00, t01, t02, t11, t12, t22;
  if (PLANE) {
    // C_A = I - (1 - 1e-3) n n^T  =>  C_t = I - (1 - 1e-3) m m^T with m = R n (probe stage): 9 + 9 operations instead of the 45-operation sandwich
    const float mx = s.c0, my = s.c1, mz = s.c2;
    const float w = 0.999f, wx = w * mx, wy = w * my;
    t00 = 1.f - wx * mx; t01 = wx * my; t02 = wx * mz;       // (off-diagonal entries carry the opposite sign: subtracted with one fma below)
    t11 = 1.f - wy * my; t12 = wy * mz; t22 = 1.f - w * mz * mz;
  } else {
    t00 = s.c0; t01 = s.c1; t02 = s.c2; t11 = s.c3; t12 = s.c4; t22 = s.c5;
  }
  // residual mu - q, both relative to the voxel centre
  const float rx = r0.x - s.qr0, ry = r0.y - s.qr1, rz = r0.z - s.qr2;
  float A00, A01, A02, A11, A12, A22;
  if (PLANE && GLIM_AMD_PLANE_SM) {
    // the record holds A_B = (C_B + I)^-1 (plane view of the table).  M = A_B + c w w^T, w = A_B m, c = 0.999 / (1 - 0.999 m.w).  A lane
    // without a match has read the table's all-zero record: w = 0, M = 0, every contribution an exact zero -- no select.
    // Explicit fmaf chains: this function is inlined into four kernels that must return the same bits.
    const float a00 = r0.w, a01 = r1.x, a02 = r1.y, a11 = r1.z, a12 = r1.w, a22 = r2c22;
    const float mx = s.c0, my = s.c1, mz = s.c2;
    const float wx = fmaf(a02, mz, fmaf(a01, my, a00 * mx));
    const float wy = fmaf(a12, mz, fmaf(a11, my, a01 * mx));
    const float wz = fmaf(a22, mz, fmaf(a12, my, a02 * mx));
    const float mw = fmaf(mz, wz, fmaf(my, wy, mx * wx));
    const float den = fmaf(-0.999f, mw, 1.0f);
    float iden = __builtin_amdgcn_rcpf(den);
    iden = fmaf(fmaf(-den, iden, 1.0f), iden, iden);
    const float c = 0.999f * iden;
    const float cx = c * wx, cy = c * wy, cz = c * wz;
    A00 = fmaf(cx, wx, a00); A01 = fmaf(cx, wy, a01); A02 = fmaf(cx, wz, a02);
    A11 = fmaf(cy, wy, a11); A12 = fmaf(cy, wz, a12); A22 = fmaf(cz, wz, a22);
    (void)hit; (void)t00; (void)t01; (void)t02; (void)t11; (void)t12; (void)t22;
  } else {
  // S = C_B + R C_A R^T (symmetric).  A lane without a match has read SOME record of the table -- another voxel's, or the zeros of an empty
  // way (voxelmap.hip initialises every record) -- so everything up to the determinant is finite for it too; idet = 0 (a select, not a
  // product) then zeroes its contributions exactly.  No per-coefficient selects.
  const float S00 = r0.w + t00, S11 = r1.z + t11, S22 = r2c22 + t22;
  float S01, S02, S12;
  if (PLANE) {
    const float mx = s.c0, my = s.c1, mz = s.c2, wx = 0.999f * mx, wy = 0.999f * my;
    S01 = fmaf(-wx, my, r1.x); S02 = fmaf(-wx, mz, r1.y); S12 = fmaf(-wy, mz, r1.w);
  } else {
    S01 = r1.x + t01; S02 = r1.y + t02; S12 = r1.w + t12;
  }
  (void)t01; (void)t02; (void)t12;
  // M = S^-1 by cofactors (symmetric, called A below)
  const float k00 = S11 * S22 - S12 * S12;
  const float k01 = S02 * S12 - S01 * S22;
  const float k02 = S01 * S12 - S02 * S11;
  const float det = S00 * k00 + S01 * k01 + S02 * k02;
  float idet = __builtin_amdgcn_rcpf(det);         // 1 ulp hardware reciprocal ...
  idet = fmaf(fmaf(-det, idet, 1.0f), idet, idet);  // ... + one Newton step (full FP32 accuracy, no division sequence)
  idet = hit ? idet : 0.f;
  A00 = k00 * idet; A01 = k01 * idet; A02 = k02 * idet;
  A11 = (S00 * S22 - S02 * S02) * idet;
  A12 = (S01 * S02 - S00 * S12) * idet;
  A22 = (S00 * S11 - S01 * S01) * idet;
  }
  // u = M r,  e = r . u   (r = mu - q is already a target-frame vector)
  const float ux = A00 * rx + A01 * ry + A02 * rz;
  const float uy = A01 * rx + A11 * ry + A12 * rz;
  const float uz = A02 * rx + A12 * ry + A22 * rz;
  acc[27] += rx * ux + ry * uy + rz * uz;
  if (MODE == MODE_LINEARIZE) {
    // J_s = [R hat(p) | -R] = [hat(q') | -I] diag(R, R) with q' = R p: everything below is accumulated for J' = [hat(q') | -I] and M in the
    // target frame; the constant diag(R, R) is applied once per factor, in FP64, by finalize_factor
    const float x = s.qp0, y = s.qp1, z = s.qp2;
    // G = hat(q') M : column j = q' x M[:,j]
    const float g00 = y * A02 - z * A01, g01 = y * A12 - z * A11, g02 = y * A22 - z * A12;
    const float g10 = z * A00 - x * A02, g11 = z * A01 - x * A12, g12 = z * A02 - x * A22;
    const float g20 = x * A01 - y * A00, g21 = x * A11 - y * A01, g22 = x * A12 - y * A02;
    // Hww = -G hat(q') : row i = q' x G[i,:]
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
    // b_w = u x q',  b_v = -u
    acc[21] += uy * z - uz * y;
    acc[22] += uz * x - ux * z;
    acc[23] += ux * y - uy * x;
    acc[24] += ux; acc[25] += uy; acc[26] += uz;
  }
}

template <bool PLANE>
struct PipeCtx {  // wave-uniform context of the pipelined loop
  const FactorDesc& d;
  const double* Tl;
  const double* Te;
  const Rot32& R;
  int base, stride, ppt, last;
  bool validate;
};

// One trip of the software-pipelined point loop.  `pr` holds the probe of point `it` on entry (its 16-byte key gather was issued in the
// previous trip) and the probe of point it+1 on exit; `nxt` holds the stream data of point it+1 on entry and of it+2 on exit (AHEAD = 1).
//   (1) resolve the probe of point `it` and issue the dependent 36-byte record gather (an on-chip hit: the key load fetched the line),
//   (2) transform point it+AHEAD (FP64), derive its key and issue ITS key gather -- the only load that normally goes to HBM at random,
//       so it gets a whole trip of algebra (this wave's and the other resident waves') to come back,
//   (3) issue the coalesced stream loads of point it+AHEAD+1,
//   (4) wait for the records of point `it` only (the OLDEST loads in flight: counted vmcnt) and run the algebra.
// The first version of the loop issued key gather t, stream t+1, then waited for the key, then for the record, inside one trip: two
// dependent memory round trips (one of them HBM) exposed per trip per wave, which 5 waves per SIMD could not cover (waves parked on
// memory 71 % of their cycles; 50 % with this form -- tools/pmc_kexp.sh, profiles/r02/probe/).
// i1 / ok1: the point this trip probes (stream data in `nxt`) and whether it exists; i2: the point whose stream loads this trip issues.
// TAILSKIP (the resident kernels: a wavefront there walks one or two trips and the caller waits for it): the probe and the stream loads of a point that
// no lane has -- the last trip's steps (2) and (3), ~80 instructions in front of the last algebra -- are skipped behind a scalar branch.  Nothing
// that is added depends on them: same bits.  (The launch-per-call kernels walk 30-60 trips and stay as they are: their loop sits at its register cap.)
template <int MODE, bool FROZEN, bool PLANE, bool TAILSKIP = false>
__device__ __forceinline__ void pipe_trip_at(const PipeCtx<PLANE>& pc, Probe<PLANE>& pr, PointIn& nxt, int i1, bool ok1, int i2, float (&acc)[NACC], int& wave_inliers,
                                             int& wave_skips) {
  const FactorDesc& d = pc.d;
  // (1) resolve point `it`
  const Probe<PLANE> cur = pr;
  unsigned long long k0 = (unsigned long long)__float_as_uint(cur.head.x) | ((unsigned long long)__float_as_uint(cur.head.y) << 32);
  unsigned long long k1 = (unsigned long long)__float_as_uint(cur.head.z) | ((unsigned long long)__float_as_uint(cur.head.w) << 32);
  unsigned int b = cur.bkt;
#if GLIM_AMD_KEYCMP4
  // four 64-bit compares per trip instead of six: each predicate is formed ONCE and reused as a lane mask -- the form below
  // this #if re-evaluates `k0 == key` / `k1 == key` after the (rare) spill walk because the walk may have changed them
  const bool valid = cur.key != EMPTY_KEY;
  bool eq0 = k0 == cur.key, eq1 = k1 == cur.key;
  if (valid && !eq0 && !eq1 && k1 != EMPTY_KEY) {
    // rare spill: both ways of the home bucket hold other keys -> walk to the next bucket (exact compare, table never full)
    do {
      b = (b + 1 == d.num_buckets) ? 0u : b + 1;
      const float4 h = gld4(reinterpret_cast<const char*>(d.buckets) + b * 128u);
      k0 = (unsigned long long)__float_as_uint(h.x) | ((unsigned long long)__float_as_uint(h.y) << 32);
      k1 = (unsigned long long)__float_as_uint(h.z) | ((unsigned long long)__float_as_uint(h.w) << 32);
      eq0 = k0 == cur.key;
      eq1 = k1 == cur.key;
    } while (!eq0 && !eq1 && k1 != EMPTY_KEY);
  }
  const bool in1 = eq1;
  const bool hit = valid && (eq0 || eq1);
#else
  if (cur.key != EMPTY_KEY) {
    // rare spill: both ways of the home bucket hold other keys -> walk to the next bucket (exact compare, table never full)
    while (k0 != cur.key && k1 != cur.key && k1 != EMPTY_KEY) {
      b = (b + 1 == d.num_buckets) ? 0u : b + 1;
      const float4 h = gld4(reinterpret_cast<const char*>(d.buckets) + b * 128u);
      k0 = (unsigned long long)__float_as_uint(h.x) | ((unsigned long long)__float_as_uint(h.y) << 32);
      k1 = (unsigned long long)__float_as_uint(h.z) | ((unsigned long long)__float_as_uint(h.w) << 32);
    }
  }
  const bool in1 = (k1 == cur.key);
  const bool hit = (cur.key != EMPTY_KEY) && (k0 == cur.key || in1);
#endif
  // A wavefront trip in which no lane has a correspondence skips the record gather and the algebra (its lanes would add exact zeros, so the
  // sums are bit-identical).  In Hilbert order misses come in runs: 17 % of the trips of the 256-submap all-pairs cost (inlier fraction
  // 0.69) have no hit at all; 13.9 -> 12.2 ms for that evaluation (BENCH_r02 `staged`).  General (36 B/pt) kernel only: under the
  // plane-form kernel's 96-register cap the branch makes the allocator spill, and scan-to-scan factors have few all-miss trips.
  const unsigned long long hit_lanes = __ballot(hit);
  wave_inliers += __popcll(hit_lanes);  // wave-uniform count: scalar registers, no per-lane counter
  const bool any_hit = PLANE || hit_lanes != 0ull;  // wave-uniform: a scalar branch
  if (!PLANE) wave_skips += any_hit ? 0 : 1;         // (scalar) trips that skip the record gather and the algebra: reported per run, glim_amd_factor_set_trip_stats
  // every lane reads a record (way 0 of the last bucket when there is no hit) so the wavefront does not diverge
  // (plane view: a lane without a correspondence reads the all-zero record of the extra bucket behind the table, so its M is exactly 0)
  const unsigned int rec_off = (PLANE && GLIM_AMD_PLANE_SM) ? (hit ? b * 128u + (in1 ? 64u : 16u) : d.num_buckets * 128u + 16u) : b * 128u + (in1 ? 64u : 16u);
  const char* rp = reinterpret_cast<const char*>(d.buckets) + rec_off;
  float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
  float r2 = 0.f;
  if (any_hit) {
    r0 = gld4(rp);        // mx my mz c00
    r1 = gld4(rp + 16);   // c01 c02 c11 c12
    r2 = gld1(rp + 32);   // c22
  }
  if (!TAILSKIP || ok1) {
    // (2) probe of point it+AHEAD (a lane past its last point probes with EMPTY_KEY at a clamped, valid address)
    pr = probe_point<FROZEN, PLANE>(d, nxt, i1, ok1, pc.Tl, pc.Te, pc.R, pc.validate, pc.last);
    // (3) stream loads of point it+AHEAD+1
    nxt = load_point<PLANE>(d, (unsigned int)min(i2, pc.last));
  }
  // (4) algebra of point `it`
  if (any_hit) accumulate_point<MODE, PLANE>(acc, hit, r0, r1, r2, cur, pc.R);
  // The key gather the NEXT trip resolves is consumed HERE, at the very end of this trip, and nowhere earlier: without this pin the
  // register allocator recycles two of its destination registers as algebra temporaries and copies them out mid-trip, which puts the
  // s_waitcnt for the HBM gather 85 instructions after its issue instead of a whole trip.
  asm volatile("" : "+v"(pr.head.x), "+v"(pr.head.y), "+v"(pr.head.z), "+v"(pr.head.w));
  // Likewise the six FP32 offsets of the point probed in this trip must EXIST here: left alone, the scheduler sinks the FP64 -> FP32
  // conversions into the next trip and carries the six FP64 values (12 VGPRs instead of 6) around the loop.
  asm volatile("" : "+v"(pr.qr0), "+v"(pr.qr1), "+v"(pr.qr2), "+v"(pr.qp0), "+v"(pr.qp1), "+v"(pr.qp2));
}
template <int MODE, bool FROZEN, bool PLANE, bool TAILSKIP = false>
__device__ __forceinline__ void pipe_trip(const PipeCtx<PLANE>& pc, Probe<PLANE>& pr, PointIn& nxt, int it, float (&acc)[NACC], int& wave_inliers, int& wave_skips) {
  constexpr int AHEAD = 1;
  pipe_trip_at<MODE, FROZEN, PLANE, TAILSKIP>(pc, pr, nxt, pc.base + (it + AHEAD) * pc.stride, it + AHEAD < pc.ppt, pc.base + (it + AHEAD + 1) * pc.stride, acc, wave_inliers,
                                    wave_skips);
}

// The SIMD's issue arbiter serves the highest user priority first and, among equals, the OLDEST wave: with every wave at priority 0 the
// blocks dispatched first ran ahead (block time rose with the block index, 105 -> 138 us, whatever the data: tools/k4_timing.py) and left
// their CUs half empty for the last quarter of the launch.  Rotating the user priority trip by trip, phase-shifted per resident block,
// gives every wave the same share of the issue slots, so the resident set finishes together (125 -> 119.5 us per 128 factors).  Measured
// level with it: five phases for five resident blocks, four trips per level; worse: a static priority that favours the younger blocks
// (profiles/r02/probe/kexp_ab_priority_schemes.jsonl).
__device__ __forceinline__ void rotate_priority(int step) {
  switch (step & 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
}

// One (factor, chunk) row of a plan, computed by the 256 threads of a block: its lanes walk the factor's points in 256-point hands dealt
// round robin to the factor's blocks, through the two-trip software pipeline of pipe_trip, with the wave priority rotated every trip; every lane
// runs the algebra (a lane without a match contributes exact zeros) and the only branch on the hot path is the rare bucket spill.  On return
// (after a block barrier) s_red[w] holds wavefront w's sums -- as its four 16-lane row sums (GLIM_AMD_ROW_SUMS_LDS; wave_value adds them) -- of
// accumulator j, [28] its inlier count.
// first: the stream data of this lane's first point when the caller has loaded it already (a resident worker does, while it waits for its pose).
// cull (CULL only): this block's four pre-cull words (cull_kernel), one per wavefront: bit t set = trip t of that wavefront has no point or cannot
// find a correspondence.  The wavefront then walks its LIVE trips only, through the same two-trip pipeline: a culled trip costs nothing at all --
// no stream load, no FP64 transform, no hash, no key gather (the in-loop skip of an all-miss trip still pays those: 40 % of a trip) -- and since
// a culled trip would have added exact zeros in every lane, the sums are the same bits.
template <int MODE, bool FROZEN, bool PLANE, bool CULL = false, bool TAILSKIP = false>
__device__ __forceinline__ void compute_row(const FactorDesc& d, const double* __restrict__ Tl, const double* __restrict__ Te, int chunk, int prio_phase,
                                            RedRow* s_red, const PointIn* first = nullptr, const unsigned long long* cull = nullptr,
                                            unsigned long long* loop_done_stamp = nullptr) {
  // rotation of the linearisation pose in FP32 (R[r][c])
  // (wave-uniform: the compiler keeps these in SGPRs; forcing readfirstlane changed nothing -- 93 VGPRs either way)
  const float R00 = (float)Tl[0], R01 = (float)Tl[1], R02 = (float)Tl[2];
  const float R10 = (float)Tl[4], R11 = (float)Tl[5], R12 = (float)Tl[6];
  const float R20 = (float)Tl[8], R21 = (float)Tl[9], R22 = (float)Tl[10];
  const Rot32 R = {R00, R01, R02, R10, R11, R12, R20, R21, R22};
  const bool validate = (d.flags & GLIM_AMD_FACTOR_SURFACE_VALIDATION) && (PLANE || d.sn != nullptr);
  const int last = d.n - 1;

  float acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; j++) acc[j] = 0.f;
  int wave_inliers = 0;  // wave-uniform count (scalar registers)
  int wave_skips = 0;    // wave-uniform: trips of this wavefront without any correspondence (general kernel)

  const int ppt = d.ppt;
  // Points of a factor are dealt to its blocks in 256-point hands, round robin: trip t of block (chunk) c covers points
  // [(t * num_blocks + c) * 256, +256).  Contiguous chunks (block c = points [c * ppt * 256, +ppt * 256)) left the launch TAIL-bound: in
  // Hilbert order a chunk is one region of the scan, and a region of sparse far-range points touches several times more bucket lines per
  // point than a dense near-range one -- per-block time stamps (tools/k4_timing.py) showed equal-sized blocks of ONE resident set taking
  // 75 ... 159 us (median 114), the kernel lasting as long as the slowest.  Dealt round robin every block sees the same mix.
  const int stride = d.num_blocks * BLOCK;
  const int base = chunk * BLOCK + threadIdx.x;
  if (d.n > 0) {
    // Software pipeline over the points of this lane (pipe_trip above): the key gather of a point is issued one trip before its algebra.
    // (Issuing it two trips ahead buys little: loads return in order, so the record gather of the trip in between would wait for it.)
    PipeCtx<PLANE> pc = {d, Tl, Te, R, base, stride, ppt, last, validate};
    if (CULL) {
      // wave-uniform throughout: the word comes by a scalar load, the live mask and the three trip numbers of the pipeline live in SGPRs
      const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
      unsigned long long live = (ppt >= 64 ? ~0ull : ((1ull << ppt) - 1ull)) & ~cull[wv];
      auto pop = [ppt](unsigned long long& m) -> int {  // lowest live trip, ppt ("none") when the mask is empty
        const int i = m ? (int)__builtin_ctzll(m) : ppt;
        m &= m - 1ull;
        return i;
      };
      int t0 = pop(live), t1 = pop(live), t2 = pop(live);
      PointIn nxt = load_point<PLANE>(d, (unsigned int)min(base + t0 * stride, last));
      Probe<PLANE> pr = probe_point<FROZEN, PLANE>(d, nxt, base + t0 * stride, t0 < ppt, Tl, Te, R, validate, last);
      nxt = load_point<PLANE>(d, (unsigned int)min(base + t1 * stride, last));
      for (int step = 0; t0 < ppt; step++) {
        rotate_priority(step + prio_phase);
        pipe_trip_at<MODE, FROZEN, PLANE>(pc, pr, nxt, base + t1 * stride, t1 < ppt, base + t2 * stride, acc, wave_inliers, wave_skips);
        t0 = t1;
        t1 = t2;
        t2 = pop(live);
      }
    } else {
    PointIn nxt = first ? *first : load_point<PLANE>(d, (unsigned int)min(base, last));
    Probe<PLANE> pr = probe_point<FROZEN, PLANE>(d, nxt, base, ppt > 0, Tl, Te, R, validate, last);
    nxt = load_point<PLANE>(d, (unsigned int)min(base + stride, last));
    for (int it = 0; it < ppt; it++) {
      rotate_priority(it + prio_phase);
      pipe_trip<MODE, FROZEN, PLANE, TAILSKIP>(pc, pr, nxt, it, acc, wave_inliers, wave_skips);
    }
    }
  }

  if (loop_done_stamp && threadIdx.x == 0) *loop_done_stamp = __builtin_amdgcn_s_memrealtime();  // (device timeline of a resident session: the algebra of the last point is issued)
  // ---- block reduction: DPP wave sums -> LDS ----
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#if GLIM_AMD_ROW_SUMS_LDS
  // every 16-lane row of the wavefront leaves ITS sum (lanes 15 / 31 / 47 / 63); row_value adds the four as the row_bcast steps did.  Values that
  // exist once per wavefront (error-only sum, inlier count, skipped trips) go to row 3, zeros to the others: (v + 0) + (0 + 0) is v.
  const bool row_end = (lane & 15) == 15;
  const int rrow = lane >> 4;
  if (MODE == MODE_LINEARIZE) {
    wave_row_sums<NACC>(acc);
    if (row_end) {
#pragma unroll
      for (int j = 0; j < NACC; j++) s_red[wave][rrow][j] = acc[j];
    }
  } else {
    const float v = wave_sum_to_lane63(acc[27]);
    if (row_end) s_red[wave][rrow][27] = rrow == 3 ? v : 0.f;
  }
  if (row_end) {
    s_red[wave][rrow][28] = rrow == 3 ? (float)wave_inliers : 0.f;  // <= 64 * ppt: exact in FP32
    s_red[wave][rrow][29] = rrow == 3 ? (float)wave_skips : 0.f;    // <= ppt
  }
#else
  if (MODE == MODE_LINEARIZE) {
#if GLIM_AMD_STEP_MAJOR_SUMS
    wave_sums_to_lane63<NACC>(acc);  // (step-major: device_math.hpp)
    if (lane == 63) {
#pragma unroll
      for (int j = 0; j < NACC; j++) s_red[wave][0][j] = acc[j];
    }
#else
#pragma unroll
    for (int j = 0; j < NACC; j++) {
      const float v = wave_sum_to_lane63(acc[j]);
      if (lane == 63) s_red[wave][0][j] = v;
    }
#endif
  } else {
    const float v = wave_sum_to_lane63(acc[27]);
    if (lane == 63) s_red[wave][0][27] = v;
  }
  {
    if (lane == 63) s_red[wave][0][28] = (float)wave_inliers;  // <= 64 * ppt: exact in FP32
    if (lane == 63) s_red[wave][0][29] = (float)wave_skips;    // <= ppt
  }
#endif
  __syncthreads();
}

// value j of the block's partial row from the four wavefront sums (the same expression wherever a row is published: same bits)
__device__ __forceinline__ float wave_value(const RedRow* s_red, int w, int j) {
#if GLIM_AMD_ROW_SUMS_LDS
  return (s_red[w][3][j] + s_red[w][2][j]) + (s_red[w][1][j] + s_red[w][0][j]);  // lane 63 after row_bcast:15 and row_bcast:31
#else
  return s_red[w][0][j];
#endif
}
template <int MODE>
__device__ __forceinline__ float row_value(const RedRow* s_red, int j) {
  const bool live = (MODE == MODE_LINEARIZE) ? (j <= 29) : (j >= 27 && j <= 29);  // 0..26 sums, 27 error, 28 inliers, 29 skipped trips
  return live ? (wave_value(s_red, 0, j) + wave_value(s_red, 1, j)) + (wave_value(s_red, 2, j) + wave_value(s_red, 3, j)) : 0.f;
}

// tagged granules (see TAG_PIECES): thread p < 10 publishes values 3p .. 3p + 2 and the call's tag with ONE write-through 16-byte store
template <int MODE>
__device__ __forceinline__ void publish_row_tagged(const RedRow* s_red, char* rows16, size_t row, unsigned int tag) {
  if (threadIdx.x < TAG_PIECES) {
    const int j = 3 * (int)threadIdx.x;
    const v4i_t piece = {__float_as_int(row_value<MODE>(s_red, j)), __float_as_int(row_value<MODE>(s_red, j + 1)), __float_as_int(row_value<MODE>(s_red, j + 2)),
                         (int)tag};
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(rows16 + row * TAG_ROW_BYTES, 0, TAG_ROW_BYTES, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(piece, rsrc, (int)threadIdx.x * 16, 0, AUX_SC1);
  }
}

// The fused factor kernel.  MODE: linearise (28 sums) or error only.  FROZEN: residual at a separate evaluation pose with correspondences
// and Mahalanobis matrices frozen at the linearisation pose.  PLANE: plane-form source stream (24 B/pt) or general (36 B/pt).  INLINE: the
// pose and the descriptor of a single-factor set arrive in the kernel arguments (block b is chunk b of factor 0).
// One block = one (factor, chunk) row of the plan (compute_row).  block_offset: first plan row of this launch's segment; blocks_per_round:
// blocks the device takes per dispatch round (its CUs), for the priority phase.
// FUSED: the single-dispatch form of a small synchronous set.  The launch carries `fin_blocks` extra blocks behind the segment's rows; block
// seg_rows + i finalises factor finmap[fin_offset + i] (fused_finalize) while the row blocks publish their partial rows as tagged granules
// instead of plain rows: no second dispatch, no kernel boundary, no arrival counter.
// (FUSED variants serve small latency-bound sets that never fill the chip: they take the general kernel's 128-register budget, which keeps
//  the finalising branch's sixteen 16-byte loads in flight out of scratch.)
// CULL: the general segment of a large set, with the pre-cull words of cull_kernel (fa.cull_words; compute_row).
template <int MODE, bool FROZEN, bool PLANE, bool INLINE, bool FUSED, bool CULL = false>
__global__ __launch_bounds__(BLOCK, (PLANE && !FUSED) ? GLIM_AMD_MINW_PLANE : GLIM_AMD_MINW_GENERAL) void vgicp_kernel(const FactorDesc* __restrict__ descs, const double* __restrict__ poses_lin,
                                                          const double* __restrict__ poses_eval, const int2* __restrict__ blockmap,
                                                          float* __restrict__ partials, const InlineArgs ip, const FinalizeArgs fa, int block_offset,
                                                          int blocks_per_round, int seg_rows, int fin_offset) {
  // one LDS object: the row blocks use its first 4 x sizeof(RedRow) bytes (s_red), a finalising block all of it (s_part, s_sum)
  __shared__ double s_lds[FUSED ? (FIN_GROUPS + 1) * PARTIAL_STRIDE : (4 * (int)sizeof(RedRow)) / (int)sizeof(double)];
  static_assert(4 * sizeof(RedRow) <= (FIN_GROUPS + 1) * PARTIAL_STRIDE * sizeof(double), "s_red fits the finalisers' LDS object");
  RedRow* s_red = reinterpret_cast<RedRow*>(s_lds);
  if (FUSED && (int)blockIdx.x >= seg_rows) {
    const int ff = INLINE ? 0 : fa.finmap[fin_offset + ((int)blockIdx.x - seg_rows)];
    const FactorDesc fd = INLINE ? ip.d : descs[ff];
    fused_finalize(fd, ff, fa, MODE, reinterpret_cast<double (*)[PARTIAL_STRIDE]>(s_lds), s_lds + FIN_GROUPS * PARTIAL_STRIDE,
                   INLINE ? ip.m : poses_lin + 12 * (size_t)ff);
    return;
  }
  const int gblock = block_offset + (int)blockIdx.x;  // row of this block in the plan (the plane-form and the general segment are separate launches)
  // INLINE (single-factor sets): pose and descriptor are read from the kernel arguments (scalar loads from the kernarg segment), the block
  // map is the identity -- no dependent blockmap -> descriptor load chain in front of the first stream load
  const int2 bm = INLINE ? make_int2(0, gblock) : blockmap[gblock];
  const int f = bm.x;
  if (f < 0) return;  // padding block of the XCD-aware map
  const FactorDesc d = INLINE ? ip.d : descs[f];
  const double* Tl = INLINE ? ip.m : poses_lin + 12 * (size_t)f;
  const double* Te = FROZEN ? poses_eval + 12 * (size_t)f : Tl;
  // prio_phase: which of the CU's resident blocks this one is (dispatch is round robin over the CUs)
  compute_row<MODE, FROZEN, PLANE, CULL>(d, Tl, Te, bm.y, gblock / blocks_per_round, s_red, nullptr, CULL ? fa.cull_words + 4 * (size_t)blockIdx.x : nullptr);
  // a factor's partial rows are consecutive: row first_block + chunk (the finalisation reads them without an index table)
  const size_t row = (size_t)(d.first_block + bm.y);
  if (FUSED) {
    publish_row_tagged<MODE>(s_red, fa.rows16, row, fa.seq);
    return;
  }
  if (threadIdx.x < PARTIAL_STRIDE) partials[row * PARTIAL_STRIDE + threadIdx.x] = row_value<MODE>(s_red, (int)threadIdx.x);
}

// ---- pre-cull of the general segment -----------------------------------------------------------------------------------
// 17.8 % of the wavefront trips of the 256-submap all-pairs cost find no correspondence in any lane; the in-loop skip saves them the record
// gather and the algebra, but only AFTER the stream loads, the FP64 transform, the hash and the key gather (40 % of a trip).  Whether a trip can
// hit at all is decided by geometry the kernel does not need to look at point by point: the 64 stream-consecutive (Hilbert-ordered) points of a
// trip lie in a small box (ensure_chunk_boxes: 24 B per chunk, once per cloud), and the target map's occupied voxels are a few-KB bit mask
// (ensure_occupancy, once per map).  This pre-pass -- one THREAD per (plan row, wavefront), so the 31 M box tests of an evaluation are ordinary
// data-parallel work, ~60 instructions each, not wave-uniform work inside the hot loop -- moves every chunk box by the evaluation's pose
// (FP64, conservative: centre + |R| half-extent, padded by 1e-7 voxel against the 1e-13 the two FP64 evaluation orders can differ by) and marks
// the trips whose box touches no occupied cell.  Conservative by construction: a marked trip has no correspondence in any lane, so it would have
// added exact zeros -- results are bit-identical with the pre-cull on or off (tests/test_gpu_edge_cases.py).
struct CullDesc {
  const float* boxes;       // source cloud's chunk boxes (6 floats per 64 stream points), or null: nothing of this factor is culled
  const unsigned int* occ;  // target map's occupancy mask, or null
  int org[3], dim[3], shift, row_words;
};

__device__ __forceinline__ bool box_misses_mask(const CullDesc& cd, const float* __restrict__ b, const double (&T)[12], double inv_res) {
  const double cx = 0.5 * ((double)b[0] + (double)b[3]), cy = 0.5 * ((double)b[1] + (double)b[4]), cz = 0.5 * ((double)b[2] + (double)b[5]);
  const double hx = 0.5 * ((double)b[3] - (double)b[0]), hy = 0.5 * ((double)b[4] - (double)b[1]), hz = 0.5 * ((double)b[5] - (double)b[2]);
  int lo[3], hi[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const double q = T[4 * a] * cx + T[4 * a + 1] * cy + T[4 * a + 2] * cz + T[4 * a + 3];
    const double e = fabs(T[4 * a]) * hx + fabs(T[4 * a + 1]) * hy + fabs(T[4 * a + 2]) * hz;
    const double tl = fmin(fmax(floor((q - e) * inv_res - 1e-7), -2097152.0), 2097152.0), th = fmin(fmax(floor((q + e) * inv_res + 1e-7), -2097152.0), 2097152.0);
    lo[a] = max(((int)tl - cd.org[a]) >> cd.shift, 0);
    hi[a] = min(((int)th - cd.org[a]) >> cd.shift, cd.dim[a] - 1);
    if (lo[a] > hi[a]) return true;  // entirely outside the box of the occupied voxels
  }
  const int w0 = lo[0] >> 5, w1 = hi[0] >> 5;
  if ((long long)(hi[2] - lo[2] + 1) * (hi[1] - lo[1] + 1) * (w1 - w0 + 1) > 192) return false;  // a box this large (in cells) is not worth walking
  for (int z = lo[2]; z <= hi[2]; z++)
    for (int y = lo[1]; y <= hi[1]; y++) {
      const unsigned int* row = cd.occ + ((size_t)z * cd.dim[1] + y) * cd.row_words;
      for (int w = w0; w <= w1; w++) {
        unsigned int m = 0xffffffffu;
        if (w == w0) m &= 0xffffffffu << (lo[0] & 31);
        if (w == w1) m &= 0xffffffffu >> (31 - (hi[0] & 31));
        if (row[w] & m) return false;
      }
    }
  return true;
}

// One LANE per (plan row, wavefront, trip): lanes [k P2, (k + 1) P2) of a wavefront test the P2 = 2^log2p >= ppt trips of ONE (row, wavefront) pair,
// so all the box tests of a pair are in flight at once (the first version walked a pair's <= 32 trips one after the other in one thread: two dependent
// loads per trip, ~1.7 us each, 0.7 ms per evaluation of configs[3] -- more than the cull saved) and the pair's word is one ballot, no atomics.
__global__ __launch_bounds__(256) void cull_kernel(const FactorDesc* __restrict__ descs, const CullDesc* __restrict__ culls, const double* __restrict__ poses_lin,
                                                   const int2* __restrict__ blockmap, int row0, int rows, int log2p, unsigned long long* __restrict__ words,
                                                   unsigned long long* __restrict__ stats) {
  const int lane = (int)(threadIdx.x & 63);
  const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int pairs_per_wave = 64 >> log2p, it = lane & ((1 << log2p) - 1);
  const long long pair = wave * pairs_per_wave + (lane >> log2p);
  bool cull = false, has_points = false;
  if (pair < 4ll * rows) {
    const int r = (int)(pair >> 2), w = (int)(pair & 3);
    const int2 bm = blockmap[row0 + r];
    if (bm.x >= 0) {
      const int n = descs[bm.x].n, nb = descs[bm.x].num_blocks, ppt = descs[bm.x].ppt;
      const long long chunk = ((long long)it * nb + bm.y) * 4 + w;  // trip `it` of this wavefront covers stream points [64 chunk, 64 chunk + 64)
      if (it >= ppt || chunk * 64 >= n) {
        cull = true;  // no such trip / no point at all
      } else {
        has_points = true;
        const CullDesc cd = culls[bm.x];
        if (cd.boxes && cd.occ) {
          double T[12];
#pragma unroll
          for (int i = 0; i < 12; i++) T[i] = poses_lin[12 * (size_t)bm.x + i];
          cull = box_misses_mask(cd, cd.boxes + 6 * chunk, T, descs[bm.x].inv_res);
        }
      }
    } else {
      cull = true;  // padding row of the XCD-aware map: its block returns at once
    }
  }
  const unsigned long long culled_lanes = __ballot(cull), point_lanes = __ballot(has_points);
  if (it == 0 && pair < 4ll * rows) {
    const unsigned long long field = log2p == 6 ? ~0ull : ((1ull << (1 << log2p)) - 1ull);
    words[pair] = (culled_lanes >> ((lane >> log2p) << log2p)) & field;
  }
  // counters (measurement only, armed by glim_amd_factor_set_cull_stats: `stats` is null otherwise -- one atomic pair per WAVEFRONT on two words
  // was 1.1 M same-address atomics per evaluation of configs[3], i.e. longer than the factor kernel itself).  One pair per block, spread over 64 slots.
  if (stats) {
    __shared__ unsigned int s_cnt[2];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    if (lane == 0 && point_lanes) {
      atomicAdd(&s_cnt[0], (unsigned int)__popcll(culled_lanes & point_lanes));
      atomicAdd(&s_cnt[1], (unsigned int)__popcll(point_lanes));
    }
    __syncthreads();
    if (threadIdx.x < 2 && s_cnt[threadIdx.x]) atomicAdd(&stats[2 * (blockIdx.x & 63) + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
  }
}

// ---- resident form of the synchronous call --------------------------------------------------------------------------------------------
// A kernel that STAYS on the device between calls and takes its requests through host-mapped memory: no launch on the request path at all
// (tools/ubench/sync_floor.hip on this box: launch -> trivial kernel -> host-mapped word -> host wake-up 6.2 us; request word -> resident
// leader -> 512 worker blocks -> 512 granules back -> completion word 3.3 us).  For GLIM's odometry, which linearises the same small factor
// list again and again (odometry_estimation_gpu.cpp:383-385, once per optimiser iteration).
//   blocks 0 .. workers-1                 compute plan rows b, b + workers, ... (compute_row) and publish them as tagged granules
//   blocks workers .. workers + nf - 1    finalise one factor each (fused_finalize: same bits as every other form); block `workers` also LEADS:
//                                         it polls the request word in host memory, reads the poses the host left beside it and re-publishes
//                                         them in device memory as {double, tag} granules -- which is also the "go" signal for everyone else
// Tags of a resident session have bit 31 set (the launch-per-call forms count from 1), 0xffffffff = exit.  The leader exits on its own after
// `idle_polls` empty polls (a few milliseconds), so a device-wide synchronise elsewhere in the process can never wait for long, and every
// other wait is bounded as well: a session that loses its leader dies instead of hanging the device.
struct ResidentArgs {
  const FactorDesc* descs;
  const int2* blockmap;
  const int* finmap;
  int total_rows, num_factors, workers, blocks_per_round;
  char* rows16;                    // tagged partial rows (the plan's)
  char* rec16;                     // host-mapped record granules (the plan's)
  char* pose16;                    // device: replicas x num_factors x 12 pose granules {double, tag, 0}
  int replicas;                    // copies of the pose granules (block b watches copy b % replicas: 512 blocks polling ONE 192-byte spot is a hot
                                   // spot on one memory channel that slows everything else on the device)
  // host-mapped request lines of 64 bytes each: {7 doubles of the pose array, tag}.  The host writes every line's doubles, then every line's
  // tag; a 64-byte line is read as a unit, so a line whose tag is the new one carries the new doubles -- tag and pose travel in ONE PCIe
  // round trip (a separate request word costs a second one, ~1.2 us, before the first worker can start).
  const unsigned long long* h_lines;
  int num_lines;
  unsigned int* mail;              // host-mapped: [16] alive (1 while the kernel serves, 0 once it has left)
  unsigned int first_tag;          // the last tag served before this launch
  unsigned int idle_polls;         // leader: empty polls before it leaves
  // Device timeline of the LAST request served (glim_amd_debug_resident_timeline; null = off): s_memrealtime stamps (100 MHz), device memory.
  //   [0] leader: request seen in host memory        [1] leader: poses re-published on the device
  //   [2] finaliser of factor 0: pose seen           [3] ... every row of its factor summed (all tags arrived)      [4] ... record stored towards the host
  //   [5] ... its 32 group sums added   [6] ... its blocks rotated (finalize_tail)   [7] shader-clock ticks (s_memtime) between [2] and [4]
  //   [8 + 4 b + {0, 1, 2, 3}] worker block b: pose seen, first row computed (block-reduced), row granules published, point loop left (before the reduction)
  unsigned long long* timeline;
};
constexpr int TL_WORKER0 = 8;
__device__ __forceinline__ void tl_stamp(unsigned long long* tl, int slot) {
  if (tl && threadIdx.x == 0) tl[slot] = __builtin_amdgcn_s_memrealtime();
}
constexpr unsigned int RES_EXIT = 0xffffffffu;

// wave 0, lanes 0..11: wait for the 12 pose granules of factor f.  exact != 0: until their tag is `exact`; otherwise until it is a session tag
// different from `last`.  Returns the tag through s_tag (RES_EXIT when the session ends or the wait gives up) and the pose through s_pose.
// (Round 6 also had every block read the HOST's request lines of its own factor when a request was due, to save the leader's re-publication hop
//  -- 1.15 us median, 1.43 us for the last worker: 257 blocks reading host memory turned a 12 us call into a 385 us one and slowed a kernel launched
//  beside the session 32x; one reader is what the link serves well.  profiles/r06/probe/resident_direct_host_poll_refuted.json)
__device__ __forceinline__ void wait_pose(const ResidentArgs& ra, int f, unsigned int last, unsigned int exact, double* s_pose, unsigned int* s_tag) {
  if (threadIdx.x < 64) {
    const int lane = (int)threadIdx.x < 12 ? (int)threadIdx.x : 0;
    const int copy = (int)blockIdx.x % ra.replicas;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(ra.pose16 + ((size_t)copy * ra.num_factors + f) * 12 * 16, 0, 12 * 16, 0x00020000);
    v4i_t g;
    unsigned int tag = RES_EXIT;
    const unsigned int limit = ra.idle_polls * 64u + (1u << 16);
    for (unsigned int spins = 0; spins < limit; spins++) {
      g = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, 0, AUX_SC1_VOLATILE);
      const unsigned int t = (unsigned int)g.z;
      const bool mine = exact ? (t == exact || t == RES_EXIT) : ((t & 0x80000000u) != 0u && t != last);
      // every lane must see the SAME new tag (the leader writes the 12 granules of a factor with one store instruction, but they travel separately)
      const unsigned int t0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
      if (__all(mine && t == t0)) {
        tag = t0;
        break;
      }
      // back off: a request that follows the previous one within microseconds is seen at once; a session that has been quiet for a while
      // polls every few microseconds instead of hammering the fabric beside whatever else runs on the device
      if (spins < 32) __builtin_amdgcn_s_sleep(4);
      else if (spins < 256) __builtin_amdgcn_s_sleep(24);
      else __builtin_amdgcn_s_sleep(100);
    }
    if (threadIdx.x < 12) s_pose[threadIdx.x] = __longlong_as_double(((long long)(unsigned int)g.y << 32) | (long long)(unsigned int)g.x);
    if (threadIdx.x == 0) *s_tag = tag;
  }
  __syncthreads();
}

// PLANE_ONLY: every factor of the plan streams a plane-form cloud (the odometry's case).  That variant stays within the plane-form kernel's 96
// registers -- what a session HOLDS while it idles is what everything else on the device cannot use: at 168 registers (both stream forms in one
// kernel) two resident blocks per CU left room for ONE wave per SIMD of any other kernel, and an 8-factor launch beside an idle session took
// 29 instead of 12 us (tools/res_probe.py).
#ifndef GLIM_AMD_RES_MINW
#define GLIM_AMD_RES_MINW 4
#endif
template <bool PLANE_ONLY>
__global__ __launch_bounds__(BLOCK, PLANE_ONLY ? GLIM_AMD_RES_MINW : 3) void resident_kernel(const ResidentArgs ra) {
  __shared__ double s_lds[(FIN_GROUPS + 1) * PARTIAL_STRIDE];
  __shared__ double s_pose[12];
  __shared__ unsigned int s_tag;
  RedRow* s_red = reinterpret_cast<RedRow*>(s_lds);
  const int b = (int)blockIdx.x;
  const bool finaliser = b >= ra.workers, leader = b == ra.workers;
  unsigned int last = ra.first_tag;
  // the first real row of a worker decides which factor's pose granules it watches between requests (a worker that owns only padding rows of
  // an XCD-aware map watches factor 0: it still has to see the session end)
  int first_row = -1;
  if (!finaliser)
    for (int r = b; r < ra.total_rows && first_row < 0; r += ra.workers)
      if (ra.blockmap[r].x >= 0) first_row = r;
  for (;;) {
    if (leader) {
      // ---- wait for a request in host memory; re-publish the poses (device granules) or the end of the session.  Word w of the request
      // lines (8 per line: 7 doubles + tag) is polled by thread w -- the first LEAD_WORDS words in one sweep; a longer request is read in
      // further sweeps once its first lines carry the new tag.
      constexpr int LEAD_WORDS = BLOCK;
      const int words = ra.num_lines * 8;
      unsigned int req = last;
      unsigned long long w0 = 0;
      for (unsigned int idle = 0;; idle++) {
        bool ok = true;
        if ((int)threadIdx.x < words) {
          w0 = __hip_atomic_load(ra.h_lines + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          // the tag word of this thread's line sits in lane (l | 7) of the same wavefront
          const unsigned int t = (unsigned int)__shfl((unsigned int)w0, (int)((threadIdx.x & 63) | 7), 64);
          if (threadIdx.x == 7) s_tag = t;  // line 0's tag decides
          ok = t != last;
        }
        const int all_new = __syncthreads_and(ok ? 1 : 0);
        const unsigned int t0 = s_tag;
        // (every polled line must carry the SAME new tag: a sweep can straddle the host's update)
        bool same = true;
        if ((int)threadIdx.x < words) same = (unsigned int)__shfl((unsigned int)w0, (int)((threadIdx.x & 63) | 7), 64) == t0;
        const int consistent = __syncthreads_and(same ? 1 : 0);
        if (all_new && consistent) {
          req = t0;
          break;
        }
        if (idle >= ra.idle_polls) {
          req = RES_EXIT;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      if (req != RES_EXIT) tl_stamp(ra.timeline, 0);  // (the exit request must not overwrite the account of the last served one)
      const int nd = ra.num_factors * 12;
      const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(ra.pose16, 0, ra.replicas * nd * 16, 0x00020000);
      for (int base_w = 0; base_w < words; base_w += LEAD_WORDS) {
        const int w = base_w + (int)threadIdx.x;
        unsigned long long v = w0;
        bool stale = false;
        if (base_w > 0 && w < words && req != RES_EXIT) {
          // later lines: read until their own tag is the request's (bounded: the host wrote every tag before it wrote line 0's)
          stale = true;
          for (int tries = 0; tries < 1000; tries++) {
            v = __hip_atomic_load(ra.h_lines + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned int t = (unsigned int)__shfl((unsigned int)v, (int)((threadIdx.x & 63) | 7), 64);
            if (__all(t == req)) {
              stale = false;
              break;
            }
          }
        }
        // a line that never showed this request's tag is not part of a request: nothing of it is published under the tag -- the session ends
        // (exit tag) and the host's call is answered by the launch-per-call path
        if (base_w > 0 && __syncthreads_or(stale ? 1 : 0)) req = RES_EXIT;
        const int line = w >> 3, slot = w & 7, i = line * 7 + slot;  // index into the pose array
        if (w < words && slot < 7 && i < nd) {
          const long long bits = req != RES_EXIT ? (long long)v : 0ll;
          const v4i_t g = {(int)(bits & 0xffffffffll), (int)(bits >> 32), (int)req, 0};
          for (int c = 0; c < ra.replicas; c++) __builtin_amdgcn_raw_buffer_store_b128(g, rsrc, (c * nd + i) * 16, 0, AUX_SC1);
        }
      }
      if (req != RES_EXIT) tl_stamp(ra.timeline, 1);
      if (threadIdx.x == 0) s_tag = req;
      if (req == RES_EXIT) {
        if (threadIdx.x == 0) __hip_atomic_store(ra.mail + 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
      }
      __syncthreads();
    }
    if (finaliser) {
      const int f = ra.finmap ? ra.finmap[b - ra.workers] : 0;
      wait_pose(ra, f, last, leader ? s_tag : 0u, s_pose, &s_tag);
      const unsigned int tag = s_tag;
      if (tag == RES_EXIT) return;
      unsigned long long* const tlf = (b == ra.workers) ? ra.timeline : nullptr;  // (the finaliser of the plan's first factor keeps the account)
      tl_stamp(tlf, 2);
      const unsigned long long shader_ticks0 = tlf ? __builtin_amdgcn_s_memtime() : 0ull;  // (shader clock counter: with the 100 MHz stamps it gives the clock the session runs at)
      FinalizeArgs fa;
      fa.out = nullptr;
      fa.out_mirror = nullptr;
      fa.out_row_offset = 0;
      fa.done_counter = nullptr;
      fa.host_flag = nullptr;
      fa.seq = tag;
      fa.num_factors = ra.num_factors;
      fa.rows16 = ra.rows16;
      fa.trip_stats = nullptr;
      fa.rec16 = ra.rec16;
      fa.finmap = ra.finmap;
      fa.cull_words = nullptr;
      const FactorDesc d = ra.descs[f];
      fused_finalize<8>(d, f, fa, MODE_LINEARIZE, reinterpret_cast<double (*)[PARTIAL_STRIDE]>(s_lds), s_lds + FIN_GROUPS * PARTIAL_STRIDE, s_pose, tlf ? tlf + 3 : nullptr);
      tl_stamp(tlf, 4);
      if (tlf && threadIdx.x == 0) tlf[7] = __builtin_amdgcn_s_memtime() - shader_ticks0;
      __syncthreads();
      last = tag;
      continue;
    }
    // ---- worker: its rows of this request
    if (first_row < 0) {
      wait_pose(ra, 0, last, 0u, s_pose, &s_tag);  // nothing to compute: this block only follows the session
      if (s_tag == RES_EXIT) return;
      last = s_tag;
      __syncthreads();
      continue;
    }
    unsigned int tag = 0;
    int have = -1;  // factor whose pose Tl holds
    double Tl[12];
    for (int r = first_row; r < ra.total_rows; r += ra.workers) {
      const int2 bm = ra.blockmap[r];
      if (bm.x < 0) continue;
      // everything that does not depend on the pose is fetched BEFORE the wait: the descriptor and this lane's first point
      const FactorDesc d = ra.descs[bm.x];
      PointIn p0;
      {
        const unsigned int i0 = (unsigned int)min(bm.y * BLOCK + (int)threadIdx.x, max(d.n - 1, 0));
        if (PLANE_ONLY || d.plane) p0 = load_point<true>(d, i0);
        else p0 = load_point<false>(d, i0);
      }
      if (bm.x != have) {
        wait_pose(ra, bm.x, last, tag, s_pose, &s_tag);
        if (s_tag == RES_EXIT) return;
        tag = s_tag;
        have = bm.x;
        // the pose as wave-uniform scalars (the launch-per-call kernels read theirs from the kernel arguments / a uniform address)
#pragma unroll
        for (int i = 0; i < 12; i++) {
          const long long bits = __double_as_longlong(s_pose[i]);
          const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(bits & 0xffffffffll));
          const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(bits >> 32));
          Tl[i] = __longlong_as_double(((long long)hi << 32) | (long long)lo);
        }
      }
      unsigned long long* const tlw = (ra.timeline && r == first_row) ? ra.timeline + TL_WORKER0 + 4 * b : nullptr;
      tl_stamp(tlw, 0);
      if (PLANE_ONLY || d.plane) compute_row<MODE_LINEARIZE, false, true, false, GLIM_AMD_RES_TAILSKIP != 0>(d, Tl, Tl, bm.y, r / ra.blocks_per_round, s_red, &p0, nullptr, tlw ? tlw + 3 : nullptr);
      else compute_row<MODE_LINEARIZE, false, false, false, GLIM_AMD_RES_TAILSKIP != 0>(d, Tl, Tl, bm.y, r / ra.blocks_per_round, s_red, &p0, nullptr, tlw ? tlw + 3 : nullptr);
      tl_stamp(tlw, 1);
      publish_row_tagged<MODE_LINEARIZE>(s_red, ra.rows16, (size_t)(d.first_block + bm.y), tag);
      tl_stamp(tlw, 2);
      __syncthreads();  // s_red and s_pose are reused by the next row
    }
    last = tag;
  }
}

// Finalisation: one 256-thread block per factor.
__global__ __launch_bounds__(BLOCK) void finalize_kernel(const FactorDesc* __restrict__ descs, const float* __restrict__ partials, const FinalizeArgs fa,
                                                         int mode, const double* __restrict__ poses_lin, const InlineArgs ip) {
  __shared__ double s_part[FIN_GROUPS][PARTIAL_STRIDE];
  __shared__ double s_sum[PARTIAL_STRIDE];
  const int f = blockIdx.x;
  const FactorDesc d = ip.valid ? ip.d : descs[f];
  finalize_factor(d, f, partials, fa, mode, s_part, s_sum, ip.valid ? ip.m : poses_lin + 12 * (size_t)f);
}

// Finalisation of a large set of short factors (configs[3]: 32 640 factors of ~8 rows): one WAVEFRONT per factor, four per block, no block
// barrier.  One 256-thread block per factor is latency-bound there (two barriers, a 32-deep LDS sum and the serial rotation per block, sixteen
// resident rounds: 145 us per evaluation).  Lane j < 32 adds value j of the factor's rows in row order -- exactly the order finalize_factor's
// group sums give a factor of at most FIN_GROUPS rows, so both kernels produce the same bits; lanes 0..3 then rotate as there.
constexpr int FIN_WAVES = BLOCK / 64;
constexpr int FIN_SHORT_MIN_FACTORS = 2048;
__global__ __launch_bounds__(BLOCK) void finalize_short_kernel(const FactorDesc* __restrict__ descs, const float* __restrict__ partials, const FinalizeArgs fa,
                                                               int mode, const double* __restrict__ poses_lin) {
  __shared__ double s_sum[FIN_WAVES][PARTIAL_STRIDE];
  __shared__ double s_rot[FIN_WAVES][32];
  const int wave = threadIdx.x >> 6, t = threadIdx.x & 63;
  const int f = blockIdx.x * FIN_WAVES + wave;
  if (f >= fa.num_factors) return;
  const int first = descs[f].first_block, nb = descs[f].num_blocks;  // nb <= FIN_GROUPS (launch site)
  double Tl[12];
  if (t < 4) {
#pragma unroll
    for (int i = 0; i < 12; i++) Tl[i] = poses_lin[12 * (size_t)f + i];
  }
  if (t < PARTIAL_STRIDE) {
    double sum = 0.0;
    constexpr int INFLIGHT = 8;
    for (int c = 0; c < nb; c += INFLIGHT) {
      float v[INFLIGHT];
#pragma unroll
      for (int u = 0; u < INFLIGHT; u++) v[u] = partials[(size_t)(first + min(c + u, nb - 1)) * PARTIAL_STRIDE + t];
#pragma unroll
      for (int u = 0; u < INFLIGHT; u++)
        if (c + u < nb) sum += (double)v[u];
    }
    s_sum[wave][t] = sum;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  double* o = fa.out + ((size_t)fa.out_row_offset + f) * COMPACT;
  // (the record once more for an asynchronous caller's host array, glim_amd_multi: round 5 added the second store to finalize_tail only, so a
  //  piece of > 2048 short factors -- every piece of configs[3] -- left the host array untouched; found by round 6's virtual-device run)
  double* om = fa.out_mirror ? fa.out_mirror + ((size_t)fa.out_row_offset + f) * COMPACT : nullptr;
  if (t == 29 && fa.trip_stats && s_sum[wave][29] > 0.0) atomicAdd(&fa.trip_stats[f & 63], (unsigned long long)s_sum[wave][29]);
  double value = 0.0;
  bool mine = false;
  if (t == 0) {
    value = s_sum[wave][28];
    mine = true;
  }
  if (t == 1) {
    value = s_sum[wave][27];
    mine = true;
  }
  if (mode == MODE_LINEARIZE) {
    if (t < 4) rotate_part(t, s_sum[wave], Tl, s_rot[wave]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (mine) {
    o[t] = value;
    if (om) om[t] = value;
  }
  if (t < 27) {  // slots 2 .. 28 of the record, by lanes 0 .. 26
    double v = 0.0;
    if (mode == MODE_LINEARIZE) v = t < 21 ? s_rot[wave][acc_of_upper(t)] : (t < 24 ? s_rot[wave][t] : -s_rot[wave][t]);
    o[2 + t] = v;
    if (om) om[2 + t] = v;
  }
}

__global__ __launch_bounds__(BLOCK) void correspondence_kernel(FactorDesc d, const double* __restrict__ pose, int32_t* __restrict__ corr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.n) return;
  const float4 p4 = d.pts[i];
  double qx, qy, qz;
  transform_point_d(pose, (double)p4.x, (double)p4.y, (double)p4.z, qx, qy, qz);
  const int cx = fast_floor_d(qx * d.inv_res), cy = fast_floor_d(qy * d.inv_res), cz = fast_floor_d(qz * d.inv_res);
  bool hit = find_slot(d.buckets, d.num_buckets, pack_key(cx, cy, cz)) >= 0;
  if (hit && (d.flags & GLIM_AMD_FACTOR_SURFACE_VALIDATION) && d.normals) {
    const float4 nn = d.normals[i];
    const float rnx = (float)pose[0] * nn.x + (float)pose[1] * nn.y + (float)pose[2] * nn.z;
    const float rny = (float)pose[4] * nn.x + (float)pose[5] * nn.y + (float)pose[6] * nn.z;
    const float rnz = (float)pose[8] * nn.x + (float)pose[9] * nn.y + (float)pose[10] * nn.z;
    if (rnx * (float)qx + rny * (float)qy + rnz * (float)qz > 0.f) hit = false;
  }
  corr[4 * (size_t)i + 0] = cx;
  corr[4 * (size_t)i + 1] = cy;
  corr[4 * (size_t)i + 2] = cz;
  corr[4 * (size_t)i + 3] = hit ? 1 : -1;
}

struct OverlapTarget {
  const VoxelBucket* buckets;
  unsigned int num_buckets;
  int pad;
  double inv_res;
  double T[12];
};
// One overlap query = one source cloud against a list of (map, delta) targets (odometry_estimation_gpu.cpp:224-231, :248).
struct OverlapQuery {
  const float4* pts;
  int n;
  int first_target, num_targets;
  int first_block, num_blocks;  // blocks [first_block, first_block + num_blocks) of the launch walk this query's points
  int lanes_shift;              // 2^lanes_shift consecutive lanes share a point and split its targets (overlap_kernel)
};
constexpr int OVERLAP_INLINE_TARGETS = 16;
struct OverlapInline {  // single-query call: everything in the kernel arguments, nothing to upload
  OverlapQuery q;
  OverlapTarget t[OVERLAP_INLINE_TARGETS];
};

// K6: a point counts once if ANY (map_j, delta_j) contains it.  One launch answers any number of queries; completion without a stream
// synchronise or a read-back copy: every block adds (1 << 32 | its hits) to its query's 64-bit counter with ONE returning atomic, the block
// that sees all the others' arrivals writes the query's total into host-mapped memory, resets the counter for the next call and -- when it
// also completes the last query -- publishes the call's sequence number, on which the host spins.
template <bool INLINE>
__global__ __launch_bounds__(BLOCK) void overlap_kernel(const OverlapInline in, const OverlapQuery* __restrict__ queries, const OverlapTarget* __restrict__ targets,
                                                         const int* __restrict__ block_query, int num_queries, unsigned long long* __restrict__ counters,
                                                         unsigned int* __restrict__ queries_done, unsigned int* __restrict__ h_hits,
                                                         unsigned int* __restrict__ h_flag, unsigned int seq) {
  __shared__ int s_tmp[16];
  const int qi = INLINE ? 0 : block_query[blockIdx.x];
  const OverlapQuery q = INLINE ? in.q : queries[qi];
  const OverlapTarget* tg = INLINE ? in.t : targets + q.first_target;
  // A point's targets are looked up by G = 2^lanes_shift CONSECUTIVE LANES at once (G = 16 for the odometry's 15-keyframe query) instead of one
  // lane walking them one after the other: a lookup is a dependent chain of memory latencies (key line, perhaps a spill), and fifteen of them in
  // sequence made the 15-target call of a 10 000-pt frame a 30 us kernel -- latency, not work.  Lane l of a group takes targets l, l + G, ...; the
  // group's verdict is a slice of the wavefront's ballot; the group's first lane counts the point.  Same count: "any target contains the point".
  int mine = 0;
  const int chunk = (int)blockIdx.x - q.first_block;
  const int shift = q.lanes_shift, G = 1 << shift, sub_lane = (int)threadIdx.x & (G - 1);
  const int lane = (int)threadIdx.x & 63, group_base = lane & ~(G - 1);
  const unsigned long long group_mask = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << group_base;
  const long long items = (long long)q.n << shift;  // (point, lane-of-group) pairs
  const int rounds = (q.num_targets + G - 1) >> shift;
  for (long long base = (long long)chunk * BLOCK; base < items; base += (long long)q.num_blocks * BLOCK) {  // (block-uniform trip count: ballots stay whole)
    const long long idx = base + (long long)threadIdx.x;
    const bool live = idx < items;
    const int i = live ? (int)(idx >> shift) : 0;
    const float4 p4 = q.pts[i];
    bool group_hit = false;
    for (int r = 0; r < rounds; r++) {
      const int t = (r << shift) + sub_lane;
      bool hit = false;
      if (live && !group_hit && t < q.num_targets) {
        double qx, qy, qz;
        transform_point_d(tg[t].T, (double)p4.x, (double)p4.y, (double)p4.z, qx, qy, qz);
        hit = find_slot(tg[t].buckets, tg[t].num_buckets, voxel_key(qx, qy, qz, tg[t].inv_res)) >= 0;
      }
      group_hit = group_hit || (__ballot(hit) & group_mask) != 0ull;
    }
    if (live && group_hit && sub_lane == 0) mine++;
  }
  const int block_hits = block_reduce_i<2>(mine, s_tmp);
  if (threadIdx.x == 0) {
    const unsigned long long prev = atomicAdd(counters + qi, (1ull << 32) | (unsigned long long)(unsigned int)block_hits);
    if ((int)(prev >> 32) == q.num_blocks - 1) {
      counters[qi] = 0ull;  // every block of this query has arrived: nobody touches the counter again in this launch
      h_hits[qi] = (unsigned int)prev + (unsigned int)block_hits;
      __threadfence_system();
      const unsigned int done = INLINE ? 0u : atomicAdd(queries_done, 1u);
      if (INLINE || (int)done == num_queries - 1) {
        if (!INLINE) *queries_done = 0u;
        __hip_atomic_store(h_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

void hat3(const double* a, double* H) {
  H[0] = 0; H[1] = -a[2]; H[2] = a[1];
  H[3] = a[2]; H[4] = 0; H[5] = -a[0];
  H[6] = -a[1]; H[7] = a[0]; H[8] = 0;
}

}  // namespace


// -----------------------------------------------------------------------------------------------------------------
// plan management
// -----------------------------------------------------------------------------------------------------------------
namespace glim_amd {

void resident_release(glim_amd_ctx* ctx, FactorPlan* plan);  // ends the resident session (below) of this context / plan, if any

namespace {

constexpr size_t PLAN_CACHE_MAX = 16;    // idle plans kept per context
constexpr int RESIDENT_MAX_FACTORS = 64;  // sets a resident session may serve (vgicp.hip "resident sessions")
constexpr int FUSED_MAX_FACTORS = 1024, FUSED_MAX_ROWS = 16384;  // sets that may take the single-dispatch form (2.6 MB of tagged rows at most)
constexpr size_t HOST_POSES_MAX = 256;   // synchronous sets up to this many factors: poses read by the kernels from host-mapped memory (96 B per factor over PCIe)

// The plan's device blocks, pinned blocks and events go back where they came from; the plan object stays (plan_build fills it again).
// Caller: no session serves the plan and nothing enqueued is still using the buffers (plan_idle).
void plan_release_cull(FactorPlan* p) {
  if (p->d_cull_descs) (void)pool_free(p->d_cull_descs);
  if (p->d_cull_words) (void)pool_free(p->d_cull_words);
  if (p->d_cull_stats) (void)pool_free(p->d_cull_stats);
  p->d_cull_descs = nullptr;
  p->d_cull_words = nullptr;
  p->d_cull_stats = nullptr;
  p->cull = false;
  p->cull_count = false;
}
void plan_release_buffers(FactorPlan* p) {
  plan_release_cull(p);
  if (p->d_upload) (void)pool_free(p->d_upload);  // d_descs | d_blockmap | d_finmap
  if (p->h_upload) (void)pinned_free(p->h_upload);
  if (p->d_zeroed) (void)pool_free(p->d_zeroed);  // d_done | d_trip_stats | d_rows16
  if (p->d_partials) (void)pool_free(p->d_partials);
  if (p->d_poses) (void)pool_free(p->d_poses);
  if (p->d_compact) (void)pool_free(p->d_compact);
  if (p->h_poses) (void)pinned_free(p->h_poses);
  if (p->h_compact) (void)pinned_free(p->h_compact);
  if (p->h_flag) (void)pinned_free(p->h_flag);
  if (p->h_rec16) (void)pinned_free(p->h_rec16);
  p->d_upload = p->h_upload = p->d_zeroed = nullptr;
  p->d_partials = nullptr;
  p->d_poses = p->d_compact = p->h_poses = p->h_poses_dev = p->h_compact = p->h_compact_dev = nullptr;
  p->h_flag = p->h_flag_dev = nullptr;
  p->h_rec16 = p->h_rec16_dev = nullptr;
  p->d_descs = nullptr;
  p->d_blockmap = nullptr;
  p->d_finmap = nullptr;
  p->d_done = nullptr;
  p->d_trip_stats = nullptr;
  p->d_rows16 = nullptr;
  for (int i = 0; i < FactorPlan::POSE_RING; i++) {
    if (p->pose_events[i]) (void)hipEventDestroy(p->pose_events[i]);
    p->pose_events[i] = nullptr;
    p->pose_pending[i] = false;
  }
  if (p->poses_free_event) (void)hipEventDestroy(p->poses_free_event);
  p->poses_free_event = nullptr;
  p->poses_free_pending = false;
  p->cap_factors = p->cap_blocks = 0;
  p->alloc_rows = -1;
}

// a resident session that serves this plan ends; nothing enqueued earlier (asynchronous entry points included) is still using its buffers
void plan_idle(FactorPlan* p) {
  resident_release(nullptr, p);
  if (p->maybe_busy && p->last_stream) (void)hipStreamSynchronize(p->last_stream);
  p->maybe_busy = false;
}

void plan_free(FactorPlan* p) {
  if (!p) return;
  plan_idle(p);
  plan_release_buffers(p);
  delete p;
}

template <class T>
bool host_device_view(T* host, T** dev) {
  *dev = nullptr;
  if (hipHostGetDevicePointer(reinterpret_cast<void**>(dev), host, 0) != hipSuccess) {
    (void)hipGetLastError();
    *dev = nullptr;
    return false;
  }
  return true;
}

}  // namespace

void ctx_release_factor_resources(glim_amd_ctx* ctx) {
  resident_release(ctx, nullptr);
  for (FactorPlan* p : ctx->plan_cache) plan_free(p);
  ctx->plan_cache.clear();
  if (ctx->ov_counters) (void)pool_free(ctx->ov_counters);
  if (ctx->ov_host) (void)pinned_free(ctx->ov_host);
  ctx->ov_counters = nullptr;
  ctx->ov_host = nullptr;
  ctx->ov_host_dev = nullptr;
}

// The set gives up its plan: parked in the context's cache (most recent first) for the next set with the same factor list.
void factor_set_park_plan(glim_amd_factor_set* set) {
  FactorPlan* p = set->plan;
  set->plan = nullptr;
  if (!p) return;
  glim_amd_ctx* ctx = set->ctx;
  if (!ctx->diag.plan_cache) {
    plan_free(p);
    return;
  }
  ctx->plan_cache.insert(ctx->plan_cache.begin(), p);
  while (ctx->plan_cache.size() > PLAN_CACHE_MAX) {
    plan_free(ctx->plan_cache.back());
    ctx->plan_cache.pop_back();
  }
}

namespace {

// Build the device plan: factor descriptors, chunking, and the block -> (factor, chunk) map.
// Factors whose source cloud is plane-form take the 24 B/pt kernel, the others the general 36 B/pt kernel -- decided per factor, so one
// cloud with averaged covariances (a merged submap) does not demote the rest of the set; the two groups occupy separate row segments of
// the plan and are launched separately.
// Workgroup b is dispatched to XCD b % 8 (observed, MI355X_MICROARCH.md "Workgroup dispatch"); when a segment holds enough factors,
// all chunks of one factor are given rows of one residue class so that the factor's voxel table stays in a single XCD's 4 MiB L2.
// This is a speed-only choice: any placement is correct.
int plan_upload(glim_amd_factor_set* set, FactorPlan* plan) {
  if (plan->uploaded) return GLIM_AMD_OK;
  const size_t nf = plan->h_descs.size();
  if (nf > 0) {
    // the three tables travel as one copy out of the plan's pinned block (the block lives as long as the plan: no synchronise needed)
    char* h = plan->h_upload;
    memcpy(h + ((char*)plan->d_descs - plan->d_upload), plan->h_descs.data(), nf * sizeof(FactorDesc));
    memcpy(h + ((char*)plan->d_blockmap - plan->d_upload), plan->h_blockmap.data(), plan->h_blockmap.size() * sizeof(int2));
    if (plan->d_finmap) memcpy(h + ((char*)plan->d_finmap - plan->d_upload), plan->h_finmap.data(), plan->h_finmap.size() * sizeof(int));
    GA_HIP(hipMemcpyAsync(plan->d_upload, h, plan->upload_bytes, hipMemcpyHostToDevice, set->stream));
    plan->last_stream = set->stream;
    plan->maybe_busy = true;  // (plan_free waits for the copy before the pinned image goes back to its pool)
  }
  plan->uploaded = true;
  return GLIM_AMD_OK;
}

int plan_build(glim_amd_factor_set* set, FactorPlan* plan) {
  const int nf = (int)set->entries.size();
  glim_amd_ctx* ctx = set->ctx;
  const Diag& diag = ctx->diag;
  const bool allow_plane = diag.plane != 0;
  // Grid sizing: the plane-form and the general factors are separate launches (segments); each aims for a WHOLE NUMBER of resident sets of
  // blocks (num_cus x waves per SIMD of that kernel variant) with equal work each, and for at most PPT_MAX points per thread; each factor
  // gets blocks in proportion to its points.  One resident set where that keeps a thread below 2 PPT_MAX points (every lane then reduces its
  // 28 accumulators exactly once: two rounds of 26-point threads cost the 128-factor launch 3 % against one round of 51-point threads);
  // otherwise shorter blocks in k >= 3 full rounds: a block that walks a whole 62 k-pt submap (243 points per
  // thread) made the 256-submap cost 12.2 ms against 10.9 ms at 32 points per thread, and the 380-factor sub-mapping bundle 0.24 against
  // 0.21 ms, while a grid that is NOT a multiple of the resident set (2048 or 4096 blocks for 1280 slots) costs the 128-factor launch 13 %
  // in a half-empty last round (profiles/r03/probe/ppt_sweep.txt).
  constexpr long long PPT_MAX = 32;
  long long seg_points[2] = {0, 0};
  long long seg_target[2] = {(long long)std::max(1, ctx->num_cus) * GLIM_AMD_MINW_PLANE, (long long)std::max(1, ctx->num_cus) * GLIM_AMD_MINW_GENERAL};
  const int forced_ppt = diag.ppt;

  plan->built_plane = diag.plane;
  plan->built_ppt = diag.ppt;
  plan->built_cull = diag.cull;
  plan->built_small_rows = diag.small_rows;
  plan->h_descs.assign(nf, FactorDesc());
  plan->h_finmap.clear();
  std::vector<int> nblocks(nf);
  for (int f = 0; f < nf; f++) {
    const auto& e = set->entries[f];
    glim_amd_cloud* src = const_cast<glim_amd_cloud*>(e.source);  // lazily built stream copies are a cache, not a change of the cloud
    if (!allow_plane && src->plane_form && !src->gs0) {
      // diagnostic switch: run a plane-form cloud through the general kernel (its covariance arrays hold the same matrix)
      src->plane_form = false;
      const int rc = ensure_factor_streams(src, ctx, set->stream);
      src->plane_form = true;
      GA_TRY(rc);
    } else {
      GA_TRY(ensure_factor_streams(src, ctx, set->stream));
    }
    FactorDesc& d = plan->h_descs[f];
    d.pts = src->pts;
    d.normals = src->has_normals ? src->normals : nullptr;
    d.plane = (allow_plane && src->plane_form && src->pn4 && src->n2) ? 1 : 0;
    if (d.plane) {
      d.s0 = src->pn4;
      d.s1 = src->n2;
      d.s2 = nullptr;
      d.sn = nullptr;
    } else {
      d.s0 = src->gs0;
      d.s1 = src->gs1;
      d.s2 = src->gs2;
      d.sn = src->gsn;
    }
    d.buckets = e.target->buckets;
    d.num_buckets = e.target->num_buckets;
    if (GLIM_AMD_PLANE_SM && d.plane) {
      glim_amd_voxelmap* tm = const_cast<glim_amd_voxelmap*>(e.target);  // the lazily built view is a cache, not a change of the map
      GA_TRY(ensure_plane_view(tm, set->stream));
      d.buckets = tm->buckets_sm;
    }
    d.n = (int)src->n;
    d.inv_res = e.target->inv_resolution;
    d.res = e.target->resolution;
    d.flags = e.flags;
    seg_points[d.plane ? 0 : 1] += d.n;
  }
  for (int seg = 0; seg < 2; seg++) {
    const long long resident = seg_target[seg], fine = (seg_points[seg] + BLOCK * PPT_MAX - 1) / (BLOCK * PPT_MAX);
    if (fine > 2 * resident) seg_target[seg] = ((fine + resident - 1) / resident) * resident;  // (up to two rounds' worth of points: one round of longer blocks)
  }
  // Small sets (the odometry's: one to a few dozen factors of a frame each) are latency-bound, and their synchronous linearisation may be
  // served by a resident session whose 2 x CUs worker blocks take one row each: the smallest points-per-thread <= 8 that brings the plan
  // down to that many rows (XCD padding included) is used for every factor -- 34 factors of 10 000 points: 4 points per thread, 400 rows,
  // 18.2 us per resident call against 21.9 us with 1 360 one-point rows (and 21.3 against 22.0 us launch-per-call); a single 131 072-pt
  // factor keeps its 512 one-point rows.  Larger sets keep the throughput rule below.
  int small_set_ppt = 0;
  if (!forced_ppt && nf >= 1 && nf <= RESIDENT_MAX_FACTORS) {
    // Round 6: ... and no FACTOR gets more rows than one per compute unit.  The device timeline of the resident call
    // (glim_amd_debug_resident_timeline) showed the rows of a 131 072-pt factor computed by 4.6 us after the request whether a thread takes one
    // point or two (the second rides in the software pipeline), while the factor's finalising block pays for every row it has to collect:
    // 512 -> 256 rows took the synchronous call 13.7 -> 11.4 us on the same box, the single-dispatch form 14.9 -> 14.0, and a session of 257
    // blocks holds ONE wave slot per SIMD instead of two -- a 128-factor launch beside it pays 1.04x instead of 1.39x
    // (profiles/r06/probe/sync_rows_ab.json).  The odometry's 34-factor sets keep their 4 points per thread: planned into 256 rows in TOTAL they
    // were 6 % slower (8 points per thread; their factors own 10 rows each, nothing for a finaliser to wait for: sync_rows_34_factor_ab.json).
    // diag small_rows=<n> overrides the per-factor cap.
    const long long cap = 2ll * std::max(1, ctx->num_cus);
    const long long factor_cap = diag.small_rows > 0 ? (long long)diag.small_rows : 1ll * std::max(1, ctx->num_cus);
    for (int p = 1; p <= 8 && !small_set_ppt; p++) {
      long long rows = 0;
      for (int seg = 0; seg < 2; seg++) {
        std::vector<long long> nb;
        for (int f = 0; f < nf; f++)
          if ((plan->h_descs[f].plane != 0) == (seg == 0)) nb.push_back(std::max(1, (plan->h_descs[f].n + BLOCK * p - 1) / (BLOCK * p)));
        for (long long b : nb)
          if (b > factor_cap) rows = cap + 1;  // (a factor with too many rows at this p: try the next)
        if (nb.size() < 16) {
          for (long long b : nb) rows += b;
        } else {  // the XCD-aware map pads every XCD's list to the longest one
          long long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          for (long long b : nb) *std::min_element(load, load + 8) += b;
          rows += 8 * *std::max_element(load, load + 8);
        }
      }
      if (rows <= cap) small_set_ppt = p;
    }
  }
  for (int f = 0; f < nf; f++) {
    FactorDesc& d = plan->h_descs[f];
    int ppt = forced_ppt ? forced_ppt : small_set_ppt;
    if (!ppt) {
      const long long target_blocks = seg_target[d.plane ? 0 : 1], total_points = seg_points[d.plane ? 0 : 1];
      const long long share = std::max(1ll, (target_blocks * (long long)d.n + total_points / 2) / std::max(1ll, total_points));
      ppt = (int)std::max(1ll, std::min(256ll, ((long long)d.n + share * BLOCK - 1) / (share * BLOCK)));
    }
    d.ppt = ppt;
    nblocks[f] = std::max(1, (d.n + BLOCK * ppt - 1) / (BLOCK * ppt));
    d.num_blocks = nblocks[f];
  }
  plan->points_per_thread = nf ? plan->h_descs[0].ppt : 1;
  plan->max_rows_per_factor = nf ? *std::max_element(nblocks.begin(), nblocks.end()) : 0;

  // block map: the plane-form segment first, then the general one
  std::vector<int2> blockmap;
  int seg_rows[2] = {0, 0};
  for (int seg = 0; seg < 2; seg++) {  // seg 0: plane-form factors, seg 1: the others
    std::vector<int> fs;
    for (int f = 0; f < nf; f++)
      if ((plan->h_descs[f].plane != 0) == (seg == 0)) fs.push_back(f);
    // Locality order: factors that stream the SAME source cloud become neighbours in the plan (sources in order of first appearance, the
    // caller's order inside a group).  The resident blocks are the next ~1000 rows of the plan, so neighbours run together and all but
    // the first of them find the 2 MB source stream in L2 / the 256 MiB Infinity Cache instead of HBM: the all-pairs cost of 256 merged
    // submaps (32 640 factors, 255 uses of every cloud) takes 13.7 ms in this order against 16.5 ms target-major and 15.1 ms in 32 x 32
    // tiles (profiles/r02/probe/global256_pair_order.jsonl; pinning a source group to ONE XCD was measured slower, 14.3 ms).  Results do not depend
    // on the order (every factor owns its rows).
    {
      std::unordered_map<const void*, int> first_use;
      std::vector<int> group(nf, 0);
      for (int f : fs) group[f] = first_use.emplace(plan->h_descs[f].s0, f).first->second;
      std::stable_sort(fs.begin(), fs.end(), [&](int a, int b) { return group[a] < group[b]; });
    }
    const size_t base = blockmap.size();
    if (fs.size() < 16) {
      for (int f : fs)
        for (int c = 0; c < nblocks[f]; c++) blockmap.push_back(make_int2(f, c));
    } else {
      std::vector<std::vector<int2>> per_xcd(8);
      long long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int f : fs) {
        int x = 0;
        for (int k = 1; k < 8; k++)
          if (load[k] < load[x]) x = k;
        load[x] += nblocks[f];
        for (int c = 0; c < nblocks[f]; c++) per_xcd[x].push_back(make_int2(f, c));
      }
      size_t longest = 0;
      for (auto& v : per_xcd) longest = std::max(longest, v.size());
      blockmap.resize(base + longest * 8, make_int2(-1, 0));
      for (int x = 0; x < 8; x++)
        for (size_t j = 0; j < per_xcd[x].size(); j++) blockmap[base + j * 8 + x] = per_xcd[x][j];
    }
    seg_rows[seg] = (int)(blockmap.size() - base);
    // the factors the trailing blocks of this segment's single-dispatch launch finalise, in plan order
    plan->fin_count[seg] = (int)fs.size();
    for (int f : fs) plan->h_finmap.push_back(f);
  }
  // a factor's partial rows are consecutive: row first_block + chunk, whatever block of the map computes the chunk
  long long total_blocks = 0;
  for (int f = 0; f < nf; f++) {
    plan->h_descs[f].first_block = (int)total_blocks;
    total_blocks += nblocks[f];
  }

  plan->plane_rows = seg_rows[0];
  plan->total_rows = (int)blockmap.size();
  const size_t nfa = (size_t)std::max(1, nf), nba = std::max<size_t>(1, blockmap.size());
  // single-dispatch form (small synchronous sets only): tagged rows, every tag 0 = "no call yet" (sequence numbers start at 1)
  const bool fused_form = nf >= 1 && nf <= FUSED_MAX_FACTORS && total_blocks <= FUSED_MAX_ROWS;
  auto up16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
  const size_t o_map = up16(nfa * sizeof(FactorDesc)), o_fin = o_map + up16(nba * sizeof(int2));
  const size_t upload_bytes = o_fin + (fused_form ? up16((size_t)nf * sizeof(int)) : 0);
  const size_t o_trip = 64, o_rows = o_trip + 64 * sizeof(unsigned long long);
  // A plan object that comes with buffers is the one the cache was about to evict (factor_set_prepare): GLIM's odometry asks for a NEW list with
  // every frame -- the new cloud against the same number of window frames and keyframes --, so the evicted plan has exactly the shape of the
  // one being built and its ten blocks, four host views and events serve again (round 5: 12 us of a live frame's first linearisation were
  // allocation, views and the clearing memset).  Tags keep counting (poll_seq is not reset), so the tagged rows and record granules of the
  // previous life can never read as this one's and nothing needs clearing; the skipped-trip counters are cleared when the plan has general rows.
  const bool reuse = plan->d_upload && plan->cap_factors == nfa && plan->cap_blocks == nba && plan->alloc_rows == total_blocks && plan->fused_alloc == fused_form &&
                     plan->upload_bytes == upload_bytes && plan->h_flag && plan->h_poses_dev && plan->h_compact_dev && (!fused_form || plan->h_rec16);
  if (plan->d_upload && !reuse) plan_release_buffers(plan);
  if (reuse) {
    if (seg_rows[1] > 0) GA_HIP(hipMemsetAsync(plan->d_trip_stats, 0, 64 * sizeof(unsigned long long), set->stream));
  } else {
    {  // descriptors | block map | finaliser map: one device block, one pinned image, one copy (plan_upload)
      plan->upload_bytes = upload_bytes;
      GA_HIP(pool_malloc(&plan->d_upload, plan->upload_bytes));
      GA_HIP(pinned_malloc(&plan->h_upload, plan->upload_bytes));
      memset(plan->h_upload, 0, plan->upload_bytes);
    }
    {  // completion counter | skipped-trip counters | tagged rows: one block, one memset
      const size_t zero_bytes = o_rows + (fused_form ? (size_t)total_blocks * TAG_ROW_BYTES : 0);
      GA_HIP(pool_malloc(&plan->d_zeroed, zero_bytes));
      GA_HIP(hipMemsetAsync(plan->d_zeroed, 0, zero_bytes, set->stream));
    }
    GA_HIP(pool_malloc(&plan->d_partials, (size_t)std::max(1ll, total_blocks) * PARTIAL_STRIDE * sizeof(float)));
    GA_HIP(pool_malloc(&plan->d_poses, nfa * 24 * sizeof(double)));
    GA_HIP(pool_malloc(&plan->d_compact, nfa * COMPACT * sizeof(double)));
    if (fused_form) {
      const size_t rec_bytes = (size_t)nf * COMPACT * 16;
      if (pinned_malloc(&plan->h_rec16, rec_bytes) == hipSuccess) {
        memset(plan->h_rec16, 0, rec_bytes);
        if (!host_device_view(plan->h_rec16, &plan->h_rec16_dev)) {
          (void)pinned_free(plan->h_rec16);
          plan->h_rec16 = nullptr;
        }
      } else {
        (void)hipGetLastError();
        plan->h_rec16 = nullptr;
      }
    }
    if (pinned_malloc(&plan->h_flag, 64) == hipSuccess) {
      *plan->h_flag = 0;
      if (!host_device_view(plan->h_flag, &plan->h_flag_dev)) {
        (void)pinned_free(plan->h_flag);
        plan->h_flag = nullptr;
      }
    } else {
      (void)hipGetLastError();
      plan->h_flag = nullptr;
    }
    GA_HIP(pinned_malloc(&plan->h_poses, (size_t)FactorPlan::POSE_RING * nfa * 24 * sizeof(double)));
    (void)host_device_view(plan->h_poses, &plan->h_poses_dev);
    GA_HIP(pinned_malloc(&plan->h_compact, nfa * COMPACT * sizeof(double)));
    (void)host_device_view(plan->h_compact, &plan->h_compact_dev);
    plan->poll_seq = 0;
  }
  plan->d_descs = reinterpret_cast<FactorDesc*>(plan->d_upload);
  plan->d_blockmap = reinterpret_cast<int2*>(plan->d_upload + o_map);
  plan->d_finmap = fused_form ? reinterpret_cast<int*>(plan->d_upload + o_fin) : nullptr;
  plan->d_done = reinterpret_cast<int*>(plan->d_zeroed);
  plan->d_trip_stats = reinterpret_cast<unsigned long long*>(plan->d_zeroed + o_trip);
  plan->d_rows16 = fused_form ? plan->d_zeroed + o_rows : nullptr;
  plan->alloc_rows = total_blocks;
  plan->fused_alloc = fused_form;
  plan->recycled = reuse;
  plan->sync_linearize_calls = 0;
  plan->cap_factors = nfa;
  plan->cap_blocks = nba;
  plan->h_blockmap = std::move(blockmap);
  plan->uploaded = false;
  // Descriptors and block map go to the device when a launch first needs them THERE (plan_upload): a single-factor set linearised
  // synchronously takes both through the kernel arguments, so the per-frame "new cloud, new map, one factor" pattern of the odometry front
  // end builds its plan without a single transfer.  The host copies live in the plan, so no synchronise is needed either way.
  plan->last_stream = set->stream;
  plan->maybe_busy = false;
  if (nf > 1) GA_TRY(plan_upload(set, plan));
  // pre-cull of the general segment: large asynchronous-sized sets only (the pre-pass is one more launch, ~5 us of stream latency + ~2 ns per
  // row: a set whose general segment runs for less than ~150 us would pay more than it gets back), every general factor at <= 64 points per thread
  plan_release_cull(plan);
  constexpr int CULL_MIN_ROWS = 16384;
  bool want_cull = diag.cull && seg_rows[1] > 0 && (seg_rows[1] >= CULL_MIN_ROWS || diag.cull == 2);  // (cull=2: any size, for the tests; single-dispatch launches never use it)
  int cull_ppt = 1;
  for (int f = 0; f < nf && want_cull; f++)
    if (!plan->h_descs[f].plane) {
      if (plan->h_descs[f].ppt > 64) want_cull = false;
      cull_ppt = std::max(cull_ppt, plan->h_descs[f].ppt);
    }
  plan->cull_log2p = 0;
  while ((1 << plan->cull_log2p) < cull_ppt) plan->cull_log2p++;  // lanes per (row, wavefront) pair of the pre-pass: the next power of two >= ppt
  if (want_cull) {
    std::vector<CullDesc> cds((size_t)nf);
    for (int f = 0; f < nf; f++) {
      CullDesc& cd = cds[(size_t)f];
      memset(&cd, 0, sizeof(cd));
      if (plan->h_descs[f].plane) continue;
      glim_amd_cloud* src = const_cast<glim_amd_cloud*>(set->entries[f].source);
      glim_amd_voxelmap* tm = const_cast<glim_amd_voxelmap*>(set->entries[f].target);  // (the mask is a cache, not a change of the map)
      if (plan->h_descs[f].s0 != src->gs0) continue;  // (not the cloud's own general stream: no boxes for it)
      GA_TRY(ensure_chunk_boxes(src, set->stream));
      GA_TRY(ensure_occupancy(tm, set->stream));
      if (!tm->occ) continue;
      cd.boxes = src->gbox;
      cd.occ = tm->occ;
      for (int a = 0; a < 3; a++) {
        cd.org[a] = tm->occ_org[a];
        cd.dim[a] = tm->occ_dim[a];
      }
      cd.shift = tm->occ_shift;
      cd.row_words = tm->occ_row_words;
    }
    hipError_t e = pool_malloc(&plan->d_cull_descs, cds.size() * sizeof(CullDesc));
    if (e == hipSuccess) e = pool_malloc(&plan->d_cull_words, (size_t)seg_rows[1] * 4 * sizeof(unsigned long long));
    if (e == hipSuccess) e = pool_malloc(&plan->d_cull_stats, 128 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemcpyAsync(plan->d_cull_descs, cds.data(), cds.size() * sizeof(CullDesc), hipMemcpyHostToDevice, set->stream);
    if (e == hipSuccess) e = hipMemsetAsync(plan->d_cull_stats, 0, 128 * sizeof(unsigned long long), set->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(set->stream);  // (`cds` is pageable and goes out of scope)
    if (e != hipSuccess) {
      (void)hipGetLastError();
      plan_release_cull(plan);  // a cache of hints, not a requirement: the plain kernel computes the same bits
    } else {
      plan->cull = true;
    }
  }
  return GLIM_AMD_OK;
}

void make_key(const glim_amd_factor_set* set, std::vector<PlanKey>& key) {
  key.resize(set->entries.size());
  for (size_t f = 0; f < key.size(); f++) key[f] = PlanKey{set->entries[f].target->uid, set->entries[f].source->uid, set->entries[f].flags, 0u};
}

}  // namespace

// Make set->plan the device plan of set->entries.  Caller holds ctx->mu.  Steady state (nothing changed since the last call): two
// compares.  Changed list: the context's cache of idle plans is searched for the same (map, cloud, flags) list first -- GLIM's
// clear -> add -> linearize per optimiser iteration then costs no allocation, no upload and no synchronise.
int factor_set_prepare(glim_amd_factor_set* set) {
  glim_amd_ctx* ctx = set->ctx;
  const uint64_t epoch = global_mutation_epoch().load();
  const Diag& diag = ctx->diag;
  if (!set->dirty && set->plan && set->seen_epoch == epoch && set->plan->built_plane == diag.plane && set->plan->built_ppt == diag.ppt && set->plan->built_cull == diag.cull && set->plan->built_small_rows == diag.small_rows) return GLIM_AMD_OK;
  std::vector<PlanKey> key;
  make_key(set, key);
  auto usable = [&](const FactorPlan* p) { return p->key == key && p->built_plane == diag.plane && p->built_ppt == diag.ppt && p->built_cull == diag.cull && p->built_small_rows == diag.small_rows; };
  if (set->plan && !usable(set->plan)) factor_set_park_plan(set);
  if (!set->plan && diag.plan_cache) {
    for (size_t i = 0; i < ctx->plan_cache.size(); i++) {
      if (!usable(ctx->plan_cache[i])) continue;
      set->plan = ctx->plan_cache[i];
      ctx->plan_cache.erase(ctx->plan_cache.begin() + (long)i);
      // work a previous owner enqueued asynchronously on another stream must not overlap this set's use of the same buffers
      if (set->plan->maybe_busy && set->plan->last_stream && set->plan->last_stream != set->stream) {
        GA_HIP(hipStreamSynchronize(set->plan->last_stream));
        set->plan->maybe_busy = false;
      }
      break;
    }
  }
  if (!set->plan) {
    // a full cache: the plan the next park would evict is taken now, buffers and all -- plan_build keeps them when the new list has its shape
    FactorPlan* p = nullptr;
    if (diag.plan_cache && diag.plan_recycle && ctx->plan_cache.size() >= PLAN_CACHE_MAX) {
      p = ctx->plan_cache.back();
      ctx->plan_cache.pop_back();
      plan_idle(p);
    } else {
      p = new FactorPlan();
    }
    p->key = key;
    const int rc = plan_build(set, p);
    if (rc != GLIM_AMD_OK) {
      plan_free(p);
      return rc;
    }
    ctx->plans_built++;
    if (p->recycled) ctx->plans_recycled++;
    set->plan = p;
  }
  // target maps whose build returned to its caller before its last kernel had finished (voxelmap.hip: polled voxel count): this set's stream waits
  // for them.  One relaxed load per entry once the build has been seen complete.
  for (const auto& e : set->entries) GA_TRY(voxelmap_wait_ready(e.target, set->stream));
  set->inline_args.valid = 0;
  set->poses_dev = set->plan->d_poses;
  if (set->entries.size() == 1) set->inline_args.d = set->plan->h_descs[0];
  set->dirty = false;
  set->seen_epoch = epoch;
  return GLIM_AMD_OK;
}

}  // namespace glim_amd

namespace {

FinalizeArgs finalize_args(const glim_amd_factor_set* set, double* out, long long row_offset, bool poll) {
  const FactorPlan* plan = set->plan;
  FinalizeArgs fa;
  fa.out = out;
  fa.out_mirror = (out && out != set->plan->h_compact_dev && out != set->plan->d_compact) ? set->record_mirror : nullptr;
  fa.out_row_offset = row_offset;
  fa.done_counter = plan->d_done;
  fa.host_flag = poll ? plan->h_flag_dev : nullptr;
  fa.seq = plan->poll_seq;
  fa.num_factors = (int)set->entries.size();
  fa.rows16 = plan->d_rows16;
  fa.trip_stats = plan->d_trip_stats;
  fa.rec16 = nullptr;
  fa.finmap = plan->d_finmap;
  fa.cull_words = plan->cull ? plan->d_cull_words : nullptr;
  return fa;
}

template <int MODE, bool FROZEN, bool INLINE, bool FUSED>
void launch_segments(glim_amd_factor_set* set, const FinalizeArgs& fa, float* partials) {
  const FactorPlan* plan = set->plan;
  const double* lin = set->poses_dev;
  const double* ev = set->poses_dev + set->entries.size() * 12;
  const int per_round = std::max(1, set->ctx->num_cus);
  const int rows0 = plan->plane_rows, rows1 = plan->total_rows - plan->plane_rows;
  const int fin0 = FUSED ? plan->fin_count[0] : 0, fin1 = FUSED ? plan->fin_count[1] : 0;
  if (rows0 > 0)
    vgicp_kernel<MODE, FROZEN, true, INLINE, FUSED><<<rows0 + fin0, BLOCK, 0, set->stream>>>(plan->d_descs, lin, ev, plan->d_blockmap, partials, set->inline_args, fa, 0,
                                                                                            per_round, rows0, 0);
  if (rows1 > 0 && plan->cull && !INLINE && !FUSED) {
    const long long cull_waves = ((4ll * rows1) + (64 >> plan->cull_log2p) - 1) / (64 >> plan->cull_log2p);
    cull_kernel<<<(unsigned int)((cull_waves + 3) / 4), 256, 0, set->stream>>>(plan->d_descs, static_cast<const CullDesc*>(plan->d_cull_descs), lin, plan->d_blockmap, rows0,
                                                                              rows1, plan->cull_log2p, plan->d_cull_words, plan->cull_count ? plan->d_cull_stats : nullptr);
    vgicp_kernel<MODE, FROZEN, false, false, false, true><<<rows1, BLOCK, 0, set->stream>>>(plan->d_descs, lin, ev, plan->d_blockmap, partials, set->inline_args, fa, rows0,
                                                                                           per_round, rows1, fin0);
  } else if (rows1 > 0)
    vgicp_kernel<MODE, FROZEN, false, INLINE, FUSED><<<rows1 + fin1, BLOCK, 0, set->stream>>>(plan->d_descs, lin, ev, plan->d_blockmap, partials, set->inline_args, fa,
                                                                                             rows0, per_round, rows1, fin0);
}

// the factor kernel(s); partials: the plan's device rows.  fused: the single-dispatch form (tagged rows, finalising blocks inside the launch)
void launch_vgicp(glim_amd_factor_set* set, int mode, bool frozen, const FinalizeArgs& fa, float* partials, bool fused) {
  const bool inl = set->inline_args.valid != 0;
#define GA_LAUNCH(MODE, FROZEN, INL)                                           \
  do {                                                                         \
    if (fused) launch_segments<MODE, FROZEN, INL, true>(set, fa, partials);    \
    else launch_segments<MODE, FROZEN, INL, false>(set, fa, partials);         \
  } while (0)
  if (mode == MODE_LINEARIZE) {
    if (inl) GA_LAUNCH(MODE_LINEARIZE, false, true);
    else GA_LAUNCH(MODE_LINEARIZE, false, false);
  } else if (frozen) {
    GA_LAUNCH(MODE_ERROR, true, false);
  } else {
    if (inl) GA_LAUNCH(MODE_ERROR, false, true);
    else GA_LAUNCH(MODE_ERROR, false, false);
  }
#undef GA_LAUNCH
  set->plan->last_stream = set->stream;
}

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield" ::: "memory");
#endif
}

bool use_single_dispatch(const glim_amd_factor_set* set, bool poll) {
  const FactorPlan* plan = set->plan;
  return poll && plan->d_rows16 && plan->h_rec16_dev && set->ctx->diag.fuse;
}

// host side of the single-dispatch form's completion: wait until every record granule carries this call's sequence number, then unpack the
// values into plan->h_compact (where every consumer of a synchronous call reads them).  false after ~200 ms.
// alive (optional): a word the producer clears when it leaves without serving (resident session): the wait ends early then.
bool collect_tagged_records(FactorPlan* plan, size_t nf, unsigned int seq, const volatile unsigned int* alive = nullptr) {
  const volatile unsigned long long* g = reinterpret_cast<const volatile unsigned long long*>(plan->h_rec16);
  const size_t n = nf * COMPACT;
  const auto t0 = std::chrono::steady_clock::now();
  size_t i = 0;
  for (unsigned long spins = 0; i < n;) {
    if ((unsigned int)g[2 * i + 1] == seq) {
      i++;
      continue;
    }
    cpu_relax();
    ++spins;
    if (alive && (spins & 0x3f) == 0x3f && *alive == 0) {
      // the session has ended; its last stores may still be in flight: look once more after a moment, then report
      for (int k = 0; k < 2000; k++) cpu_relax();
      while (i < n && (unsigned int)g[2 * i + 1] == seq) i++;
      if (i < n) return false;
      break;
    }
    if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) return false;
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  unsigned long long* out = reinterpret_cast<unsigned long long*>(plan->h_compact);
  for (size_t k = 0; k < n; k++) out[k] = g[2 * k];
  return true;
}

// the record of a factor whose finalising block gave up on a row (fused_finalize: NaN sums): its count slot is NaN, which no real result is
bool records_lost(const FactorPlan* plan, size_t nf) {
  for (size_t f = 0; f < nf; f++)
    if (std::isnan(plan->h_compact[f * COMPACT])) return true;
  return false;
}

// enqueue (no sync): poses already on the device / in the inline arguments; writes compact records to `out` rows [row_offset, row_offset + n).
// Two or three launches: the fused kernel per plan segment + the FP64 finalise.
int enqueue(glim_amd_factor_set* set, int mode, bool frozen, double* out, long long row_offset, bool poll, bool allow_fused = true) {
  const int nf = (int)set->entries.size();
  if (nf == 0) return GLIM_AMD_OK;
  FactorPlan* plan = set->plan;
  if (!set->inline_args.valid) GA_TRY(plan_upload(set, plan));  // (the inline kernels read pose and descriptor from their arguments)
  FinalizeArgs fa = finalize_args(set, out, row_offset, poll);
  // small synchronous sets whose completion the host polls: ONE dispatch per segment, the factors are finalised inside it
  const bool fused = allow_fused && use_single_dispatch(set, poll);
  if (fused) fa.rec16 = plan->h_rec16_dev;
  launch_vgicp(set, mode, frozen, fa, plan->d_partials, fused);
  if (fused) {
    GA_HIP(hipGetLastError());
    return GLIM_AMD_OK;
  }
  // more factors than one resident round of per-factor blocks, all of them short: a wavefront per factor (same bits)
  if (nf > FIN_SHORT_MIN_FACTORS && plan->max_rows_per_factor <= FIN_GROUPS && !fa.host_flag && !set->inline_args.valid)
    finalize_short_kernel<<<(nf + FIN_WAVES - 1) / FIN_WAVES, BLOCK, 0, set->stream>>>(plan->d_descs, plan->d_partials, fa, mode, set->poses_dev);
  else
    finalize_kernel<<<nf, BLOCK, 0, set->stream>>>(plan->d_descs, plan->d_partials, fa, mode, set->poses_dev, set->inline_args);
  GA_HIP(hipGetLastError());
  return GLIM_AMD_OK;
}

// Poses to the device.  Single-factor sets without an evaluation pose pass the pose in the kernel arguments (no copy at all).  Otherwise
// the poses are staged in one slot of a pinned ring and copied asynchronously; a slot is reused only after the copy that read it has
// completed (event), so back-to-back asynchronous calls never see each other's poses.
int upload_poses(glim_amd_factor_set* set, const double* T_lin, const double* T_eval, bool async_call) {
  const size_t nf = set->entries.size();
  FactorPlan* plan = set->plan;
  set->inline_args.valid = 0;
  if (nf == 1 && !T_eval && set->ctx->diag.inline_pose) {
    memcpy(set->inline_args.m, T_lin, 12 * sizeof(double));
    set->inline_args.valid = 1;
    return GLIM_AMD_OK;
  }
  const int slot = plan->pose_slot;
  plan->pose_slot = (slot + 1) % FactorPlan::POSE_RING;
  if (plan->pose_pending[slot]) {
    GA_HIP(hipEventSynchronize(plan->pose_events[slot]));
    plan->pose_pending[slot] = false;
  }
  double* h = plan->h_poses + (size_t)slot * plan->cap_factors * 24;
  const auto t_stage = std::chrono::steady_clock::now();
  memcpy(h, T_lin, nf * 12 * sizeof(double));
  if (T_eval) memcpy(h + nf * 12, T_eval, nf * 12 * sizeof(double));
  set->last_pose_stage_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_stage).count();
  if (!async_call && plan->h_poses_dev && nf <= HOST_POSES_MAX && set->ctx->diag.host_poses) {
    // small synchronous set: the kernels read the poses straight from this pinned slot (the call returns only after they have run)
    set->poses_dev = plan->h_poses_dev + (size_t)slot * plan->cap_factors * 24;
    return GLIM_AMD_OK;
  }
  set->poses_dev = plan->d_poses;
  hipStream_t up = (async_call && set->upload_stream) ? set->upload_stream : set->stream;
  if (up != set->stream && plan->poses_free_pending) GA_HIP(hipStreamWaitEvent(up, plan->poses_free_event, 0));  // the previous call's kernels still read d_poses
  GA_HIP(hipMemcpyAsync(plan->d_poses, h, nf * (T_eval ? 24 : 12) * sizeof(double), hipMemcpyHostToDevice, up));
  if (async_call) {
    if (!plan->pose_events[slot]) GA_HIP(hipEventCreateWithFlags(&plan->pose_events[slot], hipEventDisableTiming));
    GA_HIP(hipEventRecord(plan->pose_events[slot], up));
    plan->pose_pending[slot] = true;
    if (up != set->stream) GA_HIP(hipStreamWaitEvent(set->stream, plan->pose_events[slot], 0));
  }
  return GLIM_AMD_OK;
}

// spin until *word == value (acquire); false after ~200 ms
bool spin_until(const volatile unsigned int* word, unsigned int value) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned long spins = 0;; spins++) {
    if (*word == value) {
      std::atomic_thread_fence(std::memory_order_acquire);
      return true;
    }
    cpu_relax();
    if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) return false;
  }
}

// ---- resident sessions (host side; kernel: resident_kernel) ------------------------------------------------------------------------------
// ONE session per device, process-wide: a resident kernel spins on wave slots, and two of them that each wait for blocks the other keeps
// from becoming resident would only get going again when one idles out.  A plan is served through the session after RESIDENT_WARMUP
// launch-per-call linearisations (a set that is linearised once or twice never starts one); it takes the session over from another plan
// only when that one has not been asked for RESIDENT_TAKEOVER_US.
constexpr int RESIDENT_WARMUP = 3;
constexpr int RESIDENT_MAX_LINES = (RESIDENT_MAX_FACTORS * 12 + 6) / 7;
constexpr size_t RESIDENT_POSE16_BYTES = (size_t)RESIDENT_MAX_FACTORS * 12 * 16;  // (8 replicas of <= 4 factors fit as well)

constexpr long long RESIDENT_TAKEOVER_US = 1000;
struct ResidentSession {
  std::mutex mu;
  FactorPlan* plan = nullptr;
  glim_amd_ctx* ctx = nullptr;
  hipStream_t stream = nullptr;
  unsigned int *h_mail = nullptr, *h_mail_dev = nullptr;  // [16] alive
  unsigned long long *h_lines = nullptr, *h_lines_dev = nullptr;  // request lines {7 doubles, tag} (ResidentArgs::h_lines)
  char* d_pose16 = nullptr;
  unsigned int counter = 0, last_tag = 0;
  bool launched = false;
  std::atomic<bool> busy{false};
  std::chrono::steady_clock::time_point last_use;
  unsigned long long launches = 0, requests = 0;
  // device timeline (glim_amd_debug_resident_timeline): on from the next launch when `timeline_on`; the buffer outlives the sessions
  bool timeline_on = false;
  unsigned long long* d_timeline = nullptr;
  int timeline_workers = 0;
  double last_request_host_us = 0.0;  // host: from posting the last request to its last record granule
};
ResidentSession g_resident[16];
constexpr size_t TIMELINE_WORDS = TL_WORKER0 + 4 * 1024;

unsigned int next_session_tag(ResidentSession& S) {
  S.counter = (S.counter >= 0x7ffffff0u) ? 1u : S.counter + 1u;
  return 0x80000000u | S.counter;
}

// the request of a session: poses (n doubles) into the host-mapped lines -- every line's doubles first, then every line's tag, line 0's last
void resident_post(ResidentSession& S, const double* poses, size_t n, unsigned int tag) {
  volatile unsigned long long* L = S.h_lines;
  const unsigned long long* src = reinterpret_cast<const unsigned long long*>(poses);
  const size_t lines = (n + 6) / 7;
  for (size_t i = 0; i < n; i++) L[(i / 7) * 8 + (i % 7)] = src[i];
  std::atomic_thread_fence(std::memory_order_release);  // (x86: stores stay in program order; the fence keeps the compiler from moving them)
  for (size_t l = lines; l-- > 0;) L[l * 8 + 7] = tag;
  std::atomic_thread_fence(std::memory_order_release);
}

// S.mu held.  Ends the resident kernel (if any) and waits for it.
void resident_stop(ResidentSession& S) {
  if (!S.launched) return;
  if (S.h_lines) {
    const size_t lines = RESIDENT_MAX_LINES;
    for (size_t l = lines; l-- > 0;) reinterpret_cast<volatile unsigned long long*>(S.h_lines)[l * 8 + 7] = RES_EXIT;
    std::atomic_thread_fence(std::memory_order_seq_cst);
  }
  (void)hipStreamSynchronize(S.stream);
  S.launched = false;
  S.plan = nullptr;
  S.ctx = nullptr;
}

// S.mu and the context mutex held, device current.  (Re)starts the resident kernel for `plan`; it will serve the first request tag != S.last_tag.
int resident_launch(ResidentSession& S, glim_amd_factor_set* set, FactorPlan* plan) {
  glim_amd_ctx* ctx = set->ctx;
  if (!S.stream) GA_HIP(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
  // every resource on its own test: a failure half way leaves the rest to the next attempt instead of a kernel launched with null pointers
  // (ADVICE r4).  Any failure here is "this call takes the launch-per-call path" to the caller.
  auto unsupported = [](hipError_t e) {
    if (e != hipSuccess) (void)hipGetLastError();
    return e != hipSuccess;
  };
  if (!S.h_mail) {
    if (unsupported(pinned_malloc(&S.h_mail, 256))) {
      S.h_mail = nullptr;
      return GLIM_AMD_ERR_UNSUPPORTED;
    }
    memset(S.h_mail, 0, 256);
  }
  if (!S.h_mail_dev && !host_device_view(S.h_mail, &S.h_mail_dev)) return GLIM_AMD_ERR_UNSUPPORTED;
  if (!S.h_lines) {
    if (unsupported(pinned_malloc(&S.h_lines, (size_t)RESIDENT_MAX_LINES * 64))) {
      S.h_lines = nullptr;
      return GLIM_AMD_ERR_UNSUPPORTED;
    }
    memset(S.h_lines, 0, (size_t)RESIDENT_MAX_LINES * 64);
  }
  if (!S.h_lines_dev && !host_device_view(S.h_lines, &S.h_lines_dev)) return GLIM_AMD_ERR_UNSUPPORTED;
  if (!S.d_pose16 && unsupported(pool_malloc(&S.d_pose16, RESIDENT_POSE16_BYTES))) {
    S.d_pose16 = nullptr;
    return GLIM_AMD_ERR_UNSUPPORTED;
  }
  // descriptors, block map and finaliser map have to be on the device (a single-factor plan may never have uploaded them), and complete
  // before the session's stream reads them
  if (!plan->uploaded) {
    GA_TRY(plan_upload(set, plan));
    GA_HIP(hipStreamSynchronize(set->stream));
  }
  const int nf = (int)set->entries.size();
  volatile unsigned int* mail = S.h_mail;
  mail[16] = 1u;
  // (an explicit stop leaves exit tags in the request lines; a pending request stays)
  for (size_t l = 0; l < (size_t)RESIDENT_MAX_LINES; l++) {
    volatile unsigned long long* tagw = reinterpret_cast<volatile unsigned long long*>(S.h_lines) + l * 8 + 7;
    if ((unsigned int)*tagw == RES_EXIT) *tagw = S.last_tag;
  }
  std::atomic_thread_fence(std::memory_order_seq_cst);
  GA_HIP(hipMemsetAsync(S.d_pose16, 0, RESIDENT_POSE16_BYTES, S.stream));  // stale granules (an earlier session's exit tags) must not be read as news
  ResidentArgs ra;
  ra.descs = plan->d_descs;
  ra.blockmap = plan->d_blockmap;
  ra.finmap = plan->d_finmap;
  ra.total_rows = plan->total_rows;
  ra.num_factors = nf;
  ra.workers = std::min(plan->total_rows, 2 * std::max(1, ctx->num_cus));
  ra.blocks_per_round = std::max(1, ctx->num_cus);
  ra.rows16 = plan->d_rows16;
  ra.rec16 = plan->h_rec16_dev;
  ra.pose16 = S.d_pose16;
  ra.replicas = nf <= 4 ? 8 : 1;  // (a larger set's blocks watch different factors' granules anyway)
  ra.h_lines = S.h_lines_dev;
  ra.num_lines = (nf * 12 + 6) / 7;
  ra.mail = S.h_mail_dev;
  ra.first_tag = S.last_tag;
  // one empty poll of the request word is one PCIe read (~1.2 us) plus a short sleep
  ra.idle_polls = (unsigned int)std::max(100, ctx->diag.resident_idle_us * 2 / 3);
  ra.timeline = nullptr;
  if (S.timeline_on && ra.workers <= 1024) {
    if (!S.d_timeline && hipMalloc(reinterpret_cast<void**>(&S.d_timeline), TIMELINE_WORDS * sizeof(unsigned long long)) != hipSuccess) {
      (void)hipGetLastError();
      S.d_timeline = nullptr;
    }
    if (S.d_timeline) {
      GA_HIP(hipMemsetAsync(S.d_timeline, 0, TIMELINE_WORDS * sizeof(unsigned long long), S.stream));
      ra.timeline = S.d_timeline;
      S.timeline_workers = ra.workers;
    }
  }
  if (plan->plane_rows == plan->total_rows) resident_kernel<true><<<ra.workers + nf, BLOCK, 0, S.stream>>>(ra);
  else resident_kernel<false><<<ra.workers + nf, BLOCK, 0, S.stream>>>(ra);
  GA_HIP(hipGetLastError());
  S.launched = true;
  S.plan = plan;
  S.ctx = ctx;
  S.launches++;
  return GLIM_AMD_OK;
}

// The synchronous linearisation of a small set through the device's resident session.  Returns GLIM_AMD_OK (records in plan->h_compact),
// GLIM_AMD_ERR_UNSUPPORTED when this call should take the launch-per-call path instead, or an error.
int run_resident(glim_amd_factor_set* set, const double* T_lin) {
  glim_amd_ctx* ctx = set->ctx;
  if (ctx->device < 0 || ctx->device >= 16) return GLIM_AMD_ERR_UNSUPPORTED;
  ResidentSession& S = g_resident[ctx->device];
  const size_t nf = set->entries.size();
  FactorPlan* plan = nullptr;
  unsigned int tag = 0;
  std::chrono::steady_clock::time_point t_post;
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    GA_HIP(hipSetDevice(ctx->device));
    GA_TRY(factor_set_prepare(set));
    plan = set->plan;
    const Diag& diag = ctx->diag;
    if (!resident_enabled(ctx) || !diag.fuse || !diag.poll || !plan->d_rows16 || !plan->h_rec16_dev || nf > (size_t)RESIDENT_MAX_FACTORS) return GLIM_AMD_ERR_UNSUPPORTED;
    // one row per worker block: a plan with more rows than the session has workers would walk them one after the other, slower than a launch
    // whose blocks all run at once (34 factors of 131 072 points: 68 against 53 us)
    if (plan->total_rows > 2 * std::max(1, ctx->num_cus)) return GLIM_AMD_ERR_UNSUPPORTED;
    if (++plan->sync_linearize_calls <= RESIDENT_WARMUP) return GLIM_AMD_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> slock(S.mu);
    if (S.busy.load()) return GLIM_AMD_ERR_UNSUPPORTED;  // another thread's request is in flight
    const auto now = std::chrono::steady_clock::now();
    if (S.launched && S.plan != plan) {
      if (std::chrono::duration_cast<std::chrono::microseconds>(now - S.last_use).count() < RESIDENT_TAKEOVER_US) return GLIM_AMD_ERR_UNSUPPORTED;
      resident_stop(S);
    }
    if (S.launched && reinterpret_cast<volatile unsigned int*>(S.h_mail)[16] == 0u) {
      (void)hipStreamSynchronize(S.stream);  // it has idled out
      S.launched = false;
    }
    if (!S.launched) {
      // a session that cannot be started is never an error of the call: the launch-per-call path computes the same bits
      if (resident_launch(S, set, plan) != GLIM_AMD_OK) return GLIM_AMD_ERR_UNSUPPORTED;
    }
    tag = next_session_tag(S);
    t_post = std::chrono::steady_clock::now();
    resident_post(S, T_lin, nf * 12, tag);
    S.busy.store(true);
    S.last_use = now;
    S.requests++;
  }
  bool ok = collect_tagged_records(plan, nf, tag, reinterpret_cast<volatile unsigned int*>(S.h_mail) + 16);
  S.last_request_host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_post).count();
  if (!ok) {
    // the kernel left (idle time-out racing with this request) or is stuck: make sure it is gone, start a fresh one, which finds the pending
    // request in the mailbox
    std::lock_guard<std::mutex> lock(ctx->mu);
    std::lock_guard<std::mutex> slock(S.mu);
    (void)hipSetDevice(ctx->device);
    reinterpret_cast<volatile unsigned int*>(S.h_mail)[16] = 0u;
    (void)hipStreamSynchronize(S.stream);
    S.launched = false;
    int rc = resident_launch(S, set, plan);
    if (rc == GLIM_AMD_OK) ok = collect_tagged_records(plan, nf, tag, nullptr);  // (the request lines still hold this request)
    if (!ok) {
      resident_stop(S);
      S.busy.store(false);
      return GLIM_AMD_ERR_STATE;
    }
  }
  {
    std::lock_guard<std::mutex> slock(S.mu);
    S.last_tag = tag;
    S.busy.store(false);
  }
  // a finalising block that gave up on a row (bounded spin) publishes NaN under this call's tag: not a result -- the launch-per-call path answers
  if (records_lost(plan, nf)) return GLIM_AMD_ERR_UNSUPPORTED;
  return GLIM_AMD_OK;
}

}  // namespace

extern "C" int glim_amd_debug_resident_stats(int device, uint64_t* launches, uint64_t* requests, int32_t* alive) {
  if (device < 0 || device >= 16) return GLIM_AMD_ERR_INVALID;
  ResidentSession& S = g_resident[device];
  std::lock_guard<std::mutex> slock(S.mu);
  if (launches) *launches = S.launches;
  if (requests) *requests = S.requests;
  if (alive) *alive = (S.launched && S.h_mail && reinterpret_cast<volatile unsigned int*>(S.h_mail)[16] != 0u) ? 1 : 0;
  return GLIM_AMD_OK;
}

// Device timeline of the resident session.  enable: the next session launch carries s_memrealtime stamps (a running session is
// ended so that the next request starts one that does).  read: ends the session (its stamps are then in memory) and reports the LAST request,
// microseconds relative to the moment the leader saw the request in host memory:
//   [0] host: posting the request -> last record granule seen (the whole round trip, host clock)
//   [1] leader: poses re-published on the device
//   [2] [3] [4]    workers: pose seen                   min / median / max over the worker blocks
//   [5] [6] [7]    workers: first row computed           min / median / max
//   [8] [9] [10]   workers: row granules published        min / median / max
//   [11] finaliser of factor 0: pose seen   [12] every row of the factor summed   [13] record stored towards the host   [14] worker blocks
//   [15] device span: request seen -> record stored (= [13]);  [0] - [15] = host -> device -> host transit + host polling
extern "C" int glim_amd_debug_resident_timeline(int device, int enable, double* us, int32_t num_fields) {
  if (device < 0 || device >= 16) return GLIM_AMD_ERR_INVALID;
  ResidentSession& S = g_resident[device];
  std::lock_guard<std::mutex> slock(S.mu);
  if (S.busy.load()) return GLIM_AMD_ERR_STATE;
  int prev = -1;
  (void)hipGetDevice(&prev);
  GA_HIP(hipSetDevice(device));
  resident_stop(S);  // (ends the kernel: everything it stored is in memory now)
  int rc = GLIM_AMD_OK;
  if (us && num_fields > 0) {
    for (int i = 0; i < num_fields; i++) us[i] = 0.0;
    if (!S.d_timeline || S.timeline_workers <= 0) {
      rc = GLIM_AMD_ERR_STATE;
    } else {
      std::vector<unsigned long long> h(TIMELINE_WORDS, 0ull);
      GA_HIP(hipMemcpy(h.data(), S.d_timeline, TIMELINE_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      const unsigned long long t0 = h[0];
      auto rel = [&](unsigned long long t) { return t >= t0 && t0 ? (double)(t - t0) * 0.01 : -1.0; };  // 100 MHz ticks -> us
      std::vector<double> a, b, c, l;
      for (int w = 0; w < S.timeline_workers; w++) {
        const unsigned long long* p = &h[(size_t)TL_WORKER0 + 4 * (size_t)w];
        if (!p[0] || !p[1] || !p[2]) continue;
        a.push_back(rel(p[0]));
        b.push_back(rel(p[1]));
        c.push_back(rel(p[2]));
        if (p[3]) l.push_back(rel(p[3]));
      }
      auto put = [&](int i, double v) {
        if (i < num_fields) us[i] = v;
      };
      auto mmm = [&](std::vector<double>& v, int i) {
        if (v.empty()) return;
        std::sort(v.begin(), v.end());
        put(i, v.front());
        put(i + 1, v[v.size() / 2]);
        put(i + 2, v.back());
      };
      put(0, S.last_request_host_us);
      put(1, rel(h[1]));
      mmm(a, 2);
      mmm(b, 5);
      mmm(c, 8);
      put(11, rel(h[2]));
      put(12, rel(h[3]));
      put(13, rel(h[4]));
      put(14, (double)a.size());
      put(15, rel(h[4]));
      mmm(l, 16);
      put(19, rel(h[5]));
      put(20, rel(h[6]));
      put(21, (h[4] > h[2] && h[2]) ? (double)h[7] / ((double)(h[4] - h[2]) * 0.01) : 0.0);  // shader ticks per microsecond = MHz
    }
  }
  S.timeline_on = enable != 0;
  if (prev >= 0) (void)hipSetDevice(prev);
  return rc;
}

extern "C" int glim_amd_debug_plan_stats(glim_amd_ctx* ctx, uint64_t* built, uint64_t* recycled, int32_t* cached) {
  if (!ctx) return GLIM_AMD_ERR_INVALID;
  std::lock_guard<std::mutex> lock(ctx->mu);
  if (built) *built = ctx->plans_built;
  if (recycled) *recycled = ctx->plans_recycled;
  if (cached) *cached = (int32_t)ctx->plan_cache.size();
  return GLIM_AMD_OK;
}

extern "C" int glim_amd_debug_resident_stop(int device) {
  if (device < 0 || device >= 16) return GLIM_AMD_ERR_INVALID;
  ResidentSession& S = g_resident[device];
  std::lock_guard<std::mutex> slock(S.mu);
  if (S.busy.load()) return GLIM_AMD_ERR_STATE;
  (void)hipSetDevice(device);
  resident_stop(S);
  return GLIM_AMD_OK;
}

namespace glim_amd {
// Memory of a cloud / voxel map is about to be recycled (destroyed or rebuilt: quiesce_device): when the session's plan uses that object, its
// workers hold descriptors into that memory and prefetch from it while they wait, so the session ends here rather than by its idle time-out
// (ADVICE r4).  Objects the session does not use leave it alone.
// A request in flight (another thread's) is allowed to finish first.
void resident_stop_device(int device, uint64_t uid) {
  if (device < 0 || device >= 16) return;
  ResidentSession& S = g_resident[device];
  // This function must not return while the session can still read the object (the caller recycles its memory next): it waits for a request in
  // flight as long as it takes -- a request is bounded by its own give-up time -- and after FORCE_AFTER_TRIES (2 s: something is wrong) it ends the
  // kernel under the requester's feet instead of falling through (the earlier 0.2 s time-out returned with the session alive); the
  // requester then finds its line unanswered, retires the session and repeats the call as a launch.
  constexpr long FORCE_AFTER_TRIES = 200000;
  for (long tries = 0;; tries++) {
    {
      std::lock_guard<std::mutex> slock(S.mu);
      if (!S.launched) return;
      if (uid != 0 && S.plan) {  // does the session's plan use the object? (the plan is alive while the session serves it)
        bool used = false;
        for (const PlanKey& k : S.plan->key) used = used || k.target_uid == uid || k.source_uid == uid;
        if (!used) return;
      }
      if (!S.busy.load() || tries >= FORCE_AFTER_TRIES) {
        if (tries >= FORCE_AFTER_TRIES) set_hip_error(hipErrorUnknown, "resident_stop_device: session busy for 2 s, stopped by force");
        int prev = -1;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(device);
        resident_stop(S);
        if (prev >= 0) (void)hipSetDevice(prev);
        return;
      }
    }
    std::this_thread::sleep_for(std::chrono::microseconds(10));
  }
}
void resident_release(glim_amd_ctx* ctx, FactorPlan* plan) {
  for (ResidentSession& S : g_resident) {
    std::lock_guard<std::mutex> slock(S.mu);
    if (S.launched && ((plan && S.plan == plan) || (ctx && S.ctx == ctx))) resident_stop(S);
  }
}
}  // namespace glim_amd

namespace {

// One synchronous evaluation (linearise or error) of the whole set: results in plan->h_compact when this returns.
//  * small sets (<= 1024 factors, the per-frame odometry and sub-mapping cases): the records go straight into host-mapped pinned memory and
//    the block that writes the last one publishes a sequence number there; the host spins on that word (sub-microsecond wake-up) instead of
//    paying a stream-synchronise round trip.  A single factor takes its pose and descriptor through the kernel arguments: no upload at all.
//    Other small sets leave their poses in host-mapped pinned memory and the kernels read them from there: no copy-engine transfer (and no
//    wait for it) in front of the first kernel.
//  * large sets: device records + one copy + stream synchronise.
// The context mutex is held only while the work is enqueued, so factor sets of one context (different streams of its pool) overlap on
// the device when driven from different host threads, like the reference's StreamTempBufferRoundRobin factors.
// The host has seen the results of a call enqueued on the set's stream: whatever was enqueued there before (the plan's descriptor upload
// included) is done, and work of an earlier owner on another stream was waited for when the plan was adopted -- whoever frees, adopts or
// recycles the plan next need not synchronise.
inline void plan_seen_complete(const glim_amd_factor_set* set, FactorPlan* plan) {
  if (set->plan != plan) return;
  plan->last_stream = set->stream;
  plan->maybe_busy = false;
}

int run_sync(glim_amd_factor_set* set, int mode, const double* T_lin, const double* T_eval, bool allow_fast = true) {
  const size_t nf = set->entries.size();
  if (allow_fast && mode == MODE_LINEARIZE && !T_eval && nf <= (size_t)RESIDENT_MAX_FACTORS && resident_enabled(set->ctx)) {
    const int rc = run_resident(set, T_lin);
    if (rc != GLIM_AMD_ERR_UNSUPPORTED) return rc;
  }
  bool poll = false, fused = false;
  unsigned int seq = 0;
  FactorPlan* plan = nullptr;
  {
    std::lock_guard<std::mutex> lock(set->ctx->mu);
    GA_HIP(hipSetDevice(set->ctx->device));
    GA_TRY(factor_set_prepare(set));
    plan = set->plan;
    GA_TRY(upload_poses(set, T_lin, T_eval, false));
    const Diag& diag = set->ctx->diag;
    const bool mapped = plan->h_compact_dev && nf <= 1024;
    poll = mapped && plan->h_flag && diag.poll;
    if (poll) seq = ++plan->poll_seq;
    fused = allow_fast && use_single_dispatch(set, poll);
    GA_TRY(enqueue(set, mode, T_eval != nullptr, mapped ? plan->h_compact_dev : plan->d_compact, 0, poll, allow_fast));
    if (!mapped) GA_HIP(hipMemcpyAsync(plan->h_compact, plan->d_compact, nf * COMPACT * sizeof(double), hipMemcpyDeviceToHost, set->stream));
  }
  if (fused) {
    bool got = collect_tagged_records(plan, nf, seq);
    if (!got) {
      // (cannot happen short of a device fault: fall back to the stream, then the granules must be there)
      GA_HIP(hipStreamSynchronize(set->stream));
      got = collect_tagged_records(plan, nf, seq);
    }
    if (!got) return GLIM_AMD_ERR_STATE;
    // a lost row (the finaliser's bounded spin ran out) arrives as a NaN record under a valid tag: answer with the two-dispatch form
    if (records_lost(plan, nf)) return run_sync(set, mode, T_lin, T_eval, false);
    plan_seen_complete(set, plan);
    return GLIM_AMD_OK;
  }
  if (!(poll && spin_until(plan->h_flag, seq))) GA_HIP(hipStreamSynchronize(set->stream));
  plan_seen_complete(set, plan);
  return GLIM_AMD_OK;
}

int overlap_scratch(glim_amd_ctx* ctx, hipStream_t st) {
  if (ctx->ov_counters) return GLIM_AMD_OK;
  const size_t cbytes = (size_t)glim_amd_ctx::OV_MAX_QUERIES * sizeof(unsigned long long) + 64;
  GA_HIP(pool_malloc(&ctx->ov_counters, cbytes));
  GA_HIP(hipMemsetAsync(ctx->ov_counters, 0, cbytes, st));
  if (ctx->ov_host) return GLIM_AMD_OK;  // (the counters alone are re-made after a failed call: overlap_discard_counters)
  GA_HIP(pinned_malloc(&ctx->ov_host, (size_t)(1 + glim_amd_ctx::OV_MAX_QUERIES) * sizeof(unsigned int)));
  memset(ctx->ov_host, 0, (size_t)(1 + glim_amd_ctx::OV_MAX_QUERIES) * sizeof(unsigned int));
  if (!host_device_view(ctx->ov_host, &ctx->ov_host_dev)) return GLIM_AMD_ERR_HIP;
  return GLIM_AMD_OK;
}

// After a failed overlap launch / wait the arrival counters may be left non-zero, which would poison the next call's counts: wait for whatever
// did get enqueued and drop the block -- overlap_scratch hands the next call a freshly zeroed one.
void overlap_discard_counters(glim_amd_ctx* ctx, hipStream_t st) {
  (void)hipStreamSynchronize(st);
  (void)hipGetLastError();
  if (ctx->ov_counters) (void)pool_free(ctx->ov_counters);
  ctx->ov_counters = nullptr;
}

void fill_overlap_target(OverlapTarget& o, const glim_amd_voxelmap* m, const double* T12) {
  o.buckets = m->buckets;
  o.num_buckets = m->num_buckets;
  o.pad = 0;
  o.inv_res = m->inv_resolution;
  memcpy(o.T, T12, 12 * sizeof(double));
}

// blocks a query's points are spread over: one atomic per block ends the query, so few fat blocks (at most one per CU)
// lanes that share a point: the smallest power of two >= the number of targets, at most 16 (more targets take further rounds)
int overlap_lanes_shift(int num_targets) {
  int s = 0;
  while ((1 << s) < num_targets && s < 4) s++;
  return s;
}
int overlap_blocks(const glim_amd_ctx* ctx, int64_t n, int lanes_shift) {
  // two (point, lane) items per thread, at most two blocks per CU: one returning atomic per block ends the query
  const int64_t items = n << lanes_shift;
  return (int)std::max<int64_t>(1, std::min<int64_t>((items + 2 * BLOCK - 1) / (2 * BLOCK), 2 * std::max(1, ctx->num_cus)));
}

}  // namespace

extern "C" {

int glim_amd_factor_set_create(glim_amd_ctx* ctx, glim_amd_factor_set** out) {
  if (!ctx || !out) return GLIM_AMD_ERR_INVALID;
  glim_amd_factor_set* s = new glim_amd_factor_set();
  s->ctx = ctx;
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    s->stream = ctx->round_robin();
  }
  *out = s;
  return GLIM_AMD_OK;
}

int glim_amd_factor_set_destroy(glim_amd_factor_set* set) {
  if (!set) return GLIM_AMD_OK;
  {
    std::lock_guard<std::mutex> lock(set->ctx->mu);
    (void)hipSetDevice(set->ctx->device);
    factor_set_park_plan(set);  // idle plans wait in the context's cache for the next set with the same factor list
  }
  delete set;
  return GLIM_AMD_OK;
}

int glim_amd_factor_set_add(glim_amd_factor_set* set, const glim_amd_voxelmap* target, const glim_amd_cloud* source, uint32_t flags,
                            int32_t* factor_index) {
  if (!set || !target || !source) return GLIM_AMD_ERR_INVALID;
  // maps and clouds of ANY context of the set's device (GLIM's modules hand frames and maps to one another); the set runs on its own context's streams
  if (target->ctx->device != set->ctx->device || source->ctx->device != set->ctx->device) return GLIM_AMD_ERR_INVALID;
  if (!target->buckets || !source->has_covs) return GLIM_AMD_ERR_STATE;
  if (source->n > (int64_t)(1u << 28)) return GLIM_AMD_ERR_INVALID;  // 32-bit byte offsets into the point streams
  set->entries.push_back({target, source, flags});
  set->dirty = true;
  if (factor_index) *factor_index = (int32_t)set->entries.size() - 1;
  return GLIM_AMD_OK;
}

int glim_amd_factor_set_clear(glim_amd_factor_set* set) {
  if (!set) return GLIM_AMD_ERR_INVALID;
  set->entries.clear();
  set->dirty = true;
  return GLIM_AMD_OK;
}

int glim_amd_factor_set_size(const glim_amd_factor_set* set, int32_t* n) {
  if (!set || !n) return GLIM_AMD_ERR_INVALID;
  *n = (int32_t)set->entries.size();
  return GLIM_AMD_OK;
}

int glim_amd_expand_compact(const double* c, const double* T, uint32_t flags, glim_amd_linearized6* out) {
  if (!c || !out) return GLIM_AMD_ERR_INVALID;
  memset(out, 0, sizeof(*out));
  out->num_inliers = (int64_t)llround(c[0]);
  out->error = c[1];
  int k = 2;
  for (int i = 0; i < 6; i++)
    for (int j = i; j < 6; j++) {
      out->H_ss[6 * i + j] = c[k];
      out->H_ss[6 * j + i] = c[k];
      k++;
    }
  for (int i = 0; i < 6; i++) out->b_s[i] = c[k++];
  if ((flags & GLIM_AMD_FACTOR_BINARY) && T) {
    // Ad = Adjoint(delta^-1) = [R^T 0; -R^T hat(t) R^T]   ([omega; v] ordering)
    double Rt[9], Ht[9], Ad[36];
    for (int r = 0; r < 3; r++)
      for (int cc = 0; cc < 3; cc++) Rt[3 * r + cc] = T[4 * cc + r];
    const double t[3] = {T[3], T[7], T[11]};
    hat3(t, Ht);
    memset(Ad, 0, sizeof(Ad));
    for (int r = 0; r < 3; r++)
      for (int cc = 0; cc < 3; cc++) {
        Ad[6 * r + cc] = Rt[3 * r + cc];
        Ad[6 * (r + 3) + cc + 3] = Rt[3 * r + cc];
        double s = 0.0;
        for (int m = 0; m < 3; m++) s += Rt[3 * r + m] * Ht[3 * m + cc];
        Ad[6 * (r + 3) + cc] = -s;
      }
    double AtH[36];  // Ad^T H_ss
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) {
        double s = 0.0;
        for (int m = 0; m < 6; m++) s += Ad[6 * m + i] * out->H_ss[6 * m + j];
        AtH[6 * i + j] = s;
      }
    for (int i = 0; i < 6; i++) {
      for (int j = 0; j < 6; j++) {
        double s = 0.0;
        for (int m = 0; m < 6; m++) s += AtH[6 * i + m] * Ad[6 * m + j];
        out->H_tt[6 * i + j] = s;
        out->H_ts[6 * i + j] = -AtH[6 * i + j];
      }
      double s = 0.0;
      for (int m = 0; m < 6; m++) s += Ad[6 * m + i] * out->b_s[m];
      out->b_t[i] = -s;
    }
    // symmetrise H_tt against rounding
    for (int i = 0; i < 6; i++)
      for (int j = i + 1; j < 6; j++) {
        const double s = 0.5 * (out->H_tt[6 * i + j] + out->H_tt[6 * j + i]);
        out->H_tt[6 * i + j] = out->H_tt[6 * j + i] = s;
      }
  }
  return GLIM_AMD_OK;
}

int glim_amd_factor_set_linearize(glim_amd_factor_set* set, const double* T, glim_amd_linearized6* out) {
  if (!set) return GLIM_AMD_ERR_INVALID;
  const size_t nf = set->entries.size();
  if (nf == 0) return GLIM_AMD_OK;
  if (!T || !out) return GLIM_AMD_ERR_INVALID;
  GA_TRY(run_sync(set, MODE_LINEARIZE, T, nullptr));
  const double* rec = set->plan->h_compact;
  for (size_t f = 0; f < nf; f++) glim_amd_expand_compact(rec + f * COMPACT, T + 12 * f, set->entries[f].flags, &out[f]);
  return GLIM_AMD_OK;
}

int glim_amd_factor_set_profile_sync(glim_amd_factor_set* set, const double* T, int iters, float* ms_per_call) {
  if (!set || !T || iters <= 0 || !ms_per_call) return GLIM_AMD_ERR_INVALID;
  const size_t nf = set->entries.size();
  if (nf == 0) return GLIM_AMD_ERR_STATE;
  std::vector<glim_amd_linearized6> out(nf);
  for (int i = 0; i < 5; i++) GA_TRY(glim_amd_factor_set_linearize(set, T, out.data()));
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; i++) GA_TRY(glim_amd_factor_set_linearize(set, T, out.data()));
  const auto t1 = std::chrono::steady_clock::now();
  *ms_per_call = (float)(std::chrono::duration<double, std::milli>(t1 - t0).count() / iters);
  return GLIM_AMD_OK;
}

int glim_amd_factor_set_linearize_repeat(glim_amd_factor_set* set, const double* T, int num_pose_sets, int iters, glim_amd_linearized6* out_last) {
  if (!set || !T || num_pose_sets <= 0 || iters <= 0) return GLIM_AMD_ERR_INVALID;
  const size_t nf = set->entries.size();
  if (nf == 0) return GLIM_AMD_ERR_STATE;
  std::vector<glim_amd_linearized6> out(nf);
  for (int i = 0; i < iters; i++) GA_TRY(glim_amd_factor_set_linearize(set, T + (size_t)(i % num_pose_sets) * nf * 12, out.data()));
  if (out_last) memcpy(out_last, out.data(), nf * sizeof(glim_amd_linearized6));
  return GLIM_AMD_OK;
}

int glim_amd_factor_set_profile_lm(glim_amd_factor_set* set, const double* T, int iters, float* ms_linearize, float* ms_error) {
  if (!set || !T || iters <= 0) return GLIM_AMD_ERR_INVALID;
  const size_t nf = set->entries.size();
  if (nf == 0) return GLIM_AMD_ERR_STATE;
  std::vector<glim_amd_linearized6> out(nf);
  std::vector<double> err(nf);
  for (int i = 0; i < 3; i++) {
    GA_TRY(glim_amd_factor_set_linearize(set, T, out.data()));
    GA_TRY(glim_amd_factor_set_error(set, nullptr, T, err.data(), nullptr));
  }
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; i++) GA_TRY(glim_amd_factor_set_linearize(set, T, out.data()));
  auto t1 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; i++) GA_TRY(glim_amd_factor_set_error(set, nullptr, T, err.data(), nullptr));
  auto t2 = std::chrono::steady_clock::now();
  if (ms_linearize) *ms_linearize = (float)(std::chrono::duration<double, std::milli>(t1 - t0).count() / iters);
  if (ms_error) *ms_error = (float)(std::chrono::duration<double, std::milli>(t2 - t1).count() / iters);
  return GLIM_AMD_OK;
}

int glim_amd_factor_set_linearize_device_async(glim_amd_factor_set* set, const double* T, double* out_device, int64_t out_row_offset) {
  if (!set || !out_device || out_row_offset < 0) return GLIM_AMD_ERR_INVALID;
  if (set->entries.empty()) return GLIM_AMD_OK;
  if (!T) return GLIM_AMD_ERR_INVALID;
  std::lock_guard<std::mutex> lock(set->ctx->mu);
  GA_HIP(hipSetDevice(set->ctx->device));
  GA_TRY(factor_set_prepare(set));
  GA_TRY(upload_poses(set, T, nullptr, true));
  set->ctx->async_pending.store(true);  // clouds / maps destroyed later must wait for this work (glim_amd_ctx::quiesce)
  set->plan->maybe_busy = true;         // ... and so must whoever frees or adopts this plan from another stream
  GA_TRY(enqueue(set, MODE_LINEARIZE, false, out_device, out_row_offset, false));
  if (set->upload_stream) {
    FactorPlan* plan = set->plan;
    if (!plan->poses_free_event) GA_HIP(hipEventCreateWithFlags(&plan->poses_free_event, hipEventDisableTiming));
    GA_HIP(hipEventRecord(plan->poses_free_event, set->stream));
    plan->poses_free_pending = true;
  }
  return GLIM_AMD_OK;
}

int glim_amd_factor_set_error(glim_amd_factor_set* set, const double* T_lin, const double* T_eval, double* errors, int64_t* inliers) {
  if (!set) return GLIM_AMD_ERR_INVALID;
  const size_t nf = set->entries.size();
  if (nf == 0) return GLIM_AMD_OK;
  if (!T_eval || !errors) return GLIM_AMD_ERR_INVALID;
  // T_lin == NULL: correspondences at the evaluation pose (one pose per factor); otherwise frozen at T_lin and evaluated at T_eval
  if (T_lin) GA_TRY(run_sync(set, MODE_ERROR, T_lin, T_eval));
  else GA_TRY(run_sync(set, MODE_ERROR, T_eval, nullptr));
  const double* rec = set->plan->h_compact;
  for (size_t f = 0; f < nf; f++) {
    errors[f] = rec[f * COMPACT + 1];
    if (inliers) inliers[f] = (int64_t)llround(rec[f * COMPACT]);
  }
  return GLIM_AMD_OK;
}

int glim_amd_factor_set_correspondences(glim_amd_factor_set* set, int32_t fi, const double* T, int32_t* corr) {
  if (!set || !T || fi < 0 || fi >= (int32_t)set->entries.size()) return GLIM_AMD_ERR_INVALID;
  std::lock_guard<std::mutex> lock(set->ctx->mu);
  GA_HIP(hipSetDevice(set->ctx->device));
  GA_TRY(factor_set_prepare(set));
  const FactorDesc d = set->plan->h_descs[fi];
  if (d.n == 0) return GLIM_AMD_OK;
  if (!corr) return GLIM_AMD_ERR_INVALID;
  double* d_pose = nullptr;
  int32_t* d_corr = nullptr;
  GA_HIP(pool_malloc(&d_pose, 12 * sizeof(double)));
  hipError_t e = pool_malloc(&d_corr, (size_t)d.n * 4 * sizeof(int32_t));
  if (e == hipSuccess) e = hipMemcpyAsync(d_pose, T, 12 * sizeof(double), hipMemcpyHostToDevice, set->stream);
  if (e == hipSuccess) {
    correspondence_kernel<<<(d.n + BLOCK - 1) / BLOCK, BLOCK, 0, set->stream>>>(d, d_pose, d_corr);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(corr, d_corr, (size_t)d.n * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, set->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(set->stream);
  (void)pool_free(d_pose);
  if (d_corr) (void)pool_free(d_corr);
  if (e != hipSuccess) {
    set_hip_error(e, "factor_set_correspondences");
    return GLIM_AMD_ERR_HIP;
  }
  return GLIM_AMD_OK;
}

int glim_amd_factor_set_profile(glim_amd_factor_set* set, const double* T, int iters, float* ms_kernel, float* ms_linearize) {
  if (!set || !T || iters <= 0) return GLIM_AMD_ERR_INVALID;
  const int nf = (int)set->entries.size();
  if (nf == 0) return GLIM_AMD_ERR_STATE;
  std::lock_guard<std::mutex> lock(set->ctx->mu);
  GA_HIP(hipSetDevice(set->ctx->device));
  GA_TRY(factor_set_prepare(set));
  GA_TRY(upload_poses(set, T, nullptr, false));
  FactorPlan* plan = set->plan;
  hipEvent_t e0, e1;
  GA_HIP(hipEventCreate(&e0));
  GA_HIP(hipEventCreate(&e1));
  // warm-up (clocks, caches, TLBs)
  for (int i = 0; i < 10; i++) GA_TRY(enqueue(set, MODE_LINEARIZE, false, plan->d_compact, 0, false));
  GA_HIP(hipStreamSynchronize(set->stream));
  float ms = 0.f;
  GA_HIP(hipEventRecord(e0, set->stream));
  if (!set->inline_args.valid) GA_TRY(plan_upload(set, plan));
  const FinalizeArgs fa_prof = finalize_args(set, plan->d_compact, 0, false);
  for (int i = 0; i < iters; i++) launch_vgicp(set, MODE_LINEARIZE, false, fa_prof, plan->d_partials, false);
  GA_HIP(hipEventRecord(e1, set->stream));
  GA_HIP(hipEventSynchronize(e1));
  GA_HIP(hipEventElapsedTime(&ms, e0, e1));
  if (ms_kernel) *ms_kernel = ms / (float)iters;
  GA_HIP(hipEventRecord(e0, set->stream));
  for (int i = 0; i < iters; i++) GA_TRY(enqueue(set, MODE_LINEARIZE, false, plan->d_compact, 0, false));
  GA_HIP(hipEventRecord(e1, set->stream));
  GA_HIP(hipEventSynchronize(e1));
  GA_HIP(hipEventElapsedTime(&ms, e0, e1));
  if (ms_linearize) *ms_linearize = ms / (float)iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return GLIM_AMD_OK;
}

// One launch, no allocation, no copy, no stream synchronise: descriptors in the kernel arguments (single query, <= 16 targets) or read by the
// kernel from a pinned staging block (batches), results and the completion word in host-mapped memory.
int glim_amd_overlap_batch(glim_amd_ctx* ctx, int32_t num_queries, const int32_t* num_targets, const glim_amd_voxelmap* const* targets,
                           const double* T, const glim_amd_cloud* const* sources, double* overlaps) {
  if (!ctx || num_queries <= 0 || !num_targets || !targets || !T || !sources || !overlaps) return GLIM_AMD_ERR_INVALID;
  if (num_queries > glim_amd_ctx::OV_MAX_QUERIES) return GLIM_AMD_ERR_UNSUPPORTED;
  int64_t total_targets = 0;
  for (int q = 0; q < num_queries; q++) {
    if (num_targets[q] <= 0 || !sources[q] || sources[q]->ctx->device != ctx->device) return GLIM_AMD_ERR_INVALID;
    for (int t = 0; t < num_targets[q]; t++) {
      const glim_amd_voxelmap* m = targets[total_targets + t];
      if (!m || m->ctx->device != ctx->device) return GLIM_AMD_ERR_INVALID;
      if (!m->buckets) return GLIM_AMD_ERR_STATE;
    }
    total_targets += num_targets[q];
  }
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream();
  for (int64_t t = 0; t < total_targets; t++) GA_TRY(voxelmap_wait_ready(targets[t], st));  // (maps whose build the host has not seen complete)
  GA_TRY(overlap_scratch(ctx, st));
  // queries with an empty source are answered here (0.0) and take no part in the launch
  std::vector<int> live;
  for (int q = 0; q < num_queries; q++) {
    overlaps[q] = 0.0;
    if (sources[q]->n > 0) live.push_back(q);
  }
  if (live.empty()) return GLIM_AMD_OK;
  const unsigned int seq = ++ctx->ov_seq;
  std::vector<int64_t> first_target((size_t)num_queries);
  {
    int64_t acc = 0;
    for (int q = 0; q < num_queries; q++) {
      first_target[(size_t)q] = acc;
      acc += num_targets[q];
    }
  }
  const OverlapInline none{};
  if (live.size() == 1 && num_targets[live[0]] <= OVERLAP_INLINE_TARGETS) {
    const int q = live[0];
    OverlapInline in{};
    in.q.pts = sources[q]->pts;
    in.q.n = (int)sources[q]->n;
    in.q.first_target = 0;
    in.q.num_targets = num_targets[q];
    in.q.first_block = 0;
    in.q.lanes_shift = overlap_lanes_shift(num_targets[q]);
    in.q.num_blocks = overlap_blocks(ctx, sources[q]->n, in.q.lanes_shift);
    for (int t = 0; t < num_targets[q]; t++) fill_overlap_target(in.t[t], targets[first_target[(size_t)q] + t], T + 12 * (first_target[(size_t)q] + t));
    overlap_kernel<true><<<in.q.num_blocks, BLOCK, 0, st>>>(in, nullptr, nullptr, nullptr, 1, ctx->ov_counters, nullptr, ctx->ov_host_dev + 1, ctx->ov_host_dev, seq);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && !spin_until(ctx->ov_host, seq)) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
      set_hip_error(e, "overlap");
      overlap_discard_counters(ctx, st);
      return GLIM_AMD_ERR_HIP;
    }
    overlaps[q] = (double)ctx->ov_host[1] / (double)sources[q]->n;
    return GLIM_AMD_OK;
  }
  // batch: queries, targets and the block -> query table in one pinned block the kernel reads directly
  int total_blocks = 0;
  std::vector<OverlapQuery> hq(live.size());
  int64_t live_targets = 0;
  long long walking_blocks = 0;
  for (const int q : live) walking_blocks += overlap_blocks(ctx, sources[q]->n, 0);
  const bool batch_fills_device = walking_blocks >= (long long)std::max(1, ctx->num_cus);
  for (size_t i = 0; i < live.size(); i++) {
    const int q = live[i];
    hq[i].pts = sources[q]->pts;
    hq[i].n = (int)sources[q]->n;
    hq[i].first_target = (int)live_targets;
    hq[i].num_targets = num_targets[q];
    hq[i].first_block = total_blocks;
    // Spreading a point's targets over lanes trades work for latency: every target is looked up, where one lane walking them stops at the first
    // hit (keyframes overlap: two or three lookups per point on average).  A batch that fills the chip by itself (the 43-query keyframe elimination
    // loop: 860 blocks) is bound by that work, not by one query's latency -- 55-59 us walking, 92 us spread -- so only batches too small to fill
    // the device spread.
    hq[i].lanes_shift = batch_fills_device ? 0 : overlap_lanes_shift(num_targets[q]);
    hq[i].num_blocks = overlap_blocks(ctx, sources[q]->n, hq[i].lanes_shift);
    total_blocks += hq[i].num_blocks;
    live_targets += num_targets[q];
  }
  const size_t qbytes = hq.size() * sizeof(OverlapQuery), tbytes = (size_t)live_targets * sizeof(OverlapTarget), bbytes = (size_t)total_blocks * sizeof(int);
  char* stage = nullptr;
  GA_HIP(pinned_malloc(&stage, qbytes + tbytes + bbytes));
  char* stage_dev = nullptr;
  if (!host_device_view(stage, &stage_dev)) {
    (void)pinned_free(stage);
    return GLIM_AMD_ERR_HIP;
  }
  memcpy(stage, hq.data(), qbytes);
  OverlapTarget* ht = reinterpret_cast<OverlapTarget*>(stage + qbytes);
  int* hb = reinterpret_cast<int*>(stage + qbytes + tbytes);
  for (size_t i = 0; i < live.size(); i++) {
    const int q = live[i];
    for (int t = 0; t < num_targets[q]; t++)
      fill_overlap_target(ht[hq[i].first_target + t], targets[first_target[(size_t)q] + t], T + 12 * (first_target[(size_t)q] + t));
    for (int b = 0; b < hq[i].num_blocks; b++) hb[hq[i].first_block + b] = (int)i;
  }
  unsigned int* queries_done = reinterpret_cast<unsigned int*>(ctx->ov_counters + glim_amd_ctx::OV_MAX_QUERIES);
  overlap_kernel<false><<<total_blocks, BLOCK, 0, st>>>(none, reinterpret_cast<const OverlapQuery*>(stage_dev), reinterpret_cast<const OverlapTarget*>(stage_dev + qbytes),
                                                        reinterpret_cast<const int*>(stage_dev + qbytes + tbytes), (int)live.size(), ctx->ov_counters, queries_done,
                                                        ctx->ov_host_dev + 1, ctx->ov_host_dev, seq);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && !spin_until(ctx->ov_host, seq)) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    set_hip_error(e, "overlap_batch");
    overlap_discard_counters(ctx, st);  // (waits for the stream: the kernel reads `stage` directly)
    (void)pinned_free(stage);
    return GLIM_AMD_ERR_HIP;
  }
  (void)pinned_free(stage);
  for (size_t i = 0; i < live.size(); i++) overlaps[live[i]] = (double)ctx->ov_host[1 + i] / (double)sources[live[i]]->n;
  return GLIM_AMD_OK;
}

int glim_amd_factor_set_profile_fresh(glim_amd_ctx* ctx, int32_t n, const glim_amd_voxelmap* const* targets, const glim_amd_cloud* const* sources,
                                      const uint32_t* flags, const double* T, int iters, float* us_per_iteration) {
  if (!ctx || n <= 0 || !targets || !sources || !T || iters <= 0 || !us_per_iteration) return GLIM_AMD_ERR_INVALID;
  std::vector<glim_amd_linearized6> out((size_t)n);
  auto once = [&]() -> int {
    glim_amd_factor_set* set = nullptr;
    GA_TRY(glim_amd_factor_set_create(ctx, &set));
    int rc = GLIM_AMD_OK;
    for (int f = 0; f < n && rc == GLIM_AMD_OK; f++) rc = glim_amd_factor_set_add(set, targets[f], sources[f], flags ? flags[f] : 0u, nullptr);
    if (rc == GLIM_AMD_OK) rc = glim_amd_factor_set_linearize(set, T, out.data());
    (void)glim_amd_factor_set_destroy(set);
    return rc;
  };
  for (int i = 0; i < 5; i++) GA_TRY(once());
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; i++) GA_TRY(once());
  const auto t1 = std::chrono::steady_clock::now();
  *us_per_iteration = (float)(std::chrono::duration<double, std::micro>(t1 - t0).count() / iters);
  return GLIM_AMD_OK;
}

int glim_amd_factor_set_trip_stats(glim_amd_factor_set* set, uint64_t* skipped_trips, uint64_t* total_trips, int reset) {
  if (!set) return GLIM_AMD_ERR_INVALID;
  std::lock_guard<std::mutex> lock(set->ctx->mu);
  GA_HIP(hipSetDevice(set->ctx->device));
  GA_TRY(factor_set_prepare(set));
  FactorPlan* plan = set->plan;
  GA_HIP(hipStreamSynchronize(set->stream));
  unsigned long long h[64];
  GA_HIP(hipMemcpy(h, plan->d_trip_stats, sizeof(h), hipMemcpyDeviceToHost));
  unsigned long long sk = 0;
  for (unsigned long long v : h) sk += v;
  if (skipped_trips) *skipped_trips = sk;
  if (total_trips) {
    unsigned long long tot = 0;
    for (const FactorDesc& d : plan->h_descs) tot += (unsigned long long)d.num_blocks * (BLOCK / 64) * (unsigned long long)d.ppt;
    *total_trips = tot;
  }
  if (reset) GA_HIP(hipMemset(plan->d_trip_stats, 0, sizeof(h)));
  return GLIM_AMD_OK;
}

int glim_amd_factor_set_cull_stats(glim_amd_factor_set* set, uint64_t* culled_trips, uint64_t* trips_with_points, int reset) {
  if (!set) return GLIM_AMD_ERR_INVALID;
  std::lock_guard<std::mutex> lock(set->ctx->mu);
  GA_HIP(hipSetDevice(set->ctx->device));
  GA_TRY(factor_set_prepare(set));
  FactorPlan* plan = set->plan;
  unsigned long long h[128], sum[2] = {0ull, 0ull};
  if (plan->cull) {
    GA_HIP(hipStreamSynchronize(set->stream));
    GA_HIP(hipMemcpy(h, plan->d_cull_stats, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < 128; i++) sum[i & 1] += h[i];
    if (reset) GA_HIP(hipMemset(plan->d_cull_stats, 0, sizeof(h)));
    plan->cull_count = reset != 0;  // reset != 0: zero the counters and COUNT from now on; reset == 0: read and stop counting (the default state)
  }
  if (culled_trips) *culled_trips = sum[0];
  if (trips_with_points) *trips_with_points = sum[1];
  return plan->cull ? GLIM_AMD_OK : GLIM_AMD_ERR_UNSUPPORTED;
}

int glim_amd_factor_set_profile_fresh_samples(glim_amd_ctx* ctx, int32_t n, const glim_amd_voxelmap* const* targets, const glim_amd_cloud* const* sources,
                                              const uint32_t* flags, const double* T, int iters, double gap_us, float* samples_us) {
  if (!ctx || n <= 0 || !targets || !sources || !T || iters <= 0 || !samples_us || gap_us < 0.0) return GLIM_AMD_ERR_INVALID;
  std::vector<glim_amd_linearized6> out((size_t)n);
  auto once = [&]() -> int {
    glim_amd_factor_set* set = nullptr;
    GA_TRY(glim_amd_factor_set_create(ctx, &set));
    int rc = GLIM_AMD_OK;
    for (int f = 0; f < n && rc == GLIM_AMD_OK; f++) rc = glim_amd_factor_set_add(set, targets[f], sources[f], flags ? flags[f] : 0u, nullptr);
    if (rc == GLIM_AMD_OK) rc = glim_amd_factor_set_linearize(set, T, out.data());
    (void)glim_amd_factor_set_destroy(set);
    return rc;
  };
  for (int i = 0; i < 5; i++) GA_TRY(once());
  for (int i = 0; i < iters; i++) {
    const auto t0 = std::chrono::steady_clock::now();
    GA_TRY(once());
    const auto t1 = std::chrono::steady_clock::now();
    samples_us[i] = (float)std::chrono::duration<double, std::micro>(t1 - t0).count();
    while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count() < gap_us) cpu_relax();  // the optimiser's host work
  }
  return GLIM_AMD_OK;
}

int glim_amd_overlap_profile(glim_amd_ctx* ctx, int32_t num_queries, const int32_t* num_targets, const glim_amd_voxelmap* const* targets, const double* T,
                             const glim_amd_cloud* const* sources, int iters, float* us_per_call) {
  if (!ctx || num_queries <= 0 || iters <= 0 || !us_per_call) return GLIM_AMD_ERR_INVALID;
  std::vector<double> ov((size_t)num_queries);
  for (int i = 0; i < 5; i++) GA_TRY(glim_amd_overlap_batch(ctx, num_queries, num_targets, targets, T, sources, ov.data()));
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; i++) GA_TRY(glim_amd_overlap_batch(ctx, num_queries, num_targets, targets, T, sources, ov.data()));
  const auto t1 = std::chrono::steady_clock::now();
  *us_per_call = (float)(std::chrono::duration<double, std::micro>(t1 - t0).count() / iters);
  return GLIM_AMD_OK;
}

int glim_amd_overlap(glim_amd_ctx* ctx, int32_t num_targets, const glim_amd_voxelmap* const* targets, const double* T,
                     const glim_amd_cloud* source, double* overlap) {
  if (!source || !overlap) return GLIM_AMD_ERR_INVALID;
  return glim_amd_overlap_batch(ctx, 1, &num_targets, targets, T, &source, overlap);
}

}  // extern "C"
