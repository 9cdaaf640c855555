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
// Two algebraic restructurings keep the kernel HBM-bound (DESIGN.md "K4"):
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

#include "device_math.hpp"
#include "internal.hpp"

using namespace glim_amd;

namespace {

#ifndef GLIM_AMD_ABLATE
#define GLIM_AMD_ABLATE 0  // profiling-only variants (tools/ablate.sh): 1 no gathers, 2 FP32 transform, 3 no algebra, 4 key gather only, 6 linear (coalesced) bucket index
#endif

constexpr int BLOCK = 256;
constexpr int NACC = 28;  // FP32 accumulators per thread (see layout below); slot 28 of a partial row = inlier count (int bits)

// accumulator layout:  0..5  Hww (00 01 02 11 12 22)   6..14 G = hat(p) A (row-major 3x3 = H_wv)   15..20 A (00 01 02 11 12 22)
//                      21..23 u x p (= b_w)            24..26 u (b_v = -u)                          27 e
// compact record:      [count, error, 21 upper-triangular H_ss entries row-major, 6 b_s]
__constant__ int c_acc_of_upper[21] = {0, 1, 2, 6, 7, 8, 3, 4, 9, 10, 11, 5, 12, 13, 14, 15, 16, 17, 18, 19, 20};

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
  const int* rows;           // per factor: the plan rows (blocks) that hold its partial sums, in chunk order
  double* out;               // compact records
  long long out_row_offset;
  int* done_counter;         // finished factors of this launch (polling fast path)
  unsigned int* host_flag;   // host-mapped completion word or null
  unsigned int seq;
  int num_factors;
};

// Fixed-order FP64 sum of factor f's partial rows -> compact record; executed by all 256 threads of ONE block.  Thread (g, j),
// g = tid / 32, j = tid % 32, sums rows g, g + 8, g + 16, ... of value j; the 8 group sums are then added in group order.  The
// order depends only on the plan, so results are bit-reproducible whichever block happens to run this.
__device__ __forceinline__ void finalize_factor(const FactorDesc& d, int f, const float* __restrict__ partials, const FinalizeArgs& fa, int mode,
                                                double (*s_part)[PARTIAL_STRIDE], double* s_sum, const double* __restrict__ T) {
  const int j = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int first = d.first_block, nb = d.num_blocks;
  // A single factor spread over the whole chip has ~500 partial rows and ONE finalising block: with one dependent (row id -> value)
  // load pair per trip the 64 trips of a thread are 64 memory round trips (~23 us, most of a synchronous single-factor call).  The
  // loads of 16 trips are therefore issued together; the additions keep their order, so the sums are bit-identical.
  double s = 0.0;
  constexpr int INFLIGHT = 16;
  for (int c = g; c < nb; c += 8 * INFLIGHT) {
    int r[INFLIGHT];
    float v[INFLIGHT];
#pragma unroll
    for (int u = 0; u < INFLIGHT; u++) r[u] = fa.rows[first + min(c + 8 * u, nb - 1)];  // past the end: a valid row, value unused
#pragma unroll
    for (int u = 0; u < INFLIGHT; u++) v[u] = partials[(size_t)r[u] * PARTIAL_STRIDE + j];
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
  double* o = fa.out + ((size_t)fa.out_row_offset + f) * COMPACT;
  if (t == 0) o[0] = s_sum[28];
  if (t == 1) o[1] = s_sum[27];
  if (mode == MODE_LINEARIZE) {
    // The kernel accumulated H' = sum J'^T M J' and b' = sum J'^T M r for J' = [hat(R p) | -I] in the target frame; with
    // J_s = J' diag(R, R):  H_ss = diag(R, R)^T H' diag(R, R), b_s = diag(R, R)^T b', i.e. every 3x3 block B' becomes R^T B' R and
    // every 3-vector R^T v.  Thread k < 3 rotates block k (Hww, Hwv, Hvv), thread 3 the two vectors; the results land in s_sum.
    __shared__ double s_rot[32];
    if (t < 4) {
      double R[9];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) R[3 * r + c] = T[4 * r + c];
      if (t < 3) {
        double B[9];
        if (t == 0) {
          B[0] = s_sum[0]; B[1] = s_sum[1]; B[2] = s_sum[2]; B[3] = s_sum[1]; B[4] = s_sum[3]; B[5] = s_sum[4]; B[6] = s_sum[2]; B[7] = s_sum[4]; B[8] = s_sum[5];
        } else if (t == 1) {
#pragma unroll
          for (int i = 0; i < 9; i++) B[i] = s_sum[6 + i];
        } else {
          B[0] = s_sum[15]; B[1] = s_sum[16]; B[2] = s_sum[17]; B[3] = s_sum[16]; B[4] = s_sum[18]; B[5] = s_sum[19]; B[6] = s_sum[17]; B[7] = s_sum[19]; B[8] = s_sum[20];
        }
        double BR[9], O[9];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) BR[3 * r + c] = B[3 * r] * R[c] + B[3 * r + 1] * R[3 + c] + B[3 * r + 2] * R[6 + c];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) O[3 * r + c] = R[r] * BR[c] + R[3 + r] * BR[3 + c] + R[6 + r] * BR[6 + c];  // (R^T BR)[r][c]
        if (t == 0) {
          s_rot[0] = O[0]; s_rot[1] = O[1]; s_rot[2] = O[2]; s_rot[3] = O[4]; s_rot[4] = O[5]; s_rot[5] = O[8];
        } else if (t == 1) {
#pragma unroll
          for (int i = 0; i < 9; i++) s_rot[6 + i] = O[i];
        } else {
          s_rot[15] = O[0]; s_rot[16] = O[1]; s_rot[17] = O[2]; s_rot[18] = O[4]; s_rot[19] = O[5]; s_rot[20] = O[8];
        }
      } else {
#pragma unroll
        for (int c = 0; c < 3; c++) {
          s_rot[21 + c] = R[c] * s_sum[21] + R[3 + c] * s_sum[22] + R[6 + c] * s_sum[23];   // R^T (sum u x q')
          s_rot[24 + c] = R[c] * s_sum[24] + R[3 + c] * s_sum[25] + R[6 + c] * s_sum[26];   // R^T (sum u)
        }
      }
    }
    __syncthreads();
    if (t < 21) o[2 + t] = s_rot[c_acc_of_upper[t]];
    if (t >= 21 && t < 24) o[2 + t] = s_rot[t];          // b_w = R^T sum u x q'
    if (t >= 24 && t < 27) o[2 + t] = -s_rot[t];         // b_v = -R^T sum u
  } else if (t >= 2 && t < COMPACT) {
    o[t] = 0.0;
  }
  if (fa.host_flag) {
    // completion flag for the polling fast path: every finalising block makes its record visible system-wide, the one that
    // completes the launch publishes `seq` into host-mapped memory (the host spins on it instead of a stream synchronise)
    __threadfence_system();
    __syncthreads();
    if (t == 0) {
      const int prev = atomicAdd(fa.done_counter, 1);
      if (prev == fa.num_factors - 1) {
        *fa.done_counter = 0;
        __threadfence_system();
        __hip_atomic_store(fa.host_flag, fa.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// MODE: linearise (28 sums) or error only.  FROZEN: residual at a separate evaluation pose with correspondences and
// Mahalanobis matrices frozen at the linearisation pose.  U: points per loop trip.  MINW: occupancy hint (waves per SIMD).
//
// The loop is software-pipelined and branch-free on the hot path: the coalesced point/covariance loads of trip t+1 and the
// 48-byte voxel-slot gathers of trip t are issued back to back BEFORE the algebra of trip t, every lane runs the algebra
// and the accumulation is predicated by `hit` (93 % of lanes hit, so predication beats divergence); the only branch left is
// the rare hash-collision re-probe.  Each lane thus exposes one gather round trip per trip instead of three dependent ones.
template <int MODE, bool FROZEN, bool PLANE, bool INLINE>
__global__ __launch_bounds__(BLOCK, 5) void vgicp_kernel(const FactorDesc* __restrict__ descs, const double* __restrict__ poses_lin,
                                                          const double* __restrict__ poses_eval, const int2* __restrict__ blockmap,
                                                          float* __restrict__ partials, const InlinePose ip, const FinalizeArgs fa, int block_offset) {
  constexpr int U = 1;  // points per loop trip (2 and 4 were measured slower: more registers, fewer resident waves)
  __shared__ float s_red[4][PARTIAL_STRIDE];
  const int gblock = block_offset + (int)blockIdx.x;  // row of this block in the plan (the plane-form and the general segment are separate launches)
  const int2 bm = blockmap[gblock];
  const int f = bm.x;
  if (f < 0) return;  // padding block of the XCD-aware map
  const FactorDesc d = descs[f];
  // INLINE (single-factor sets): the pose is read from the kernel arguments (scalar registers), no device pose array involved
  const double* Tl = INLINE ? ip.m : poses_lin + 12 * (size_t)f;
  const double* Te = FROZEN ? poses_eval + 12 * (size_t)f : Tl;

  // rotation of the linearisation pose in FP32 (R[r][c])
  // (wave-uniform: the compiler keeps these in SGPRs; forcing readfirstlane changed nothing -- 93 VGPRs either way)
  const float R00 = (float)Tl[0], R01 = (float)Tl[1], R02 = (float)Tl[2];
  const float R10 = (float)Tl[4], R11 = (float)Tl[5], R12 = (float)Tl[6];
  const float R20 = (float)Tl[8], R21 = (float)Tl[9], R22 = (float)Tl[10];
  const bool validate = (d.flags & GLIM_AMD_FACTOR_SURFACE_VALIDATION) && (PLANE || d.sn != nullptr);
  const float resf = (float)d.res;
  const int last = d.n - 1;

  float acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; j++) acc[j] = 0.f;
  int inliers = 0;

  const int ppt = d.ppt;
  const int base = bm.y * (BLOCK * ppt) + threadIdx.x;
  if (d.n > 0) {
    // prologue: points of trip 0 (indices clamped so every load is in bounds; validity is tracked separately)
    PointIn nxt[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      nxt[u] = load_point<PLANE>(d, (unsigned int)min(base + u * BLOCK, last));
    }
    for (int it0 = 0; it0 < ppt; it0 += U) {
      PointIn cur[U];
      float4 head[U];
      float qr[U][3];  // q relative to the centre of its voxel (|.| <= res/2): FP32 without cancellation
      float qp[U][3];  // q' = R p = q - t: the rotated source point, the lever arm of the target-frame Jacobian
      unsigned long long key[U];
      unsigned int bkt[U];

      // ---- FP64 transform -> voxel key -> 16-byte gather of the home bucket's two keys ----
#pragma unroll
      for (int u = 0; u < U; u++) {
        cur[u] = nxt[u];
        const int i = base + (it0 + u) * BLOCK;
        const bool ok = (it0 + u < ppt) && (i < d.n);
#if GLIM_AMD_ABLATE == 2
        const float qxf = R00 * cur[u].p.x + R01 * cur[u].p.y + R02 * cur[u].p.z + (float)Tl[3];
        const float qyf = R10 * cur[u].p.x + R11 * cur[u].p.y + R12 * cur[u].p.z + (float)Tl[7];
        const float qzf = R20 * cur[u].p.x + R21 * cur[u].p.y + R22 * cur[u].p.z + (float)Tl[11];
        const float ir = (float)d.inv_res;
        const float tx = qxf * ir, ty = qyf * ir, tz = qzf * ir;
        const double qx = qxf, qy = qyf, qz = qzf;
        (void)qx; (void)qy; (void)qz;
#else
        double qx, qy, qz;
        transform_point_d(Tl, (double)cur[u].p.x, (double)cur[u].p.y, (double)cur[u].p.z, qx, qy, qz);
        const double tx = qx * d.inv_res, ty = qy * d.inv_res, tz = qz * d.inv_res;
#endif
        qp[u][0] = (float)(qx - Tl[3]);
        qp[u][1] = (float)(qy - Tl[7]);
        qp[u][2] = (float)(qz - Tl[11]);
        // floor(t) == fast_floor(t) for every in-range coordinate (same integer, bit-exact); v_floor_f64 + one subtraction also
        // gives the in-voxel fraction for free
#if GLIM_AMD_ABLATE == 2
        const float fx = floorf(tx), fy = floorf(ty), fz = floorf(tz);
        const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
#else
        const double fx = floor(tx), fy = floor(ty), fz = floor(tz);
        const int cx = __double2int_rz(fx), cy = __double2int_rz(fy), cz = __double2int_rz(fz);
#endif
        // out-of-range coordinates saturate in v_cvt_i32_f64 and fail the unsigned 21-bit range check
        const unsigned int ux = (unsigned int)(cx + KEY_OFFSET), uy = (unsigned int)(cy + KEY_OFFSET), uz = (unsigned int)(cz + KEY_OFFSET);
        const bool valid = ok && (((ux | uy | uz) >> KEY_BITS) == 0u);
        key[u] = valid ? (((unsigned long long)ux << (2 * KEY_BITS)) | ((unsigned long long)uy << KEY_BITS) | (unsigned long long)uz) : EMPTY_KEY;
        unsigned int hsh = hash_fields(ux, uy, uz);
        if (FROZEN) {
          double ex, ey, ez;
          transform_point_d(Te, (double)cur[u].p.x, (double)cur[u].p.y, (double)cur[u].p.z, ex, ey, ez);
          qr[u][0] = (float)(ex - ((double)cx + 0.5) * d.res);
          qr[u][1] = (float)(ey - ((double)cy + 0.5) * d.res);
          qr[u][2] = (float)(ez - ((double)cz + 0.5) * d.res);
        } else {
          qr[u][0] = ((float)(tx - fx) - 0.5f) * resf;
          qr[u][1] = ((float)(ty - fy) - 0.5f) * resf;
          qr[u][2] = ((float)(tz - fz) - 0.5f) * resf;
        }
        if (validate) {
          // surface validation (upstream predicate unverified -- SURVEY.md App. B.5): the source normal faces the source sensor
          // (p . n <= 0, cloud_covariance_estimation.cpp:98-101); reject when the transformed surface faces away from the target origin.
          const float4 nn = PLANE ? make_float4(cur[u].p.w, cur[u].cb.x, cur[u].cb.y, 0.f)
                                  : gld4(reinterpret_cast<const char*>(d.sn) + (unsigned int)min(i, last) * 16u);
          const float rnx = R00 * nn.x + R01 * nn.y + R02 * nn.z;
          const float rny = R10 * nn.x + R11 * nn.y + R12 * nn.z;
          const float rnz = R20 * nn.x + R21 * nn.y + R22 * nn.z;
          if (rnx * (float)qx + rny * (float)qy + rnz * (float)qz > 0.f) key[u] = EMPTY_KEY;
        }
        bkt[u] = __umulhi(hsh, d.num_buckets);
#if GLIM_AMD_ABLATE == 6
        bkt[u] = (unsigned int)(base + (it0 + u) * BLOCK) % d.num_buckets;  // coalesced stand-in for the hashed bucket
        key[u] = EMPTY_KEY - 1;
#endif
#if GLIM_AMD_ABLATE == 1
        head[u] = make_float4(__uint_as_float((unsigned int)key[u]), __uint_as_float((unsigned int)(key[u] >> 32)), 0.f, 0.f);
#else
        head[u] = gld4(reinterpret_cast<const char*>(d.buckets) + bkt[u] * 128u);  // both keys of the home bucket (32-bit offset: <= 2^25 buckets)
#endif
      }
      // ---- coalesced loads of the NEXT trip, issued behind the gathers (counted vmcnt lets the gathers be consumed first) ----
#pragma unroll
      for (int u = 0; u < U; u++) {
        nxt[u] = load_point<PLANE>(d, (unsigned int)min(base + (it0 + U + u) * BLOCK, last));
      }
      // ---- resolve the probe, then the per-point algebra (all lanes; accumulation predicated by hit) ----
#pragma unroll
      for (int u = 0; u < U; u++) {
        unsigned long long k0 = (unsigned long long)__float_as_uint(head[u].x) | ((unsigned long long)__float_as_uint(head[u].y) << 32);
        unsigned long long k1 = (unsigned long long)__float_as_uint(head[u].z) | ((unsigned long long)__float_as_uint(head[u].w) << 32);
        unsigned int b = bkt[u];
        if (key[u] != EMPTY_KEY) {
          // rare spill: both ways of the home bucket hold other keys -> walk to the next bucket (exact compare, table never full)
          while (k0 != key[u] && k1 != key[u] && k1 != EMPTY_KEY) {
            b = (b + 1 == d.num_buckets) ? 0u : b + 1;
            const float4 h = gld4(reinterpret_cast<const char*>(d.buckets) + b * 128u);
            k0 = (unsigned long long)__float_as_uint(h.x) | ((unsigned long long)__float_as_uint(h.y) << 32);
            k1 = (unsigned long long)__float_as_uint(h.z) | ((unsigned long long)__float_as_uint(h.w) << 32);
          }
        }
        const bool in1 = (k1 == key[u]);
        const bool hit = (key[u] != EMPTY_KEY) && (k0 == key[u] || in1);
        inliers += hit ? 1 : 0;
        // 48-byte record of the matching way: the line was just fetched by the key load, so this dependent read stays on chip;
        // every lane reads (way 0 when there is no hit) so the wavefront does not diverge
        const char* rp = reinterpret_cast<const char*>(d.buckets) + (b * 128u + (in1 ? 64u : 16u));
#if GLIM_AMD_ABLATE == 1 || GLIM_AMD_ABLATE == 4
        (void)rp;
        const float4 r0 = make_float4(0.01f * cur[u].p.x, 0.01f, -0.02f, 1.0f);
        const float4 r1 = make_float4(0.01f, 0.02f, 0.9f, 0.03f);
        const float4 r2 = make_float4(0.8f + 0.001f * cur[u].p.y, 0.f, 0.f, 0.f);
#else
        const float4 r0 = gld4(rp);        // mx my mz c00
        const float4 r1 = gld4(rp + 16);   // c01 c02 c11 c12
        const float4 r2 = gld4(rp + 32);   // c22 count - -
#endif
        // C_t = R C_A R^T, the source covariance in the TARGET frame (symmetric: t00 t01 t02 t11 t12 t22)
        float t00, t01, t02, t11, t12, t22;
        if (PLANE) {
          // C_A = I - (1 - 1e-3) n n^T  =>  C_t = I - (1 - 1e-3) m m^T with m = R n: 9 + 9 operations instead of a 45-operation sandwich
          const float nx = cur[u].p.w, ny = cur[u].cb.x, nz = cur[u].cb.y;
          const float mx = R00 * nx + R01 * ny + R02 * nz;
          const float my = R10 * nx + R11 * ny + R12 * nz;
          const float mz = R20 * nx + R21 * ny + R22 * nz;
          const float w = 0.999f, wx = w * mx, wy = w * my;
          t00 = 1.f - wx * mx; t01 = -wx * my; t02 = -wx * mz;
          t11 = 1.f - wy * my; t12 = -wy * mz; t22 = 1.f - w * mz * mz;
        } else {
          const float c00 = cur[u].p.w, c01 = cur[u].ca.x, c02 = cur[u].ca.y, c11 = cur[u].ca.z, c12 = cur[u].ca.w, c22 = cur[u].cb.x;
          const float a00 = R00 * c00 + R01 * c01 + R02 * c02, a01 = R00 * c01 + R01 * c11 + R02 * c12, a02 = R00 * c02 + R01 * c12 + R02 * c22;
          const float a10 = R10 * c00 + R11 * c01 + R12 * c02, a11 = R10 * c01 + R11 * c11 + R12 * c12, a12 = R10 * c02 + R11 * c12 + R12 * c22;
          const float a20 = R20 * c00 + R21 * c01 + R22 * c02, a21 = R20 * c01 + R21 * c11 + R22 * c12, a22 = R20 * c02 + R21 * c12 + R22 * c22;
          t00 = a00 * R00 + a01 * R01 + a02 * R02; t01 = a00 * R10 + a01 * R11 + a02 * R12; t02 = a00 * R20 + a01 * R21 + a02 * R22;
          t11 = a10 * R10 + a11 * R11 + a12 * R12; t12 = a10 * R20 + a11 * R21 + a12 * R22; t22 = a20 * R20 + a21 * R21 + a22 * R22;
        }

        // residual mu - q, both relative to the voxel centre
        const float rx = r0.x - qr[u][0];
        const float ry = r0.y - qr[u][1];
        const float rz = r0.z - qr[u][2];

#if GLIM_AMD_ABLATE == 3
        acc[0] += rx + ry + rz + r0.w + r1.x + r1.y + r1.z + r1.w + r2.x + t00 + t01 + t02 + t11 + t12 + t22 + cur[u].p.x;
        continue;
#endif
        // S = C_B + R C_A R^T   (TARGET frame, symmetric).  Non-hit lanes get C_B = I so the algebra stays finite.
#if GLIM_AMD_ABLATE == 8  // no selects on the voxel covariance: a lane without a match reads some finite record (another voxel's or zeros)
        const float S00 = r0.w + t00, S01 = r1.x + t01, S02 = r1.y + t02;
        const float S11 = r1.z + t11, S12 = r1.w + t12, S22 = r2.x + t22;
#else
        const float S00 = (hit ? r0.w : 1.f) + t00, S01 = (hit ? r1.x : 0.f) + t01, S02 = (hit ? r1.y : 0.f) + t02;
        const float S11 = (hit ? r1.z : 1.f) + t11, S12 = (hit ? r1.w : 0.f) + t12, S22 = (hit ? r2.x : 1.f) + t22;
#endif

        // M = S^-1 by cofactors (symmetric, called A below); idet = 0 on non-hit lanes zeroes every contribution below
        const float k00 = S11 * S22 - S12 * S12;
        const float k01 = S02 * S12 - S01 * S22;
        const float k02 = S01 * S12 - S02 * S11;
        const float det = S00 * k00 + S01 * k01 + S02 * k02;
        float idet = __builtin_amdgcn_rcpf(det);         // 1 ulp hardware reciprocal ...
        idet = fmaf(fmaf(-det, idet, 1.0f), idet, idet);  // ... + one Newton step (full FP32 accuracy, no division sequence)
        idet = hit ? idet : 0.f;
        const float A00 = k00 * idet, A01 = k01 * idet, A02 = k02 * idet;
        const float A11 = (S00 * S22 - S02 * S02) * idet;
        const float A12 = (S01 * S02 - S00 * S12) * idet;
        const float A22 = (S00 * S11 - S01 * S01) * idet;

        // u = M r,  e = r . u   (r = mu - q is already a target-frame vector)
        const float ux = A00 * rx + A01 * ry + A02 * rz;
        const float uy = A01 * rx + A11 * ry + A12 * rz;
        const float uz = A02 * rx + A12 * ry + A22 * rz;
        acc[27] += rx * ux + ry * uy + rz * uz;

        if (MODE == MODE_LINEARIZE) {
          // J_s = [R hat(p) | -R] = [hat(q') | -I] diag(R, R) with q' = R p: everything below is accumulated for J' = [hat(q') | -I] and M in
          // the target frame; the constant diag(R, R) is applied once per factor, in FP64, by finalize_factor
          const float x = qp[u][0], y = qp[u][1], z = qp[u][2];
          // G = hat(q') M : column j = q' x M[:,j]
          const float g00 = y * A02 - z * A01, g01 = y * A12 - z * A11, g02 = y * A22 - z * A12;
          const float g10 = z * A00 - x * A02, g11 = z * A01 - x * A12, g12 = z * A02 - x * A22;
          const float g20 = x * A01 - y * A00, g21 = x * A11 - y * A01, g22 = x * A12 - y * A02;
          // Hww = -G hat(p) : row i = p x G[i,:]
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
          // b_w = u x p,  b_v = -u
          acc[21] += uy * z - uz * y;
          acc[22] += uz * x - ux * z;
          acc[23] += ux * y - uy * x;
          acc[24] += ux; acc[25] += uy; acc[26] += uz;
        }
      }
    }
  }

  // ---- block reduction: DPP wave sums -> LDS -> one partial row ----
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (MODE == MODE_LINEARIZE) {
#pragma unroll
    for (int j = 0; j < NACC; j++) {
      const float v = wave_sum_to_lane63(acc[j]);
      if (lane == 63) s_red[wave][j] = v;
    }
  } else {
    const float v = wave_sum_to_lane63(acc[27]);
    if (lane == 63) s_red[wave][27] = v;
  }
  {
    const float v = wave_sum_to_lane63((float)inliers);  // <= 64 * ppt: exact in FP32
    if (lane == 63) s_red[wave][28] = v;
  }
  __syncthreads();
  if (threadIdx.x < PARTIAL_STRIDE) {
    const int j = threadIdx.x;
    float v = 0.f;
    const bool live = (MODE == MODE_LINEARIZE) ? (j <= 28) : (j == 27 || j == 28);
    if (live) v = (s_red[0][j] + s_red[1][j]) + (s_red[2][j] + s_red[3][j]);
    partials[(size_t)gblock * PARTIAL_STRIDE + j] = v;
  }
}

// Finalisation: one block of 256 threads per factor.
__global__ __launch_bounds__(256) void finalize_kernel(const FactorDesc* __restrict__ descs, const float* __restrict__ partials, const FinalizeArgs fa,
                                                       int mode, const double* __restrict__ poses_lin, const InlinePose ip) {
  __shared__ double s_part[8][PARTIAL_STRIDE];
  __shared__ double s_sum[PARTIAL_STRIDE];
  const int f = blockIdx.x;
  const FactorDesc d = descs[f];
  finalize_factor(d, f, partials, fa, mode, s_part, s_sum, ip.valid ? ip.m : poses_lin + 12 * (size_t)f);
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

// K6: a point counts once if ANY (map_j, delta_j) contains it (odometry_estimation_gpu.cpp:224-231).
__global__ __launch_bounds__(BLOCK) void overlap_kernel(int n, const float4* __restrict__ pts, const OverlapTarget* __restrict__ targets,
                                                         int num_targets, unsigned int* __restrict__ hits) {
  int mine = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 p4 = pts[i];
    for (int t = 0; t < num_targets; t++) {
      double qx, qy, qz;
      transform_point_d(targets[t].T, (double)p4.x, (double)p4.y, (double)p4.z, qx, qy, qz);
      if (find_slot(targets[t].buckets, targets[t].num_buckets, voxel_key(qx, qy, qz, targets[t].inv_res)) >= 0) {
        mine++;
        break;
      }
    }
  }
  const float v = wave_sum_to_lane63((float)mine);  // per-thread counts are tiny: exact in FP32
  if ((threadIdx.x & 63) == 63 && v > 0.f) atomicAdd(hits, (unsigned int)v);
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

const CallSwitches& call_switches() {
  static const CallSwitches s = {getenv("GLIM_AMD_NO_POLL") != nullptr, getenv("GLIM_AMD_NO_INLINE_POSE") != nullptr};
  return s;
}

void factor_set_release_plan(glim_amd_factor_set* set) {
  // the plan's buffers go back to the pool: nothing enqueued earlier (asynchronous entry points included) may still be using them
  if (set->d_descs && set->stream) (void)hipStreamSynchronize(set->stream);
  if (set->d_descs) (void)pool_free(set->d_descs);
  if (set->d_blockmap) (void)pool_free(set->d_blockmap);
  if (set->d_partials) (void)pool_free(set->d_partials);
  if (set->d_poses) (void)pool_free(set->d_poses);
  if (set->d_compact) (void)pool_free(set->d_compact);
  if (set->d_done) (void)pool_free(set->d_done);
  if (set->h_poses) (void)pinned_free(set->h_poses);
  if (set->h_compact) (void)pinned_free(set->h_compact);
  set->d_descs = nullptr;
  set->d_blockmap = nullptr;
  set->d_partials = nullptr;
  set->d_poses = nullptr;
  set->d_compact = nullptr;
  set->d_done = nullptr;
  set->h_poses = nullptr;
  set->h_compact = nullptr;
  set->h_compact_dev = nullptr;
  for (int i = 0; i < glim_amd_factor_set::POSE_RING; i++) set->pose_pending[i] = false;
  set->cap_factors = set->cap_blocks = 0;
}

// Build the device plan: factor descriptors, chunking, and the block -> (factor, chunk) map.
// Factors whose source cloud is plane-form take the 24 B/pt kernel, the others the general 36 B/pt kernel -- decided per factor, so one
// cloud with averaged covariances (a merged submap) does not demote the rest of the set; the two groups occupy separate row segments of
// the plan and are launched separately.
// Workgroup b is dispatched to XCD b % 8 (observed, MI355X_MICROARCH.md "Workgroup dispatch"); when a segment holds enough factors,
// all chunks of one factor are given rows of one residue class so that the factor's voxel table stays in a single XCD's 4 MiB L2.
// This is a speed-only choice: any placement is correct.
int factor_set_prepare(glim_amd_factor_set* set) {
  if (!set->dirty) return GLIM_AMD_OK;
  const int nf = (int)set->entries.size();
  glim_amd_ctx* ctx = set->ctx;
  long long total_points = 0;
  for (auto& e : set->entries) total_points += e.source->n;
  // Grid sizing: aim for ONE resident set of blocks (num_cus x blocks_per_cu) with equal work each, so every lane reduces its 28
  // accumulators exactly once; each factor gets blocks in proportion to its points.  (Finer grids were measured level: 1280 blocks
  // 142.7 us, 2560 142.6, 5120 138.6, 10240 142.9 per 128 factors.)
  const int blocks_per_cu = 5;  // 96 VGPRs -> 5 waves/SIMD -> 5 blocks of 4 waves per CU
  long long target_blocks = (long long)std::max(1, ctx->num_cus) * blocks_per_cu;
  if (const char* env = getenv("GLIM_AMD_TARGET_BLOCKS")) target_blocks = std::max(1, atoi(env));
  int forced_ppt = 0;
  if (const char* env = getenv("GLIM_AMD_PPT")) forced_ppt = std::max(1, std::min(256, atoi(env)));
  const bool allow_plane = getenv("GLIM_AMD_NO_PLANE") == nullptr;

  set->h_descs.assign(nf, FactorDesc());
  std::vector<int> nblocks(nf);
  for (int f = 0; f < nf; f++) {
    const auto& e = set->entries[f];
    glim_amd_cloud* src = const_cast<glim_amd_cloud*>(e.source);  // lazily built stream copies are a cache, not a change of the cloud
    if (!allow_plane && src->plane_form && !src->gs0) {
      // diagnostic switch: run a plane-form cloud through the general kernel (its covariance arrays hold the same matrix)
      src->plane_form = false;
      const int rc = ensure_factor_streams(src, set->stream);
      src->plane_form = true;
      GA_TRY(rc);
    } else {
      GA_TRY(ensure_factor_streams(src, set->stream));
    }
    FactorDesc& d = set->h_descs[f];
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
    d.n = (int)src->n;
    d.inv_res = e.target->inv_resolution;
    d.res = e.target->resolution;
    d.flags = e.flags;
    int ppt = forced_ppt;
    if (!ppt) {
      const long long share = std::max(1ll, (target_blocks * (long long)d.n + total_points / 2) / std::max(1ll, total_points));
      ppt = (int)std::max(1ll, std::min(256ll, ((long long)d.n + share * BLOCK - 1) / (share * BLOCK)));
    }
    d.ppt = ppt;
    nblocks[f] = std::max(1, (d.n + BLOCK * ppt - 1) / (BLOCK * ppt));
    d.num_blocks = nblocks[f];
  }
  set->points_per_thread = nf ? set->h_descs[0].ppt : 1;

  // block map: the plane-form segment first, then the general one
  std::vector<int2> blockmap;
  const bool want_xcd = getenv("GLIM_AMD_NO_XCD_MAP") == nullptr;
  int seg_rows[2] = {0, 0};
  for (int seg = 0; seg < 2; seg++) {  // seg 0: plane-form factors, seg 1: the others
    std::vector<int> fs;
    for (int f = 0; f < nf; f++)
      if ((set->h_descs[f].plane != 0) == (seg == 0)) fs.push_back(f);
    const size_t base = blockmap.size();
    if (fs.size() < 16 || !want_xcd) {
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
  }
  // rows[]: for each factor the plan rows that hold its partial sums, in chunk order
  long long total_blocks = 0;
  for (int f = 0; f < nf; f++) {
    set->h_descs[f].first_block = (int)total_blocks;
    total_blocks += nblocks[f];
  }
  std::vector<int> rows((size_t)total_blocks);
  for (size_t b = 0; b < blockmap.size(); b++) {
    if (blockmap[b].x < 0) continue;
    rows[(size_t)set->h_descs[blockmap[b].x].first_block + blockmap[b].y] = (int)b;
  }

  factor_set_release_plan(set);
  set->plane_rows = seg_rows[0];
  set->total_rows = (int)blockmap.size();
  const size_t nfa = (size_t)std::max(1, nf), nba = std::max<size_t>(1, blockmap.size());
  GA_HIP(pool_malloc(&set->d_descs, nfa * sizeof(FactorDesc)));
  GA_HIP(pool_malloc(&set->d_blockmap, nba * sizeof(int2) + std::max<size_t>(1, rows.size()) * sizeof(int)));
  GA_HIP(pool_malloc(&set->d_partials, nba * PARTIAL_STRIDE * sizeof(float)));
  GA_HIP(pool_malloc(&set->d_poses, nfa * 24 * sizeof(double)));
  GA_HIP(pool_malloc(&set->d_compact, nfa * COMPACT * sizeof(double)));
  GA_HIP(pool_malloc(&set->d_done, sizeof(int)));
  GA_HIP(hipMemsetAsync(set->d_done, 0, sizeof(int), set->stream));
  if (!set->h_flag) {
    if (pinned_malloc(&set->h_flag, 64) == hipSuccess) {
      *set->h_flag = 0;
      if (hipHostGetDevicePointer(reinterpret_cast<void**>(&set->h_flag_dev), set->h_flag, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)pinned_free(set->h_flag);
        set->h_flag = nullptr;
        set->h_flag_dev = nullptr;
      }
    } else {
      (void)hipGetLastError();
      set->h_flag = nullptr;
    }
  }
  GA_HIP(pinned_malloc(&set->h_poses, (size_t)glim_amd_factor_set::POSE_RING * nfa * 24 * sizeof(double)));
  GA_HIP(pinned_malloc(&set->h_compact, nfa * COMPACT * sizeof(double)));
  set->h_compact_dev = nullptr;
  if (hipHostGetDevicePointer(reinterpret_cast<void**>(&set->h_compact_dev), set->h_compact, 0) != hipSuccess) {
    (void)hipGetLastError();
    set->h_compact_dev = nullptr;
  }
  set->cap_factors = nfa;
  set->cap_blocks = nba;
  if (nf > 0) {
    GA_HIP(hipMemcpyAsync(set->d_descs, set->h_descs.data(), (size_t)nf * sizeof(FactorDesc), hipMemcpyHostToDevice, set->stream));
    GA_HIP(hipMemcpyAsync(set->d_blockmap, blockmap.data(), blockmap.size() * sizeof(int2), hipMemcpyHostToDevice, set->stream));
    GA_HIP(hipMemcpyAsync(reinterpret_cast<char*>(set->d_blockmap) + nba * sizeof(int2), rows.data(), rows.size() * sizeof(int),
                          hipMemcpyHostToDevice, set->stream));
    GA_HIP(hipStreamSynchronize(set->stream));  // the staging vectors above die with this scope
  }
  set->dirty = false;
  return GLIM_AMD_OK;
}

}  // namespace glim_amd

namespace {

const int* rows_ptr(const glim_amd_factor_set* set) {
  return reinterpret_cast<const int*>(reinterpret_cast<const char*>(set->d_blockmap) + set->cap_blocks * sizeof(int2));
}

FinalizeArgs finalize_args(const glim_amd_factor_set* set, double* out, long long row_offset, bool poll) {
  FinalizeArgs fa;
  fa.rows = rows_ptr(set);
  fa.out = out;
  fa.out_row_offset = row_offset;
  fa.done_counter = set->d_done;
  fa.host_flag = poll ? set->h_flag_dev : nullptr;
  fa.seq = set->poll_seq;
  fa.num_factors = (int)set->entries.size();
  return fa;
}

template <int MODE, bool FROZEN, bool INLINE>
void launch_segments(glim_amd_factor_set* set, const FinalizeArgs& fa) {
  const double* lin = set->d_poses;
  const double* ev = set->d_poses + set->entries.size() * 12;
  if (set->plane_rows > 0)
    vgicp_kernel<MODE, FROZEN, true, INLINE><<<set->plane_rows, BLOCK, 0, set->stream>>>(set->d_descs, lin, ev, set->d_blockmap, set->d_partials,
                                                                                       set->inline_pose, fa, 0);
  if (set->total_rows > set->plane_rows)
    vgicp_kernel<MODE, FROZEN, false, INLINE><<<set->total_rows - set->plane_rows, BLOCK, 0, set->stream>>>(
      set->d_descs, lin, ev, set->d_blockmap, set->d_partials, set->inline_pose, fa, set->plane_rows);
}

// the fused kernel(s) alone (no finalisation): used by the profiling entry point
void launch_vgicp(glim_amd_factor_set* set, int mode, bool frozen, const FinalizeArgs& fa) {
  const bool inl = set->inline_pose.valid != 0;
  if (mode == MODE_LINEARIZE) {
    if (inl) launch_segments<MODE_LINEARIZE, false, true>(set, fa);
    else launch_segments<MODE_LINEARIZE, false, false>(set, fa);
  } else if (frozen) {
    launch_segments<MODE_ERROR, true, false>(set, fa);
  } else {
    if (inl) launch_segments<MODE_ERROR, false, true>(set, fa);
    else launch_segments<MODE_ERROR, false, false>(set, fa);
  }
}

// enqueue (no sync): poses already in d_poses / the inline pose; writes compact records to `out` rows [row_offset, row_offset + n).
// Two or three launches: the fused kernel per plan segment + the FP64 finalise.
int enqueue(glim_amd_factor_set* set, int mode, bool frozen, double* out, long long row_offset, bool poll) {
  const int nf = (int)set->entries.size();
  if (nf == 0) return GLIM_AMD_OK;
  const FinalizeArgs fa = finalize_args(set, out, row_offset, poll);
  launch_vgicp(set, mode, frozen, fa);
  finalize_kernel<<<nf, 256, 0, set->stream>>>(set->d_descs, set->d_partials, fa, mode, set->d_poses, set->inline_pose);
  GA_HIP(hipGetLastError());
  return GLIM_AMD_OK;
}

// Poses to the device.  Single-factor sets without an evaluation pose pass the pose in the kernel arguments (no copy at all).  Otherwise
// the poses are staged in one slot of a pinned ring and copied asynchronously; a slot is reused only after the copy that read it has
// completed (event), so back-to-back asynchronous calls never see each other's poses.
int upload_poses(glim_amd_factor_set* set, const double* T_lin, const double* T_eval, bool async_call) {
  const size_t nf = set->entries.size();
  set->inline_pose.valid = 0;
  if (nf == 1 && !T_eval && !call_switches().no_inline_pose) {
    memcpy(set->inline_pose.m, T_lin, 12 * sizeof(double));
    set->inline_pose.valid = 1;
    return GLIM_AMD_OK;
  }
  const int slot = set->pose_slot;
  set->pose_slot = (slot + 1) % glim_amd_factor_set::POSE_RING;
  if (set->pose_pending[slot]) {
    GA_HIP(hipEventSynchronize(set->pose_events[slot]));
    set->pose_pending[slot] = false;
  }
  double* h = set->h_poses + (size_t)slot * set->cap_factors * 24;
  memcpy(h, T_lin, nf * 12 * sizeof(double));
  if (T_eval) memcpy(h + nf * 12, T_eval, nf * 12 * sizeof(double));
  GA_HIP(hipMemcpyAsync(set->d_poses, h, nf * (T_eval ? 24 : 12) * sizeof(double), hipMemcpyHostToDevice, set->stream));
  if (async_call) {
    if (!set->pose_events[slot]) GA_HIP(hipEventCreateWithFlags(&set->pose_events[slot], hipEventDisableTiming));
    GA_HIP(hipEventRecord(set->pose_events[slot], set->stream));
    set->pose_pending[slot] = true;
  }
  return GLIM_AMD_OK;
}

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield" ::: "memory");
#endif
}

// One synchronous evaluation (linearise or error) of the whole set: results in set->h_compact when this returns.
// Small sets (the per-frame odometry case): the finalise kernel writes the 232-byte records straight into host-mapped pinned memory and
// then publishes a sequence number there; the host spins on that word (sub-microsecond wake-up) instead of paying a stream-synchronise
// round trip -- no device-to-host copy, and for a single factor no host-to-device copy either.  The context mutex is held only while
// the work is enqueued, so factor sets of one context (different streams of its pool) overlap on the device when driven from
// different host threads, like the reference's StreamTempBufferRoundRobin factors.
int run_sync(glim_amd_factor_set* set, int mode, const double* T_lin, const double* T_eval) {
  const size_t nf = set->entries.size();
  bool poll = false;
  {
    std::lock_guard<std::mutex> lock(set->ctx->mu);
    GA_HIP(hipSetDevice(set->ctx->device));
    GA_TRY(factor_set_prepare(set));
    GA_TRY(upload_poses(set, T_lin, T_eval, false));
    const bool mapped = set->h_compact_dev && nf <= 1024;
    poll = mapped && set->h_flag && !call_switches().no_poll;
    if (poll) set->poll_seq++;
    GA_TRY(enqueue(set, mode, T_eval != nullptr, mapped ? set->h_compact_dev : set->d_compact, 0, poll));
    if (!mapped) GA_HIP(hipMemcpyAsync(set->h_compact, set->d_compact, nf * COMPACT * sizeof(double), hipMemcpyDeviceToHost, set->stream));
  }
  bool done = false;
  if (poll) {
    const auto t0 = std::chrono::steady_clock::now();
    volatile unsigned int* flag = set->h_flag;
    for (unsigned long spins = 0;; spins++) {
      if (*flag == set->poll_seq) {
        done = true;
        break;
      }
      cpu_relax();
      if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;  // fall back
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  if (!done) GA_HIP(hipStreamSynchronize(set->stream));
  return GLIM_AMD_OK;
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
  (void)hipSetDevice(set->ctx->device);
  (void)hipStreamSynchronize(set->stream);
  factor_set_release_plan(set);
  for (int i = 0; i < glim_amd_factor_set::POSE_RING; i++)
    if (set->pose_events[i]) (void)hipEventDestroy(set->pose_events[i]);
  if (set->h_flag) (void)pinned_free(set->h_flag);
  delete set;
  return GLIM_AMD_OK;
}

int glim_amd_factor_set_add(glim_amd_factor_set* set, const glim_amd_voxelmap* target, const glim_amd_cloud* source, uint32_t flags,
                            int32_t* factor_index) {
  if (!set || !target || !source) return GLIM_AMD_ERR_INVALID;
  if (target->ctx != set->ctx || source->ctx != set->ctx) return GLIM_AMD_ERR_INVALID;
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
  for (size_t f = 0; f < nf; f++) glim_amd_expand_compact(set->h_compact + f * COMPACT, T + 12 * f, set->entries[f].flags, &out[f]);
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
  return enqueue(set, MODE_LINEARIZE, false, out_device, out_row_offset, false);
}

int glim_amd_factor_set_error(glim_amd_factor_set* set, const double* T_lin, const double* T_eval, double* errors, int64_t* inliers) {
  if (!set) return GLIM_AMD_ERR_INVALID;
  const size_t nf = set->entries.size();
  if (nf == 0) return GLIM_AMD_OK;
  if (!T_eval || !errors) return GLIM_AMD_ERR_INVALID;
  // T_lin == NULL: correspondences at the evaluation pose (one pose per factor); otherwise frozen at T_lin and evaluated at T_eval
  if (T_lin) GA_TRY(run_sync(set, MODE_ERROR, T_lin, T_eval));
  else GA_TRY(run_sync(set, MODE_ERROR, T_eval, nullptr));
  for (size_t f = 0; f < nf; f++) {
    errors[f] = set->h_compact[f * COMPACT + 1];
    if (inliers) inliers[f] = (int64_t)llround(set->h_compact[f * COMPACT]);
  }
  return GLIM_AMD_OK;
}

int glim_amd_factor_set_correspondences(glim_amd_factor_set* set, int32_t fi, const double* T, int32_t* corr) {
  if (!set || !T || fi < 0 || fi >= (int32_t)set->entries.size()) return GLIM_AMD_ERR_INVALID;
  std::lock_guard<std::mutex> lock(set->ctx->mu);
  GA_HIP(hipSetDevice(set->ctx->device));
  GA_TRY(factor_set_prepare(set));
  const FactorDesc d = set->h_descs[fi];
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
  hipEvent_t e0, e1;
  GA_HIP(hipEventCreate(&e0));
  GA_HIP(hipEventCreate(&e1));
  // warm-up (clocks, caches, TLBs)
  for (int i = 0; i < 10; i++) GA_TRY(enqueue(set, MODE_LINEARIZE, false, set->d_compact, 0, false));
  GA_HIP(hipStreamSynchronize(set->stream));
  float ms = 0.f;
  GA_HIP(hipEventRecord(e0, set->stream));
  const FinalizeArgs fa_prof = finalize_args(set, set->d_compact, 0, false);
  for (int i = 0; i < iters; i++) launch_vgicp(set, MODE_LINEARIZE, false, fa_prof);
  GA_HIP(hipEventRecord(e1, set->stream));
  GA_HIP(hipEventSynchronize(e1));
  GA_HIP(hipEventElapsedTime(&ms, e0, e1));
  if (ms_kernel) *ms_kernel = ms / (float)iters;
  GA_HIP(hipEventRecord(e0, set->stream));
  for (int i = 0; i < iters; i++) GA_TRY(enqueue(set, MODE_LINEARIZE, false, set->d_compact, 0, false));
  GA_HIP(hipEventRecord(e1, set->stream));
  GA_HIP(hipEventSynchronize(e1));
  GA_HIP(hipEventElapsedTime(&ms, e0, e1));
  if (ms_linearize) *ms_linearize = ms / (float)iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return GLIM_AMD_OK;
}

int glim_amd_overlap(glim_amd_ctx* ctx, int32_t num_targets, const glim_amd_voxelmap* const* targets, const double* T,
                     const glim_amd_cloud* source, double* overlap) {
  if (!ctx || num_targets <= 0 || !targets || !T || !source || !overlap) return GLIM_AMD_ERR_INVALID;
  for (int t = 0; t < num_targets; t++) {
    if (!targets[t] || targets[t]->ctx != ctx) return GLIM_AMD_ERR_INVALID;
    if (!targets[t]->buckets) return GLIM_AMD_ERR_STATE;
  }
  if (source->ctx != ctx) return GLIM_AMD_ERR_INVALID;
  if (source->n == 0) {
    *overlap = 0.0;
    return GLIM_AMD_OK;
  }
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream();
  std::vector<OverlapTarget> h(num_targets);
  for (int t = 0; t < num_targets; t++) {
    h[t].buckets = targets[t]->buckets;
    h[t].num_buckets = targets[t]->num_buckets;
    h[t].pad = 0;
    h[t].inv_res = targets[t]->inv_resolution;
    memcpy(h[t].T, T + 12 * (size_t)t, 12 * sizeof(double));
  }
  OverlapTarget* d_t = nullptr;
  unsigned int* d_hits = nullptr;
  GA_HIP(pool_malloc(&d_t, (size_t)num_targets * sizeof(OverlapTarget)));
  hipError_t e = pool_malloc(&d_hits, sizeof(unsigned int));
  if (e == hipSuccess) e = hipMemsetAsync(d_hits, 0, sizeof(unsigned int), st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_t, h.data(), (size_t)num_targets * sizeof(OverlapTarget), hipMemcpyHostToDevice, st);
  unsigned int hits = 0;
  if (e == hipSuccess) {
    const int n = (int)source->n;
    const int blocks = std::min((n + BLOCK - 1) / BLOCK, std::max(1, ctx->num_cus) * 8);
    overlap_kernel<<<blocks, BLOCK, 0, st>>>(n, source->pts, d_t, num_targets, d_hits);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(&hits, d_hits, sizeof(hits), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)pool_free(d_t);
  if (d_hits) (void)pool_free(d_hits);
  if (e != hipSuccess) {
    set_hip_error(e, "overlap");
    return GLIM_AMD_ERR_HIP;
  }
  *overlap = (double)hits / (double)source->n;
  return GLIM_AMD_OK;
}

}  // extern "C"
