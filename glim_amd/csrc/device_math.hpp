// device_math.hpp -- device-side helpers shared by the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>

#include "internal.hpp"

namespace glim_amd {

// FP64 multiply / add that the compiler must NOT fuse into an FMA.  HIP compiles with -ffp-contract=fast-honor-pragmas and the
// header's __dmul_rn / __dadd_rn are plain `x * y` / `x + y` (contractable), so the parity-critical expressions -- the ones the
// CPU oracle evaluates with separate roundings (gcc -ffp-contract=off) -- are written with these two helpers.
__device__ __forceinline__ double dmul(double a, double b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ double dadd(double a, double b) {
#pragma clang fp contract(off)
  return a + b;
}

// fast_floor(x) = (int)x - (x < (int)x): the voxel-coordinate rule of the reference
// (gtsam_points util/fast_floor.hpp; in-tree twin src/glim/viewer/editor/points_selector.cpp:177).
__device__ __forceinline__ int fast_floor_d(double x) {
  const int i = __double2int_rz(x);
  return i - (x < (double)i);
}

// q = R p + t in FP64 with the exact fma order of oracle/vgicp_oracle.c:orc_transform_point -- this expression is
// the parity contract that makes voxel coordinates (and hence correspondences) bit-identical to the CPU path.
__device__ __forceinline__ void transform_point_d(const double* __restrict__ T, double px, double py, double pz, double& qx,
                                                  double& qy, double& qz) {
  qx = __fma_rn(T[0], px, __fma_rn(T[1], py, __fma_rn(T[2], pz, T[3])));
  qy = __fma_rn(T[4], px, __fma_rn(T[5], py, __fma_rn(T[6], pz, T[7])));
  qz = __fma_rn(T[8], px, __fma_rn(T[9], py, __fma_rn(T[10], pz, T[11])));
}

// 3 x 21-bit packed voxel coordinate.  Returns EMPTY_KEY when a coordinate is outside [-2^20, 2^20).
__device__ __forceinline__ unsigned long long pack_key(int cx, int cy, int cz) {
  const unsigned int ux = (unsigned int)(cx + KEY_OFFSET);
  const unsigned int uy = (unsigned int)(cy + KEY_OFFSET);
  const unsigned int uz = (unsigned int)(cz + KEY_OFFSET);
  if ((ux | uy | uz) >> KEY_BITS) return EMPTY_KEY;
  return ((unsigned long long)ux << (2 * KEY_BITS)) | ((unsigned long long)uy << KEY_BITS) | (unsigned long long)uz;
}

__device__ __forceinline__ void unpack_key(unsigned long long key, int& cx, int& cy, int& cz) {
  const unsigned int m = (1u << KEY_BITS) - 1u;
  cx = (int)((key >> (2 * KEY_BITS)) & m) - KEY_OFFSET;
  cy = (int)((key >> KEY_BITS) & m) - KEY_OFFSET;
  cz = (int)(key & m) - KEY_OFFSET;
}

__device__ __forceinline__ unsigned long long voxel_key(double qx, double qy, double qz, double inv_res) {
  return pack_key(fast_floor_d(qx * inv_res), fast_floor_d(qy * inv_res), fast_floor_d(qz * inv_res));
}

// 63-bit key -> 32-bit hash with well-mixed HIGH bits (the bucket index is the multiply-shift range reduction
// (hash * num_buckets) >> 32, so any table size works and no power-of-two rounding wastes memory).  The three 21-bit axis
// fields are combined with full-rate 24-bit multiply-adds, then one xor-shift-multiply round.  Any hash is valid because
// lookups compare the full key.
// GLIM_AMD_PAIR_SHIFT = 1: the two x-adjacent voxels 2m and 2m+1 hash to the same bucket and share its two ways, i.e. one 128-byte
// line.  The L2 fetches whole 128-byte lines from HBM (TCC_EA0_RDREQ_128B = all read requests of the factor kernel, none 32/64-byte), so
// a lookup costs a line per touched BUCKET; with the source stream in Hilbert order the two voxels of a pair are looked up close in
// time and the pair costs one line instead of two.  Measured on MI355X (128 x 131 072-pt factors, 0.5 m maps): 143 -> 131 us per
// launch, provided the table is large enough (6 buckets per voxel) that a bucket rarely receives two different pairs -- at 3 buckets
// per voxel the extra spills to the next bucket eat the gain (145 us), which is what the first trial of this idea measured.
#ifndef GLIM_AMD_PAIR_SHIFT
#define GLIM_AMD_PAIR_SHIFT 1
#endif
// ux, uy, uz: the three 21-bit fields, already confined to 21 bits (a lane whose coordinate is out of range carries EMPTY_KEY and only
// needs SOME in-range bucket)
__device__ __forceinline__ unsigned int hash_fields(unsigned int ux, unsigned int uy, unsigned int uz) {
  unsigned int h = __umul24(ux >> GLIM_AMD_PAIR_SHIFT, 0x9E3779u) + __umul24(uy, 0x85EBCBu) + __umul24(uz, 0xC2B2AFu);
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 13;
  return h;
}
__device__ __forceinline__ unsigned int hash_key(unsigned long long k) {
  const unsigned int m = (1u << KEY_BITS) - 1u;
  return hash_fields((unsigned int)(k >> (2 * KEY_BITS)) & m, (unsigned int)(k >> KEY_BITS) & m, (unsigned int)k & m);
}
__device__ __forceinline__ unsigned int bucket_of(unsigned long long key, unsigned int num_buckets) {
  return __umulhi(hash_key(key), num_buckets);
}

// Exact lookup: returns 2 * bucket + way, or -1.  Buckets fill way 0 first, then way 1, then spill to the next bucket, so the
// first EMPTY key met ends the search.  The table always holds free ways (>= 4 ways per key).
__device__ __forceinline__ int find_slot(const VoxelBucket* __restrict__ buckets, unsigned int num_buckets, unsigned long long key) {
  if (key == EMPTY_KEY) return -1;
  unsigned int b = bucket_of(key, num_buckets);
  for (;;) {
    const unsigned long long k0 = buckets[b].key[0], k1 = buckets[b].key[1];
    if (k0 == key) return (int)(2 * b);
    if (k0 == EMPTY_KEY) return -1;
    if (k1 == key) return (int)(2 * b + 1);
    if (k1 == EMPTY_KEY) return -1;
    b = (b + 1 == num_buckets) ? 0u : b + 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 64-lane wavefront sum with DPP (no LDS traffic): quad xor-1, xor-2, row_half_mirror, row_mirror give every lane of a
// 16-lane row the row sum; row_bcast:15 and row_bcast:31 fold the four rows, the total lands in lane 63.
// ---------------------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}

// N wavefront sums at once, STEP-major: step k of all N values before step k + 1 of any.  Called value by value (wave_sum_to_lane63 in a loop) the
// compiler emits N dependent chains one after the other, every DPP read two wait states behind the write it depends on (s_nop) -- 28 values: ~400
// instructions of which 170 are s_nop / register copies, ~3 400 cycles of ONE wavefront, 1.35 us on the device timeline of the synchronous call
// (round 6: the point loop is left 2.7 us after the request, the row is block-reduced at 4.05).  Step-major the N values of a step are independent:
// no wait states, 6 N additions.  Every value goes through the same six additions in the same order as in wave_sum_to_lane63: the same bits.
template <int N>
__device__ __forceinline__ void wave_sums_to_lane63(float (&v)[N]) {
#pragma unroll
  for (int j = 0; j < N; j++) v[j] += dpp_f<0xB1, 0xF>(v[j]);   // quad_perm [1,0,3,2]
#pragma unroll
  for (int j = 0; j < N; j++) v[j] += dpp_f<0x4E, 0xF>(v[j]);   // quad_perm [2,3,0,1]
#pragma unroll
  for (int j = 0; j < N; j++) v[j] += dpp_f<0x141, 0xF>(v[j]);  // row_half_mirror
#pragma unroll
  for (int j = 0; j < N; j++) v[j] += dpp_f<0x140, 0xF>(v[j]);  // row_mirror
#pragma unroll
  for (int j = 0; j < N; j++) v[j] += dpp_f<0x142, 0xA>(v[j]);  // row_bcast:15 -> rows 1,3
#pragma unroll
  for (int j = 0; j < N; j++) v[j] += dpp_f<0x143, 0xC>(v[j]);  // row_bcast:31 -> rows 2,3
  // (the row_bcast steps stay three instructions per value -- zero, masked move, add: giving the masked-out rows -0.0, the addition's identity,
  //  does not make the compiler fold them into one v_add_f32_dpp either)
}

// The first four steps only: every lane of a 16-lane row ends up with its ROW's sum (lanes 15 / 31 / 47 / 63 are where the two row_bcast steps of
// wave_sums_to_lane63 would read them).  A caller that hands the four row sums r0..r3 to somebody who adds them as (r3 + r2) + (r1 + r0) -- what lane 63
// holds after row_bcast:15 and row_bcast:31 -- gets the same bits for 4 N instead of 10 N instructions per wavefront.
template <int N>
__device__ __forceinline__ void wave_row_sums(float (&v)[N]) {
#pragma unroll
  for (int j = 0; j < N; j++) v[j] += dpp_f<0xB1, 0xF>(v[j]);   // quad_perm [1,0,3,2]
#pragma unroll
  for (int j = 0; j < N; j++) v[j] += dpp_f<0x4E, 0xF>(v[j]);   // quad_perm [2,3,0,1]
#pragma unroll
  for (int j = 0; j < N; j++) v[j] += dpp_f<0x141, 0xF>(v[j]);  // row_half_mirror
#pragma unroll
  for (int j = 0; j < N; j++) v[j] += dpp_f<0x140, 0xF>(v[j]);  // row_mirror
}

__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v += dpp_f<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += dpp_f<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += dpp_f<0x141, 0xF>(v);  // row_half_mirror
  v += dpp_f<0x140, 0xF>(v);  // row_mirror
  v += dpp_f<0x142, 0xA>(v);  // row_bcast:15 -> rows 1,3
  v += dpp_f<0x143, 0xC>(v);  // row_bcast:31 -> rows 2,3
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// block-level integer reductions (blocks of <= 1024 threads): wavefront butterflies, then one LDS round; the result is
// valid in thread 0.  Used so that a kernel issues ONE set of global atomics per block: atomics of many wavefronts on
// the same address serialise in L2 (measured ~10 ns each on MI355X -- 12 000 of them cost more than the kernel body).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// OP: 0 = min, 1 = max, 2 = sum.  s_tmp: >= 16 ints of LDS per call site (a __syncthreads() separates consecutive calls).
template <int OP>
__device__ __forceinline__ int block_reduce_i(int v, int* s_tmp) {
  v = OP == 0 ? wave_min_i(v) : (OP == 1 ? wave_max_i(v) : wave_sum_i(v));
  const int wave = threadIdx.x >> 6, waves = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) s_tmp[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < waves; w++) v = OP == 0 ? min(v, s_tmp[w]) : (OP == 1 ? max(v, s_tmp[w]) : v + s_tmp[w]);
  }
  __syncthreads();
  return v;
}

// Reset of a bounding-box accumulator (3 minima, 3 maxima as order-preserving ints) on the device: copying six ints from a stack array is a
// staged, blocking host-to-device transfer (~10 us), a one-wave kernel is an ordinary asynchronous launch.
static __global__ void init_bbox_kernel(int* __restrict__ bb) {
  if (threadIdx.x < 6) bb[threadIdx.x] = threadIdx.x < 3 ? 0x7fffffff : (int)0x80000000;
}

}  // namespace glim_amd
