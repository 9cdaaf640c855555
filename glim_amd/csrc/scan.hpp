// scan.hpp -- exclusive prefix sum of an int array on the device in three launches (per-tile sums, scan of the tile sums in
// one block, per-tile scan with the tile offset).  Shared by the kNN grid build (knn.hip), the radix sort (sort.hip) and the scan
// preprocessing (preprocess.hip).  Hand-written: no rocPRIM / hipCUB.
#pragma once
#include <hip/hip_runtime.h>

namespace glim_amd {
namespace scan_detail {

// exclusive scan of counts[T] -> starts[T] in three launches: per-tile sums, scan of the tile sums (one block), per-tile scan
// with the tile offset.  A tile is SCAN_TILE consecutive entries read coalesced (lane-contiguous int4).
constexpr int SCAN_TILE = 4096;  // 1024 threads x int4

__device__ __forceinline__ int block_exclusive_scan_1024(int v, int* s_tmp, int* total) {
  // wave-level inclusive scan by shuffles, then a scan of the 16 wave sums
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) s_tmp[wave] = inc;
  __syncthreads();
  if (threadIdx.x < 16) {
    int w = s_tmp[threadIdx.x];
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const int t = __shfl_up(w, off, 16);
      if ((int)threadIdx.x >= off) w += t;
    }
    s_tmp[16 + threadIdx.x] = w;  // inclusive scan of wave sums
  }
  __syncthreads();
  const int wave_off = wave ? s_tmp[16 + wave - 1] : 0;
  if (total) *total = s_tmp[31];
  return wave_off + inc - v;
}

static __global__ __launch_bounds__(1024) void scan_tile_sums_kernel(const int* __restrict__ counts, unsigned int T, int* __restrict__ tile_sums) {
  __shared__ int s_tmp[32];
  const unsigned int i = blockIdx.x * SCAN_TILE + threadIdx.x * 4;
  int4 c = make_int4(0, 0, 0, 0);
  if (i + 3 < T) c = *reinterpret_cast<const int4*>(counts + i);
  else {
    if (i < T) c.x = counts[i];
    if (i + 1 < T) c.y = counts[i + 1];
    if (i + 2 < T) c.z = counts[i + 2];
  }
  int total = 0;
  block_exclusive_scan_1024(c.x + c.y + c.z + c.w, s_tmp, &total);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

static __global__ __launch_bounds__(1024) void scan_tile_offsets_kernel(int* __restrict__ tile_sums, int num_tiles) {
  // exclusive scan of the tile sums in place (num_tiles <= 1024 * chunk handled by a serial carry over chunks of 1024)
  __shared__ int s_tmp[32];
  int carry = 0;
  for (int base = 0; base < num_tiles; base += 1024) {
    const int i = base + (int)threadIdx.x;
    const int v = i < num_tiles ? tile_sums[i] : 0;
    int total = 0;
    const int ex = block_exclusive_scan_1024(v, s_tmp, &total);
    if (i < num_tiles) tile_sums[i] = carry + ex;
    carry += total;
    __syncthreads();
  }
}

static __global__ __launch_bounds__(1024) void scan_apply_kernel(const int* __restrict__ counts, unsigned int T, const int* __restrict__ tile_offsets,
                                                          int* __restrict__ starts) {
  __shared__ int s_tmp[32];
  const unsigned int i = blockIdx.x * SCAN_TILE + threadIdx.x * 4;
  int4 c = make_int4(0, 0, 0, 0);
  if (i + 3 < T) c = *reinterpret_cast<const int4*>(counts + i);
  else {
    if (i < T) c.x = counts[i];
    if (i + 1 < T) c.y = counts[i + 1];
    if (i + 2 < T) c.z = counts[i + 2];
  }
  const int ex = tile_offsets[blockIdx.x] + block_exclusive_scan_1024(c.x + c.y + c.z + c.w, s_tmp, nullptr);
  const int4 o = make_int4(ex, ex + c.x, ex + c.x + c.y, ex + c.x + c.y + c.z);
  if (i + 3 < T) *reinterpret_cast<int4*>(starts + i) = o;
  else {
    if (i < T) starts[i] = o.x;
    if (i + 1 < T) starts[i + 1] = o.y;
    if (i + 2 < T) starts[i + 2] = o.z;
  }
}

}  // namespace scan_detail

// number of ints of scratch `tile_sums` must hold for an array of T entries
inline size_t scan_scratch_ints(unsigned int T) { return (size_t)((T + scan_detail::SCAN_TILE - 1) / scan_detail::SCAN_TILE + 1); }

// out[i] = sum(in[0..i)), i < T.  Enqueues on `st`; no synchronisation.
inline hipError_t exclusive_scan_int(hipStream_t st, const int* in, unsigned int T, int* tile_sums, int* out) {
  if (T == 0) return hipSuccess;
  const int tiles = (int)((T + scan_detail::SCAN_TILE - 1) / scan_detail::SCAN_TILE);
  scan_detail::scan_tile_sums_kernel<<<tiles, 1024, 0, st>>>(in, T, tile_sums);
  scan_detail::scan_tile_offsets_kernel<<<1, 1024, 0, st>>>(tile_sums, tiles);
  scan_detail::scan_apply_kernel<<<tiles, 1024, 0, st>>>(in, T, tile_sums, out);
  return hipGetLastError();
}

}  // namespace glim_amd
