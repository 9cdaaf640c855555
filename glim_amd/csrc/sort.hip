// sort.hip -- stable LSD radix sort of (u64 key, u32 value) pairs on the device, hand-written for gfx950 (no rocPRIM / hipCUB).
//
// Used by the scan preprocessing (preprocess.hip): sorting points by voxel key (gtsam_points voxelgrid_sampling /
// randomgrid_sampling sort their (coord, index) pairs, cloud_preprocessor.cpp:104-109) and by time stamp
// (std::sort, cloud_preprocessor.cpp:135).  Stability is what makes the device result reproduce the oracle's
// (key, original index) order exactly.
//
// One pass = 8 key bits, three launches:
//   rs_hist     every block owns a tile of RS_TILE consecutive pairs and counts its 256 digit values in LDS;
//               hist[digit][block] (digit-major) goes to global memory
//   rs_offsets  ONE block turns the 256 x B table into global exclusive offsets (digit-major order == the output order of
//               a stable sort: all pairs with a smaller digit first, then the same digit in earlier blocks)
//   rs_scatter  every block re-reads its tile in 8 rounds of 256 threads; inside a round the rank of a pair among the equal
//               digits is  (equal digits in earlier rounds)  +  (equal digits in earlier waves of this round)  +
//               (equal digits in lower lanes of this wave); the last term comes from 8 ballots (one per digit bit), so the
//               order inside a tile is exactly the index order
// Only the key bits the caller declares significant are sorted (`bits`): voxel keys are compacted to the bounding box of the
// scan first (~20 bits instead of 63), so a 131 072-point scan needs 3 passes, not 8.
#include "internal.hpp"

using namespace glim_amd;

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_ROUNDS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ROUNDS;

__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const unsigned long long* __restrict__ keys, int n, int shift, int mask, int* __restrict__ hist,
                                                             int B) {
  __shared__ int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int base = blockIdx.x * RS_TILE;
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; r++) {
    const int i = base + r * RS_THREADS + (int)threadIdx.x;
    if (i < n) atomicAdd(&h[(int)(keys[i] >> shift) & mask], 1);
  }
  __syncthreads();
  hist[threadIdx.x * B + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of E = 256 * B ints in place by one block of 1024 threads (thread t owns a run of consecutive entries)
__global__ __launch_bounds__(1024) void rs_offsets_kernel(int* __restrict__ hist, int E) {
  __shared__ int s_wave[16];
  const int per = (E + 1023) / 1024;
  const int begin = (int)threadIdx.x * per, end = min(E, begin + per);
  int sum = 0;
  for (int i = begin; i < end; i++) sum += hist[i];
  // block exclusive scan of the 1024 thread sums
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  int wave_off = 0;
  for (int w = 0; w < wave; w++) wave_off += s_wave[w];
  int run = wave_off + inc - sum;
  for (int i = begin; i < end; i++) {
    const int v = hist[i];
    hist[i] = run;
    run += v;
  }
}

__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const unsigned long long* __restrict__ keys_in, const unsigned int* __restrict__ vals_in,
                                                                unsigned long long* __restrict__ keys_out, unsigned int* __restrict__ vals_out, int n,
                                                                int shift, int mask, const int* __restrict__ offsets, int B) {
  __shared__ int wave_cnt[RS_THREADS / 64][256];
  __shared__ int running[256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  running[tid] = offsets[tid * B + blockIdx.x];
#pragma unroll
  for (int w = 0; w < RS_THREADS / 64; w++) wave_cnt[w][tid] = 0;
  __syncthreads();
  const int base = blockIdx.x * RS_TILE;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  for (int r = 0; r < RS_ROUNDS; r++) {
    const int i = base + r * RS_THREADS + tid;
    const bool valid = i < n;
    const unsigned long long key = valid ? keys_in[i] : 0ull;
    const unsigned int val = valid ? (vals_in ? vals_in[i] : (unsigned int)i) : 0u;
    const int d = (int)(key >> shift) & mask;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const bool bit = (d >> b) & 1;
      const unsigned long long bal = __ballot(bit);
      peers &= bit ? bal : ~bal;
    }
    const int rank_in_wave = __popcll(peers & lt_mask);
    if (valid && rank_in_wave == 0) wave_cnt[wave][d] = __popcll(peers);
    __syncthreads();
    if (valid) {
      int off = running[d];
      for (int w = 0; w < wave; w++) off += wave_cnt[w][d];
      const int dst = off + rank_in_wave;
      keys_out[dst] = key;
      vals_out[dst] = val;
    }
    __syncthreads();
    int add = 0;
#pragma unroll
    for (int w = 0; w < RS_THREADS / 64; w++) {
      add += wave_cnt[w][tid];
      wave_cnt[w][tid] = 0;
    }
    running[tid] += add;
    __syncthreads();
  }
}

}  // namespace

namespace glim_amd {

size_t radix_sort_scratch_bytes(int n) { return (size_t)256 * (size_t)((n + RS_TILE - 1) / RS_TILE + 1) * sizeof(int); }

// Sorts n pairs by the low `bits` bits of the key, stable.  Ping-pongs between (keys_a, vals_a) and (keys_b, vals_b); the
// sorted pairs end up in (*keys_sorted, *vals_sorted), which is one of the two.  vals_a_is_iota: the values are 0..n-1 and
// vals_a does not need to be initialised.  Enqueues on `st`; no synchronisation.
hipError_t radix_sort_pairs(hipStream_t st, int n, int bits, unsigned long long* keys_a, unsigned int* vals_a, unsigned long long* keys_b,
                            unsigned int* vals_b, bool vals_a_is_iota, int* scratch, unsigned long long** keys_sorted, unsigned int** vals_sorted) {
  unsigned long long *kin = keys_a, *kout = keys_b;
  unsigned int *vin = vals_a, *vout = vals_b;
  if (n > 0) {
    const int B = (n + RS_TILE - 1) / RS_TILE;
    const int passes = bits <= 0 ? 1 : (bits + 7) / 8;  // bits == 0: one identity pass (materialises iota values)
    for (int p = 0; p < passes; p++) {
      const int shift = 8 * p;
      const int rem = bits - shift;
      const int mask = rem >= 8 ? 255 : (rem <= 0 ? 0 : (1 << rem) - 1);  // key bits above `bits` are ignored
      rs_hist_kernel<<<B, RS_THREADS, 0, st>>>(kin, n, shift, mask, scratch, B);
      rs_offsets_kernel<<<1, 1024, 0, st>>>(scratch, 256 * B);
      rs_scatter_kernel<<<B, RS_THREADS, 0, st>>>(kin, (p == 0 && vals_a_is_iota) ? nullptr : vin, kout, vout, n, shift, mask, scratch, B);
      std::swap(kin, kout);
      std::swap(vin, vout);
    }
  }
  *keys_sorted = kin;
  *vals_sorted = vin;
  return hipGetLastError();
}

}  // namespace glim_amd

extern "C" {

// parity / debug only: sort host pairs on the device
int glim_amd_debug_sort_pairs(glim_amd_ctx* ctx, int64_t n, int32_t bits, const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out,
                              uint32_t* vals_out) {
  if (!ctx || n < 0 || n > (int64_t)(1 << 28) || bits < 0 || bits > 64 || (n > 0 && (!keys_in || !keys_out || !vals_out))) return GLIM_AMD_ERR_INVALID;
  if (n == 0) return GLIM_AMD_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream();
  unsigned long long *ka = nullptr, *kb = nullptr, *ks = nullptr;
  unsigned int *va = nullptr, *vb = nullptr, *vs = nullptr;
  int* scratch = nullptr;
  hipError_t e = pool_malloc(&ka, (size_t)n * 8);
  if (e == hipSuccess) e = pool_malloc(&kb, (size_t)n * 8);
  if (e == hipSuccess) e = pool_malloc(&va, (size_t)n * 4);
  if (e == hipSuccess) e = pool_malloc(&vb, (size_t)n * 4);
  if (e == hipSuccess) e = pool_malloc(&scratch, radix_sort_scratch_bytes((int)n));
  if (e == hipSuccess) e = hipMemcpyAsync(ka, keys_in, (size_t)n * 8, hipMemcpyHostToDevice, st);
  if (e == hipSuccess && vals_in) e = hipMemcpyAsync(va, vals_in, (size_t)n * 4, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = radix_sort_pairs(st, (int)n, bits, ka, va, kb, vb, vals_in == nullptr, scratch, &ks, &vs);
  if (e == hipSuccess) e = hipMemcpyAsync(keys_out, ks, (size_t)n * 8, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(vals_out, vs, (size_t)n * 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  for (void* p : {(void*)ka, (void*)kb, (void*)va, (void*)vb, (void*)scratch})
    if (p) (void)pool_free(p);
  if (e != hipSuccess) {
    set_hip_error(e, "debug_sort_pairs");
    return GLIM_AMD_ERR_HIP;
  }
  return GLIM_AMD_OK;
}

}  // extern "C"
