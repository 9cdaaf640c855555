// sort.hip -- stable LSD radix sort of (u64 key, u32 value) pairs on the device, hand-written for gfx950 (no rocPRIM / hipCUB).
//
// Used by the scan preprocessing (preprocess.hip): sorting points by voxel key (gtsam_points voxelgrid_sampling /
// randomgrid_sampling sort their (coord, index) pairs, cloud_preprocessor.cpp:104-109) and by time stamp
// (std::sort, cloud_preprocessor.cpp:135).  Stability is what makes the device result reproduce the oracle's
// (key, original index) order exactly.
//
// One pass = 8 key bits, two launches (three for very large inputs):
//   rs_hist     every block owns a tile of 256 x ROUNDS consecutive pairs and counts its 256 digit values in LDS;
//               hist[digit][block] (digit-major) goes to global memory
//   rs_scatter  thread d of every block first derives the block's global offset for digit d from row d of that table (digit-major
//               order == the output order of a stable sort: all pairs with a smaller digit first, then the same digit in
//               earlier blocks) -- the table has <= 256 columns and sits in L2, so re-deriving it per block is cheaper than
//               a separate single-block scan launch; above 256 blocks a separate rs_offsets launch does it once.
//               Every wavefront of the block then ranks its own chunk of the tile (ROUNDS rounds of 64 pairs): rank among
//               the equal digits =  (equal digits in earlier wavefronts)  +  (equal digits in this wavefront's earlier
//               rounds)  +  (equal digits in lower lanes of this round); the last term comes from 8 ballots (one per digit
//               bit), so the order inside a tile is exactly the index order.  Two block barriers per pass.
// ROUNDS (1, 2, 4, 8) is the smallest tile that keeps the grid at <= 128 blocks (GLIM_AMD_RS_BLOCKS): small inputs still spread
// over the chip, while the per-block re-derivation of the offsets (the whole table is read by every block) stays cheap.
// Only the key bits the caller declares significant are sorted (`bits`): voxel keys are compacted to the bounding box of the
// scan first (~20 bits instead of 63), so a 131 072-point scan needs 3 passes, not 8.
#include <algorithm>
#include <cstdlib>

#include "internal.hpp"

using namespace glim_amd;

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_FUSED_MAX_BLOCKS = 256;

template <int ROUNDS>
__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const unsigned long long* __restrict__ keys, int n, int shift, int mask, int* __restrict__ hist,
                                                             int stride) {
  __shared__ int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int base = blockIdx.x * (RS_THREADS * ROUNDS);
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const int i = base + r * RS_THREADS + (int)threadIdx.x;
    if (i < n) atomicAdd(&h[(int)(keys[i] >> shift) & mask], 1);
  }
  __syncthreads();
  hist[threadIdx.x * stride + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of E = 256 * stride ints in place by one block of 1024 threads (thread t owns a run of consecutive entries);
// only used when the grid has more than RS_FUSED_MAX_BLOCKS blocks
__global__ __launch_bounds__(1024) void rs_offsets_kernel(int* __restrict__ hist, int E) {
  __shared__ int s_wave[16];
  const int per = (E + 1023) / 1024;
  const int begin = (int)threadIdx.x * per, end = min(E, begin + per);
  int sum = 0;
  for (int i = begin; i < end; i++) sum += hist[i];
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

// Tile layout: wavefront w of a block owns the ROUNDS * 64 consecutive pairs [w * ROUNDS * 64, (w + 1) * ROUNDS * 64) of the tile and
// walks them in ROUNDS rounds of 64, so the stable order inside a tile is (wavefront, round, lane) == the index order.  Phase 1 needs
// no block barrier: a wavefront ranks its own pairs against its own LDS counters (lock-step execution orders the counter read of
// round r + 1 after the leader's update of round r).  Two barriers per pass in total.
template <int ROUNDS, bool FUSED>
__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const unsigned long long* __restrict__ keys_in, const unsigned int* __restrict__ vals_in,
                                                                unsigned long long* __restrict__ keys_out, unsigned int* __restrict__ vals_out, int n,
                                                                int shift, int mask, const int* __restrict__ table, int stride) {
  constexpr int WAVES = RS_THREADS / 64;
  __shared__ int wave_cnt[WAVES][256];   // phase 1: digit counts of each wavefront's chunk; phase 2: its base offset per digit
  __shared__ int s_wave[WAVES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int w = 0; w < WAVES; w++) wave_cnt[w][tid] = 0;
  __syncthreads();

  // ---- phase 1: load, digit, rank inside the wavefront's chunk ----
  const int base = blockIdx.x * (RS_THREADS * ROUNDS) + wave * (ROUNDS * 64) + lane;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  unsigned long long key[ROUNDS];
  unsigned int val[ROUNDS];
  int rank[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const int i = base + r * 64;
    const bool valid = i < n;
    key[r] = valid ? keys_in[i] : 0ull;
    val[r] = valid ? (vals_in ? vals_in[i] : (unsigned int)i) : 0u;
  }
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const bool valid = base + r * 64 < n;
    const int d = (int)(key[r] >> shift) & mask;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const bool bit = (d >> b) & 1;
      const unsigned long long bal = __ballot(bit);
      peers &= bit ? bal : ~bal;
    }
    const int below = __popcll(peers & lt_mask);
    const int prior = wave_cnt[wave][d];  // equal digits in this wavefront's earlier rounds
    __builtin_amdgcn_wave_barrier();
    if (valid && below == 0) wave_cnt[wave][d] = prior + __popcll(peers);
    __builtin_amdgcn_wave_barrier();
    rank[r] = prior + below;
  }
  __syncthreads();

  // ---- phase 2: thread d turns the per-wavefront counts of digit d into base offsets ----
  int offset;
  if (FUSED) {
    // row `tid` of the histogram table (stride is a multiple of 4, padding columns are zero): digit total and the part in earlier blocks
    const int4* row = reinterpret_cast<const int4*>(table + (size_t)tid * stride);
    int total = 0, before = 0;
    const int my4 = (int)blockIdx.x >> 2, my_r = (int)blockIdx.x & 3;
    for (int c = 0; c < stride / 4; c++) {
      const int4 v = row[c];
      const int s4 = (v.x + v.y) + (v.z + v.w);
      total += s4;
      if (c < my4) before += s4;
      else if (c == my4) before += (my_r > 0 ? v.x : 0) + (my_r > 1 ? v.y : 0) + (my_r > 2 ? v.z : 0);
    }
    int inc = total;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int wave_off = 0;
    for (int w = 0; w < wave; w++) wave_off += s_wave[w];
    offset = wave_off + inc - total + before;
  } else {
    offset = table[(size_t)tid * stride + blockIdx.x];
  }
#pragma unroll
  for (int w = 0; w < WAVES; w++) {
    const int c = wave_cnt[w][tid];
    wave_cnt[w][tid] = offset;
    offset += c;
  }
  __syncthreads();

  // ---- phase 3: scatter ----
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    if (base + r * 64 < n) {
      const int d = (int)(key[r] >> shift) & mask;
      const int dst = wave_cnt[wave][d] + rank[r];
      keys_out[dst] = key[r];
      vals_out[dst] = val[r];
    }
  }
}

template <int ROUNDS>
void launch_pass(hipStream_t st, int B, int stride, const unsigned long long* kin, const unsigned int* vin, unsigned long long* kout, unsigned int* vout, int n,
                 int shift, int mask, int* table) {
  rs_hist_kernel<ROUNDS><<<B, RS_THREADS, 0, st>>>(kin, n, shift, mask, table, stride);
  if (B <= RS_FUSED_MAX_BLOCKS) {
    rs_scatter_kernel<ROUNDS, true><<<B, RS_THREADS, 0, st>>>(kin, vin, kout, vout, n, shift, mask, table, stride);
  } else {
    rs_offsets_kernel<<<1, 1024, 0, st>>>(table, 256 * stride);
    rs_scatter_kernel<ROUNDS, false><<<B, RS_THREADS, 0, st>>>(kin, vin, kout, vout, n, shift, mask, table, stride);
  }
}

// largest grid the tile choice aims for: measured on MI355X, 131 072 pairs: 64 blocks 16.5 us / pass, 128: 14.8, 256: 18.9 (every block
// reads the whole table)
inline int block_target() { return std::min(128, RS_FUSED_MAX_BLOCKS); }
inline int pick_rounds(int n) {
  for (int r = 1; r < 8; r <<= 1)
    if ((n + RS_THREADS * r - 1) / (RS_THREADS * r) <= block_target()) return r;
  return 8;
}
inline int table_stride(int n) {
  const int r = pick_rounds(n);
  const int B = (n + RS_THREADS * r - 1) / (RS_THREADS * r);
  // fused offsets read the rows as int4 (padding columns stay zero); the separate rs_offsets launch scans the table in place, so it
  // must not have padding columns (they would carry prefix values into the next pass)
  return B <= RS_FUSED_MAX_BLOCKS ? ((B + 3) & ~3) : B;
}

}  // namespace

namespace glim_amd {

// (sized so that the same scratch also serves any smaller sort: a smaller input may pick a smaller tile and up to 256 blocks)
size_t radix_sort_scratch_bytes(int n) { return (size_t)256 * (size_t)std::max(table_stride(n > 0 ? n : 1), RS_FUSED_MAX_BLOCKS) * sizeof(int); }

// Sorts n pairs by the low `bits` bits of the key, stable.  Ping-pongs between (keys_a, vals_a) and (keys_b, vals_b); the
// sorted pairs end up in (*keys_sorted, *vals_sorted), which is one of the two.  vals_a_is_iota: the values are 0..n-1 and
// vals_a does not need to be initialised.  `scratch`: radix_sort_scratch_bytes(n' >= n).  Enqueues on `st`; no synchronisation.
hipError_t radix_sort_pairs(hipStream_t st, int n, int bits, unsigned long long* keys_a, unsigned int* vals_a, unsigned long long* keys_b,
                            unsigned int* vals_b, bool vals_a_is_iota, int* scratch, unsigned long long** keys_sorted, unsigned int** vals_sorted) {
  unsigned long long *kin = keys_a, *kout = keys_b;
  unsigned int *vin = vals_a, *vout = vals_b;
  if (n > 0) {
    const int rounds = pick_rounds(n);
    const int B = (n + RS_THREADS * rounds - 1) / (RS_THREADS * rounds);
    const int stride = table_stride(n);
    if (stride != B) {  // the padding columns of the table are read by the fused offset computation: keep them zero
      hipError_t e = hipMemsetAsync(scratch, 0, (size_t)256 * stride * sizeof(int), st);
      if (e != hipSuccess) return e;
    }
    const int passes = bits <= 0 ? 1 : (bits + 7) / 8;  // bits == 0: one identity pass (materialises iota values)
    for (int p = 0; p < passes; p++) {
      const int shift = 8 * p;
      const int rem = bits - shift;
      const int mask = rem >= 8 ? 255 : (rem <= 0 ? 0 : (1 << rem) - 1);  // key bits above `bits` are ignored
      const unsigned int* v = (p == 0 && vals_a_is_iota) ? nullptr : vin;
      switch (rounds) {
        case 1: launch_pass<1>(st, B, stride, kin, v, kout, vout, n, shift, mask, scratch); break;
        case 2: launch_pass<2>(st, B, stride, kin, v, kout, vout, n, shift, mask, scratch); break;
        case 4: launch_pass<4>(st, B, stride, kin, v, kout, vout, n, shift, mask, scratch); break;
        default: launch_pass<8>(st, B, stride, kin, v, kout, vout, n, shift, mask, scratch); break;
      }
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
