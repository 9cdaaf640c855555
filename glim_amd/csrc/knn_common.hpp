// knn_common.hpp -- shared by the four translation units of kernel group K2 (knn.hip: host side, grid and exhaustive kernels, curve order;
// knn_qgroup.hip: the query-group kernel, the default; knn_chunks.hip: the 64-query chunk kernel; knn_pairs.hip: the pair-lane chunk kernel).
// The chunk kernels are the slow ones to compile (every list size is its own instantiation), so they build in parallel with the rest.
#pragma once
#include <cstdint>

#include "device_math.hpp"
#include "internal.hpp"
#include "scope_sync.hpp"

namespace glim_amd {
// 64-query chunk kernel over the Hilbert-ordered points (`sorted`, C chunks of 64 with boxes `box`, which also has room for the boxes of the
// groups of 64 chunks: this call fills them first).  dbg: optional per-wavefront counters.  select: see knn_chunks.hip.
// guard: one device int, non-zero = do nothing (the caller found the cloud's extent unusable for the FP32 mask pass)
void knn_launch_chunks(hipStream_t st, int n, int C, const float4* sorted, float* box, int k, int32_t* out, int* dbg, bool select, const int* guard);
// boxes of the groups of 64 chunks, behind the chunk boxes (knn_launch_chunks does it itself)
void knn_launch_group_boxes(hipStream_t st, int C, float* box);
// query-group kernel (knn_qgroup.hip): lanes are candidates, a wavefront answers `queries_per_wave` (2 or 4) consecutive queries; fills the group boxes first
// dbg: optional 5 zeroed ints (wavefronts, chunk scans, exact query-chunk evaluations, insertions, chunk-test rounds)
void knn_launch_qgroup(hipStream_t st, int n, int C, const float4* sorted, float* box, int k, int32_t* out, const int* guard, int queries_per_wave, int* dbg);
constexpr int KNN_KERNEL_QGROUP = 3;  // diag knn_kernel=qgroup (internal.hpp has auto / wave64 / pair)
// pair-lane kernel over 32-point half chunks (C32 of them, boxes `box32`); k <= 16
void knn_launch_pairs(hipStream_t st, int n, int C32, const float4* sorted, const float* box32, int k, int32_t* out, bool select, const int* guard);
}  // namespace glim_amd

namespace {

using namespace glim_amd;

constexpr int CHUNK = 64;  // points per chunk of the curve order (one wavefront's queries)
constexpr int QCH = 32;    // the pair-lane kernel's half chunks

typedef float v2f __attribute__((ext_vector_type(2)));

template <int K>
struct TopK {
  double d[K];
  int idx[K];
  __device__ __forceinline__ void init(int self) {
#pragma unroll
    for (int j = 0; j < K; j++) {
      d[j] = __longlong_as_double(0x7ff0000000000000ll);  // +inf
      idx[j] = self;
    }
  }
  __device__ __forceinline__ void push(double dn, int in) {
    if (dn < d[K - 1] || (dn == d[K - 1] && in < idx[K - 1])) {
      d[K - 1] = dn;
      idx[K - 1] = in;
#pragma unroll
      for (int j = K - 1; j > 0; j--) {
        const bool better = d[j] < d[j - 1] || (d[j] == d[j - 1] && idx[j] < idx[j - 1]);
        if (!__any(better)) break;  // no lane's new entry moves further up: late candidates settle after a step or two
        const double td = better ? d[j - 1] : d[j];
        const int ti = better ? idx[j - 1] : idx[j];
        d[j - 1] = better ? d[j] : d[j - 1];
        idx[j - 1] = better ? idx[j] : idx[j - 1];
        d[j] = td;
        idx[j] = ti;
      }
    }
  }
};

__device__ __forceinline__ double sqdist(double qx, double qy, double qz, double x, double y, double z) {
  const double dx = qx - x, dy = qy - y, dz = qz - z;
  return dadd(dadd(dmul(dx, dx), dmul(dy, dy)), dmul(dz, dz));
}


// k <= 16 only (the pair-lane kernel keeps two top-k lists per query in registers: beyond 16 entries it spills)
#define DISPATCH_K16(FN, ...)                    \
  do {                                           \
    if (k <= 8) FN<8>(__VA_ARGS__);              \
    else if (k <= 10) FN<10>(__VA_ARGS__);       \
    else FN<16>(__VA_ARGS__);                    \
  } while (0)
#define DISPATCH_K(FN, ...)                      \
  do {                                           \
    if (k <= 8) FN<8>(__VA_ARGS__);              \
    else if (k <= 10) FN<10>(__VA_ARGS__);       \
    else if (k <= 16) FN<16>(__VA_ARGS__);       \
    else if (k <= 24) FN<24>(__VA_ARGS__);       \
    else FN<32>(__VA_ARGS__);                    \
  } while (0)

}  // namespace
