// knn.hip -- kernel K2: exact k-nearest-neighbour search on the device.
// Replaces CloudPreprocessor::find_neighbors (src/glim/preprocess/cloud_preprocessor.cpp:190-221, a nanoflann kd-tree):
// for every point the k nearest points among ALL points including itself (buffers pre-filled with i, :197), ascending
// squared distance; ties are ordered by ascending index (the oracle's rule; the reference leaves exact ties to the kd-tree).
//
// Distances are evaluated in FP64 as (dx*dx + dy*dy) + dz*dz on the FP32-representable inputs -- the same expression, in
// the same order and without fma contraction, as oracle/vgicp_oracle.c:sqdist3 -- so neighbour SETS are bit-identical.
//
// Round-1 implementation: LDS-tiled exhaustive scan (every query against every point), one query per lane, tile of 1024
// candidates staged as FP64 SoA in LDS and read as wave-uniform broadcasts.  O(N^2) but exact and branch-light:
// 131 072 points = 1.7e10 pair tests.  A grid-hashed search for the 300k-point stream (BASELINE config 5) is the next step.
#include "internal.hpp"

using namespace glim_amd;

namespace {

constexpr int TILE = 1024;

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

template <int K>
__global__ __launch_bounds__(256) void knn_bruteforce_kernel(int n, const float4* __restrict__ pts, int k, int32_t* __restrict__ out) {
  __shared__ double s_x[TILE], s_y[TILE], s_z[TILE];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  double qx = 0, qy = 0, qz = 0;
  if (live) {
    const float4 p = pts[i];
    qx = p.x; qy = p.y; qz = p.z;
  }
  TopK<K> best;
  best.init(i);
  int found = 0;
  for (int t0 = 0; t0 < n; t0 += TILE) {
    const int tn = min(TILE, n - t0);
    __syncthreads();
    for (int j = threadIdx.x; j < tn; j += blockDim.x) {
      const float4 p = pts[t0 + j];
      s_x[j] = p.x; s_y[j] = p.y; s_z[j] = p.z;
    }
    __syncthreads();
    if (live) {
      for (int j = 0; j < tn; j++) {
        const double dx = qx - s_x[j], dy = qy - s_y[j], dz = qz - s_z[j];
        const double d2 = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
        best.push(d2, t0 + j);
      }
      found += tn;
    }
  }
  if (live) {
    const int valid = min(found, K);
#pragma unroll
    for (int j = 0; j < K; j++)
      if (j < k) out[(size_t)i * k + j] = (j < valid) ? best.idx[j] : i;
  }
}

template <int K>
void launch_knn(hipStream_t st, int n, const float4* pts, int k, int32_t* out) {
  knn_bruteforce_kernel<K><<<(n + 255) / 256, 256, 0, st>>>(n, pts, k, out);
}

}  // namespace

extern "C" {

int glim_amd_cloud_find_neighbors(glim_amd_cloud* c, int k, int32_t* neighbors_out) {
  if (!c || k <= 0) return GLIM_AMD_ERR_INVALID;
  if (k > 32) return GLIM_AMD_ERR_UNSUPPORTED;
  glim_amd_ctx* ctx = c->ctx;
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  if (c->neighbors) {
    (void)hipFree(c->neighbors);
    c->neighbors = nullptr;
  }
  c->k = k;
  GA_HIP(hipMalloc(&c->neighbors, (size_t)(c->n > 0 ? c->n : 1) * k * sizeof(int32_t)));
  if (c->n == 0) return GLIM_AMD_OK;
  const int n = (int)c->n;
  hipStream_t st = ctx->stream();
  if (k <= 8) launch_knn<8>(st, n, c->pts, k, c->neighbors);
  else if (k <= 10) launch_knn<10>(st, n, c->pts, k, c->neighbors);
  else if (k <= 16) launch_knn<16>(st, n, c->pts, k, c->neighbors);
  else if (k <= 24) launch_knn<24>(st, n, c->pts, k, c->neighbors);
  else launch_knn<32>(st, n, c->pts, k, c->neighbors);
  GA_HIP(hipGetLastError());
  if (neighbors_out) GA_HIP(hipMemcpyAsync(neighbors_out, c->neighbors, (size_t)n * k * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  GA_HIP(hipStreamSynchronize(st));
  return GLIM_AMD_OK;
}

}  // extern "C"
