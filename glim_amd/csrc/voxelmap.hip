// voxelmap.hip -- GaussianVoxelMapGPU equivalent (kernel group K3): lossless open-addressing voxel hash built on the device.
// Replaces gtsam_points::GaussianVoxelMapGPU(resolution, ...)::insert(frame) as called at
// src/glim/odometry/odometry_estimation_gpu.cpp:103-104, src/glim/mapping/sub_mapping.cpp:398-399 and
// src/glim/mapping/global_mapping.cpp:265-266,747-748.  Statistic per voxel = mean of member means and mean of member
// covariances, voxel identity = integer coordinate fast_floor(p * (1/resolution)).
//
// Build (all on the context stream):
//   1. insert_keys   : every point CASes its packed 64-bit coordinate key into an over-sized scratch table (2N slots,
//                      never full) and the distinct keys are counted                        -> V
//   2. move_keys     : the V distinct keys are re-inserted into the final table of 2^ceil(log2(2V)) 64-byte slots
//   3. accumulate    : every point adds its mean / covariance as 64-bit FIXED-POINT integers with atomics -- integer
//                      addition is associative, so the sums (and the map) are bit-reproducible whatever the atomic order
//   4. finalise      : sums / count in FP64, stored as FP32 in the slot (key + statistic share one cache line)
#include <vector>

#include "device_math.hpp"
#include "internal.hpp"

using namespace glim_amd;

namespace {

constexpr double MEAN_SCALE = 268435456.0;      // 2^28  (3.7e-9 m resolution, |sum| < 3.4e10 m)
constexpr double COV_SCALE = 68719476736.0;     // 2^36  (1.5e-11 resolution, |sum| < 1.3e8)
constexpr int ACC_STRIDE = 10;                  // 3 mean + 6 cov + count

__global__ __launch_bounds__(256) void fill_u64_kernel(unsigned long long* __restrict__ p, size_t n, unsigned long long v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ __launch_bounds__(256) void init_slots_kernel(VoxelSlot* __restrict__ slots, unsigned int n) {
  const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  VoxelSlot s;
  s.key = EMPTY_KEY;
  s.mx = s.my = s.mz = 0.f;
  s.c00 = s.c01 = s.c02 = s.c11 = s.c12 = s.c22 = 0.f;
  s.count = 0;
  s.pad[0] = s.pad[1] = s.pad[2] = s.pad[3] = 0;
  slots[i] = s;
}

// stats[0] = distinct keys, stats[1] = points whose coordinate does not fit the 21-bit key range
__global__ __launch_bounds__(256) void insert_keys_kernel(int n, const float4* __restrict__ pts, double inv_res,
                                                          unsigned long long* __restrict__ tkeys, unsigned int tmask,
                                                          unsigned long long* __restrict__ pkeys, int* __restrict__ stats) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  const unsigned long long key = voxel_key((double)p.x, (double)p.y, (double)p.z, inv_res);
  pkeys[i] = key;
  if (key == EMPTY_KEY) {
    atomicAdd(&stats[1], 1);
    return;
  }
  unsigned int s = hash_key(key) & tmask;
  for (;;) {
    const unsigned long long prev = atomicCAS(&tkeys[s], EMPTY_KEY, key);
    if (prev == EMPTY_KEY) {
      atomicAdd(&stats[0], 1);
      return;
    }
    if (prev == key) return;
    s = (s + 1) & tmask;
  }
}

__global__ __launch_bounds__(256) void move_keys_kernel(const unsigned long long* __restrict__ tkeys, unsigned int tsize,
                                                        VoxelSlot* __restrict__ slots, unsigned int mask) {
  const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= tsize) return;
  const unsigned long long key = tkeys[i];
  if (key == EMPTY_KEY) return;
  unsigned int s = hash_key(key) & mask;
  for (;;) {
    const unsigned long long prev = atomicCAS(&slots[s].key, EMPTY_KEY, key);
    if (prev == EMPTY_KEY) return;  // keys in the scratch table are distinct: no equal-key case
    s = (s + 1) & mask;
  }
}

__device__ __forceinline__ void atomic_add_fixed(long long* p, double v, double scale) {
  const long long q = __double2ll_rn(v * scale);
  atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)q);
}

__global__ __launch_bounds__(256) void accumulate_kernel(int n, const float4* __restrict__ pts, const float4* __restrict__ covA,
                                                         const float2* __restrict__ covB, const unsigned long long* __restrict__ pkeys,
                                                         const VoxelSlot* __restrict__ slots, unsigned int mask, long long* __restrict__ acc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = find_slot(slots, mask, pkeys[i]);
  if (s < 0) return;
  const float4 p = pts[i];
  const float4 a = covA[i];
  const float2 b = covB[i];
  long long* dst = acc + (size_t)s * ACC_STRIDE;
  atomic_add_fixed(dst + 0, (double)p.x, MEAN_SCALE);
  atomic_add_fixed(dst + 1, (double)p.y, MEAN_SCALE);
  atomic_add_fixed(dst + 2, (double)p.z, MEAN_SCALE);
  atomic_add_fixed(dst + 3, (double)a.x, COV_SCALE);
  atomic_add_fixed(dst + 4, (double)a.y, COV_SCALE);
  atomic_add_fixed(dst + 5, (double)a.z, COV_SCALE);
  atomic_add_fixed(dst + 6, (double)a.w, COV_SCALE);
  atomic_add_fixed(dst + 7, (double)b.x, COV_SCALE);
  atomic_add_fixed(dst + 8, (double)b.y, COV_SCALE);
  atomicAdd(reinterpret_cast<unsigned long long*>(dst + 9), 1ull);
}

__global__ __launch_bounds__(256) void finalize_kernel(VoxelSlot* __restrict__ slots, unsigned int tsize, const long long* __restrict__ acc) {
  const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= tsize) return;
  if (slots[i].key == EMPTY_KEY) return;
  const long long* a = acc + (size_t)i * ACC_STRIDE;
  const long long cnt = a[9];
  const double inv_n = 1.0 / (double)cnt;
  const double im = inv_n / MEAN_SCALE, ic = inv_n / COV_SCALE;
  VoxelSlot s = slots[i];
  s.mx = (float)((double)a[0] * im);
  s.my = (float)((double)a[1] * im);
  s.mz = (float)((double)a[2] * im);
  s.c00 = (float)((double)a[3] * ic);
  s.c01 = (float)((double)a[4] * ic);
  s.c02 = (float)((double)a[5] * ic);
  s.c11 = (float)((double)a[6] * ic);
  s.c12 = (float)((double)a[7] * ic);
  s.c22 = (float)((double)a[8] * ic);
  s.count = (int)cnt;
  slots[i] = s;
}

unsigned int next_pow2(unsigned long long v) {
  unsigned long long p = 1;
  while (p < v) p <<= 1;
  return (unsigned int)p;
}

struct DeviceTemp {
  void* p = nullptr;
  ~DeviceTemp() {
    if (p) (void)hipFree(p);
  }
};

}  // namespace

extern "C" {

int glim_amd_voxelmap_create(glim_amd_ctx* ctx, double resolution, int /*init_num_buckets*/, int /*max_bucket_scan_count*/,
                             double /*target_points_drop_rate*/, glim_amd_voxelmap** out) {
  if (!ctx || !out || !(resolution > 0.0)) return GLIM_AMD_ERR_INVALID;
  glim_amd_voxelmap* m = new glim_amd_voxelmap();
  m->ctx = ctx;
  m->resolution = resolution;
  m->inv_resolution = 1.0 / resolution;  // same FP64 expression as the oracle (orc_voxelmap_create)
  *out = m;
  return GLIM_AMD_OK;
}

int glim_amd_voxelmap_destroy(glim_amd_voxelmap* m) {
  if (!m) return GLIM_AMD_OK;
  if (m->ctx) (void)hipSetDevice(m->ctx->device);
  if (m->slots) (void)hipFree(m->slots);
  delete m;
  return GLIM_AMD_OK;
}

int glim_amd_voxelmap_insert(glim_amd_voxelmap* m, const glim_amd_cloud* cloud) {
  if (!m || !cloud || cloud->ctx != m->ctx) return GLIM_AMD_ERR_INVALID;
  if (!cloud->has_covs) return GLIM_AMD_ERR_STATE;
  if (m->slots) return GLIM_AMD_ERR_UNSUPPORTED;  // GLIM's GPU path builds each map with a single insert()
  if (cloud->n > (int64_t)(1u << 30)) return GLIM_AMD_ERR_INVALID;
  glim_amd_ctx* ctx = m->ctx;
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream();
  const int n = (int)cloud->n;

  DeviceTemp tkeys, pkeys, stats, acc;
  const unsigned int tsize0 = next_pow2((unsigned long long)(n > 32 ? n : 32) * 2);
  GA_HIP(hipMalloc(&tkeys.p, (size_t)tsize0 * sizeof(unsigned long long)));
  GA_HIP(hipMalloc(&pkeys.p, (size_t)(n > 0 ? n : 1) * sizeof(unsigned long long)));
  GA_HIP(hipMalloc(&stats.p, 2 * sizeof(int)));
  GA_HIP(hipMemsetAsync(stats.p, 0, 2 * sizeof(int), st));
  fill_u64_kernel<<<1024, 256, 0, st>>>((unsigned long long*)tkeys.p, tsize0, EMPTY_KEY);
  GA_HIP(hipGetLastError());
  if (n > 0) {
    insert_keys_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, cloud->pts, m->inv_resolution, (unsigned long long*)tkeys.p, tsize0 - 1,
                                                         (unsigned long long*)pkeys.p, (int*)stats.p);
    GA_HIP(hipGetLastError());
  }
  int h_stats[2] = {0, 0};
  GA_HIP(hipMemcpyAsync(h_stats, stats.p, sizeof(h_stats), hipMemcpyDeviceToHost, st));
  GA_HIP(hipStreamSynchronize(st));
  if (h_stats[1] != 0) return GLIM_AMD_ERR_RANGE;

  const int num_voxels = h_stats[0];
  const unsigned int tsize = next_pow2((unsigned long long)(num_voxels > 32 ? num_voxels : 32) * 2);
  VoxelSlot* slots = nullptr;
  GA_HIP(hipMalloc(&slots, (size_t)tsize * sizeof(VoxelSlot)));
  hipError_t e = hipMalloc(&acc.p, (size_t)tsize * ACC_STRIDE * sizeof(long long));
  if (e == hipSuccess) e = hipMemsetAsync(acc.p, 0, (size_t)tsize * ACC_STRIDE * sizeof(long long), st);
  if (e == hipSuccess) {
    init_slots_kernel<<<(tsize + 255) / 256, 256, 0, st>>>(slots, tsize);
    move_keys_kernel<<<(tsize0 + 255) / 256, 256, 0, st>>>((const unsigned long long*)tkeys.p, tsize0, slots, tsize - 1);
    if (n > 0)
      accumulate_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, cloud->pts, cloud->covA, cloud->covB, (const unsigned long long*)pkeys.p, slots,
                                                          tsize - 1, (long long*)acc.p);
    finalize_kernel<<<(tsize + 255) / 256, 256, 0, st>>>(slots, tsize, (const long long*)acc.p);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    set_hip_error(e, "voxelmap_insert");
    (void)hipFree(slots);
    return GLIM_AMD_ERR_HIP;
  }
  m->slots = slots;
  m->table_size = tsize;
  m->num_voxels = num_voxels;
  return GLIM_AMD_OK;
}

int glim_amd_voxelmap_info(const glim_amd_voxelmap* m, int32_t* num_voxels, int32_t* num_buckets, double* resolution, size_t* bytes) {
  if (!m) return GLIM_AMD_ERR_INVALID;
  if (num_voxels) *num_voxels = m->num_voxels;
  if (num_buckets) *num_buckets = (int32_t)m->table_size;
  if (resolution) *resolution = m->resolution;
  if (bytes) *bytes = (size_t)m->table_size * sizeof(VoxelSlot);
  return GLIM_AMD_OK;
}

int glim_amd_voxelmap_download(const glim_amd_voxelmap* m, int32_t* coords, int32_t* counts, float* means, float* cov33) {
  if (!m) return GLIM_AMD_ERR_INVALID;
  if (!m->slots) return GLIM_AMD_ERR_STATE;
  glim_amd_ctx* ctx = m->ctx;
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  std::vector<VoxelSlot> host(m->table_size);
  GA_HIP(hipMemcpyAsync(host.data(), m->slots, (size_t)m->table_size * sizeof(VoxelSlot), hipMemcpyDeviceToHost, ctx->stream()));
  GA_HIP(hipStreamSynchronize(ctx->stream()));
  const unsigned int msk = (1u << KEY_BITS) - 1u;
  int v = 0;
  for (const VoxelSlot& s : host) {
    if (s.key == EMPTY_KEY) continue;
    if (v >= m->num_voxels) return GLIM_AMD_ERR_STATE;
    if (coords) {
      coords[3 * v + 0] = (int)((s.key >> (2 * KEY_BITS)) & msk) - KEY_OFFSET;
      coords[3 * v + 1] = (int)((s.key >> KEY_BITS) & msk) - KEY_OFFSET;
      coords[3 * v + 2] = (int)(s.key & msk) - KEY_OFFSET;
    }
    if (counts) counts[v] = s.count;
    if (means) {
      means[3 * v] = s.mx;
      means[3 * v + 1] = s.my;
      means[3 * v + 2] = s.mz;
    }
    if (cov33) {
      float* c = cov33 + 9 * v;
      c[0] = s.c00; c[1] = s.c01; c[2] = s.c02;
      c[3] = s.c01; c[4] = s.c11; c[5] = s.c12;
      c[6] = s.c02; c[7] = s.c12; c[8] = s.c22;
    }
    v++;
  }
  return v == m->num_voxels ? GLIM_AMD_OK : GLIM_AMD_ERR_STATE;
}

}  // extern "C"
