// context.hip -- library, error and context entry points of the C ABI (include/glim_amd.h).
#include <atomic>

#include "internal.hpp"

namespace glim_amd {

static thread_local char g_hip_error[512] = "";
static std::atomic<int> g_live_contexts{0};

void set_hip_error(hipError_t e, const char* what) {
  snprintf(g_hip_error, sizeof(g_hip_error), "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
  (void)hipGetLastError();  // clear the sticky error
}

}  // namespace glim_amd

using namespace glim_amd;

extern "C" {

int glim_amd_version(void) { return GLIM_AMD_VERSION; }

const char* glim_amd_error_string(int code) {
  switch (code) {
    case GLIM_AMD_OK: return "ok";
    case GLIM_AMD_ERR_INVALID: return "invalid argument";
    case GLIM_AMD_ERR_HIP: return "HIP runtime error";
    case GLIM_AMD_ERR_NO_DEVICE: return "no HIP device";
    case GLIM_AMD_ERR_RANGE: return "voxel coordinate out of key range";
    case GLIM_AMD_ERR_STATE: return "invalid state for this call";
    case GLIM_AMD_ERR_UNSUPPORTED: return "unsupported";
    case GLIM_AMD_ERR_NOMEM: return "out of memory";
    default: return "unknown error";
  }
}

const char* glim_amd_last_hip_error(void) { return g_hip_error; }

int glim_amd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int glim_amd_ctx_create(int device, int num_streams, void* external_stream, glim_amd_ctx** out) {
  if (!out || num_streams < 0) return GLIM_AMD_ERR_INVALID;
  *out = nullptr;
  const int ndev = glim_amd_device_count();
  if (ndev <= 0) return GLIM_AMD_ERR_NO_DEVICE;
  if (device < 0 || device >= ndev) return GLIM_AMD_ERR_INVALID;
  GA_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  GA_HIP(hipGetDeviceProperties(&prop, device));
  glim_amd_ctx* ctx = new glim_amd_ctx();
  ctx->device = device;
  ctx->num_cus = prop.multiProcessorCount;
  if (external_stream) {
    ctx->owns_streams = false;
    ctx->streams.push_back((hipStream_t)external_stream);
  } else {
    ctx->owns_streams = true;
    if (num_streams == 0) num_streams = 1;
    for (int i = 0; i < num_streams; i++) {
      hipStream_t s;
      hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
      if (e != hipSuccess) {
        set_hip_error(e, "hipStreamCreateWithFlags");
        for (auto t : ctx->streams) (void)hipStreamDestroy(t);
        delete ctx;
        return GLIM_AMD_ERR_HIP;
      }
      ctx->streams.push_back(s);
    }
  }
  g_live_contexts++;
  *out = ctx;
  return GLIM_AMD_OK;
}

int glim_amd_ctx_destroy(glim_amd_ctx* ctx) {
  if (!ctx) return GLIM_AMD_OK;
  if (ctx->live_children.load() != 0) return GLIM_AMD_ERR_STATE;  // children must be destroyed first; the context stays valid
  (void)hipSetDevice(ctx->device);
  for (auto s : ctx->streams) (void)hipStreamSynchronize(s);
  if (ctx->owns_streams)
    for (auto s : ctx->streams) (void)hipStreamDestroy(s);
  if (--g_live_contexts == 0) pool_trim(ctx->device);  // last context gone: give the cached device memory back
  delete ctx;
  return GLIM_AMD_OK;
}

int glim_amd_ctx_synchronize(glim_amd_ctx* ctx) {
  if (!ctx) return GLIM_AMD_ERR_INVALID;
  GA_HIP(hipSetDevice(ctx->device));
  for (auto s : ctx->streams) GA_HIP(hipStreamSynchronize(s));
  return GLIM_AMD_OK;
}

int glim_amd_device_info(glim_amd_ctx* ctx, char* name, size_t name_len, size_t* free_bytes, size_t* total_bytes, int* num_cus) {
  if (!ctx) return GLIM_AMD_ERR_INVALID;
  GA_HIP(hipSetDevice(ctx->device));
  hipDeviceProp_t prop;
  GA_HIP(hipGetDeviceProperties(&prop, ctx->device));
  if (name && name_len) {
    snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  size_t f = 0, t = 0;
  GA_HIP(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  if (num_cus) *num_cus = prop.multiProcessorCount;
  return GLIM_AMD_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// Device memory pool (per device, process-wide): hipMalloc / hipFree cost 10-100+ us each and hipFree synchronises the
// device, which is what made the per-frame path (upload -> kNN -> covariance -> voxel map) jittery (p99 46 ms on the 300k-point
// stream).  Freed blocks are cached by size and handed back to later requests of a similar size; every API call that frees
// scratch has synchronised its stream before returning, so re-use is ordered.  GLIM_AMD_NO_POOL=1 disables the cache.
// ---------------------------------------------------------------------------------------------------------------------
#include <map>
#include <unordered_map>

namespace glim_amd {

namespace {
struct DevicePool {
  std::mutex mu;
  std::multimap<size_t, void*> free_blocks;
  std::unordered_map<void*, size_t> live;
  size_t cached_bytes = 0;
};
DevicePool& pool_of(int device) {
  // intentionally leaked: contexts held in static storage by callers may be destroyed after any static of this library
  static DevicePool* pools = new DevicePool[64];
  return pools[(device >= 0 && device < 64) ? device : 0];
}
constexpr size_t kMaxCachedBytes = 32ull << 30;
bool pool_disabled() {
  static const bool off = getenv("GLIM_AMD_NO_POOL") != nullptr;
  return off;
}
}  // namespace

hipError_t pool_malloc_impl(void** p, size_t bytes) {
  if (bytes == 0) bytes = 1;
  const size_t want = (bytes + 255) & ~(size_t)255;
  int dev = 0;
  (void)hipGetDevice(&dev);
  DevicePool& P = pool_of(dev);
  if (!pool_disabled()) {
    std::lock_guard<std::mutex> lock(P.mu);
    auto it = P.free_blocks.lower_bound(want);
    if (it != P.free_blocks.end() && it->first <= want + want / 2 + 4096) {
      *p = it->second;
      P.live[*p] = it->first;
      P.cached_bytes -= it->first;
      P.free_blocks.erase(it);
      return hipSuccess;
    }
  }
  static const bool trace = getenv("GLIM_AMD_POOL_TRACE") != nullptr;
  if (trace) fprintf(stderr, "[glim_amd pool] miss: hipMalloc(%zu)\n", want);
  hipError_t e = hipMalloc(p, want);
  if (e != hipSuccess && !pool_disabled()) {  // out of memory: drop the cache and retry once
    (void)hipGetLastError();
    pool_trim(dev);
    e = hipMalloc(p, want);
  }
  if (e == hipSuccess && !pool_disabled()) {
    std::lock_guard<std::mutex> lock(P.mu);
    P.live[*p] = want;
  }
  return e;
}

hipError_t pool_free(void* p) {
  if (!p) return hipSuccess;
  if (pool_disabled()) return hipFree(p);
  int dev = 0;
  (void)hipGetDevice(&dev);
  DevicePool& P = pool_of(dev);
  size_t sz = 0;
  {
    std::lock_guard<std::mutex> lock(P.mu);
    auto it = P.live.find(p);
    if (it == P.live.end()) return hipFree(p);  // not ours (allocated before the pool was enabled)
    sz = it->second;
    P.live.erase(it);
    if (P.cached_bytes + sz <= kMaxCachedBytes) {
      P.free_blocks.emplace(sz, p);
      P.cached_bytes += sz;
      return hipSuccess;
    }
  }
  return hipFree(p);
}

// Pinned, device-mapped host memory (completion flags, pose / result staging of a factor set): hipHostMalloc / hipHostFree cost
// 100+ us each, and GLIM builds a fresh NonlinearFactorSetGPU for every linearisation, so these blocks are cached by size too.
namespace {
struct PinnedPool {
  std::mutex mu;
  std::multimap<size_t, void*> free_blocks;
  std::unordered_map<void*, size_t> live;
  size_t cached_bytes = 0;
};
PinnedPool& pinned_pool() {
  static PinnedPool* pool = new PinnedPool();  // leaked on purpose, like the device pools
  return *pool;
}
constexpr size_t kMaxCachedPinnedBytes = 256ull << 20;
}  // namespace

hipError_t pinned_malloc_impl(void** p, size_t bytes) {
  size_t want = 256;
  while (want < bytes) want <<= 1;  // power-of-two size classes: a set that grows by one factor still hits the cache
  PinnedPool& P = pinned_pool();
  if (!pool_disabled()) {
    std::lock_guard<std::mutex> lock(P.mu);
    auto it = P.free_blocks.find(want);
    if (it != P.free_blocks.end()) {
      *p = it->second;
      P.live[*p] = want;
      P.cached_bytes -= want;
      P.free_blocks.erase(it);
      return hipSuccess;
    }
  }
  hipError_t e = hipHostMalloc(p, want, hipHostMallocMapped | hipHostMallocPortable);  // cached blocks may be handed to a context on another device
  if (e == hipSuccess && !pool_disabled()) {
    std::lock_guard<std::mutex> lock(P.mu);
    P.live[*p] = want;
  }
  return e;
}

hipError_t pinned_free(void* p) {
  if (!p) return hipSuccess;
  if (pool_disabled()) return hipHostFree(p);
  PinnedPool& P = pinned_pool();
  {
    std::lock_guard<std::mutex> lock(P.mu);
    auto it = P.live.find(p);
    if (it == P.live.end()) return hipHostFree(p);
    const size_t sz = it->second;
    P.live.erase(it);
    if (P.cached_bytes + sz <= kMaxCachedPinnedBytes) {
      P.free_blocks.emplace(sz, p);
      P.cached_bytes += sz;
      return hipSuccess;
    }
  }
  return hipHostFree(p);
}

void pool_trim(int device) {
  {
    PinnedPool& H = pinned_pool();
    std::vector<void*> blocks;
    {
      std::lock_guard<std::mutex> lock(H.mu);
      for (auto& kv : H.free_blocks) blocks.push_back(kv.second);
      H.free_blocks.clear();
      H.cached_bytes = 0;
    }
    for (void* b : blocks) (void)hipHostFree(b);
  }
  DevicePool& P = pool_of(device);
  std::vector<void*> blocks;
  {
    std::lock_guard<std::mutex> lock(P.mu);
    for (auto& kv : P.free_blocks) blocks.push_back(kv.second);
    P.free_blocks.clear();
    P.cached_bytes = 0;
  }
  for (void* b : blocks) (void)hipFree(b);
}

}  // namespace glim_amd
