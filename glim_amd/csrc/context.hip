// context.hip -- library, error and context entry points of the C ABI (include/glim_amd.h).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <string>

#include "internal.hpp"

namespace glim_amd {

static thread_local char g_hip_error[512] = "";
static std::atomic<int> g_live_contexts{0};
// every live context (quiesce_device walks it; guarded by g_ctx_registry_mu)
static std::mutex g_ctx_registry_mu;
static std::vector<glim_amd_ctx*> g_ctx_registry;

std::atomic<uint64_t>& global_mutation_epoch() {
  static std::atomic<uint64_t> e{1};
  return e;
}
void quiesce_device(int device, uint64_t uid) {
  resident_stop_device(device, uid);  // its workers hold descriptors into memory that is about to be recycled
  std::lock_guard<std::mutex> lock(g_ctx_registry_mu);
  for (glim_amd_ctx* c : g_ctx_registry)
    if (c->device == device) c->quiesce();
}

void set_hip_error(hipError_t e, const char* what) {
  snprintf(g_hip_error, sizeof(g_hip_error), "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
  (void)hipGetLastError();  // clear the sticky error
}

// ---- diagnostic switches (internal.hpp "Diag") --------------------------------------------------------------------------------
namespace {
struct DiagKey {
  const char* name;
  int Diag::*field;
  const char* const* words;  // value words (index = value), or null for a plain integer
  int lo, hi;
};
const char* const kPathWords[] = {"auto", "grid", "chunks", "brute", nullptr};
const char* const kKernelWords[] = {"auto", "wave64", "pair", "qgroup", nullptr};
const char* const kResidentWords[] = {"0", "1", "auto", nullptr};
const DiagKey kDiagKeys[] = {
  {"knn_path", &Diag::knn_path, kPathWords, 0, 3},
  {"knn_kernel", &Diag::knn_kernel, kKernelWords, 0, 3},
  {"knn_select", &Diag::knn_select, nullptr, 0, 1},
  {"plane", &Diag::plane, nullptr, 0, 1},
  {"curve_order", &Diag::curve_order, nullptr, 0, 1},
  {"ppt", &Diag::ppt, nullptr, 0, 256},
  {"poll", &Diag::poll, nullptr, 0, 1},
  {"inline_pose", &Diag::inline_pose, nullptr, 0, 1},
  {"bucket_factor", &Diag::bucket_factor, nullptr, 0, 64},
  {"plan_cache", &Diag::plan_cache, nullptr, 0, 1},
  {"plan_recycle", &Diag::plan_recycle, nullptr, 0, 1},
  {"host_poses", &Diag::host_poses, nullptr, 0, 1},
  {"host_pack", &Diag::host_pack, nullptr, 0, 1},
  {"pull_gated", &Diag::pull_gated, nullptr, 0, 1},
  {"frame_fused", &Diag::frame_fused, nullptr, 0, 1},
  {"view_fused", &Diag::view_fused, nullptr, 0, 1},
  {"fuse", &Diag::fuse, nullptr, 0, 1},
  {"pp_fast", &Diag::pp_fast, nullptr, 0, 1},
  {"resident", &Diag::resident, kResidentWords, 0, 2},
  {"resident_idle_us", &Diag::resident_idle_us, nullptr, 100, 1000000},
  {"cull", &Diag::cull, nullptr, 0, 2},
  {"small_rows", &Diag::small_rows, nullptr, 0, 4096},
  {"pool", &Diag::pool, nullptr, 0, 1},
  {"multi_rccl", &Diag::multi_rccl, nullptr, 0, 1},
  {"multi_host_gather", &Diag::multi_host_gather, nullptr, 0, 1},
  {"multi_virtual", &Diag::multi_virtual, nullptr, 0, 1},
};
}  // namespace

int diag_parse(Diag& d, const char* key_values) {
  Diag out = d;
  std::string s = key_values ? key_values : "";
  size_t pos = 0;
  while (pos < s.size()) {
    size_t end = s.find(',', pos);
    if (end == std::string::npos) end = s.size();
    const std::string item = s.substr(pos, end - pos);
    pos = end + 1;
    if (item.empty()) continue;
    const size_t eq = item.find('=');
    if (eq == std::string::npos) return GLIM_AMD_ERR_INVALID;
    const std::string key = item.substr(0, eq), val = item.substr(eq + 1);
    if (key == "knn_debug") {
      if (val.size() >= sizeof(out.knn_debug)) return GLIM_AMD_ERR_INVALID;
      snprintf(out.knn_debug, sizeof(out.knn_debug), "%s", val.c_str());
      continue;
    }
    bool found = false;
    for (const DiagKey& k : kDiagKeys) {
      if (key != k.name) continue;
      int v = -1;
      if (k.words) {
        for (int i = 0; k.words[i]; i++)
          if (val == k.words[i]) v = i;
      } else {
        char* e = nullptr;
        const long lv = strtol(val.c_str(), &e, 10);
        if (e != val.c_str() && *e == 0) v = (int)lv;
      }
      if (v < k.lo || v > k.hi) return GLIM_AMD_ERR_INVALID;
      out.*(k.field) = v;
      found = true;
    }
    if (!found) return GLIM_AMD_ERR_INVALID;
  }
  d = out;
  return GLIM_AMD_OK;
}

int diag_print(const Diag& d, char* buf, size_t len) {
  std::string s;
  for (const DiagKey& k : kDiagKeys) {
    if (!s.empty()) s += ",";
    s += k.name;
    s += "=";
    const int v = d.*(k.field);
    if (k.words) s += k.words[v];
    else s += std::to_string(v);
  }
  if (d.knn_debug[0]) s += std::string(",knn_debug=") + d.knn_debug;
  if (!buf || len < s.size() + 1) return GLIM_AMD_ERR_INVALID;
  memcpy(buf, s.c_str(), s.size() + 1);
  return GLIM_AMD_OK;
}

// the context's 1 KiB pinned scratch block as the host sees it and as kernels see it (mapped); false: not available here
bool pinned_scratch_views(::glim_amd_ctx* ctx, void** host, void** device) {
  constexpr size_t SCRATCH = 1024;
  if (!ctx->pinned_scratch && pinned_malloc(&ctx->pinned_scratch, SCRATCH) != hipSuccess) {
    (void)hipGetLastError();
    ctx->pinned_scratch = nullptr;
    return false;
  }
  if (!ctx->pinned_scratch_dev && hipHostGetDevicePointer(&ctx->pinned_scratch_dev, ctx->pinned_scratch, 0) != hipSuccess) {
    (void)hipGetLastError();
    ctx->pinned_scratch_dev = nullptr;
    return false;
  }
  *host = ctx->pinned_scratch;
  *device = ctx->pinned_scratch_dev;
  return true;
}

hipError_t read_back_sync(::glim_amd_ctx* ctx, hipStream_t st, void* dst_host, const void* src_device, size_t bytes) {
  constexpr size_t SCRATCH = 1024;
  if (bytes > SCRATCH || (!ctx->pinned_scratch && pinned_malloc(&ctx->pinned_scratch, SCRATCH) != hipSuccess)) {
    (void)hipGetLastError();
    hipError_t e = hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, st);  // pageable fallback
    return e != hipSuccess ? e : hipStreamSynchronize(st);
  }
  hipError_t e = hipMemcpyAsync(ctx->pinned_scratch, src_device, bytes, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e == hipSuccess) memcpy(dst_host, ctx->pinned_scratch, bytes);
  return e;
}

const Diag& process_diag() {
  static const Diag d = [] {
    Diag x;
    if (const char* env = getenv("GLIM_AMD_DIAG")) {
      if (diag_parse(x, env) != GLIM_AMD_OK) fprintf(stderr, "[glim_amd] GLIM_AMD_DIAG=\"%s\" not understood: ignored\n", env);
    }
    return x;
  }();
  return d;
}

}  // namespace glim_amd

using namespace glim_amd;

extern "C" {

int glim_amd_ctx_set_diag(glim_amd_ctx* ctx, const char* key_values) {
  if (!ctx) return GLIM_AMD_ERR_INVALID;
  std::lock_guard<std::mutex> lock(ctx->mu);
  if (!key_values || !key_values[0]) {  // back to the process defaults
    ctx->diag = process_diag();
    return GLIM_AMD_OK;
  }
  // pool / multi_rccl / multi_host_gather / multi_virtual are read from the PROCESS defaults only (pool_disabled, glim_amd_multi_create): setting
  // them on a context would be accepted and do nothing, so it is refused (they belong in GLIM_AMD_DIAG)
  for (const char* key : {"pool=", "multi_rccl=", "multi_host_gather=", "multi_virtual="}) {
    const size_t len = strlen(key);
    for (const char* q = key_values; (q = strstr(q, key)) != nullptr; q += len)
      if (q == key_values || q[-1] == ',') return GLIM_AMD_ERR_INVALID;
  }
  return diag_parse(ctx->diag, key_values);
}

// test hook: what an earlier read-back may have left in the context's shared pinned scratch (the polled voxel-map builds keep their completion
// word there: tests/test_gpu_edge_cases.py poisons it with the NEXT build's sequence number)
int glim_amd_debug_scratch_poke(glim_amd_ctx* ctx, int32_t word, uint32_t value) {
  if (!ctx || word < 0 || word >= 256) return GLIM_AMD_ERR_INVALID;
  std::lock_guard<std::mutex> lock(ctx->mu);
  void *h = nullptr, *d = nullptr;
  GA_HIP(hipSetDevice(ctx->device));
  if (!pinned_scratch_views(ctx, &h, &d)) return GLIM_AMD_ERR_UNSUPPORTED;
  if (value == 0xffffffffu) value = (ctx->map_seq + 1u) ? ctx->map_seq + 1u : 1u;  // "the sequence number the next polled build will wait for"
  reinterpret_cast<volatile uint32_t*>(h)[word] = value;
  return GLIM_AMD_OK;
}

int glim_amd_ctx_get_diag(glim_amd_ctx* ctx, char* buf, size_t len) {
  if (!ctx) return GLIM_AMD_ERR_INVALID;
  std::lock_guard<std::mutex> lock(ctx->mu);
  return diag_print(ctx->diag, buf, len);
}

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
  return glim_amd_ctx_create_ex(device, num_streams, external_stream, 0, out);
}

int glim_amd_ctx_create_ex(int device, int num_streams, void* external_stream, int priority, glim_amd_ctx** out) {
  if (!out || num_streams < 0 || priority < -1 || priority > 1) return GLIM_AMD_ERR_INVALID;
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
  ctx->diag = process_diag();
  ctx->priority = priority;
  if (external_stream) {
    ctx->owns_streams = false;
    ctx->streams.push_back((hipStream_t)external_stream);
  } else {
    ctx->owns_streams = true;
    if (num_streams == 0) num_streams = 1;
    int least = 0, greatest = 0;  // (numerically: greatest priority <= least priority)
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    for (int i = 0; i < num_streams; i++) {
      hipStream_t s;
      hipError_t e = priority == 0 ? hipStreamCreateWithFlags(&s, hipStreamNonBlocking)
                                   : hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority > 0 ? greatest : least);
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
  {
    std::lock_guard<std::mutex> lock(g_ctx_registry_mu);
    g_ctx_registry.push_back(ctx);
  }
  *out = ctx;
  return GLIM_AMD_OK;
}

int glim_amd_ctx_destroy(glim_amd_ctx* ctx) {
  if (!ctx) return GLIM_AMD_OK;
  if (ctx->live_children.load() != 0) return GLIM_AMD_ERR_STATE;  // children must be destroyed first; the context stays valid
  (void)hipSetDevice(ctx->device);
  {
    std::lock_guard<std::mutex> lock(g_ctx_registry_mu);
    g_ctx_registry.erase(std::remove(g_ctx_registry.begin(), g_ctx_registry.end(), ctx), g_ctx_registry.end());
  }
  for (auto s : ctx->streams) (void)hipStreamSynchronize(s);
  ctx_release_factor_resources(ctx);
  if (ctx->pinned_scratch) (void)pinned_free(ctx->pinned_scratch);
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

}  // extern "C"
namespace glim_amd {
void pool_stats(unsigned long long* device_mallocs, unsigned long long* device_frees, unsigned long long* pinned_mallocs, unsigned long long* cached_bytes);  // below
}
extern "C" {
int glim_amd_debug_pool_stats(uint64_t* device_mallocs, uint64_t* device_frees, uint64_t* pinned_mallocs, uint64_t* cached_bytes) {
  unsigned long long a = 0, b = 0, c = 0, d = 0;
  glim_amd::pool_stats(&a, &b, &c, cached_bytes ? &d : nullptr);
  if (device_mallocs) *device_mallocs = a;
  if (device_frees) *device_frees = b;
  if (pinned_mallocs) *pinned_mallocs = c;
  if (cached_bytes) *cached_bytes = d;
  return GLIM_AMD_OK;
}

int glim_amd_debug_ctx_query_streams(glim_amd_ctx* ctx, int32_t* busy) {
  if (!ctx) return GLIM_AMD_ERR_INVALID;
  GA_HIP(hipSetDevice(ctx->device));
  int n = 0;
  for (auto s : ctx->streams) {
    const hipError_t e = hipStreamQuery(s);
    if (e == hipErrorNotReady) n++;
    else if (e != hipSuccess) {
      set_hip_error(e, "hipStreamQuery");
      return GLIM_AMD_ERR_HIP;
    }
  }
  (void)hipGetLastError();
  if (busy) *busy = n;
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
// scratch has synchronised its stream before returning, so re-use is ordered.  GLIM_AMD_DIAG="pool=0" disables the cache.
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
// calls that went past the caches to the runtime (glim_amd_debug_pool_stats): a real hipMalloc / hipHostMalloc in a steady-state loop is a latency
// event for everything on the device, not only for the caller
std::atomic<unsigned long long> g_device_mallocs{0}, g_device_frees{0}, g_pinned_mallocs{0};
bool pool_disabled() {
  static const bool off = process_diag().pool == 0;
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
  g_device_mallocs++;
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
  g_device_frees++;
  return hipFree(p);
}

void pool_stats(unsigned long long* device_mallocs, unsigned long long* device_frees, unsigned long long* pinned_mallocs, unsigned long long* cached_bytes) {
  if (device_mallocs) *device_mallocs = g_device_mallocs.load();
  if (device_frees) *device_frees = g_device_frees.load();
  if (pinned_mallocs) *pinned_mallocs = g_pinned_mallocs.load();
  if (cached_bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    DevicePool& P = pool_of(dev);
    std::lock_guard<std::mutex> lock(P.mu);
    *cached_bytes = P.cached_bytes;
  }
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
  g_pinned_mallocs++;
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

void voxelmap_drop_cleared_tables(int device);  // voxelmap.hip: its cache of pre-cleared tables holds pool blocks

void pool_trim(int device) {
  voxelmap_drop_cleared_tables(device);
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
