// multi.hip -- single-process, multi-device evaluation of a multi-scan VGICP cost (C ABI: glim_amd_multi_*).
//
// GLIM's GlobalMapping is ONE process that creates the matching-cost factors of every overlapping submap pair on one device with a pool of 64
// streams (src/glim/mapping/global_mapping.cpp:110, :430-484) and lets the optimiser linearise them all.  This is the MI355X-node extension of
// that call: one process, one context + one host worker thread + one RCCL communicator (ncclCommInitAll) per device.
//   - submap clouds and voxel maps are REPLICATED on every device (256 submaps x 64k points = 0.7 GB against 288 GB of HBM each),
//   - the FACTOR LIST is sharded into contiguous, cost-balanced chunks (cost = source points), so no point data ever crosses xGMI,
//   - every device linearises its chunk with the fused kernel and writes the 29-double compact records into its slot of a
//     [devices x max_rows x 29] array; ONE in-place ncclAllGather over xGMI completes the array on every device (it moves (N-1)/N of the
//     bytes once; a dense all-reduce of zero-padded rows would move twice that), device 0 copies it to the host and the records are expanded
//     to glim_amd_linearized6 (binary blocks by the adjoint identity) in the original factor order.
// librccl is opened lazily with dlopen: libglim_amd.so has no link-time dependency on it, and single-device users never load it.
#include <dlfcn.h>
// Only function pointers into librccl are used (dlopen below), so a ROCm install without the RCCL development headers still builds this
// library: the handful of types and enumerators the calls need are then declared here with RCCL's own values.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclDouble = 8 } ncclDataType_t;
#endif

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <thread>

#include "internal.hpp"

using namespace glim_amd;

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok() const { return CommInitAll && CommDestroy && AllGather; }
};

RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (api.handle) break;
    }
    if (!api.handle) return;
    api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(dlsym(api.handle, "ncclCommInitAll"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
    api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(api.handle, "ncclCommAbort"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(api.handle, "ncclAllGather"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
  });
  return api;
}

struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> task;
  bool has_task = false, done = false, quit = false;
  int rc = 0;
};

}  // namespace

struct glim_amd_multi {
  int ndev = 0;
  std::vector<int> devices;
  std::vector<glim_amd_ctx*> ctxs;
  std::vector<std::vector<glim_amd_cloud*>> clouds;   // [device][cloud id]
  std::vector<std::vector<glim_amd_voxelmap*>> maps;  // [device][map id]
  std::vector<glim_amd_factor_set*> sets;             // [device]
  std::vector<int64_t> bounds;                        // ndev + 1: device d owns factors [bounds[d], bounds[d + 1])
  std::vector<uint32_t> flags;
  int64_t nf = 0, max_rows = 0;
  std::vector<double*> d_gather;  // [device]: ndev x max_rows x COMPACT
  double* h_gather = nullptr;     // pinned
  std::vector<ncclComm_t> comms;
  bool use_rccl = false;
  bool broken = false;  // a collective failed and the communicators were aborted: only destroy is valid from here on
  std::vector<Worker*> workers;
  // HIP events per device around the two phases of the LAST evaluation (kernels, then collective + copy-out): glim_amd_multi_last_timing
  std::vector<hipEvent_t> ev;  // 3 per device: start, kernels enqueued-and-done boundary, end
  std::vector<float> kernel_ms, gather_ms;

  // run fn(device index) on every device's worker thread concurrently; first non-zero return code wins
  int run_all(const std::function<int(int)>& fn) {
    for (int d = 0; d < ndev; d++) {
      Worker* w = workers[d];
      std::lock_guard<std::mutex> lock(w->mu);
      w->task = [fn, d] { return fn(d); };
      w->has_task = true;
      w->done = false;
      w->cv.notify_all();
    }
    int rc = GLIM_AMD_OK;
    for (int d = 0; d < ndev; d++) {
      Worker* w = workers[d];
      std::unique_lock<std::mutex> lock(w->mu);
      w->cv.wait(lock, [w] { return w->done; });
      if (rc == GLIM_AMD_OK && w->rc != GLIM_AMD_OK) rc = w->rc;
    }
    return rc;
  }
};

namespace {

void worker_loop(Worker* w, int device) {
  (void)hipSetDevice(device);
  for (;;) {
    std::function<int()> task;
    {
      std::unique_lock<std::mutex> lock(w->mu);
      w->cv.wait(lock, [w] { return w->has_task || w->quit; });
      if (w->quit) return;
      task = std::move(w->task);
      w->has_task = false;
    }
    const int rc = task();
    {
      std::lock_guard<std::mutex> lock(w->mu);
      w->rc = rc;
      w->done = true;
      w->cv.notify_all();
    }
  }
}

void release_factors(glim_amd_multi* m) {
  for (int d = 0; d < m->ndev; d++) {
    (void)hipSetDevice(m->devices[d]);
    if (d < (int)m->ctxs.size() && m->ctxs[d]) (void)glim_amd_ctx_synchronize(m->ctxs[d]);  // asynchronous linearisations write into d_gather
    if (d < (int)m->sets.size() && m->sets[d]) (void)glim_amd_factor_set_destroy(m->sets[d]);
    if (d < (int)m->d_gather.size() && m->d_gather[d]) (void)pool_free(m->d_gather[d]);
  }
  m->sets.assign(m->ndev, nullptr);
  m->d_gather.assign(m->ndev, nullptr);
  if (m->h_gather) (void)pinned_free(m->h_gather);
  m->h_gather = nullptr;
  m->nf = m->max_rows = 0;
}

}  // namespace

extern "C" {

// Contiguous, cost-balanced split of a factor list (pure host arithmetic; also what glim_amd/multi.py computes for the one-process-per-GPU
// harness): bounds[r] = the boundary whose cumulative cost is nearest to r / world of the total, kept monotone.
int glim_amd_shard_bounds(const double* costs, int64_t n, int32_t world, int64_t* bounds) {
  if (n < 0 || world <= 0 || !bounds || (n > 0 && !costs)) return GLIM_AMD_ERR_INVALID;
  bounds[0] = 0;
  if (n == 0) {
    for (int r = 1; r <= world; r++) bounds[r] = 0;
    return GLIM_AMD_OK;
  }
  std::vector<double> cum((size_t)n + 1, 0.0);
  for (int64_t i = 0; i < n; i++) cum[(size_t)i + 1] = cum[(size_t)i] + costs[i];
  const double total = cum[(size_t)n];
  for (int r = 1; r < world; r++) {
    const double target = total * (double)r / (double)world;
    int64_t b = (int64_t)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());  // first index with cum >= target
    if (b > 0 && std::abs(cum[(size_t)b - 1] - target) <= std::abs(cum[(size_t)std::min<int64_t>(b, n)] - target)) b -= 1;
    bounds[r] = std::min<int64_t>(std::max<int64_t>(b, bounds[r - 1]), n);
  }
  bounds[world] = n;
  return GLIM_AMD_OK;
}

int glim_amd_multi_create(const int32_t* devices, int32_t num_devices, glim_amd_multi** out) {
  if (!out || num_devices <= 0 || !devices) return GLIM_AMD_ERR_INVALID;
  *out = nullptr;
  const int ndev_visible = glim_amd_device_count();
  if (ndev_visible <= 0) return GLIM_AMD_ERR_NO_DEVICE;
  for (int i = 0; i < num_devices; i++) {
    if (devices[i] < 0 || devices[i] >= ndev_visible) return GLIM_AMD_ERR_INVALID;
    for (int j = 0; j < i; j++)
      if (devices[j] == devices[i]) return GLIM_AMD_ERR_INVALID;
  }
  glim_amd_multi* m = new glim_amd_multi();
  m->ndev = num_devices;
  m->devices.assign(devices, devices + num_devices);
  m->clouds.resize(num_devices);
  m->maps.resize(num_devices);
  m->sets.assign(num_devices, nullptr);
  m->d_gather.assign(num_devices, nullptr);
  for (int d = 0; d < num_devices; d++) {
    glim_amd_ctx* ctx = nullptr;
    const int rc = glim_amd_ctx_create(devices[d], 1, nullptr, &ctx);
    if (rc != GLIM_AMD_OK) {
      for (auto c : m->ctxs) (void)glim_amd_ctx_destroy(c);
      delete m;
      return rc;
    }
    m->ctxs.push_back(ctx);
  }
  // RCCL: one communicator per device, created together in this process.  A single device still goes through the collective (it is a
  // copy there) unless diag multi_rccl=0 (GLIM_AMD_DIAG), so that the path the 8-GPU node takes is the path a 1-GPU box tests.
  const Diag& diag = process_diag();
  if (diag.multi_rccl && rccl().ok()) {
    m->comms.assign(num_devices, nullptr);
    const ncclResult_t r = rccl().CommInitAll(m->comms.data(), num_devices, m->devices.data());
    if (r == ncclSuccess) {
      m->use_rccl = true;
    } else {
      char msg[256];
      snprintf(msg, sizeof(msg), "ncclCommInitAll: %s", rccl().GetErrorString ? rccl().GetErrorString(r) : "error");
      set_hip_error(hipErrorUnknown, msg);
      m->comms.clear();
      if (num_devices > 1 && !diag.multi_host_gather) {
        for (auto c : m->ctxs) (void)glim_amd_ctx_destroy(c);
        delete m;
        return GLIM_AMD_ERR_HIP;  // refuse to silently fall back to a PCIe gather on a multi-device node
      }
    }
  }
  for (int d = 0; d < num_devices; d++) {
    Worker* w = new Worker();
    w->th = std::thread(worker_loop, w, devices[d]);
    m->workers.push_back(w);
  }
  m->kernel_ms.assign(num_devices, 0.f);
  m->gather_ms.assign(num_devices, 0.f);
  for (int d = 0; d < num_devices; d++) {
    (void)hipSetDevice(devices[d]);
    for (int e = 0; e < 3; e++) {
      hipEvent_t ev = nullptr;
      if (hipEventCreate(&ev) != hipSuccess) {
        (void)hipGetLastError();
        ev = nullptr;
      }
      m->ev.push_back(ev);
    }
  }
  for (hipEvent_t e : m->ev)
    if (!e) {  // no timing then
      for (hipEvent_t x : m->ev)
        if (x) (void)hipEventDestroy(x);
      m->ev.clear();
      break;
    }
  *out = m;
  return GLIM_AMD_OK;
}

int glim_amd_multi_last_timing(const glim_amd_multi* m, float* kernel_ms, float* gather_ms) {
  if (!m) return GLIM_AMD_ERR_INVALID;
  if (m->ev.empty()) return GLIM_AMD_ERR_UNSUPPORTED;
  for (int d = 0; d < m->ndev; d++) {
    if (kernel_ms) kernel_ms[d] = m->kernel_ms[d];
    if (gather_ms) gather_ms[d] = m->gather_ms[d];
  }
  return GLIM_AMD_OK;
}

int glim_amd_multi_destroy(glim_amd_multi* m) {
  if (!m) return GLIM_AMD_OK;
  for (hipEvent_t e : m->ev)
    if (e) (void)hipEventDestroy(e);
  m->ev.clear();
  for (Worker* w : m->workers) {
    {
      std::lock_guard<std::mutex> lock(w->mu);
      w->quit = true;
      w->cv.notify_all();
    }
    w->th.join();
    delete w;
  }
  release_factors(m);
  for (int d = 0; d < m->ndev; d++) {
    (void)hipSetDevice(m->devices[d]);
    for (auto v : m->maps[d]) (void)glim_amd_voxelmap_destroy(v);
    for (auto c : m->clouds[d]) (void)glim_amd_cloud_destroy(c);
  }
  if (m->use_rccl)
    for (auto c : m->comms)
      if (c) (void)rccl().CommDestroy(c);
  for (auto c : m->ctxs) (void)glim_amd_ctx_destroy(c);
  delete m;
  return GLIM_AMD_OK;
}

int glim_amd_multi_info(const glim_amd_multi* m, int32_t* num_devices, int32_t* uses_rccl, int64_t* num_factors) {
  if (!m) return GLIM_AMD_ERR_INVALID;
  if (num_devices) *num_devices = m->ndev;
  if (uses_rccl) *uses_rccl = m->use_rccl ? 1 : 0;
  if (num_factors) *num_factors = m->nf;
  return GLIM_AMD_OK;
}

int glim_amd_multi_add_cloud_f32(glim_amd_multi* m, int64_t n, const float* xyz, const float* cov33, const float* normals3, int32_t* cloud_id) {
  if (!m || !cloud_id) return GLIM_AMD_ERR_INVALID;
  std::vector<glim_amd_cloud*> made(m->ndev, nullptr);
  const int rc = m->run_all([&](int d) -> int { return glim_amd_cloud_create_f32(m->ctxs[d], n, xyz, cov33, normals3, &made[d]); });
  if (rc != GLIM_AMD_OK) {
    for (auto c : made) (void)glim_amd_cloud_destroy(c);
    return rc;
  }
  for (int d = 0; d < m->ndev; d++) m->clouds[d].push_back(made[d]);
  *cloud_id = (int32_t)m->clouds[0].size() - 1;
  return GLIM_AMD_OK;
}

int glim_amd_multi_add_cloud(glim_amd_multi* m, int64_t n, const double* points4, const double* covs16, const double* normals4, int32_t* cloud_id) {
  if (!m || !cloud_id) return GLIM_AMD_ERR_INVALID;
  std::vector<glim_amd_cloud*> made(m->ndev, nullptr);
  const int rc = m->run_all([&](int d) -> int { return glim_amd_cloud_create(m->ctxs[d], n, points4, covs16, normals4, &made[d]); });
  if (rc != GLIM_AMD_OK) {
    for (auto c : made) (void)glim_amd_cloud_destroy(c);
    return rc;
  }
  for (int d = 0; d < m->ndev; d++) m->clouds[d].push_back(made[d]);
  *cloud_id = (int32_t)m->clouds[0].size() - 1;
  return GLIM_AMD_OK;
}

int glim_amd_multi_cloud_estimate_covariances(glim_amd_multi* m, int32_t cloud_id, int k) {
  if (!m || cloud_id < 0 || cloud_id >= (int32_t)m->clouds[0].size()) return GLIM_AMD_ERR_INVALID;
  // deterministic kernels: every replica ends up with bit-identical neighbours, covariances and normals
  return m->run_all([&](int d) -> int {
    GA_TRY(glim_amd_cloud_find_neighbors(m->clouds[d][cloud_id], k, nullptr));
    return glim_amd_cloud_estimate_covariances(m->clouds[d][cloud_id], k);
  });
}

int glim_amd_multi_add_voxelmap(glim_amd_multi* m, int32_t cloud_id, double resolution, int32_t* map_id) {
  if (!m || !map_id || cloud_id < 0 || cloud_id >= (int32_t)m->clouds[0].size()) return GLIM_AMD_ERR_INVALID;
  std::vector<glim_amd_voxelmap*> made(m->ndev, nullptr);
  const int rc = m->run_all([&](int d) -> int {
    GA_TRY(glim_amd_voxelmap_create(m->ctxs[d], resolution, 8192 * 2, 10, 1e-3, &made[d]));
    return glim_amd_voxelmap_insert(made[d], m->clouds[d][cloud_id]);
  });
  if (rc != GLIM_AMD_OK) {
    for (auto v : made) (void)glim_amd_voxelmap_destroy(v);
    return rc;
  }
  for (int d = 0; d < m->ndev; d++) m->maps[d].push_back(made[d]);
  *map_id = (int32_t)m->maps[0].size() - 1;
  return GLIM_AMD_OK;
}

int glim_amd_multi_set_factors(glim_amd_multi* m, int64_t num_factors, const int32_t* target_map_ids, const int32_t* source_cloud_ids, const uint32_t* flags) {
  if (!m || num_factors < 0 || (num_factors > 0 && (!target_map_ids || !source_cloud_ids))) return GLIM_AMD_ERR_INVALID;
  const int32_t nmaps = (int32_t)m->maps[0].size(), nclouds = (int32_t)m->clouds[0].size();
  std::vector<double> costs((size_t)num_factors);
  for (int64_t f = 0; f < num_factors; f++) {
    if (target_map_ids[f] < 0 || target_map_ids[f] >= nmaps || source_cloud_ids[f] < 0 || source_cloud_ids[f] >= nclouds) return GLIM_AMD_ERR_INVALID;
    costs[(size_t)f] = (double)m->clouds[0][source_cloud_ids[f]]->n;
  }
  release_factors(m);  // the handle is at "no factors" from here until everything below has succeeded
  m->nf = 0;
  m->bounds.clear();
  m->flags.clear();
  std::vector<int64_t> bounds((size_t)m->ndev + 1, 0);
  GA_TRY(glim_amd_shard_bounds(costs.data(), num_factors, m->ndev, bounds.data()));
  int64_t max_rows = 1;
  for (int d = 0; d < m->ndev; d++) max_rows = std::max<int64_t>(max_rows, bounds[(size_t)d + 1] - bounds[(size_t)d]);
  const size_t gather_doubles = (size_t)m->ndev * (size_t)max_rows * COMPACT;
  if (pinned_malloc(&m->h_gather, gather_doubles * sizeof(double)) != hipSuccess) {
    (void)hipGetLastError();
    m->h_gather = nullptr;
    return GLIM_AMD_ERR_NOMEM;
  }
  m->bounds = bounds;
  m->flags.assign((size_t)num_factors, 0u);
  for (int64_t f = 0; f < num_factors; f++) m->flags[(size_t)f] = flags ? flags[f] : 0u;
  m->nf = num_factors;
  m->max_rows = max_rows;
  const int rc = m->run_all([&](int d) -> int {
    GA_HIP(hipSetDevice(m->devices[d]));
    GA_TRY(glim_amd_factor_set_create(m->ctxs[d], &m->sets[d]));
    for (int64_t f = m->bounds[d]; f < m->bounds[d + 1]; f++)
      GA_TRY(glim_amd_factor_set_add(m->sets[d], m->maps[d][target_map_ids[f]], m->clouds[d][source_cloud_ids[f]], m->flags[(size_t)f], nullptr));
    GA_HIP(pool_malloc(&m->d_gather[d], gather_doubles * sizeof(double)));
    GA_HIP(hipMemsetAsync(m->d_gather[d], 0, gather_doubles * sizeof(double), m->ctxs[d]->stream()));
    GA_HIP(hipStreamSynchronize(m->ctxs[d]->stream()));
    return (int)GLIM_AMD_OK;
  });
  if (rc != GLIM_AMD_OK) {  // no half-built factor list: the handle is back to "no factors"
    release_factors(m);
    m->nf = 0;
    m->bounds.clear();
    m->flags.clear();
  }
  return rc;
}

int glim_amd_multi_shard(const glim_amd_multi* m, int64_t* bounds) {
  if (!m || !bounds || m->bounds.empty()) return GLIM_AMD_ERR_INVALID;
  for (int d = 0; d <= m->ndev; d++) bounds[d] = m->bounds[d];
  return GLIM_AMD_OK;
}

// One evaluation of the whole cost: H / b / error of every factor at T_target_source (n x 12), records in the original factor order.
int glim_amd_multi_linearize(glim_amd_multi* m, const double* T, glim_amd_linearized6* out, double* total_error) {
  if (!m) return GLIM_AMD_ERR_INVALID;
  if (m->nf == 0) {
    if (total_error) *total_error = 0.0;
    return GLIM_AMD_OK;
  }
  if (!T) return GLIM_AMD_ERR_INVALID;
  if (m->broken) return GLIM_AMD_ERR_STATE;
  const size_t slot = (size_t)m->max_rows * COMPACT;
  // Two rounds over the workers: the kernels are enqueued first and the collective only if EVERY device managed to -- a device that failed
  // before its ncclAllGather would leave the others waiting in theirs for ever.  (The first round only enqueues; the extra hand-over between
  // the host threads costs ~20 us per evaluation.)
  GA_TRY(m->run_all([&](int d) -> int {
    GA_HIP(hipSetDevice(m->devices[d]));
    const int64_t lo = m->bounds[d], hi = m->bounds[d + 1];
    hipStream_t st = m->ctxs[d]->stream();
    if (m->ev.size() == (size_t)3 * m->ndev) GA_HIP(hipEventRecord(m->ev[3 * d], st));
    if (hi > lo) GA_TRY(glim_amd_factor_set_linearize_device_async(m->sets[d], T + 12 * lo, m->d_gather[d], (int64_t)d * m->max_rows));
    if (m->ev.size() == (size_t)3 * m->ndev) GA_HIP(hipEventRecord(m->ev[3 * d + 1], st));
    return (int)GLIM_AMD_OK;
  }));
  const int rc = m->run_all([&](int d) -> int {
    GA_HIP(hipSetDevice(m->devices[d]));
    hipStream_t st = m->ctxs[d]->stream();
    if (m->use_rccl) {
      // in place: this device's slot is both the send buffer and its own segment of the receive buffer
      const ncclResult_t r = rccl().AllGather(m->d_gather[d] + (size_t)d * slot, m->d_gather[d], slot, ncclDouble, m->comms[d], st);
      if (r != ncclSuccess) {
        set_hip_error(hipErrorUnknown, "ncclAllGather");
        return (int)GLIM_AMD_ERR_HIP;
      }
      if (d == 0) GA_HIP(hipMemcpyAsync(m->h_gather, m->d_gather[0], (size_t)m->ndev * slot * sizeof(double), hipMemcpyDeviceToHost, st));
    } else {
      // no collective library: every device hands its own slot to the host (PCIe); single-device and explicitly allowed setups only
      GA_HIP(hipMemcpyAsync(m->h_gather + (size_t)d * slot, m->d_gather[d] + (size_t)d * slot, slot * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    if (m->ev.size() == (size_t)3 * m->ndev) GA_HIP(hipEventRecord(m->ev[3 * d + 2], st));
    GA_HIP(hipStreamSynchronize(st));
    if (m->ev.size() == (size_t)3 * m->ndev) {
      (void)hipEventElapsedTime(&m->kernel_ms[d], m->ev[3 * d], m->ev[3 * d + 1]);
      (void)hipEventElapsedTime(&m->gather_ms[d], m->ev[3 * d + 1], m->ev[3 * d + 2]);
    }
    return (int)GLIM_AMD_OK;
  });
  if (rc != GLIM_AMD_OK) {
    // A device that failed before or inside its ncclAllGather leaves the others blocked in theirs: abort every communicator so that their
    // streams drain, and retire the handle (a communicator cannot be used after an abort).
    if (m->use_rccl && rccl().CommAbort) {
      for (auto& c : m->comms)
        if (c) {
          (void)rccl().CommAbort(c);
          c = nullptr;
        }
      m->broken = true;
    }
    return rc;
  }
  double total = 0.0;
  for (int d = 0; d < m->ndev; d++)
    for (int64_t f = m->bounds[d]; f < m->bounds[d + 1]; f++) {
      const double* rec = m->h_gather + (size_t)d * slot + (size_t)(f - m->bounds[d]) * COMPACT;
      total += rec[1];
      if (out) glim_amd_expand_compact(rec, T + 12 * f, m->flags[(size_t)f], &out[f]);
    }
  if (total_error) *total_error = total;
  return GLIM_AMD_OK;
}

int glim_amd_multi_profile(glim_amd_multi* m, const double* T, int iters, float* ms_per_evaluation) {
  if (!m || !T || iters <= 0 || !ms_per_evaluation) return GLIM_AMD_ERR_INVALID;
  for (int i = 0; i < 3; i++) GA_TRY(glim_amd_multi_linearize(m, T, nullptr, nullptr));
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; i++) GA_TRY(glim_amd_multi_linearize(m, T, nullptr, nullptr));
  const auto t1 = std::chrono::steady_clock::now();
  *ms_per_evaluation = (float)(std::chrono::duration<double, std::milli>(t1 - t0).count() / iters);
  return GLIM_AMD_OK;
}

}  // extern "C"
