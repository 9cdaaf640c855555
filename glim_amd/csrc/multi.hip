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
#include <atomic>
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

// A device's host thread.  Tasks are posted under the mutex; the thread SPINS on `posted` for a while after each task before it goes to
// sleep on the condition variable: back-to-back evaluations (an optimiser's relinearisations, 1-10 ms apart) then never pay a futex
// wake-up (30-60 us on these hosts, twice per evaluation in round 4's form), while an idle handle costs no CPU.
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> task;
  std::atomic<uint32_t> posted{0}, finished{0};
  bool quit = false;
  int rc = 0;
};
constexpr int WORKER_SPIN_US = 3000;
constexpr int EV_PER_DEVICE = 4;
constexpr int MAX_PIECES = 8;  // a device's shard is evaluated as up to this many factor sets (glim_amd_multi_set_split)

// Where a shard's records sit in the gathered array (see glim_amd_multi: one region per piece, equal slots per device inside a region).
// Pure arithmetic, shared by the evaluation, the device-side error sum and glim_amd_shard_layout (which the CPU tests drive for 8 devices).
inline int64_t layout_rows_of_piece(int64_t piece_rows, int64_t max_rows, int p) { return std::max<int64_t>(0, std::min(piece_rows, max_rows - (int64_t)p * piece_rows)); }
inline int64_t layout_row(int ndev, int64_t piece_rows, int64_t max_rows, int d, int64_t k) {
  const int p = (int)(k / piece_rows);
  return (int64_t)ndev * (int64_t)p * piece_rows + (int64_t)d * layout_rows_of_piece(piece_rows, max_rows, p) + (k - (int64_t)p * piece_rows);
}
inline void layout_pieces(int64_t max_rows, int ndev, int split_mode, int* pieces, int64_t* piece_rows) {
  (void)ndev;
  // default: pieces of >= 2048 factors (16 rounds of resident blocks on configs[3]'s submaps: the launch tail stays a few per cent of the piece), at
  // most 4 -- measured on one device for the 32 640 pairs of configs[3]: 10.92 / 10.74 / 10.66 / 10.73 ms per evaluation with 1 / 2 / 4 / 8 pieces
  // (the exposed exchange shrinks with the last piece, every further launch adds its tail) --, 1-2 per device on an 8-device node
  int want = split_mode < 0 ? (int)std::min<int64_t>(4, std::max<int64_t>(1, max_rows / 2048)) : std::max(1, split_mode);
  want = (int)std::min<int64_t>(std::min(want, MAX_PIECES), std::max<int64_t>(1, max_rows / 64));  // (a piece of a few rows is all launch tail)
  *piece_rows = (max_rows + want - 1) / want;
  *pieces = (int)((max_rows + *piece_rows - 1) / *piece_rows);
}

inline double us_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}
inline void relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#endif
}

}  // namespace

// host-side account of one evaluation on one device's thread (microseconds); glim_amd_multi_last_breakdown
enum {
  BD_POST = 0,        // caller: handing the evaluation to the other devices' threads
  BD_WAKE,            // this thread: from the caller's entry to the start of its task (0 for the caller's own device)
  BD_POSE_STAGE,      // poses of this shard copied into the plan's pinned ring (both halves)
  BD_ENQUEUE,         // plan check + H2D pose copy + kernel launches (both halves), pose staging excluded
  BD_BARRIER,         // waiting until EVERY device has enqueued its kernels (a collective must not start otherwise)
  BD_COLLECTIVE,      // ncclAllGather calls + copy-out enqueue
  BD_WAIT,            // hipStreamSynchronize: the device working
  BD_JOIN,            // caller: waiting for the other devices' threads after its own device is done
  BD_SCAN,            // caller: total error + expansion of the records in factor order
  BD_TOTAL,           // caller: the whole call
  BD_LIBRARY_CALLS,   // inside the ncclAllGather calls (part of BD_COLLECTIVE)
  BD_DEVICE_GATHER,   // HIP events: from the last kernel to the end of the last all-gather on the collective's stream
  BD_DEVICE_COPY_OUT, // HIP events: from there to the end of the copy-out and the error sum
  BD_FIELDS
};

struct glim_amd_multi {
  int ndev = 0;
  std::vector<int> devices;
  std::vector<glim_amd_ctx*> ctxs;
  std::vector<std::vector<glim_amd_cloud*>> clouds;   // [device][cloud id]
  std::vector<std::vector<glim_amd_voxelmap*>> maps;  // [device][map id]
  std::vector<glim_amd_factor_set*> sets;             // [device * MAX_PIECES + piece]: a device's shard as `pieces` factor sets
  std::vector<int64_t> bounds;                        // ndev + 1: device d owns factors [bounds[d], bounds[d + 1])
  std::vector<uint32_t> flags;
  int64_t nf = 0, max_rows = 0;
  // A shard is evaluated in `pieces` consecutive pieces of piece_rows rows (the last one shorter).  The gathered array holds one REGION per
  // piece: region p = the p-th pieces of every shard, device d's at d * rows_of_piece(p) inside it -- so that each piece is one in-place
  // ncclAllGather of equal slots, issued as soon as that piece's kernels are done while the next piece's kernels run.
  int pieces = 1;
  int64_t piece_rows = 0;
  int split_mode = -1;            // -1: default (pieces of >= 2048 factors, at most 4); 0 / 1: one piece; n >= 2: n pieces (glim_amd_multi_set_split)
  std::vector<double*> d_gather;  // [device]: ndev x max_rows x COMPACT
  double* h_gather = nullptr;     // pinned
  // [device]: device view of h_gather when the finalising kernels of that device store their records THERE as well (no device-to-host copy behind
  // the kernels), or null: copy-out by hipMemcpyAsync on the collective's stream (glim_amd_multi_set_host_records)
  std::vector<double*> h_gather_dev;
  bool mirror_records = true;
  std::vector<ncclComm_t> comms;
  bool use_rccl = false;
  bool one_rank_collective = false;  // ONE device: make the (no-op) library call in every evaluation all the same (glim_amd_multi_set_one_rank_collective)
  bool broken = false;  // a collective failed and the communicators were aborted: only destroy is valid from here on
  // "Virtual devices": one physical device listed several times (GLIM_AMD_DIAG multi_virtual=1 or glim_amd_debug_multi_create_virtual).  Every
  // entry gets its own context, host thread, shard, pieces, streams and gathered array -- the whole N > 1 path of this file -- and the exchange is
  // a same-device stand-in for the in-place ncclAllGather (RCCL refuses one device twice): every "device" copies its slot of a piece into the
  // others' arrays on its collective stream, behind the piece's event, exactly where the library call sits for distinct devices.
  bool virtual_devices = false;
  // What the exchange is FOR.  The host optimiser (GLIM's ISAM2 / LM run on the host: global_mapping.cpp:501) gets every record through the
  // finalising kernels' second store into the pinned host array; the gathered DEVICE array serves device-side consumers
  // (glim_amd_multi_gathered_device).  1 (default): the exchange is enqueued behind the pieces and completes BEHIND the call -- the call waits for
  // the kernels and the host records only; the next evaluation's kernels wait for it on the device before they overwrite the send slots, and
  // glim_amd_multi_gathered_device / _wait_gather wait for it on the host.  2: the call waits for the exchange as well (rounds 4-5).  0: no exchange.
  int gather_mode = 1;
  std::vector<hipEvent_t> gather_done_ev;  // [device]: the last evaluation's exchange (and copy-out) on the collective's stream is over
  std::vector<char> gather_pending;        // [device]: gather_done_ev has a record the device's kernels have not waited for yet
  bool gather_timing_pending = false;      // the last evaluation's collective-stream events have not been read (they complete behind the call)
  // failure injection (glim_amd_debug_multi_inject_failure): the next evaluation fails on this device before the barrier (1) / inside its exchange (2)
  int inject_device = -1, inject_where = 0;
  std::mutex abort_mu;
  std::vector<Worker*> workers;   // [device]; workers[0] is null: the CALLER's thread drives device 0 (no hand-over at all on one device)
  std::vector<hipStream_t> cstream;  // [device]: the collective's stream (the factor kernels of the next piece run beside it)
  std::vector<hipStream_t> ustream;  // [device]: pose uploads (the upload of the next piece runs beside the kernels of this one)
  std::vector<double*> d_sum_scratch;  // [device]: the blocks' partial sums + the arrival counter of sum_error_kernel
  std::vector<hipEvent_t> sum_ev;    // [device]: "this shard's error sum is in host memory", recorded on the factor sets' stream
  std::vector<hipEvent_t> piece_ev;  // [device * MAX_PIECES + piece]: "this piece's records are written", recorded on the factor sets' stream
  std::vector<double*> h_total, h_total_dev;  // [device]: this shard's error sum, written by the device into host-mapped memory (sum_error_kernel)
  // HIP events per device around the two phases of the LAST evaluation (kernels, then what is left of collective + copy-out): glim_amd_multi_last_timing
  std::vector<hipEvent_t> ev;  // 4 per device: start (sets' stream), kernels done (sets' stream), last all-gather done, end (both: collective stream)
  std::vector<float> kernel_ms, gather_ms;
  std::vector<double> breakdown;  // [device][BD_FIELDS]
  // host barrier of one evaluation: every device's thread arrives after it has enqueued its kernels
  std::atomic<uint64_t> arrived{0};
  std::atomic<int> failed{0};
  uint64_t generation = 0;

  int64_t rows_of_piece(int p) const { return layout_rows_of_piece(piece_rows, max_rows, p); }
  int64_t region_start(int p) const { return (int64_t)ndev * (int64_t)p * piece_rows; }  // (every earlier region is full)
  int64_t row_of(int d, int64_t k) const { return layout_row(ndev, piece_rows, max_rows, d, k); }  // row of the k-th factor of device d's shard

  // run fn(device index) on every device concurrently -- device 0 on the calling thread -- ; first non-zero return code wins
  int run_all(const std::function<int(int)>& fn, double* post_us = nullptr, double* join_us = nullptr) {
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<uint32_t> ticket((size_t)ndev, 0u);
    for (int d = 1; d < ndev; d++) {
      Worker* w = workers[d];
      std::lock_guard<std::mutex> lock(w->mu);
      w->task = [fn, d] { return fn(d); };
      ticket[(size_t)d] = w->posted.load(std::memory_order_relaxed) + 1;
      w->posted.store(ticket[(size_t)d], std::memory_order_release);
      w->cv.notify_one();
    }
    if (post_us) *post_us = us_since(t0);
    int prev = -1;
    (void)hipGetDevice(&prev);
    int rc = fn(0);
    if (prev >= 0) (void)hipSetDevice(prev);
    const auto t1 = std::chrono::steady_clock::now();
    for (int d = 1; d < ndev; d++) {
      Worker* w = workers[d];
      for (unsigned long spins = 0; w->finished.load(std::memory_order_acquire) != ticket[(size_t)d]; spins++) {
        if (spins < 200000) relax();
        else std::this_thread::sleep_for(std::chrono::microseconds(50));
      }
      if (rc == GLIM_AMD_OK && w->rc != GLIM_AMD_OK) rc = w->rc;
    }
    if (join_us) *join_us = us_since(t1);
    return rc;
  }
};

namespace {

void worker_loop(Worker* w, int device) {
  (void)hipSetDevice(device);
  uint32_t seen = 0;
  for (;;) {
    // spin first (see Worker), then sleep
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned long spins = 0; w->posted.load(std::memory_order_acquire) == seen; spins++) {
      relax();
      if ((spins & 0xff) == 0xff && us_since(t0) > WORKER_SPIN_US) break;
    }
    std::function<int()> task;
    {
      std::unique_lock<std::mutex> lock(w->mu);
      w->cv.wait(lock, [&] { return w->posted.load(std::memory_order_relaxed) != seen || w->quit; });
      if (w->quit) return;
      seen = w->posted.load(std::memory_order_relaxed);
      task = std::move(w->task);
    }
    w->rc = task();
    w->finished.store(seen, std::memory_order_release);
  }
}

void release_factors(glim_amd_multi* m) {
  for (int d = 0; d < m->ndev; d++) {  // every stream of every device first: a virtual device's exchange writes into the OTHERS' gathered arrays
    (void)hipSetDevice(m->devices[d]);
    if (d < (int)m->ctxs.size() && m->ctxs[d]) (void)glim_amd_ctx_synchronize(m->ctxs[d]);  // asynchronous linearisations write into d_gather
    if (d < (int)m->cstream.size() && m->cstream[d]) (void)hipStreamSynchronize(m->cstream[d]);
    if (d < (int)m->ustream.size() && m->ustream[d]) (void)hipStreamSynchronize(m->ustream[d]);
  }
  for (int d = 0; d < m->ndev; d++) {
    (void)hipSetDevice(m->devices[d]);
    for (int h = 0; h < MAX_PIECES; h++)
      if ((size_t)(MAX_PIECES * d + h) < m->sets.size() && m->sets[MAX_PIECES * d + h]) (void)glim_amd_factor_set_destroy(m->sets[MAX_PIECES * d + h]);
    if (d < (int)m->d_gather.size() && m->d_gather[d]) (void)pool_free(m->d_gather[d]);
  }
  m->sets.assign((size_t)MAX_PIECES * m->ndev, nullptr);
  m->d_gather.assign(m->ndev, nullptr);
  m->gather_pending.assign((size_t)m->ndev, 0);  // (every collective stream was synchronised above)
  m->gather_timing_pending = false;
  if (m->h_gather) (void)pinned_free(m->h_gather);
  m->h_gather = nullptr;
  m->h_gather_dev.assign((size_t)m->ndev, nullptr);
  m->nf = m->max_rows = m->piece_rows = 0;
  m->pieces = 1;
}

// A device failed before or inside its ncclAllGather: the others would wait in theirs for ever.  Abort every communicator (their kernels
// leave, their streams drain) and retire the handle -- a communicator cannot be used after an abort.  Any device's thread may call it.
void abort_collectives(glim_amd_multi* m) {
  std::lock_guard<std::mutex> lock(m->abort_mu);
  if (m->broken) return;
  if (m->use_rccl && rccl().CommAbort)
    for (auto& c : m->comms)
      if (c) {
        (void)rccl().CommAbort(c);
        c = nullptr;
      }
  m->broken = true;  // (virtual devices have no communicator to abort; the handle is retired all the same: its gathered arrays are half-exchanged)
}

// collective-stream timing of the last evaluation (HIP events 1 -> 2 -> 3 of every device), once those events have completed
void harvest_gather_timing(glim_amd_multi* m) {
  if (m->ev.size() != (size_t)EV_PER_DEVICE * m->ndev) return;
  for (int d = 0; d < m->ndev; d++) {
    float ms = 0.f;
    double* bd = &m->breakdown[(size_t)d * BD_FIELDS];
    if (hipEventElapsedTime(&ms, m->ev[EV_PER_DEVICE * d + 1], m->ev[EV_PER_DEVICE * d + 3]) == hipSuccess) m->gather_ms[d] = std::max(ms, 0.f);
    if (hipEventElapsedTime(&ms, m->ev[EV_PER_DEVICE * d + 1], m->ev[EV_PER_DEVICE * d + 2]) == hipSuccess) bd[BD_DEVICE_GATHER] = 1e3 * std::max(ms, 0.f);
    if (hipEventElapsedTime(&ms, m->ev[EV_PER_DEVICE * d + 2], m->ev[EV_PER_DEVICE * d + 3]) == hipSuccess) bd[BD_DEVICE_COPY_OUT] = 1e3 * std::max(ms, 0.f);
    (void)hipGetLastError();
  }
  m->gather_timing_pending = false;
}

// host wait for the exchange of the last evaluation on EVERY device (a virtual device's array is written by the others' streams)
int wait_gather(glim_amd_multi* m) {
  int prev = -1;
  (void)hipGetDevice(&prev);
  int rc = GLIM_AMD_OK;
  for (int d = 0; d < m->ndev; d++) {
    if (!m->gather_pending[(size_t)d]) continue;
    (void)hipSetDevice(m->devices[d]);
    const hipError_t e = hipEventSynchronize(m->gather_done_ev[(size_t)d]);
    if (e != hipSuccess) {
      set_hip_error(e, "glim_amd_multi: waiting for the exchange");
      rc = GLIM_AMD_ERR_HIP;
    }
  }
  if (rc == GLIM_AMD_OK && m->gather_timing_pending) harvest_gather_timing(m);
  if (prev >= 0) (void)hipSetDevice(prev);
  return rc;
}


// error sum of device d's own rows of the gathered array (column 1 of the 29-double records), bit-reproducible: thread g of the grid adds rows
// g, g + G, ... in FP64 (G = threads of the grid: ONE row each for configs[3]'s 32 640), a block adds its 256 values in a fixed tree, and the block
// that arrives last at the counter adds the blocks' sums in block order.  Many blocks because the rows are 232 B apart -- every load its own cache
// line, and ONE compute unit pulls lines at ~100 GB/s: the one-block form of the earlier takes (1 024 threads, eight loads in flight) kept the
// stream busy for 70 us behind the last kernel, the first version (a 256-thread dependent chain) for 0.2 ms.
constexpr int SUM_BLOCKS_MAX = 128;
__global__ __launch_bounds__(256) void sum_error_kernel(const double* __restrict__ gathered, int d, long long own_rows, long long piece_rows, long long max_rows,
                                                        int ndev, double* __restrict__ partials, unsigned int* __restrict__ counter, double* __restrict__ host_total) {
  __shared__ double s_part[256];
  __shared__ int s_last;
  double acc = 0.0;
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < own_rows; k += (long long)gridDim.x * 256) {
    const long long p = k / piece_rows;
    const long long rows_p = min(piece_rows, max_rows - p * piece_rows);
    const long long row = (long long)ndev * p * piece_rows + (long long)d * rows_p + (k - p * piece_rows);
    acc += gathered[(size_t)row * COMPACT + 1];
  }
  s_part[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) s_part[threadIdx.x] += s_part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    __hip_atomic_store(&partials[blockIdx.x], s_part[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned int ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ticket == gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  s_part[threadIdx.x] = (threadIdx.x < gridDim.x) ? __hip_atomic_load(&partials[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) s_part[threadIdx.x] += s_part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (the next evaluation's launch comes behind this one on the stream)
    __hip_atomic_store(host_total, s_part[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// scratch: SUM_BLOCKS_MAX doubles, then the arrival counter (zero between launches)
void sum_error_launch(hipStream_t st, const double* gathered, int d, int64_t own_rows, int64_t piece_rows, int64_t max_rows, int ndev, double* scratch, double* host_total) {
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(SUM_BLOCKS_MAX, (own_rows + 255) / 256));
  sum_error_kernel<<<blocks, 256, 0, st>>>(gathered, d, (long long)own_rows, (long long)piece_rows, (long long)max_rows, ndev, scratch,
                                           reinterpret_cast<unsigned int*>(scratch + SUM_BLOCKS_MAX), host_total);
}

}  // namespace

namespace {
// ONE device: the evaluation makes no library call (nothing to gather), so the binding is exercised here, once: the in-place all-gather of a
// record-sized buffer on the collective's stream must come back unchanged.  (On several devices every evaluation is the test.)
bool rccl_self_test(glim_amd_multi* m) {
  if (m->comms.empty() || !m->comms[0] || m->cstream.empty() || !m->cstream[0]) return false;
  double host[COMPACT], back[COMPACT];
  for (int i = 0; i < COMPACT; i++) host[i] = 1.0 + i, back[i] = 0.0;
  double* dev = nullptr;
  if (hipMalloc(&dev, sizeof(host)) != hipSuccess) return false;
  bool ok = hipMemcpy(dev, host, sizeof(host), hipMemcpyHostToDevice) == hipSuccess &&
            rccl().AllGather(dev, dev, COMPACT, ncclDouble, m->comms[0], m->cstream[0]) == ncclSuccess && hipStreamSynchronize(m->cstream[0]) == hipSuccess &&
            hipMemcpy(back, dev, sizeof(back), hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(dev);
  for (int i = 0; ok && i < COMPACT; i++) ok = back[i] == host[i];
  if (!ok) {
    (void)hipGetLastError();
    set_hip_error(hipErrorUnknown, "glim_amd_multi_create: one-rank ncclAllGather self-test failed; evaluating without the library");
  }
  return ok;
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

int glim_amd_shard_layout(const int64_t* bounds, int32_t world, int32_t split_mode, int64_t* rows, int64_t* max_rows_out, int32_t* pieces_out,
                          int64_t* piece_rows_out) {
  if (!bounds || world <= 0 || split_mode < -1 || split_mode > MAX_PIECES || bounds[0] != 0) return GLIM_AMD_ERR_INVALID;
  int64_t max_rows = 1;
  for (int d = 0; d < world; d++) {
    if (bounds[d + 1] < bounds[d]) return GLIM_AMD_ERR_INVALID;
    max_rows = std::max<int64_t>(max_rows, bounds[d + 1] - bounds[d]);
  }
  int pieces = 1;
  int64_t piece_rows = max_rows;
  layout_pieces(max_rows, world, split_mode, &pieces, &piece_rows);
  if (rows)
    for (int d = 0; d < world; d++)
      for (int64_t f = bounds[d]; f < bounds[d + 1]; f++) rows[f] = layout_row(world, piece_rows, max_rows, d, f - bounds[d]);
  if (max_rows_out) *max_rows_out = max_rows;
  if (pieces_out) *pieces_out = pieces;
  if (piece_rows_out) *piece_rows_out = piece_rows;
  return GLIM_AMD_OK;
}

static int multi_create_impl(const int32_t* devices, int32_t num_devices, bool allow_virtual, glim_amd_multi** out) {
  if (!out || num_devices <= 0 || num_devices > 64 || !devices) return GLIM_AMD_ERR_INVALID;
  *out = nullptr;
  const int ndev_visible = glim_amd_device_count();
  if (ndev_visible <= 0) return GLIM_AMD_ERR_NO_DEVICE;
  bool repeated = false;
  for (int i = 0; i < num_devices; i++) {
    if (devices[i] < 0 || devices[i] >= ndev_visible) return GLIM_AMD_ERR_INVALID;
    for (int j = 0; j < i; j++)
      if (devices[j] == devices[i]) repeated = true;
  }
  if (repeated && !allow_virtual) return GLIM_AMD_ERR_INVALID;
  glim_amd_multi* m = new glim_amd_multi();
  m->virtual_devices = repeated;
  m->ndev = num_devices;
  m->devices.assign(devices, devices + num_devices);
  m->clouds.resize(num_devices);
  m->maps.resize(num_devices);
  m->sets.assign((size_t)MAX_PIECES * num_devices, nullptr);
  m->d_gather.assign(num_devices, nullptr);
  m->breakdown.assign((size_t)num_devices * BD_FIELDS, 0.0);
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
  if (diag.multi_rccl && rccl().ok() && !m->virtual_devices) {  // (ncclCommInitAll refuses one device twice: virtual devices exchange by copies)
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
  m->workers.assign(num_devices, nullptr);
  for (int d = 1; d < num_devices; d++) {  // (device 0 is driven by the calling thread)
    Worker* w = new Worker();
    w->th = std::thread(worker_loop, w, devices[d]);
    m->workers[d] = w;
  }
  m->kernel_ms.assign(num_devices, 0.f);
  m->gather_ms.assign(num_devices, 0.f);
  m->cstream.assign(num_devices, nullptr);
  m->ustream.assign(num_devices, nullptr);
  m->piece_ev.assign((size_t)MAX_PIECES * num_devices, nullptr);
  m->sum_ev.assign(num_devices, nullptr);
  m->gather_done_ev.assign(num_devices, nullptr);
  m->gather_pending.assign(num_devices, 0);
  m->d_sum_scratch.assign(num_devices, nullptr);
  m->h_total.assign(num_devices, nullptr);
  m->h_total_dev.assign(num_devices, nullptr);
  int prev_device = -1;
  (void)hipGetDevice(&prev_device);
  bool streams_ok = true;
  for (int d = 0; d < num_devices; d++) {
    (void)hipSetDevice(devices[d]);
    if (hipStreamCreateWithFlags(&m->cstream[d], hipStreamNonBlocking) != hipSuccess) streams_ok = false;
    if (hipStreamCreateWithFlags(&m->ustream[d], hipStreamNonBlocking) != hipSuccess) streams_ok = false;
    for (int h = 0; h < MAX_PIECES; h++)
      if (hipEventCreateWithFlags(&m->piece_ev[MAX_PIECES * d + h], hipEventDisableTiming) != hipSuccess) streams_ok = false;
    if (hipEventCreateWithFlags(&m->sum_ev[d], hipEventDisableTiming) != hipSuccess) streams_ok = false;
    if (hipEventCreateWithFlags(&m->gather_done_ev[d], hipEventDisableTiming) != hipSuccess) streams_ok = false;
    if (hipMalloc(reinterpret_cast<void**>(&m->d_sum_scratch[d]), (SUM_BLOCKS_MAX + 1) * sizeof(double)) != hipSuccess ||
        hipMemset(m->d_sum_scratch[d], 0, (SUM_BLOCKS_MAX + 1) * sizeof(double)) != hipSuccess)
      streams_ok = false;
    if (pinned_malloc(&m->h_total[d], 64) != hipSuccess || hipHostGetDevicePointer(reinterpret_cast<void**>(&m->h_total_dev[d]), m->h_total[d], 0) != hipSuccess)
      streams_ok = false;
  }
  if (!streams_ok) {
    set_hip_error(hipGetLastError(), "glim_amd_multi_create: collective streams / events");
    if (prev_device >= 0) (void)hipSetDevice(prev_device);
    (void)glim_amd_multi_destroy(m);
    return GLIM_AMD_ERR_HIP;
  }
  for (int d = 0; d < num_devices; d++) {
    (void)hipSetDevice(devices[d]);
    for (int e = 0; e < EV_PER_DEVICE; e++) {
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
  if (num_devices == 1 && m->use_rccl && !rccl_self_test(m)) {  // (one device has nothing to gather: the library is not needed to evaluate)
    (void)rccl().CommDestroy(m->comms[0]);
    m->comms.clear();
    m->use_rccl = false;
  }
  if (prev_device >= 0) (void)hipSetDevice(prev_device);
  *out = m;
  return GLIM_AMD_OK;
}

int glim_amd_multi_create(const int32_t* devices, int32_t num_devices, glim_amd_multi** out) {
  return multi_create_impl(devices, num_devices, process_diag().multi_virtual != 0, out);
}

int glim_amd_debug_multi_create_virtual(const int32_t* devices, int32_t num_devices, glim_amd_multi** out) {
  return multi_create_impl(devices, num_devices, true, out);
}

int glim_amd_debug_multi_inject_failure(glim_amd_multi* m, int32_t device, int32_t where) {
  if (!m || device < -1 || device >= m->ndev || where < 0 || where > 2) return GLIM_AMD_ERR_INVALID;
  m->inject_device = where ? device : -1;
  m->inject_where = device >= 0 ? where : 0;
  return GLIM_AMD_OK;
}

int glim_amd_debug_multi_set_diag(glim_amd_multi* m, const char* key_values) {
  if (!m) return GLIM_AMD_ERR_INVALID;
  for (glim_amd_ctx* c : m->ctxs) GA_TRY(glim_amd_ctx_set_diag(c, key_values));
  return GLIM_AMD_OK;
}

int glim_amd_multi_set_gather_mode(glim_amd_multi* m, int32_t mode) {
  if (!m || mode < 0 || mode > 2) return GLIM_AMD_ERR_INVALID;
  GA_TRY(wait_gather(m));
  m->gather_mode = mode;
  return GLIM_AMD_OK;
}

int glim_amd_multi_wait_gather(glim_amd_multi* m) {
  if (!m) return GLIM_AMD_ERR_INVALID;
  return wait_gather(m);
}

int glim_amd_multi_gathered_device(glim_amd_multi* m, int32_t device, const double** gathered, int64_t* rows) {
  if (!m || device < 0 || device >= m->ndev || !gathered) return GLIM_AMD_ERR_INVALID;
  if (m->broken || m->nf == 0 || !m->d_gather[(size_t)device]) return GLIM_AMD_ERR_STATE;
  GA_TRY(wait_gather(m));
  *gathered = m->d_gather[(size_t)device];
  if (rows) *rows = (int64_t)m->ndev * m->max_rows;
  return GLIM_AMD_OK;
}

int glim_amd_debug_multi_gathered_download(glim_amd_multi* m, int32_t device, int64_t first, int64_t count, double* compact29) {
  if (!m || device < 0 || device >= m->ndev || !compact29 || first < 0 || count < 0 || first + count > m->nf) return GLIM_AMD_ERR_INVALID;
  const double* dev = nullptr;
  GA_TRY(glim_amd_multi_gathered_device(m, device, &dev, nullptr));
  const size_t doubles = (size_t)m->ndev * (size_t)m->max_rows * COMPACT;
  std::vector<double> host(doubles);
  int prev = -1;
  (void)hipGetDevice(&prev);
  GA_HIP(hipSetDevice(m->devices[device]));
  const hipError_t e = hipMemcpy(host.data(), dev, doubles * sizeof(double), hipMemcpyDeviceToHost);
  if (prev >= 0) (void)hipSetDevice(prev);
  GA_HIP(e);
  for (int64_t f = first; f < first + count; f++) {
    int d = 0;
    while (d + 1 < m->ndev && f >= m->bounds[d + 1]) d++;
    memcpy(compact29 + (size_t)(f - first) * COMPACT, host.data() + (size_t)m->row_of(d, f - m->bounds[d]) * COMPACT, COMPACT * sizeof(double));
  }
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
    if (!w) continue;
    {
      std::lock_guard<std::mutex> lock(w->mu);
      w->quit = true;
      w->posted.fetch_add(1, std::memory_order_release);  // (ends the spin phase)
      w->cv.notify_all();
    }
    w->th.join();
    delete w;
  }
  m->workers.clear();
  release_factors(m);
  for (int d = 0; d < m->ndev; d++) {
    (void)hipSetDevice(m->devices[d]);
    if (d < (int)m->cstream.size() && m->cstream[d]) (void)hipStreamDestroy(m->cstream[d]);
    if (d < (int)m->ustream.size() && m->ustream[d]) (void)hipStreamDestroy(m->ustream[d]);
    for (int h = 0; h < MAX_PIECES; h++)
      if ((size_t)(MAX_PIECES * d + h) < m->piece_ev.size() && m->piece_ev[MAX_PIECES * d + h]) (void)hipEventDestroy(m->piece_ev[MAX_PIECES * d + h]);
    if (d < (int)m->sum_ev.size() && m->sum_ev[d]) (void)hipEventDestroy(m->sum_ev[d]);
    if (d < (int)m->gather_done_ev.size() && m->gather_done_ev[d]) (void)hipEventDestroy(m->gather_done_ev[d]);
    if (d < (int)m->d_sum_scratch.size() && m->d_sum_scratch[d]) (void)hipFree(m->d_sum_scratch[d]);
    if (d < (int)m->h_total.size() && m->h_total[d]) (void)pinned_free(m->h_total[d]);
  }
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
  // Several pieces per shard: the all-gather and the copy-out of piece p (the collective's stream) run beside the kernels of piece p + 1, and
  // the pose upload of piece p + 1 beside the kernels of piece p; only the LAST piece's exchange is exposed.  glim_amd/multi.py
  // `gather_device_halves` is the two-piece form of the same exchange for one process per GPU.  More pieces = a shorter exposed tail but one
  // more launch tail each: pieces of >= 2048 factors by default (layout_pieces).
  int pieces = 1;
  int64_t piece_rows = max_rows;
  layout_pieces(max_rows, m->ndev, m->split_mode, &pieces, &piece_rows);
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
  m->piece_rows = piece_rows;
  m->pieces = pieces;
  m->h_gather_dev.assign((size_t)m->ndev, nullptr);
  const int rc = m->run_all([&](int d) -> int {
    GA_HIP(hipSetDevice(m->devices[d]));
    if (m->mirror_records && hipHostGetDevicePointer(reinterpret_cast<void**>(&m->h_gather_dev[d]), m->h_gather, 0) != hipSuccess) {
      (void)hipGetLastError();
      m->h_gather_dev[d] = nullptr;  // (no device view of the pinned array here: copies behind the kernels)
    }
    const int64_t lo = m->bounds[d], hi = m->bounds[d + 1];
    for (int h = 0; h < m->pieces; h++) {
      const int64_t f0 = std::min(lo + (int64_t)h * piece_rows, hi), f1 = std::min(f0 + piece_rows, hi);
      if (f1 <= f0) continue;
      GA_TRY(glim_amd_factor_set_create(m->ctxs[d], &m->sets[MAX_PIECES * d + h]));
      m->sets[MAX_PIECES * d + h]->upload_stream = m->ustream[d];
      m->sets[MAX_PIECES * d + h]->record_mirror = m->h_gather_dev[d];
      for (int64_t f = f0; f < f1; f++)
        GA_TRY(glim_amd_factor_set_add(m->sets[MAX_PIECES * d + h], m->maps[d][target_map_ids[f]], m->clouds[d][source_cloud_ids[f]], m->flags[(size_t)f], nullptr));
    }
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

int glim_amd_multi_set_split(glim_amd_multi* m, int32_t mode) {
  if (!m || mode < -1 || mode > MAX_PIECES) return GLIM_AMD_ERR_INVALID;
  m->split_mode = mode;  // takes effect with the next glim_amd_multi_set_factors
  return GLIM_AMD_OK;
}

int glim_amd_multi_set_host_records(glim_amd_multi* m, int32_t mode) {
  if (!m || mode < 0 || mode > 1) return GLIM_AMD_ERR_INVALID;
  m->mirror_records = mode == 1;  // takes effect with the next glim_amd_multi_set_factors
  return GLIM_AMD_OK;
}

int glim_amd_multi_set_one_rank_collective(glim_amd_multi* m, int32_t on) {
  if (!m) return GLIM_AMD_ERR_INVALID;
  m->one_rank_collective = on != 0;
  return GLIM_AMD_OK;
}

int glim_amd_multi_shard(const glim_amd_multi* m, int64_t* bounds) {
  if (!m || !bounds || m->bounds.empty()) return GLIM_AMD_ERR_INVALID;
  for (int d = 0; d <= m->ndev; d++) bounds[d] = m->bounds[d];
  return GLIM_AMD_OK;
}

// One evaluation of the whole cost: H / b / error of every factor at T_target_source (n x 12), records in the original factor order.
//
// ONE hand-over per evaluation (round 4 handed the devices' threads two tasks -- enqueue, then collective -- and woke them through a
// condition variable each time; the driver's box showed 2.2 ms of host time per evaluation nobody could name).  Every device's thread -- the
// caller's own for device 0 -- now runs the whole sequence:
//   for every piece p of its shard: poses -> pinned ring -> H2D, kernels, event p          (the factor sets' stream; the pose staging of
//                                                                                          piece p + 1 overlaps the kernels of piece p)
//   host barrier: has EVERY device enqueued its kernels?                                   (a device that failed must keep the others out of the collective)
//   the shard's error sum, by the device, into host-mapped memory                           (the same stream, behind the last piece)
//   for every piece p, behind event p: ncclAllGather of the p-th pieces over xGMI          (the collective's stream; a device's OWN rows reach the
//                                                                                          host array by a second store of its finalising kernels --
//                                                                                          or, glim_amd_multi_set_host_records(0), by a copy here)
//   one synchronise                                                                        (of the collective's stream; ONE device that has nothing
//                                                                                          to exchange waits for the kernels' own stream instead)
// The host barrier costs no device time: the kernels are running while the threads meet.  glim_amd_multi_last_breakdown names every phase.
int glim_amd_multi_linearize(glim_amd_multi* m, const double* T, glim_amd_linearized6* out, double* total_error) {
  if (!m) return GLIM_AMD_ERR_INVALID;
  if (m->nf == 0) {
    if (total_error) *total_error = 0.0;
    return GLIM_AMD_OK;
  }
  if (!T) return GLIM_AMD_ERR_INVALID;
  if (m->broken) return GLIM_AMD_ERR_STATE;
  const auto t_call = std::chrono::steady_clock::now();
  const int ndev = m->ndev, P = m->pieces;
  const bool timed = m->ev.size() == (size_t)EV_PER_DEVICE * ndev;
  m->generation += 1;
  const uint64_t all_arrived = m->generation * (uint64_t)ndev;
  m->failed.store(0);
  std::fill(m->breakdown.begin(), m->breakdown.end(), 0.0);
  const int inject_device = m->inject_device, inject_where = m->inject_where;  // one shot
  m->inject_device = -1;
  m->inject_where = 0;
  // is there an exchange at all?  Several devices with a communicator (or virtual devices, whose stand-in copies take its place), unless switched
  // off; ONE device only when the no-op library call was asked for (glim_amd_multi_set_one_rank_collective)
  const bool exchange = m->gather_mode != 0 && ((ndev > 1 && (m->use_rccl || m->virtual_devices)) || (ndev == 1 && m->use_rccl && m->one_rank_collective));
  double post_us = 0.0, join_us = 0.0;
  const int rc = m->run_all(
    [&](int d) -> int {
      double* bd = &m->breakdown[(size_t)d * BD_FIELDS];
      bd[BD_WAKE] = d ? us_since(t_call) : 0.0;
      int rc_d = GLIM_AMD_OK;
      hipStream_t sst = nullptr, last = nullptr;  // the factor sets' stream; the stream of the last piece enqueued
      bool summed = false;
      const int64_t lo = m->bounds[d], hi = m->bounds[d + 1];
      auto enqueue_kernels = [&]() -> int {
        GA_HIP(hipSetDevice(m->devices[d]));
        sst = m->ctxs[d]->stream();
        for (int h = 0; h < P; h++)
          if (m->sets[MAX_PIECES * d + h]) {
            sst = m->sets[MAX_PIECES * d + h]->stream;
            break;
          }
        last = sst;
        if (inject_device == d && inject_where == 1) {
          set_hip_error(hipErrorUnknown, "glim_amd_multi_linearize: injected failure before the barrier");
          return (int)GLIM_AMD_ERR_HIP;
        }
        // the previous evaluation's exchange reads this device's slots of the gathered array (they are its send buffer): the kernels that
        // overwrite them wait for it ON THE DEVICE (it has normally finished long ago: the host has been away choosing the next poses)
        if (m->gather_pending[(size_t)d]) {
          for (int h = 0; h < P; h++)
            if (m->sets[MAX_PIECES * d + h]) GA_HIP(hipStreamWaitEvent(m->sets[MAX_PIECES * d + h]->stream, m->gather_done_ev[(size_t)d], 0));
          m->gather_pending[(size_t)d] = 0;
        }
        if (timed) GA_HIP(hipEventRecord(m->ev[EV_PER_DEVICE * d], sst));
        for (int h = 0; h < P; h++) {
          glim_amd_factor_set* set = m->sets[MAX_PIECES * d + h];
          const int64_t f0 = std::min(lo + (int64_t)h * m->piece_rows, hi);
          if (set) {
            const auto t0 = std::chrono::steady_clock::now();
            GA_TRY(glim_amd_factor_set_linearize_device_async(set, T + 12 * f0, m->d_gather[d], m->row_of(d, f0 - lo)));
            bd[BD_POSE_STAGE] += set->last_pose_stage_us;
            bd[BD_ENQUEUE] += us_since(t0) - set->last_pose_stage_us;
            last = set->stream;
          }
          GA_HIP(hipEventRecord(m->piece_ev[MAX_PIECES * d + h], last));
        }
        if (timed) GA_HIP(hipEventRecord(m->ev[EV_PER_DEVICE * d + 1], last));
        // the shard's error sum, by the device (the host would stream the whole 232-B-per-factor array through its caches for it: 190 us for
        // 32 640 factors on the round's boxes), on the stream of the kernels whose rows it reads: it needs neither the gather nor a second stream
        // (round 5 take 2 launched it on the collective's stream, behind a cross-stream wait: 68 us after the last kernel on one device)
        summed = false;
        if (total_error && !out) {
          *m->h_total[d] = 0.0;
          if (hi > lo) {
            sum_error_launch(last, m->d_gather[d], d, hi - lo, m->piece_rows, m->max_rows, ndev, m->d_sum_scratch[d], m->h_total_dev[d]);
            GA_HIP(hipGetLastError());
            GA_HIP(hipEventRecord(m->sum_ev[d], last));
            summed = true;
          }
        }
        return (int)GLIM_AMD_OK;
      };
      rc_d = enqueue_kernels();
      if (rc_d != GLIM_AMD_OK) {
        int none = 0;
        (void)m->failed.compare_exchange_strong(none, rc_d);  // the first failure's own code is what every device reports
      }
      // ---- every device has enqueued (or one has failed: then nobody starts a collective) ----
      const auto t_bar = std::chrono::steady_clock::now();
      m->arrived.fetch_add(1, std::memory_order_acq_rel);
      for (unsigned long spins = 0; m->arrived.load(std::memory_order_acquire) < all_arrived; spins++) {
        if (spins < 100000) relax();
        else std::this_thread::sleep_for(std::chrono::microseconds(20));
      }
      bd[BD_BARRIER] = us_since(t_bar);
      if (m->failed.load()) {
        if (sst) (void)hipStreamSynchronize(sst);  // what this device did enqueue must not outlive the call
        if (last && last != sst) (void)hipStreamSynchronize(last);
        return rc_d != GLIM_AMD_OK ? rc_d : m->failed.load();
      }
      // ---- exchange (+ copy-out) on the collective's stream ----
      const auto t_col = std::chrono::steady_clock::now();
      hipStream_t cst = m->cstream[d];
      const bool mirrored = m->h_gather_dev[d] != nullptr;
      // nothing for the collective's stream to do: no exchange, and the kernels store their records to the host themselves
      const bool own_stream_only = !exchange && mirrored && sst != nullptr;
      // the call waits for the exchange only when asked to (gather_mode 2) or when the host records come by copies on its stream
      const bool wait_for_exchange = !own_stream_only && (m->gather_mode == 2 || !mirrored || !exchange);
      if (own_stream_only) cst = last;
      auto collective = [&]() -> int {
        if (own_stream_only) {
          if (timed) {
            GA_HIP(hipEventRecord(m->ev[EV_PER_DEVICE * d + 2], cst));
            GA_HIP(hipEventRecord(m->ev[EV_PER_DEVICE * d + 3], cst));
          }
          return (int)GLIM_AMD_OK;
        }
        for (int h = 0; h < P; h++) {
          const int64_t rows = m->rows_of_piece(h);
          if (rows == 0) {
            if (timed && h == P - 1) GA_HIP(hipEventRecord(m->ev[EV_PER_DEVICE * d + 2], cst));
            continue;
          }
          const size_t start = (size_t)m->region_start(h) * COMPACT, slot = (size_t)rows * COMPACT;
          double* region = m->d_gather[d] + start;
          GA_HIP(hipStreamWaitEvent(cst, m->piece_ev[MAX_PIECES * d + h], 0));
          if (inject_device == d && inject_where == 2 && h == P - 1) {
            set_hip_error(hipErrorUnknown, "glim_amd_multi_linearize: injected failure inside the exchange");
            return (int)GLIM_AMD_ERR_HIP;
          }
          if (exchange && m->use_rccl && ndev > 1) {
            // in place: this device's slot is both the send buffer and its own segment of the receive buffer
            const auto t_lib = std::chrono::steady_clock::now();
            const ncclResult_t r = rccl().AllGather(region + (size_t)d * slot, region, slot, ncclDouble, m->comms[d], cst);
            bd[BD_LIBRARY_CALLS] += us_since(t_lib);
            if (r != ncclSuccess) {
              set_hip_error(hipErrorUnknown, "ncclAllGather");
              return (int)GLIM_AMD_ERR_HIP;
            }
          } else if (exchange && m->virtual_devices && ndev > 1) {
            // the stand-in (virtual devices share one physical device: plain copies reach every array): this "device" sends its slot of the piece
            // to the same slot of every other array -- the bytes an all-gather moves, in the place and stream order of the library call.  (Every
            // device has enqueued its kernels -- the host barrier above -- and a receiver only WRITES its own slot, so no other event is needed;
            // who READS a gathered array waits for every device's exchange: wait_gather.)
            const auto t_lib = std::chrono::steady_clock::now();
            for (int e = 0; e < ndev; e++)
              if (e != d)
                GA_HIP(hipMemcpyAsync(m->d_gather[e] + start + (size_t)d * slot, region + (size_t)d * slot, slot * sizeof(double), hipMemcpyDeviceToDevice, cst));
            bd[BD_LIBRARY_CALLS] += us_since(t_lib);
          } else if (exchange && m->use_rccl && m->one_rank_collective && h == 0) {
            // ONE device has nothing to gather: its records are where they belong.  The library is exercised once, when the handle is created
            // (rccl_self_test); with glim_amd_multi_set_one_rank_collective the no-op call is ALSO made in every evaluation, over the first
            // piece -- round 5 measured what it costs inside a process that has torch's librccl loaded: 0.3 ms of host time per call.
            const auto t_lib = std::chrono::steady_clock::now();
            const ncclResult_t r = rccl().AllGather(region, region, slot, ncclDouble, m->comms[d], cst);
            bd[BD_LIBRARY_CALLS] += us_since(t_lib);
            if (r != ncclSuccess) {
              set_hip_error(hipErrorUnknown, "ncclAllGather");
              return (int)GLIM_AMD_ERR_HIP;
            }
          }
          if (timed && h == P - 1) GA_HIP(hipEventRecord(m->ev[EV_PER_DEVICE * d + 2], cst));
          // copy-out: every device hands ITS OWN rows of the piece to the host over its own PCIe link (round 4: device 0 copied the whole
          // gathered array, 7.6 MB behind the collective; the devices' links work in parallel and need not wait for xGMI)
          const int64_t own = std::max<int64_t>(0, std::min(rows, (hi - lo) - (int64_t)h * m->piece_rows));
          if (own > 0 && !mirrored)  // (otherwise the finalising kernels have stored the rows there themselves)
            GA_HIP(hipMemcpyAsync(m->h_gather + start + (size_t)d * slot, region + (size_t)d * slot, (size_t)own * COMPACT * sizeof(double),
                                  hipMemcpyDeviceToHost, cst));
        }
        if (summed && wait_for_exchange) GA_HIP(hipStreamWaitEvent(cst, m->sum_ev[d], 0));
        if (timed) GA_HIP(hipEventRecord(m->ev[EV_PER_DEVICE * d + 3], cst));
        GA_HIP(hipEventRecord(m->gather_done_ev[(size_t)d], cst));
        m->gather_pending[(size_t)d] = 1;
        return (int)GLIM_AMD_OK;
      };
      rc_d = collective();
      bd[BD_COLLECTIVE] = us_since(t_col);
      if (rc_d != GLIM_AMD_OK) {
        abort_collectives(m);  // the others are inside (or about to enter) their ncclAllGather
        if (last) (void)hipStreamSynchronize(last);
        (void)hipStreamSynchronize(m->cstream[d]);
        return rc_d;
      }
      // ---- the one wait: the kernels (+ the error sum) whose records the finalising blocks have also stored into the host array; the exchange
      //      only when the call was asked to wait for it (it completes behind the call otherwise) ----
      const auto t_wait = std::chrono::steady_clock::now();
      hipError_t e = hipStreamSynchronize(wait_for_exchange ? cst : last);
      if (e == hipSuccess && !wait_for_exchange && sst && sst != last) e = hipStreamSynchronize(sst);
      bd[BD_WAIT] = us_since(t_wait);
      if (e != hipSuccess) {
        set_hip_error(e, "glim_amd_multi_linearize: synchronise");
        abort_collectives(m);
        return (int)GLIM_AMD_ERR_HIP;
      }
      if (timed) {
        (void)hipEventElapsedTime(&m->kernel_ms[d], m->ev[EV_PER_DEVICE * d], m->ev[EV_PER_DEVICE * d + 1]);
        (void)hipGetLastError();
      }
      return (int)GLIM_AMD_OK;
    },
    &post_us, &join_us);
  double* bd0 = &m->breakdown[0];
  bd0[BD_POST] = post_us;
  bd0[BD_JOIN] = join_us;
  if (rc != GLIM_AMD_OK) return rc;
  if (timed) {
    m->gather_timing_pending = true;
    const bool behind = exchange && m->gather_mode == 1;
    if (!behind) harvest_gather_timing(m);  // (every collective stream was synchronised above; behind the call: glim_amd_multi_wait_gather reads them)
    else std::fill(m->gather_ms.begin(), m->gather_ms.end(), 0.f);
  }
  const auto t_scan = std::chrono::steady_clock::now();
  double total = 0.0;
  if (out) {
    for (int d = 0; d < ndev; d++) {
      const int64_t lo = m->bounds[d], hi = m->bounds[d + 1];
      for (int64_t f = lo; f < hi; f++) {
        const double* rec = m->h_gather + (size_t)m->row_of(d, f - lo) * COMPACT;
        total += rec[1];
        glim_amd_expand_compact(rec, T + 12 * f, m->flags[(size_t)f], &out[f]);
      }
    }
  } else if (total_error) {
    for (int d = 0; d < ndev; d++) total += *m->h_total[d];
  }
  if (total_error) *total_error = total;
  bd0[BD_SCAN] = us_since(t_scan);
  bd0[BD_TOTAL] = us_since(t_call);
  return GLIM_AMD_OK;
}

int glim_amd_multi_records(const glim_amd_multi* m, int64_t first, int64_t count, double* compact29) {
  if (!m || !compact29 || first < 0 || count < 0 || first + count > m->nf) return GLIM_AMD_ERR_INVALID;
  for (int64_t f = first; f < first + count; f++) {
    int d = 0;
    while (d + 1 < m->ndev && f >= m->bounds[d + 1]) d++;
    memcpy(compact29 + (size_t)(f - first) * COMPACT, m->h_gather + (size_t)m->row_of(d, f - m->bounds[d]) * COMPACT, COMPACT * sizeof(double));
  }
  return GLIM_AMD_OK;
}

int glim_amd_multi_last_breakdown(const glim_amd_multi* m, int32_t device, double* us, int32_t num_fields) {
  if (!m || !us || device < 0 || device >= m->ndev || num_fields <= 0) return GLIM_AMD_ERR_INVALID;
  for (int i = 0; i < num_fields; i++) us[i] = i < BD_FIELDS ? m->breakdown[(size_t)device * BD_FIELDS + i] : 0.0;
  return GLIM_AMD_OK;
}

int glim_amd_multi_profile(glim_amd_multi* m, const double* T, int iters, float* ms_per_evaluation) {
  if (!m || !T || iters <= 0 || !ms_per_evaluation) return GLIM_AMD_ERR_INVALID;
  double total = 0.0;  // (the cost itself is part of an evaluation: the devices sum their shards' errors)
  for (int i = 0; i < 3; i++) GA_TRY(glim_amd_multi_linearize(m, T, nullptr, &total));
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; i++) GA_TRY(glim_amd_multi_linearize(m, T, nullptr, &total));
  const auto t1 = std::chrono::steady_clock::now();
  *ms_per_evaluation = (float)(std::chrono::duration<double, std::milli>(t1 - t0).count() / iters);
  return GLIM_AMD_OK;
}

}  // extern "C"
