// sync_floor.hip -- what a synchronous call costs on this box before any factor arithmetic: host launch -> kernel -> host-mapped completion
// word -> host spin, for the dispatch shapes of the synchronous VGICP call (glim_amd/csrc/vgicp.hip run_sync), and a RESIDENT kernel that
// takes its requests through a host-mapped mailbox instead of a launch.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/sync_floor.hip -o /tmp/sync_floor && /tmp/sync_floor
// Prints microseconds per round trip (median of 2000 after 200 warm-up):
//   null_1      one block, writes the word                                  = launch + start + PCIe write + wake-up
//   null_513    513 blocks, block 512 writes the word                       = + grid dispatch
//   granule_513 512 blocks publish one tagged 16-B sc1 granule each, block 512 sweeps them (sc1) and then writes the word
//                                                                            = + the in-launch hand-off of vgicp.hip's single-dispatch form
//   two_kernels 512-block kernel, then a 1-block kernel that writes the word = the two-dispatch form's skeleton
//   mailbox_1 / mailbox_512   resident kernel (1 block / 512 blocks + leader): the host writes a sequence number into a host-mapped word, the
//                             leader block polls it (sc0 sc1 loads over PCIe), releases the other blocks through a device word, collects
//                             their granules and writes the completion word: NO launch on the request path
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e = (x);                                                         \
    if (e != hipSuccess) {                                                      \
      printf("%s failed: %s\n", #x, hipGetErrorString(e));                      \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr unsigned int SC1 = 16u, SC1_VOL = 16u | (1u << 31);

__global__ void null_kernel(unsigned int* flag, unsigned int seq, int writer) {
  if ((int)blockIdx.x == writer && threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void granule_kernel(char* rows, unsigned int* flag, unsigned int seq, int nrows) {
  if ((int)blockIdx.x < nrows) {
    if (threadIdx.x == 0) {
      const v4i piece = {1, 2, 3, (int)seq};
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(rows + (size_t)blockIdx.x * 16, 0, 16, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(piece, rs, 0, 0, SC1);
    }
    return;
  }
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(rows, 0, nrows * 16, 0x00020000);
  for (int r = threadIdx.x; r < nrows; r += blockDim.x) {
    for (unsigned int spins = 0; spins < (1u << 22); spins++) {
      const v4i v = __builtin_amdgcn_raw_buffer_load_b128(rs, r * 16, 0, SC1_VOL);
      if (v.w == (int)seq) break;
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Resident kernel.  mailbox (host-mapped): [0] request sequence (host writes), [1] completion (device writes), [2] alive (device writes 0 on exit).
// go: device word the leader releases the workers through.  Exits on request == 0xffffffff or after `idle_limit` empty polls.
__global__ void resident_kernel(volatile unsigned int* mailbox, unsigned int* go, char* rows, int nworkers, unsigned int idle_limit) {
  const bool leader = (int)blockIdx.x == nworkers;
  unsigned int last = 0;
  for (;;) {
    unsigned int req = last;
    if (leader) {
      if (threadIdx.x == 0) {
        unsigned int idle = 0;
        for (;;) {
          req = __hip_atomic_load((unsigned int*)mailbox, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (req != last) break;
          if (++idle > idle_limit) {
            req = 0xffffffffu;
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
        __hip_atomic_store(go, req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      req = __shfl(req, 0);
      __shared__ unsigned int s_req;
      if (threadIdx.x == 0) s_req = req;
      __syncthreads();
      req = s_req;
    } else {
      if (threadIdx.x == 0) {
        for (;;) {
          req = __hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (req != last) break;
          __builtin_amdgcn_s_sleep(2);
        }
      }
      __shared__ unsigned int s_req2;
      if (threadIdx.x == 0) s_req2 = req;
      __syncthreads();
      req = s_req2;
    }
    if (req == 0xffffffffu) {
      if (leader && threadIdx.x == 0) __hip_atomic_store((unsigned int*)mailbox + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    last = req;
    if (!leader) {
      if (threadIdx.x == 0) {
        const v4i piece = {1, 2, 3, (int)req};
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(rows + (size_t)blockIdx.x * 16, 0, 16, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(piece, rs, 0, 0, SC1);
      }
    } else {
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(rows, 0, nworkers * 16, 0x00020000);
      for (int r = threadIdx.x; r < nworkers; r += blockDim.x) {
        for (unsigned int spins = 0; spins < (1u << 22); spins++) {
          const v4i v = __builtin_amdgcn_raw_buffer_load_b128(rs, r * 16, 0, SC1_VOL);
          if (v.w == (int)req) break;
          __builtin_amdgcn_s_sleep(2);
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store((unsigned int*)mailbox + 1, req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
  }
}

static inline bool spin(volatile unsigned int* w, unsigned int v) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned long s = 0;; s++) {
    if (*w == v) return true;
    __builtin_ia32_pause();
    if ((s & 0xffff) == 0xffff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) return false;
  }
}

template <class F>
static double median_us(F&& f, int warm, int iters) {
  for (int i = 0; i < warm; i++) f();
  std::vector<double> t((size_t)iters);
  for (int i = 0; i < iters; i++) {
    const auto a = std::chrono::steady_clock::now();
    f();
    t[(size_t)i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main(int argc, char** argv) {
  // optional argument: spin | yield | blocking -> hipSetDeviceFlags(hipDeviceSchedule...) before anything else touches the device:
  // what hipStreamSynchronize costs under each wait policy (the row null_1_stream_synchronize)
  if (argc > 1) {
    const char c = argv[1][0];
    const unsigned int flag = c == 's' ? hipDeviceScheduleSpin : c == 'y' ? hipDeviceScheduleYield : hipDeviceScheduleBlockingSync;
    const hipError_t e = hipSetDeviceFlags(flag);
    printf("hipSetDeviceFlags(%s): %s\n", argv[1], hipGetErrorString(e));
  }
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned int *h = nullptr, *d = nullptr;
  CK(hipHostMalloc(&h, 256, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void**)&d, h, 0));
  h[0] = h[1] = 0;
  h[2] = 1;
  char* rows = nullptr;
  unsigned int* go = nullptr;
  CK(hipMalloc(&rows, 1 << 20));
  CK(hipMemset(rows, 0, 1 << 20));
  CK(hipMalloc(&go, 64));
  CK(hipMemset(go, 0, 64));
  unsigned int seq = 0;
  bool ok = true;
  const double null1 = median_us([&] { null_kernel<<<1, 256, 0, st>>>(d + 1, ++seq, 0); ok &= spin(h + 1, seq); }, 200, 2000);
  const double null513 = median_us([&] { null_kernel<<<513, 256, 0, st>>>(d + 1, ++seq, 512); ok &= spin(h + 1, seq); }, 200, 2000);
  const double gran = median_us([&] { granule_kernel<<<513, 256, 0, st>>>(rows, d + 1, ++seq, 512); ok &= spin(h + 1, seq); }, 200, 2000);
  const double two = median_us([&] { null_kernel<<<512, 256, 0, st>>>(d + 3, ++seq, 0); null_kernel<<<1, 256, 0, st>>>(d + 1, seq, 0); ok &= spin(h + 1, seq); }, 200, 2000);
  const double syncd = median_us([&] { null_kernel<<<1, 256, 0, st>>>(d + 1, ++seq, 0); (void)hipStreamSynchronize(st); }, 200, 2000);
  printf("null_1 %.2f us  null_513 %.2f us  granule_513 %.2f us  two_kernels %.2f us  null_1_stream_synchronize %.2f us  (ok=%d)\n", null1, null513, gran, two, syncd, (int)ok);
  // completion WITHOUT a word written by the kernel itself (i.e. ordered after the kernel's end, as hipStreamSynchronize is):
  //   write_value   null kernel (touches nothing) + hipStreamWriteValue32 of the sequence number into the host-mapped word, host spins on it
  //   event_query   null kernel + hipEventRecord, host spins on hipEventQuery
  {
    unsigned int* sink = nullptr;
    CK(hipMalloc(&sink, 64));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t we = hipSuccess;
    const double wv = median_us([&] { null_kernel<<<1, 256, 0, st>>>(sink, ++seq, 0); we = hipStreamWriteValue32(st, d + 4, seq, 0); ok &= (we == hipSuccess) && spin(h + 4, seq); }, 200, 2000);
    const double eq = median_us([&] { null_kernel<<<1, 256, 0, st>>>(sink, ++seq, 0); (void)hipEventRecord(ev, st); while (hipEventQuery(ev) == hipErrorNotReady) __builtin_ia32_pause(); }, 200, 2000);
    printf("null_1_write_value32 %.2f us (%s)  null_1_event_query %.2f us  (ok=%d)\n", wv, hipGetErrorString(we), eq, (int)ok);
    (void)hipGetLastError();
  }
  CK(hipStreamSynchronize(st));
  for (int workers : {0, 255, 512}) {
    h[0] = 0; h[1] = 0; h[2] = 1;
    CK(hipMemset(go, 0, 64));
    CK(hipMemset(rows, 0, 1 << 20));
    // idle limit: ~2^22 polls of ~1-2 us = a few seconds at most; the run below ends it with the exit request long before
    resident_kernel<<<workers + 1, 256, 0, st>>>((volatile unsigned int*)d, go, rows, workers, 1u << 22);
    unsigned int s = 0;
    bool ok2 = true;
    const double mb = median_us([&] { h[0] = ++s; ok2 &= spin(h + 1, s); }, 200, 2000);
    h[0] = 0xffffffffu;
    CK(hipStreamSynchronize(st));
    printf("mailbox with %d worker blocks: %.2f us per request (ok=%d, alive word %u)\n", workers, mb, (int)ok2, h[2]);
  }
  return 0;
}
