// bar_request.hip -- can the HOST post a request straight into DEVICE memory (large BAR) instead of leaving it in host memory for the resident
// kernel's leader to fetch with PCIe READS?  (the request side of the synchronous call is a PCIe read round trip, ~1.2 us, polled.)
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/bar_request.hip -o /tmp/bar_request && /tmp/bar_request
// A 1-block resident kernel polls a request word and answers by writing the sequence number into a host-mapped completion word:
//   host_word      the request word lives in pinned host memory (what vgicp.hip's resident session does): the kernel polls over PCIe
//   device_word    the request word lives in fine-grained DEVICE memory (hipExtMallocWithFlags, hipDeviceMallocFinegrained) and the host stores
//                  into it through the same pointer -- works only where the device memory is host-visible (large BAR); the program reports and
//                  skips the row when the allocation or the first host store is refused
// Prints the median round trip of 2000 requests after 200 warm-ups.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <vector>

#define CK(x)                                              \
  do {                                                     \
    hipError_t e = (x);                                    \
    if (e != hipSuccess) {                                 \
      printf("%s failed: %s\n", #x, hipGetErrorString(e)); \
      return 1;                                            \
    }                                                      \
  } while (0)

__global__ void poll_kernel(const unsigned int* request, unsigned int* completion, unsigned int idle_limit) {
  unsigned int last = 0;
  for (unsigned int idle = 0; idle < idle_limit; idle++) {
    const unsigned int r = __hip_atomic_load(request, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (r == 0xffffffffu) return;
    if (r != last) {
      last = r;
      idle = 0;
      __hip_atomic_store(completion, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

// The same responder with the poll spread over the four wavefronts of a 256-thread block: every wavefront keeps ONE read of the request word in
// flight, started a quarter of a round trip after its neighbour's, and the first to see a new value answers (LDS compare-and-swap).  The expected
// wait of a request for the next poll to START drops from half a read round trip to an eighth.
__global__ void poll_kernel_staggered(const unsigned int* request, unsigned int* completion, unsigned int idle_limit, int stagger_sleep) {
  __shared__ unsigned int s_last;
  if (threadIdx.x == 0) s_last = 0u;
  __syncthreads();
  if ((threadIdx.x & 63) != 0) return;
  const int wave = (int)threadIdx.x >> 6;
  for (int k = 0; k < wave * stagger_sleep; k++) __builtin_amdgcn_s_sleep(1);  // (64 cycles each: the s_sleep argument has to be a constant)
  for (unsigned int idle = 0; idle < idle_limit; idle++) {
    const unsigned int r = __hip_atomic_load(request, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (r == 0xffffffffu) return;
    const unsigned int last = __hip_atomic_load(&s_last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (r != last) {
      unsigned int expected = last;
      if (__hip_atomic_compare_exchange_strong(&s_last, &expected, r, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
        idle = 0;
        __hip_atomic_store(completion, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

static sigjmp_buf g_jmp;
static void on_segv(int) { siglongjmp(g_jmp, 1); }

static double run(volatile unsigned int* request_host_view, const unsigned int* request_dev, volatile unsigned int* h_done, unsigned int* d_done, hipStream_t st, int stagger = -1) {
  *request_host_view = 0;
  *h_done = 0;
  if (stagger < 0) poll_kernel<<<1, 64, 0, st>>>(request_dev, d_done, 1u << 22);
  else poll_kernel_staggered<<<1, 256, 0, st>>>(request_dev, d_done, 1u << 22, stagger);
  std::vector<double> us;
  unsigned int s = 0;
  for (int i = 0; i < 2200; i++) {
    const auto t0 = std::chrono::steady_clock::now();
    *request_host_view = ++s;
    __builtin_ia32_sfence();
    unsigned long spins = 0;
    while (*h_done != s && ++spins < (1ul << 28)) __builtin_ia32_pause();
    const double dt = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (i >= 200) us.push_back(dt);
  }
  *request_host_view = 0xffffffffu;
  __builtin_ia32_sfence();
  (void)hipStreamSynchronize(st);
  std::sort(us.begin(), us.end());
  return us[us.size() / 2];
}

int main() {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned int *h = nullptr, *d = nullptr;
  CK(hipHostMalloc(&h, 256, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void**)&d, h, 0));
  printf("host_word   (request polled over PCIe):      %.2f us per request\n", run(h, d, h + 16, d + 16, st));
  for (int stagger : {0, 4, 8, 12}) printf("host_word, four staggered wavefronts (s_sleep %2d between their first polls): %.2f us per request\n", stagger, run(h, d, h + 16, d + 16, st, stagger));
  printf("host_word   (again):                         %.2f us per request\n", run(h, d, h + 16, d + 16, st));
  unsigned int* fine = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&fine, 256, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    printf("device_word: hipExtMallocWithFlags(hipDeviceMallocFinegrained) refused: %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
    return 0;
  }
  CK(hipMemset(fine, 0, 256));
  signal(SIGSEGV, on_segv);
  signal(SIGBUS, on_segv);
  if (sigsetjmp(g_jmp, 1)) {
    printf("device_word: the host cannot store into device memory through this pointer (no large BAR mapping): not available here\n");
    return 0;
  }
  *(volatile unsigned int*)fine = 0;  // faults where the memory is not host-visible
  printf("device_word (request stored through the BAR): %.2f us per request\n", run(fine, fine, h + 16, d + 16, st));
  return 0;
}
