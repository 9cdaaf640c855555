// scope_sync.hpp -- error exits of host functions that have ENQUEUED work.
// DeviceTemp / pinned staging blocks go back to process-wide pools when their scope ends; a block handed back while kernels that use it are
// still in flight can be given to another context's stream.  Every such function therefore declares a SyncOnExit AFTER its temporaries
// (destroyed first) and dismisses it on the paths that have synchronised themselves; any other exit -- a GA_HIP / GA_TRY early return after
// the first launch -- waits for the stream before the temporaries are released.
#pragma once
#include <hip/hip_runtime.h>

#include "internal.hpp"

namespace glim_amd {
struct SyncOnExit {
  hipStream_t st;
  bool armed = true;
  explicit SyncOnExit(hipStream_t s) : st(s) {}
  SyncOnExit(const SyncOnExit&) = delete;
  SyncOnExit& operator=(const SyncOnExit&) = delete;
  void dismiss() { armed = false; }
  ~SyncOnExit() {
    if (armed) {
      (void)hipStreamSynchronize(st);
      (void)hipGetLastError();
    }
  }
};
// pinned host staging block (pinned_malloc / pinned_free) with scope lifetime
struct PinnedTemp {
  void* p = nullptr;
  PinnedTemp() = default;
  PinnedTemp(const PinnedTemp&) = delete;
  PinnedTemp& operator=(const PinnedTemp&) = delete;
  ~PinnedTemp() {
    if (p) (void)pinned_free(p);
  }
  template <class T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};
}  // namespace glim_amd
