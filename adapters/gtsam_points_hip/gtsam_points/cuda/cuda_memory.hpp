// gtsam_points/cuda/cuda_memory.hpp on MI355X: cuda_mem_get_info(&free, &total) as GLIM's memory monitor polls it every five seconds
// (src/glim/viewer/memory_monitor.cpp:39), answered by the C ABI (glim_amd_device_info on device 0, the device GLIM's modules use).
#pragma once
#include <cstddef>

#include <glim_amd.h>

namespace gtsam_points {

inline void cuda_mem_get_info(size_t* free_bytes, size_t* total_bytes) {
  static glim_amd_ctx* ctx = [] {
    glim_amd_ctx* c = nullptr;
    return glim_amd_ctx_create(0, 1, nullptr, &c) == GLIM_AMD_OK ? c : nullptr;
  }();
  size_t f = 0, t = 1;
  if (ctx) {
    char name[8];
    int cus = 0;
    if (glim_amd_device_info(ctx, name, sizeof(name), &f, &t, &cus) != GLIM_AMD_OK) f = 0, t = 1;
  }
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t ? t : 1;
}

}  // namespace gtsam_points
