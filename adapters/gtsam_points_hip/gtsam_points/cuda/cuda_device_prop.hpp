// gtsam_points/cuda/cuda_device_prop.hpp on MI355X: cuda_device_names() as GLIM's system-info dump calls it (src/glim/util/debug.cpp:84),
// answered by the C ABI (glim_amd_device_count / glim_amd_device_info).  The version macros debug.cpp prints (:83) describe this library.
#pragma once
#include <string>
#include <vector>

#include <glim_amd.h>

#ifndef GTSAM_POINTS_CUDA_VERSION_MAJOR
#define GTSAM_POINTS_CUDA_VERSION_MAJOR 0  // no CUDA toolkit is involved: HIP / ROCm underneath (glim_amd_version() is the library's own version)
#define GTSAM_POINTS_CUDA_VERSION_MINOR 0
#define GTSAM_POINTS_CUDA_VERSION_PATCH 0
#endif

namespace gtsam_points {

inline std::vector<std::string> cuda_device_names() {
  std::vector<std::string> names;
  const int n = glim_amd_device_count();
  for (int d = 0; d < n; d++) {
    glim_amd_ctx* ctx = nullptr;
    if (glim_amd_ctx_create(d, 1, nullptr, &ctx) != GLIM_AMD_OK) continue;
    char name[256] = {0};
    size_t free_bytes = 0, total_bytes = 0;
    int cus = 0;
    if (glim_amd_device_info(ctx, name, sizeof(name), &free_bytes, &total_bytes, &cus) == GLIM_AMD_OK) names.emplace_back(name);
    glim_amd_ctx_destroy(ctx);
  }
  return names;
}

}  // namespace gtsam_points
