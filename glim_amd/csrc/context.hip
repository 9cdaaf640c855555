// context.hip -- library, error and context entry points of the C ABI (include/glim_amd.h).
#include "internal.hpp"

namespace glim_amd {

static thread_local char g_hip_error[512] = "";

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
  *out = ctx;
  return GLIM_AMD_OK;
}

int glim_amd_ctx_destroy(glim_amd_ctx* ctx) {
  if (!ctx) return GLIM_AMD_OK;
  (void)hipSetDevice(ctx->device);
  for (auto s : ctx->streams) (void)hipStreamSynchronize(s);
  if (ctx->owns_streams)
    for (auto s : ctx->streams) (void)hipStreamDestroy(s);
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
