// cloud.hip -- PointCloudGPU equivalent: upload + FP64-AoS -> FP32-SoA pack on the device (kernel K7).
// Replaces gtsam_points::PointCloudGPU::clone as called at src/glim/odometry/odometry_estimation_gpu.cpp:96,
// src/glim/mapping/sub_mapping.cpp:168,393 and src/glim/mapping/global_mapping.cpp:253,260,743.
#include <string>

#include "internal.hpp"
#include "pull.hpp"
#include "scope_sync.hpp"

using namespace glim_amd;

namespace {

// One thread per point: reads the reference's host layouts (Vector4d / column-major Matrix4d) from a raw device
// staging copy and writes the SoA the hot kernels stream: float4 xyz1, float4 (c00 c01 c02 c11), float2 (c12 c22), float4 n.
__global__ __launch_bounds__(256) void pack_f64_kernel(int64_t n, const double* __restrict__ points4, const double* __restrict__ covs16,
                                                       const double* __restrict__ normals4, float4* __restrict__ pts,
                                                       float4* __restrict__ covA, float2* __restrict__ covB, float4* __restrict__ nrm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double4 p = reinterpret_cast<const double4*>(points4)[i];
  pts[i] = make_float4((float)p.x, (float)p.y, (float)p.z, 1.0f);
  if (covs16) {
    const double* c = covs16 + 16 * i;  // column-major 4x4: (r,c) at c*4 + r
    covA[i] = make_float4((float)c[0], (float)c[4], (float)c[8], (float)c[5]);
    covB[i] = make_float2((float)c[9], (float)c[10]);
  }
  if (normals4) {
    const double4 v = reinterpret_cast<const double4*>(normals4)[i];
    nrm[i] = make_float4((float)v.x, (float)v.y, (float)v.z, 0.0f);
  }
}

__global__ __launch_bounds__(256) void pack_f32_kernel(int64_t n, const float* __restrict__ xyz, const float* __restrict__ cov33,
                                                       const float* __restrict__ normals3, float4* __restrict__ pts,
                                                       float4* __restrict__ covA, float2* __restrict__ covB, float4* __restrict__ nrm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  pts[i] = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 1.0f);
  if (cov33) {
    const float* c = cov33 + 9 * i;
    covA[i] = make_float4(c[0], c[1], c[2], c[4]);
    covB[i] = make_float2(c[5], c[8]);
  }
  if (normals3) nrm[i] = make_float4(normals3[3 * i], normals3[3 * i + 1], normals3[3 * i + 2], 0.0f);
}

__global__ __launch_bounds__(256) void unpack_kernel(int64_t n, const float4* __restrict__ pts, const float4* __restrict__ covA,
                                                     const float2* __restrict__ covB, const float4* __restrict__ nrm, float* __restrict__ xyz,
                                                     float* __restrict__ cov33, float* __restrict__ normals3) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (xyz) {
    const float4 p = pts[i];
    xyz[3 * i] = p.x;
    xyz[3 * i + 1] = p.y;
    xyz[3 * i + 2] = p.z;
  }
  if (cov33 && covA) {
    const float4 a = covA[i];
    const float2 b = covB[i];
    float* c = cov33 + 9 * i;
    c[0] = a.x; c[1] = a.y; c[2] = a.z;
    c[3] = a.y; c[4] = a.w; c[5] = b.x;
    c[6] = a.z; c[7] = b.x; c[8] = b.y;
  }
  if (normals3 && nrm) {
    const float4 v = nrm[i];
    normals3[3 * i] = v.x;
    normals3[3 * i + 1] = v.y;
    normals3[3 * i + 2] = v.z;
  }
}

__global__ __launch_bounds__(256) void plane_form_kernel(int64_t n, const float4* __restrict__ covA, const float2* __restrict__ covB,
                                                         const float4* __restrict__ nrm, unsigned int* __restrict__ violations) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool bad = false;
  if (i < n) {
    const float4 a = covA[i];
    const float2 b = covB[i];
    const float4 v = nrm[i];
    bad = off_plane_form(a.x, a.y, a.z, a.w, b.x, b.y, v.x, v.y, v.z);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicAdd(violations, 1u);
}

// Small clouds (the odometry front end clones one 10 000-point frame per scan, odometry_estimation_gpu.cpp:96).  The general path -- three
// pageable host-to-device copies of the reference layouts (192 B per point), the pack kernel, the plane-form kernel and their two
// synchronisations -- costs 140 us there, most of it fixed.  Here the host converts straight into the FP32 device layout (56 B per point, the
// same round-to-nearest casts as pack_f64_kernel) in a pinned, device-mapped staging block, and ONE kernel pulls the block over PCIe into the
// cloud's arrays and evaluates the plane-form predicate on the way (violations land in a word of the same block): one launch, one
// synchronise, 58 us (odometry_frame bench).  Above HOST_PACK_MAX_POINTS the runtime's pageable copy path (43 GB/s measured at 25 MB) beats a single host thread.
constexpr int64_t HOST_PACK_MAX_POINTS = 32768;
std::atomic<unsigned int> g_pull_gate_seq{1u};
std::atomic<bool> g_pull_gate_broken{false};  // a gated pull kernel once gave up waiting for this process' host side: no more gating
__global__ __launch_bounds__(256) void unstage_kernel(const PullArgs pa) {
  __shared__ int s_ok;
  if (!pull_wait(pa, &s_ok)) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool bad = false;
  if (i < pa.n) {
    float4 p, a, v;
    float2 b;
    bad = pull_point(pa, i, p, a, b, v);
  }
  pull_report(pa, bad);
}

// host half: the reference layouts (Vector4d, column-major Matrix4d) -> the FP32 sections of the staging block
// points [lo, hi) of every section
void host_pack_f64(int64_t lo, int64_t hi, const double* points4, const double* covs16, const double* normals4, float* pts, float* covA, float* covB, float* nrm) {
  for (int64_t i = lo; i < hi; i++) {
    const double* p = points4 + 4 * i;
    pts[4 * i + 0] = (float)p[0];
    pts[4 * i + 1] = (float)p[1];
    pts[4 * i + 2] = (float)p[2];
    pts[4 * i + 3] = 1.0f;
  }
  if (covs16)
    for (int64_t i = lo; i < hi; i++) {
      const double* c = covs16 + 16 * i;  // column-major 4x4: (r,c) at c*4 + r
      covA[4 * i + 0] = (float)c[0];
      covA[4 * i + 1] = (float)c[4];
      covA[4 * i + 2] = (float)c[8];
      covA[4 * i + 3] = (float)c[5];
      covB[2 * i + 0] = (float)c[9];
      covB[2 * i + 1] = (float)c[10];
    }
  if (normals4)
    for (int64_t i = lo; i < hi; i++) {
      const double* v = normals4 + 4 * i;
      nrm[4 * i + 0] = (float)v[0];
      nrm[4 * i + 1] = (float)v[1];
      nrm[4 * i + 2] = (float)v[2];
      nrm[4 * i + 3] = 0.0f;
    }
}

void drop_general_streams(glim_amd_cloud* c) {
  if (c->gs0) (void)pool_free(c->gs0);
  if (c->gs1) (void)pool_free(c->gs1);
  if (c->gs2) (void)pool_free(c->gs2);
  if (c->gsn) (void)pool_free(c->gsn);
  if (c->gbox) (void)pool_free(c->gbox);
  c->gs0 = c->gs1 = nullptr;
  c->gs2 = nullptr;
  c->gsn = nullptr;
  c->gbox = nullptr;
}
void drop_plane_streams(glim_amd_cloud* c) {
  if (c->pn4) (void)pool_free(c->pn4);
  if (c->n2) (void)pool_free(c->n2);
  c->pn4 = nullptr;
  c->n2 = nullptr;
}

// upload of a small cloud through the pinned staging block; GLIM_AMD_ERR_UNSUPPORTED: no device view of pinned memory here (caller takes the general path)
}  // namespace
namespace glim_amd {
thread_local double g_frame_stage_us[FRAME_STAGES] = {0, 0, 0, 0, 0, 0, 0};
thread_local double g_frame_t0_us = 0.0;
// The small upload in two halves (glim_amd_frame_create enqueues the frame's voxel maps between them and synchronises ONCE): enqueue = host
// conversion into the pinned staging block + the pull kernel on ctx->stream(); finish (after the caller has synchronised that stream) = the
// plane-form verdict, the unused stream copy back to the pool, the staging block back to its pool.
int cloud_small_prepare(glim_amd_ctx* ctx, glim_amd_cloud* c, const double* points4, const double* covs16, const double* normals4, SmallUpload* up) {
  const int64_t n = c->n;
  float* stage = nullptr;
  if (pinned_malloc(&stage, (size_t)n * 14 * sizeof(float) + 64) != hipSuccess) {
    (void)hipGetLastError();
    return GLIM_AMD_ERR_UNSUPPORTED;
  }
  float* dev = nullptr;
  if (hipHostGetDevicePointer(reinterpret_cast<void**>(&dev), stage, 0) != hipSuccess) {
    (void)hipGetLastError();
    (void)pinned_free(stage);
    return GLIM_AMD_ERR_UNSUPPORTED;
  }
  // sections: pts [0, 4n), covA [4n, 8n), nrm [8n, 12n), covB [12n, 14n) -- the float4 sections first, so that every section is 16-byte
  // aligned -- then the tail words: [0] violations of the plane-form test, [1] PULL_GAVE_UP, [4 .. 7] the gate words of the pieces
  volatile unsigned int* tail = reinterpret_cast<unsigned int*>(stage + 14 * n);
  tail[0] = 0u;
  tail[1] = 0u;
  // factor streams written by the same kernel (pull_point); an allocation that fails only means they are built on first use instead
  if (covs16) {
    const size_t nn = (size_t)n;
    bool ok = pool_malloc(&c->gs0, nn * sizeof(float4)) == hipSuccess && pool_malloc(&c->gs1, nn * sizeof(float4)) == hipSuccess &&
              pool_malloc(&c->gs2, nn * sizeof(float)) == hipSuccess && (!normals4 || pool_malloc(&c->gsn, nn * sizeof(float4)) == hipSuccess);
    if (ok && normals4) ok = pool_malloc(&c->pn4, nn * sizeof(float4)) == hipSuccess && pool_malloc(&c->n2, nn * sizeof(float2)) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      drop_general_streams(c);
      drop_plane_streams(c);
    }
  }
  frame_stamp(1);
  // Gated form (default): the pull kernel goes out FIRST and the host converts while the launch travels; the cloud is cut into up to four pieces
  // and the blocks of a piece start pulling the moment the host has published it.  (Cutting the cloud into four LAUNCHES instead was measured
  // and rejected, +23 us: profiles/r05/probe/upload_in_pieces_rejected.json.)
  const bool gated = ctx->diag.pull_gated && !g_pull_gate_broken.load(std::memory_order_relaxed);
  unsigned int seq = 0u;
  while (gated && (seq == 0u || seq == PULL_GAVE_UP)) seq = g_pull_gate_seq.fetch_add(1u, std::memory_order_relaxed);
  const int64_t pieces_wanted = n >= 4096 ? 4 : 1, piece_len = ((n + pieces_wanted - 1) / pieces_wanted + 255) / 256 * 256;
  PullArgs& pa = up->args;
  pa = PullArgs();
  pa.n = (int)n;
  pa.s_pts = reinterpret_cast<const float4*>(dev);
  pa.s_covA = covs16 ? reinterpret_cast<const float4*>(dev + 4 * n) : nullptr;
  pa.s_covB = reinterpret_cast<const float2*>(dev + 12 * n);
  pa.s_nrm = normals4 ? reinterpret_cast<const float4*>(dev + 8 * n) : nullptr;
  pa.pts = c->pts;
  pa.covA = c->covA;
  pa.covB = c->covB;
  pa.nrm = c->normals;
  pa.host_tail = reinterpret_cast<unsigned int*>(dev + 14 * n);
  pa.pn4 = c->pn4;
  pa.n2 = c->n2;
  pa.gs0 = c->gs0;
  pa.gs1 = c->gs1;
  pa.gs2 = c->gs2;
  pa.gsn = c->gsn;
  pa.gate = gated ? reinterpret_cast<const unsigned int*>(dev + 14 * n) + 4 : nullptr;
  pa.gate_seq = seq;
  pa.piece_len = (int)piece_len;
  up->stage = stage;
  up->violations = tail;
  up->maybe_plane = covs16 && normals4;
  up->gated = gated;
  up->packed = false;
  up->points4 = points4;
  up->covs16 = covs16;
  up->normals4 = normals4;
  return GLIM_AMD_OK;
}

// host conversion into the staging block; gated form: piece by piece, every piece published through its gate word the moment it is complete
void cloud_small_pack(SmallUpload* up) {
  if (up->packed || !up->stage) return;
  const int64_t n = up->args.n, piece_len = up->gated ? up->args.piece_len : std::max<int64_t>(n, 1);
  float* stage = up->stage;
  int piece = 0;
  for (int64_t lo = 0; lo < n; lo += piece_len, piece++) {
    host_pack_f64(lo, std::min(n, lo + piece_len), up->points4, up->covs16, up->normals4, stage, stage + 4 * n, stage + 12 * n, stage + 8 * n);
    if (up->gated) {
      std::atomic_thread_fence(std::memory_order_release);  // (x86: stores stay in program order; the fence keeps the compiler from moving them)
      up->violations[4 + piece] = up->args.gate_seq;
    }
  }
  std::atomic_thread_fence(std::memory_order_release);
  up->packed = true;
  frame_stamp(3);
}

int cloud_small_launch(SmallUpload* up, hipStream_t st) {
  unstage_kernel<<<(up->args.n + 255) / 256, 256, 0, st>>>(up->args);
  const hipError_t e = hipGetLastError();
  frame_stamp(2);
  if (e != hipSuccess) {
    set_hip_error(e, "cloud_create small upload");
    return GLIM_AMD_ERR_HIP;
  }
  return GLIM_AMD_OK;
}

int cloud_small_enqueue(glim_amd_ctx* ctx, glim_amd_cloud* c, const double* points4, const double* covs16, const double* normals4, SmallUpload* up) {
  GA_TRY(cloud_small_prepare(ctx, c, points4, covs16, normals4, up));
  if (!up->gated) cloud_small_pack(up);
  const int rc = cloud_small_launch(up, ctx->stream());
  if (rc != GLIM_AMD_OK) {  // (nothing waits for the gate then)
    (void)hipStreamSynchronize(ctx->stream());
    (void)pinned_free(up->stage);
    up->stage = nullptr;
    return rc;
  }
  cloud_small_pack(up);
  return GLIM_AMD_OK;
}
int cloud_small_finish(glim_amd_cloud* c, SmallUpload* up) {
  const bool gave_up = up->violations[1] == PULL_GAVE_UP;
  const bool plane = !gave_up && up->maybe_plane && up->violations[0] == 0u;
  (void)pinned_free(up->stage);
  up->stage = nullptr;
  c->plane_form = plane;
  if (plane && c->pn4) drop_general_streams(c);  // the factor kernel reads the form the cloud has; the other copy goes back to the pool
  else drop_plane_streams(c);
  if (gave_up) {  // (the host was held up for seconds between the launch and its conversion: the cloud holds nothing; callers repeat the upload ungated)
    g_pull_gate_broken.store(true);
    set_hip_error(hipErrorUnknown, "small-cloud upload: the gated pull kernel gave up waiting for the host; gating is off from now on");
    return GLIM_AMD_ERR_UNSUPPORTED;
  }
  return GLIM_AMD_OK;
}
}  // namespace glim_amd
namespace {
int create_small_f64(glim_amd_ctx* ctx, glim_amd_cloud* c, const double* points4, const double* covs16, const double* normals4) {
  SmallUpload up;
  const int rc = cloud_small_enqueue(ctx, c, points4, covs16, normals4, &up);
  if (rc != GLIM_AMD_OK) return rc;
  const hipError_t e = hipStreamSynchronize(ctx->stream());
  if (e != hipSuccess) {
    (void)pinned_free(up.stage);
    set_hip_error(e, "cloud_create small upload");
    return GLIM_AMD_ERR_HIP;
  }
  if (cloud_small_finish(c, &up) == GLIM_AMD_OK) return GLIM_AMD_OK;
  // the gated pull gave up (gating is off now): once more, conversion first
  drop_general_streams(c);
  drop_plane_streams(c);
  GA_TRY(cloud_small_enqueue(ctx, c, points4, covs16, normals4, &up));
  GA_HIP(hipStreamSynchronize(ctx->stream()));
  return cloud_small_finish(c, &up);
}

// factor streams in the Hilbert order of the cloud (rank == null: arrival order)
__global__ __launch_bounds__(256) void plane_stream_kernel(int n, const float4* __restrict__ pts, const float4* __restrict__ nrm,
                                                           const unsigned int* __restrict__ rank, float4* __restrict__ pn4, float2* __restrict__ n2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned int o = rank ? rank[i] : (unsigned int)i;
  const float4 p = pts[i], v = nrm[i];
  pn4[o] = make_float4(p.x, p.y, p.z, v.x);
  n2[o] = make_float2(v.y, v.z);
}
__global__ __launch_bounds__(256) void general_stream_kernel(int n, const float4* __restrict__ pts, const float4* __restrict__ covA,
                                                             const float2* __restrict__ covB, const float4* __restrict__ nrm,
                                                             const unsigned int* __restrict__ rank, float4* __restrict__ gs0, float4* __restrict__ gs1,
                                                             float* __restrict__ gs2, float4* __restrict__ gsn) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned int o = rank ? rank[i] : (unsigned int)i;
  const float4 p = pts[i], a = covA[i];
  const float2 b = covB[i];
  gs0[o] = make_float4(p.x, p.y, p.z, a.x);
  gs1[o] = make_float4(a.y, a.z, a.w, b.x);
  gs2[o] = b.y;
  if (gsn) gsn[o] = nrm[i];
}

int alloc_cloud(glim_amd_ctx* ctx, int64_t n, bool covs, bool normals, glim_amd_cloud** out) {
  if (n > (int64_t)(1u << 28)) return GLIM_AMD_ERR_INVALID;  // kernels address points with 32-bit byte offsets (16 B per point)
  glim_amd_cloud* c = new glim_amd_cloud();
  c->ctx = ctx;
  c->n = n;
  const size_t nn = (size_t)(n > 0 ? n : 1);
  hipError_t e = pool_malloc(&c->pts, nn * sizeof(float4));
  if (e == hipSuccess && covs) e = pool_malloc(&c->covA, nn * sizeof(float4));
  if (e == hipSuccess && covs) e = pool_malloc(&c->covB, nn * sizeof(float2));
  if (e == hipSuccess && normals) e = pool_malloc(&c->normals, nn * sizeof(float4));
  if (e != hipSuccess) {
    set_hip_error(e, "pool_malloc(cloud)");
    glim_amd_cloud_destroy(c);
    return e == hipErrorOutOfMemory ? GLIM_AMD_ERR_NOMEM : GLIM_AMD_ERR_HIP;
  }
  c->has_covs = covs;
  c->has_normals = normals;
  *out = c;
  return GLIM_AMD_OK;
}

}  // namespace

namespace glim_amd {

int alloc_cloud_for_frame(glim_amd_ctx* ctx, int64_t n, bool covs, bool normals, glim_amd_cloud** out) { return alloc_cloud(ctx, n, covs, normals, out); }

int detect_plane_form(glim_amd_cloud* c, hipStream_t st) {
  c->plane_form = false;
  if (c->n <= 0 || !c->covA || !c->covB || !c->normals || !c->has_covs || !c->has_normals) return GLIM_AMD_OK;
  unsigned int* d_bad = nullptr;
  GA_HIP(pool_malloc(&d_bad, sizeof(unsigned int)));
  unsigned int bad = 1;
  hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(unsigned int), st);
  if (e == hipSuccess) {
    plane_form_kernel<<<(int)((c->n + 255) / 256), 256, 0, st>>>(c->n, c->covA, c->covB, c->normals, d_bad);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)pool_free(d_bad);
  if (e != hipSuccess) {
    set_hip_error(e, "detect_plane_form");
    return GLIM_AMD_ERR_HIP;
  }
  c->plane_form = (bad == 0);
  return GLIM_AMD_OK;
}

// min | max of the xyz of every 64 consecutive points of the general stream: one wavefront per chunk (xor-butterfly over the lanes; min / max of
// floats are exact, so every point of the chunk lies inside its box bit for bit)
__global__ __launch_bounds__(256) void chunk_box_kernel(int n, const float4* __restrict__ gs0, float* __restrict__ box) {
  const int chunk = (int)((blockIdx.x * 256u + threadIdx.x) >> 6), lane = (int)(threadIdx.x & 63);
  if (chunk * 64 >= n) return;  // (wave-uniform)
  const float4 p = gs0[min(chunk * 64 + lane, n - 1)];  // a lane past the end repeats the last point: inside the box of the others
  float lo[3] = {p.x, p.y, p.z}, hi[3] = {p.x, p.y, p.z};
#pragma unroll
  for (int m = 1; m < 64; m <<= 1)
#pragma unroll
    for (int a = 0; a < 3; a++) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], m, 64));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], m, 64));
    }
  const float v = lane == 0 ? lo[0] : lane == 1 ? lo[1] : lane == 2 ? lo[2] : lane == 3 ? hi[0] : lane == 4 ? hi[1] : hi[2];
  if (lane < 6) box[6 * (size_t)chunk + lane] = v;
}

int ensure_chunk_boxes(glim_amd_cloud* c, hipStream_t st) {
  if (c->n <= 0 || !c->gs0) return GLIM_AMD_ERR_STATE;
  std::lock_guard<std::mutex> build_lock(c->build_mu);
  if (c->gbox) return GLIM_AMD_OK;
  const int n = (int)c->n, chunks = (n + 63) / 64;
  float* box = nullptr;
  GA_HIP(pool_malloc(&box, (size_t)chunks * 6 * sizeof(float)));
  chunk_box_kernel<<<(chunks * 64 + 255) / 256, 256, 0, st>>>(n, c->gs0, box);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);  // once per cloud: complete before anybody sees the pointer (any stream, any context)
  if (e != hipSuccess) {
    (void)pool_free(box);
    set_hip_error(e, "ensure_chunk_boxes");
    return GLIM_AMD_ERR_HIP;
  }
  c->gbox = box;
  return GLIM_AMD_OK;
}

int ensure_factor_streams(glim_amd_cloud* c, glim_amd_ctx* held, hipStream_t st) {
  if (!c->has_covs || c->n <= 0) return GLIM_AMD_OK;
  // a cloud may be reached from factor sets of several contexts / streams at once (GLIM's modules share frames): one builder, and the streams
  // are COMPLETE before anybody sees their pointers
  std::lock_guard<std::mutex> build_lock(c->build_mu);
  const bool want_plane = c->plane_form && c->normals;
  if (want_plane ? (c->pn4 && c->n2) : (c->gs0 != nullptr)) return GLIM_AMD_OK;
  const int n = (int)c->n;
  GA_TRY(cloud_curve_rank(c, held, st));
  if (want_plane) {
    if (!c->pn4) GA_HIP(pool_malloc(&c->pn4, (size_t)n * sizeof(float4)));
    if (!c->n2) GA_HIP(pool_malloc(&c->n2, (size_t)n * sizeof(float2)));
    plane_stream_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, c->pts, c->normals, c->curve_rank, c->pn4, c->n2);
  } else {
    GA_HIP(pool_malloc(&c->gs0, (size_t)n * sizeof(float4)));
    GA_HIP(pool_malloc(&c->gs1, (size_t)n * sizeof(float4)));
    GA_HIP(pool_malloc(&c->gs2, (size_t)n * sizeof(float)));
    if (c->has_normals && c->normals) GA_HIP(pool_malloc(&c->gsn, (size_t)n * sizeof(float4)));
    general_stream_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, c->pts, c->covA, c->covB, c->normals, c->curve_rank, c->gs0, c->gs1, c->gs2, c->gsn);
  }
  GA_HIP(hipGetLastError());
  // once per cloud (~10 us): the next reader may sit on another stream, of this context or of another one
  GA_HIP(hipStreamSynchronize(st));
  return GLIM_AMD_OK;
}

}  // namespace glim_amd

extern "C" {

int glim_amd_cloud_create(glim_amd_ctx* ctx, int64_t n, const double* points4, const double* covs16, const double* normals4,
                          glim_amd_cloud** out) {
  if (!ctx || !out || n < 0 || (n > 0 && !points4)) return GLIM_AMD_ERR_INVALID;
  *out = nullptr;
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  glim_amd_cloud* c = nullptr;
  GA_TRY(alloc_cloud(ctx, n, covs16 != nullptr, normals4 != nullptr, &c));
  if (n > 0 && n <= HOST_PACK_MAX_POINTS && ctx->diag.host_pack) {
    const int rc = create_small_f64(ctx, c, points4, covs16, normals4);
    if (rc == GLIM_AMD_OK) {
      *out = c;
      return GLIM_AMD_OK;
    }
    if (rc != GLIM_AMD_ERR_UNSUPPORTED) {
      glim_amd_cloud_destroy(c);
      return rc;
    }
  }
  if (n > 0) {
    hipStream_t s = ctx->stream();
    DeviceTemp dp, dc, dn;
    hipError_t e = pool_malloc(&dp.p, (size_t)n * 4 * sizeof(double));
    if (e == hipSuccess && covs16) e = pool_malloc(&dc.p, (size_t)n * 16 * sizeof(double));
    if (e == hipSuccess && normals4) e = pool_malloc(&dn.p, (size_t)n * 4 * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(dp.p, points4, (size_t)n * 4 * sizeof(double), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && covs16) e = hipMemcpyAsync(dc.p, covs16, (size_t)n * 16 * sizeof(double), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && normals4) e = hipMemcpyAsync(dn.p, normals4, (size_t)n * 4 * sizeof(double), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
      const int blocks = (int)((n + 255) / 256);
      pack_f64_kernel<<<blocks, 256, 0, s>>>(n, (const double*)dp.p, (const double*)dc.p, (const double*)dn.p, c->pts, c->covA, c->covB,
                                             c->normals);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
      set_hip_error(e, "cloud_create upload/pack");
      (void)hipStreamSynchronize(s);  // copies that did get enqueued must not outlive their staging blocks or the cloud's arrays
      glim_amd_cloud_destroy(c);
      return GLIM_AMD_ERR_HIP;
    }
    if (covs16 && normals4) {
      const int rc = detect_plane_form(c, s);
      if (rc != GLIM_AMD_OK) {
        glim_amd_cloud_destroy(c);
        return rc;
      }
    }
  }
  *out = c;
  return GLIM_AMD_OK;
}

int glim_amd_cloud_create_f32(glim_amd_ctx* ctx, int64_t n, const float* xyz, const float* cov33, const float* normals3,
                              glim_amd_cloud** out) {
  if (!ctx || !out || n < 0 || (n > 0 && !xyz)) return GLIM_AMD_ERR_INVALID;
  *out = nullptr;
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  glim_amd_cloud* c = nullptr;
  GA_TRY(alloc_cloud(ctx, n, cov33 != nullptr, normals3 != nullptr, &c));
  if (n > 0) {
    hipStream_t s = ctx->stream();
    DeviceTemp dp, dc, dn;
    hipError_t e = pool_malloc(&dp.p, (size_t)n * 3 * sizeof(float));
    if (e == hipSuccess && cov33) e = pool_malloc(&dc.p, (size_t)n * 9 * sizeof(float));
    if (e == hipSuccess && normals3) e = pool_malloc(&dn.p, (size_t)n * 3 * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(dp.p, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && cov33) e = hipMemcpyAsync(dc.p, cov33, (size_t)n * 9 * sizeof(float), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && normals3) e = hipMemcpyAsync(dn.p, normals3, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
      const int blocks = (int)((n + 255) / 256);
      pack_f32_kernel<<<blocks, 256, 0, s>>>(n, (const float*)dp.p, (const float*)dc.p, (const float*)dn.p, c->pts, c->covA, c->covB,
                                             c->normals);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
      set_hip_error(e, "cloud_create_f32 upload/pack");
      (void)hipStreamSynchronize(s);  // copies that did get enqueued must not outlive their staging blocks or the cloud's arrays
      glim_amd_cloud_destroy(c);
      return GLIM_AMD_ERR_HIP;
    }
    if (cov33 && normals3) {
      const int rc = detect_plane_form(c, s);
      if (rc != GLIM_AMD_OK) {
        glim_amd_cloud_destroy(c);
        return rc;
      }
    }
  }
  *out = c;
  return GLIM_AMD_OK;
}

int glim_amd_cloud_destroy(glim_amd_cloud* c) {
  if (!c) return GLIM_AMD_OK;
  if (c->ctx) {
    (void)hipSetDevice(c->ctx->device);
    quiesce_device(c->ctx->device, c->uid);  // asynchronous factor launches (of any context) may still be reading this cloud: its memory goes back to the pool below
    global_mutation_epoch()++;  // factor sets re-validate their plans
  }
  if (c->gs0) (void)pool_free(c->gs0);
  if (c->gs1) (void)pool_free(c->gs1);
  if (c->gs2) (void)pool_free(c->gs2);
  if (c->gsn) (void)pool_free(c->gsn);
  if (c->gbox) (void)pool_free(c->gbox);
  if (c->pts) (void)pool_free(c->pts);
  if (c->covA) (void)pool_free(c->covA);
  if (c->covB) (void)pool_free(c->covB);
  if (c->normals) (void)pool_free(c->normals);
  if (c->neighbors) (void)pool_free(c->neighbors);
  if (c->pn4) (void)pool_free(c->pn4);
  if (c->n2) (void)pool_free(c->n2);
  if (c->curve_rank) (void)pool_free(c->curve_rank);
  if (c->pts64) (void)pool_free(c->pts64);
  if (c->times) (void)pool_free(c->times);
  if (c->intensities) (void)pool_free(c->intensities);
  if (c->cov64) (void)pool_free(c->cov64);
  delete c;
  return GLIM_AMD_OK;
}

int glim_amd_cloud_size(const glim_amd_cloud* c, int64_t* n) {
  if (!c || !n) return GLIM_AMD_ERR_INVALID;
  *n = c->n;
  return GLIM_AMD_OK;
}

int glim_amd_cloud_memory_usage(const glim_amd_cloud* c, size_t* bytes) {
  if (!c || !bytes) return GLIM_AMD_ERR_INVALID;
  *bytes = c->bytes();
  return GLIM_AMD_OK;
}

int glim_amd_cloud_download(const glim_amd_cloud* c, float* xyz, float* cov33, float* normals3, int32_t* neighbors) {
  if (!c) return GLIM_AMD_ERR_INVALID;
  if (cov33 && !c->has_covs) return GLIM_AMD_ERR_STATE;
  if (normals3 && !c->has_normals) return GLIM_AMD_ERR_STATE;
  if (neighbors && !c->neighbors) return GLIM_AMD_ERR_STATE;
  if (c->n == 0) return GLIM_AMD_OK;
  glim_amd_ctx* ctx = c->ctx;
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream();
  const int64_t n = c->n;
  DeviceTemp dx, dc, dn;
  SyncOnExit in_flight(s);
  if (xyz) GA_HIP(pool_malloc(&dx.p, (size_t)n * 3 * sizeof(float)));
  if (cov33) GA_HIP(pool_malloc(&dc.p, (size_t)n * 9 * sizeof(float)));
  if (normals3) GA_HIP(pool_malloc(&dn.p, (size_t)n * 3 * sizeof(float)));
  if (xyz || cov33 || normals3) {
    unpack_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>(n, c->pts, c->covA, c->covB, c->normals, (float*)dx.p, (float*)dc.p, (float*)dn.p);
    GA_HIP(hipGetLastError());
  }
  if (xyz) GA_HIP(hipMemcpyAsync(xyz, dx.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, s));
  if (cov33) GA_HIP(hipMemcpyAsync(cov33, dc.p, (size_t)n * 9 * sizeof(float), hipMemcpyDeviceToHost, s));
  if (normals3) GA_HIP(hipMemcpyAsync(normals3, dn.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, s));
  if (neighbors) GA_HIP(hipMemcpyAsync(neighbors, c->neighbors, (size_t)n * c->k * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  GA_HIP(hipStreamSynchronize(s));
  in_flight.dismiss();
  return GLIM_AMD_OK;
}

// gtsam_points::PointCloud::save_compact (sub_map.cpp:62): FP32 files next to data.txt -- points_compact.bin (x y z), covs_compact.bin
// (c00 c01 c02 c11 c12 c22), normals_compact.bin (x y z), times_compact.bin, intensities_compact.bin.  The device SoA already
// holds exactly these values, so saving is a download plus a repack of the covariance halves.
static int write_file(const std::string& path, const void* data, size_t bytes) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return GLIM_AMD_ERR_INVALID;
  const size_t w = bytes ? fwrite(data, 1, bytes, f) : 0;
  fclose(f);
  return w == bytes ? GLIM_AMD_OK : GLIM_AMD_ERR_INVALID;
}
static bool read_file(const std::string& path, std::vector<char>& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize(sz > 0 ? (size_t)sz : 0);
  const size_t r = sz > 0 ? fread(out.data(), 1, (size_t)sz, f) : 0;
  fclose(f);
  return r == out.size();
}

int glim_amd_cloud_save_compact(const glim_amd_cloud* c, const char* dir) {
  if (!c || !dir) return GLIM_AMD_ERR_INVALID;
  const size_t n = (size_t)c->n;
  const std::string d(dir);
  std::vector<float> xyz(n * 3), cov9(c->has_covs ? n * 9 : 0), nrm(c->has_normals ? n * 3 : 0);
  GA_TRY(glim_amd_cloud_download(c, xyz.data(), c->has_covs ? cov9.data() : nullptr, c->has_normals ? nrm.data() : nullptr, nullptr));
  GA_TRY(write_file(d + "/points_compact.bin", xyz.data(), xyz.size() * sizeof(float)));
  if (c->has_covs) {
    std::vector<float> cov6(n * 6);
    for (size_t i = 0; i < n; i++) {
      const float* a = &cov9[9 * i];
      float* o = &cov6[6 * i];
      o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[4]; o[4] = a[5]; o[5] = a[8];
    }
    GA_TRY(write_file(d + "/covs_compact.bin", cov6.data(), cov6.size() * sizeof(float)));
  }
  if (c->has_normals) GA_TRY(write_file(d + "/normals_compact.bin", nrm.data(), nrm.size() * sizeof(float)));
  if (c->times || c->intensities) {
    std::vector<double> t(c->times ? n : 0), it(c->intensities ? n : 0);
    GA_TRY(glim_amd_cloud_download_frame(c, nullptr, c->times ? t.data() : nullptr, c->intensities ? it.data() : nullptr, nullptr));
    std::vector<float> f(n);
    if (c->times) {
      for (size_t i = 0; i < n; i++) f[i] = (float)t[i];
      GA_TRY(write_file(d + "/times_compact.bin", f.data(), n * sizeof(float)));
    }
    if (c->intensities) {
      for (size_t i = 0; i < n; i++) f[i] = (float)it[i];
      GA_TRY(write_file(d + "/intensities_compact.bin", f.data(), n * sizeof(float)));
    }
  }
  return GLIM_AMD_OK;
}

// gtsam_points::PointCloudCPU::load (sub_map.cpp:142) followed by PointCloudGPU::clone: compact FP32 files, or the full-precision
// points.bin (Vector4d) / covs.bin (Matrix4d) / normals.bin (Vector4d) pair when the compact ones are absent.
int glim_amd_cloud_load_compact(glim_amd_ctx* ctx, const char* dir, glim_amd_cloud** out) {
  if (!ctx || !dir || !out) return GLIM_AMD_ERR_INVALID;
  *out = nullptr;
  const std::string d(dir);
  std::vector<char> pb, cb, nb;
  if (read_file(d + "/points_compact.bin", pb)) {
    if (pb.size() % (3 * sizeof(float))) return GLIM_AMD_ERR_INVALID;
    const size_t n = pb.size() / (3 * sizeof(float));
    const bool has_c = read_file(d + "/covs_compact.bin", cb), has_n = read_file(d + "/normals_compact.bin", nb);
    if ((has_c && cb.size() != n * 6 * sizeof(float)) || (has_n && nb.size() != n * 3 * sizeof(float))) return GLIM_AMD_ERR_INVALID;
    std::vector<float> cov9(has_c ? n * 9 : 0);
    if (has_c) {
      const float* c6 = reinterpret_cast<const float*>(cb.data());
      for (size_t i = 0; i < n; i++) {
        const float* a = c6 + 6 * i;
        float* o = &cov9[9 * i];
        o[0] = a[0]; o[1] = o[3] = a[1]; o[2] = o[6] = a[2]; o[4] = a[3]; o[5] = o[7] = a[4]; o[8] = a[5];
      }
    }
    return glim_amd_cloud_create_f32(ctx, (int64_t)n, reinterpret_cast<const float*>(pb.data()), has_c ? cov9.data() : nullptr,
                                     has_n ? reinterpret_cast<const float*>(nb.data()) : nullptr, out);
  }
  if (read_file(d + "/points.bin", pb)) {
    if (pb.size() % (4 * sizeof(double))) return GLIM_AMD_ERR_INVALID;
    const size_t n = pb.size() / (4 * sizeof(double));
    const bool has_c = read_file(d + "/covs.bin", cb), has_n = read_file(d + "/normals.bin", nb);
    if ((has_c && cb.size() != n * 16 * sizeof(double)) || (has_n && nb.size() != n * 4 * sizeof(double))) return GLIM_AMD_ERR_INVALID;
    return glim_amd_cloud_create(ctx, (int64_t)n, reinterpret_cast<const double*>(pb.data()), has_c ? reinterpret_cast<const double*>(cb.data()) : nullptr,
                                 has_n ? reinterpret_cast<const double*>(nb.data()) : nullptr, out);
  }
  return GLIM_AMD_ERR_INVALID;
}

int glim_amd_cloud_set_neighbors(glim_amd_cloud* c, int k, const int32_t* neighbors) {
  if (!c || k <= 0 || (c->n > 0 && !neighbors)) return GLIM_AMD_ERR_INVALID;
  {
    // the covariance kernel gathers pts[neighbors[..]]: an index outside [0, n) would be an out-of-bounds device read
    const uint32_t un = (uint32_t)c->n;
    uint32_t bad = 0;
    const size_t total = (size_t)c->n * (size_t)k;
    for (size_t i = 0; i < total; i++) bad |= (uint32_t)((uint32_t)neighbors[i] >= un);
    if (bad) return GLIM_AMD_ERR_INVALID;
  }
  glim_amd_ctx* ctx = c->ctx;
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  if (c->neighbors) {
    (void)pool_free(c->neighbors);
    c->neighbors = nullptr;
  }
  c->k = k;
  const size_t bytes = (size_t)(c->n > 0 ? c->n : 1) * k * sizeof(int32_t);
  GA_HIP(pool_malloc(&c->neighbors, bytes));
  if (c->n > 0) {
    GA_HIP(hipMemcpyAsync(c->neighbors, neighbors, (size_t)c->n * k * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream()));
    GA_HIP(hipStreamSynchronize(ctx->stream()));
  }
  return GLIM_AMD_OK;
}

}  // extern "C"
