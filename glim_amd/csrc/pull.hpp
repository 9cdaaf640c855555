// Device half of the small-cloud upload (cloud.hip unstage_kernel, voxelmap.hip frame_build_kernel): a block waits for its piece of the pinned
// staging block, then every lane pulls one point over PCIe into the cloud's arrays and factor streams and tests the plane form on the way.
#pragma once
#include "internal.hpp"

namespace glim_amd {

// Plane-form test: does every stored covariance equal I - (1 - 1e-3) n n^T for the stored unit normal, within FP32 rounding of the two
// uploads?  (GLIM's CloudCovarianceEstimation only emits this form, cloud_covariance_estimation.cpp:20,:181-196, together with the
// eigenvector it is built from as the normal, :98-101; a frame that keeps the CPU estimator and is uploaded with
// PointCloudGPU::clone(frame) therefore qualifies for the 24 B/pt factor kernel.)  Tolerance: C and n are each rounded to FP32
// independently (<= 6e-8 per coefficient), n n^T then differs by <= 1.3e-7 per entry; 4e-7 leaves margin and is far below any covariance a
// merged / averaged cloud would show (those differ from the form by 1e-3 or more).
__device__ __forceinline__ bool off_plane_form(float c00, float c01, float c02, float c11, float c12, float c22, float nx, float ny, float nz) {
  const float w = 0.999f, tol = 4e-7f;
  return !(fabsf(c00 - (1.f - w * nx * nx)) <= tol && fabsf(c01 + w * nx * ny) <= tol && fabsf(c02 + w * nx * nz) <= tol &&
           fabsf(c11 - (1.f - w * ny * ny)) <= tol && fabsf(c12 + w * ny * nz) <= tol && fabsf(c22 - (1.f - w * nz * nz)) <= tol);
}

// Gated form: the kernel was launched BEFORE the host converted anything (cloud_small_pack): the block waits until the host has published its
// piece of the staging block -- gate word of the piece == this upload's sequence number, host-mapped memory, polled over PCIe by ONE lane.
// Every block of the grid is resident at once (<= 128 blocks), so waiting blocks keep nobody out.  The wait is bounded (seconds): a host that
// never comes back ends in PULL_GAVE_UP in the word behind the violations word, not in a hung device.  All lanes of the block must call;
// false: gave up (the block must not touch the staging block).
__device__ __forceinline__ bool pull_wait(const PullArgs& pa, int* s_ok) {
  if (!pa.gate) return true;
  if (threadIdx.x == 0) {
    const unsigned int* g = pa.gate + (blockIdx.x * 256) / pa.piece_len;
    bool ok = false;
    for (unsigned int spins = 0; spins < (1u << 21) && !ok; spins++) {
      ok = __hip_atomic_load(g, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == pa.gate_seq;
      if (!ok) __builtin_amdgcn_s_sleep(16);
    }
    *s_ok = ok ? 1 : 0;
    if (!ok) __hip_atomic_store(pa.host_tail + 1, PULL_GAVE_UP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  return *s_ok != 0;
}

// point i (< pa.n) from the staging block into the cloud's arrays and into its factor streams (plane_stream_kernel / general_stream_kernel;
// arrival order: clouds of this size carry no Hilbert rank), in BOTH forms while the values are in registers -- which form the factor kernel
// reads is only known when every block has tested its points; the host drops the other.  Returns "not in plane form".
__device__ __forceinline__ bool pull_point(const PullArgs& pa, int i, float4& p, float4& a, float2& b, float4& v) {
  p = pa.s_pts[i];
  pa.pts[i] = p;
  a = make_float4(0.f, 0.f, 0.f, 0.f);
  v = a;
  b = make_float2(0.f, 0.f);
  if (pa.s_covA) {
    pa.covA[i] = a = pa.s_covA[i];
    pa.covB[i] = b = pa.s_covB[i];
  }
  if (pa.s_nrm) pa.nrm[i] = v = pa.s_nrm[i];
  if (pa.pn4) {
    pa.pn4[i] = make_float4(p.x, p.y, p.z, v.x);
    pa.n2[i] = make_float2(v.y, v.z);
  }
  if (pa.gs0) {
    pa.gs0[i] = make_float4(p.x, p.y, p.z, a.x);
    pa.gs1[i] = make_float4(a.y, a.z, a.w, b.x);
    pa.gs2[i] = b.y;
    if (pa.gsn) pa.gsn[i] = v;
  }
  return pa.s_covA && pa.s_nrm && off_plane_form(a.x, a.y, a.z, a.w, b.x, b.y, v.x, v.y, v.z);
}

// (one lane per wavefront that saw a point off the form)
__device__ __forceinline__ void pull_report(const PullArgs& pa, bool bad) {
  if (__any(bad) && (threadIdx.x & 63) == 0) {
    __hip_atomic_store(pa.host_tail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();  // (visible before anything this block publishes later: frame_build_kernel's arrival)
  }
}

}  // namespace glim_amd
