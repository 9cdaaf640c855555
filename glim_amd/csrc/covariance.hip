// covariance.hip -- kernel K1: per-point covariance + sensor-facing normal from k neighbour indices.
// Replaces glim::CloudCovarianceEstimation::estimate (src/glim/common/cloud_covariance_estimation.cpp:43-122) with the
// PLANE regularisation hard-wired by its constructor (:20, :181-196), as called from
// src/glim/odometry/odometry_estimation_imu.cpp:189,320 and src/glim/mapping/sub_mapping.cpp:374.
//
// Per point (FP64, like the reference):  s = sum p_j, S = sum p_j p_j^T over the first k_neighbors neighbours (:84-89);
// mean = s/k, cov = (S - mean s^T)/k (:91-92, population form); eigenvector e0 of the smallest eigenvalue by the closed-form
// trigonometric solver the reference calls (Eigen SelfAdjointEigenSolver::computeDirect, :183); regularised covariance
// V diag(1e-3,1,1) V^T = I - (1 - 1e-3) e0 e0^T, so only e0 is needed: computeDirect's V is orthonormal to rounding in every branch (two
// unit kernel vectors of shifted matrices and their normalised cross product), and tests/test_ref.py checks on the COMPILED reference that
// its own covariance differs from I - 0.999 n n^T built from its own normal by < 1e-12 on every point, degenerate neighbourhoods included.
// normal = e0 flipped so p.n <= 0 (:98-101).
// One thread per point; the neighbour gathers are 16-byte loads that hit L2 (neighbours are spatially close).
#include "internal.hpp"

// Every FP64 expression below is evaluated with separate roundings, in the order of the reference's code (Eigen's computeDirect as restated
// in oracle/vgicp_oracle.c: eig3_roots, eig3_extract_kernel, orc_eigen3_direct): the sums, the shifted / scaled matrix, the characteristic
// polynomial and the cross products then have the reference's bits, and what is left between the two eigenvectors is the last-place
// difference of atan2 / cos / sin between the device's and the host's math library, amplified by 1 / (eigenvalue gap) -- below 1e-5 unless
// the two smallest eigenvalues agree to ~1e-11 of the largest (tests/test_ref.py reports the fraction).
#pragma clang fp contract(off)

using namespace glim_amd;

namespace {

struct V3 {
  double x, y, z;
};
__device__ __forceinline__ V3 cross(const V3& a, const V3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// null vector of the (rank-2) symmetric matrix m (rows r0 r1 r2 = columns); `rep` = the column with the largest |diagonal|.
__device__ __forceinline__ V3 extract_kernel(const V3& c0, const V3& c1, const V3& c2, V3& rep) {
  const double d0 = fabs(c0.x), d1 = fabs(c1.y), d2 = fabs(c2.z);
  V3 a, b;
  if (d0 >= d1 && d0 >= d2) {
    rep = c0; a = c1; b = c2;
  } else if (d1 >= d2) {
    rep = c1; a = c2; b = c0;
  } else {
    rep = c2; a = c0; b = c1;
  }
  const V3 x0 = cross(rep, a), x1 = cross(rep, b);
  const double n0 = dot(x0, x0), n1 = dot(x1, x1);
  if (n0 > n1) {
    const double s = sqrt(n0);
    return {x0.x / s, x0.y / s, x0.z / s};
  }
  const double s = sqrt(n1);
  return {x1.x / s, x1.y / s, x1.z / s};
}

// eigenvector of the smallest eigenvalue of the symmetric matrix (m00 m01 m02 m11 m12 m22), computeDirect semantics.
__device__ V3 smallest_eigenvector(double m00, double m01, double m02, double m11, double m12, double m22) {
  const double eps = 2.220446049250313e-16;
  const double shift = (m00 + (m11 + m22)) / 3.0;  // Eigen's trace(): a0 + (a1 + a2)
  m00 -= shift; m11 -= shift; m22 -= shift;
  double scale = fmax(fmax(fabs(m00), fabs(m11)), fmax(fabs(m22), fmax(fabs(m01), fmax(fabs(m02), fabs(m12)))));
  if (scale > 0.0) {
    m00 /= scale; m01 /= scale; m02 /= scale; m11 /= scale; m12 /= scale; m22 /= scale;
  }
  // roots of the characteristic polynomial (ascending)
  const double c0 = m00 * m11 * m22 + 2.0 * m01 * m02 * m12 - m00 * m12 * m12 - m11 * m02 * m02 - m22 * m01 * m01;
  const double c1 = m00 * m11 - m01 * m01 + m00 * m22 - m02 * m02 + m11 * m22 - m12 * m12;
  const double c2 = m00 + m11 + m22;
  const double c2_3 = c2 * (1.0 / 3.0);
  double a_3 = (c2 * c2_3 - c1) * (1.0 / 3.0);
  if (a_3 < 0.0) a_3 = 0.0;
  const double half_b = 0.5 * (c0 + c2_3 * (2.0 * c2_3 * c2_3 - c1));
  double q = a_3 * a_3 * a_3 - half_b * half_b;
  if (q < 0.0) q = 0.0;
  const double rho = sqrt(a_3);
  const double theta = atan2(sqrt(q), half_b) * (1.0 / 3.0);
  const double ct = cos(theta), st = sin(theta);
  const double sqrt3 = sqrt(3.0);
  const double ev0 = c2_3 - rho * (ct + sqrt3 * st);
  const double ev1 = c2_3 - rho * (ct - sqrt3 * st);
  const double ev2 = c2_3 + 2.0 * rho * ct;

  if ((ev2 - ev0) <= eps) return {1.0, 0.0, 0.0};  // numerically isotropic: eigenvectors = identity

  const double d0 = ev2 - ev1, d1 = ev1 - ev0;
  V3 rep;
  const V3 k0 = {m00 - ev0, m01, m02}, k1 = {m01, m11 - ev0, m12}, k2 = {m02, m12, m22 - ev0};
  if (d0 > d1 && !(d1 > 2.0 * eps * d1)) {
    // lambda0 == lambda1 exactly while lambda2 is distinct: the reference ortho-normalises the representative column saved
    // while extracting the eigenvector of lambda2 (mirrors the solver's own formula, including its col(l) self-reference)
    const V3 h0 = {m00 - ev2, m01, m02}, h1 = {m01, m11 - ev2, m12}, h2 = {m02, m12, m22 - ev2};
    const V3 v2 = extract_kernel(h0, h1, h2, rep);
    const double dd = dot(v2, rep);
    V3 e = {rep.x - dd * rep.x, rep.y - dd * rep.y, rep.z - dd * rep.z};
    const double nn = sqrt(dot(e, e));
    return {e.x / nn, e.y / nn, e.z / nn};
  }
  return extract_kernel(k0, k1, k2, rep);
}

// FP64_POINTS: the cloud keeps the exact FP64 points its FP32 image was rounded from (preprocessed, deskewed and merged clouds: pts64) and
// the sums are taken over THOSE, as the reference estimates from the FP64 deskewed points (odometry_estimation_imu.cpp:320,
// sub_mapping.cpp:374); a cloud uploaded through PointCloudGPU::clone is an FP32 object in the reference too and is read as such.
template <bool FP64_POINTS>
__global__ __launch_bounds__(256) void covariance_kernel(int n, const float4* __restrict__ pts, const double4* __restrict__ pts64,
                                                         const int32_t* __restrict__ nbrs, int k_corr,
                                                         int k_nbr, float4* __restrict__ covA, float2* __restrict__ covB,
                                                         float4* __restrict__ normals, float4* __restrict__ pn4, float2* __restrict__ n2,
                                                         const unsigned int* __restrict__ rank) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double sx = 0, sy = 0, sz = 0, sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
  const int32_t* row = nbrs + (size_t)k_corr * i;
  for (int j = 0; j < k_nbr; j++) {
    double x, y, z;
    if (FP64_POINTS) {
      const double4 p = pts64[row[j]];
      x = p.x; y = p.y; z = p.z;
    } else {
      const float4 p = pts[row[j]];
      x = p.x; y = p.y; z = p.z;
    }
    sx += x; sy += y; sz += z;
    sxx += x * x; sxy += x * y; sxz += x * z; syy += y * y; syz += y * z; szz += z * z;
  }
  const double kk = (double)k_nbr;
  const double mx = sx / kk, my = sy / kk, mz = sz / kk;
  const double c00 = (sxx - mx * sx) / kk, c01 = (sxy - mx * sy) / kk, c02 = (sxz - mx * sz) / kk;
  const double c11 = (syy - my * sy) / kk, c12 = (syz - my * sz) / kk, c22 = (szz - mz * sz) / kk;
  // the reference reads the lower triangle of (S - mean s^T)/k: (1,0) = (sxy' - my*sx)/k.  Use the same entries.
  const double l10 = (sxy - my * sx) / kk, l20 = (sxz - mz * sx) / kk, l21 = (syz - mz * sy) / kk;
  (void)c01; (void)c02; (void)c12;
  V3 e = smallest_eigenvector(c00, l10, l20, c11, l21, c22);
  const double w = 1.0 - 1e-3;
  covA[i] = make_float4((float)(1.0 - w * e.x * e.x), (float)(-w * e.x * e.y), (float)(-w * e.x * e.z), (float)(1.0 - w * e.y * e.y));
  covB[i] = make_float2((float)(-w * e.y * e.z), (float)(1.0 - w * e.z * e.z));
  const float4 p = pts[i];
  double px = p.x, py = p.y, pz = p.z;
  if (FP64_POINTS) {
    const double4 q = pts64[i];
    px = q.x; py = q.y; pz = q.z;
  }
  if (px * e.x + py * e.y + pz * e.z > 0.0) {  // :98-101  points[i].dot(normal) > 0 -> flip
    e.x = -e.x; e.y = -e.y; e.z = -e.z;
  }
  normals[i] = make_float4((float)e.x, (float)e.y, (float)e.z, 0.0f);
  // plane-form stream of the factor kernel (24 B per point: xyz + the unit normal the covariance is a function of), in Hilbert order when the cloud has one, so that the 64 lanes of a wavefront look up a
  // handful of neighbouring voxels instead of voxels spread along a scan line (the factor sums over all points: order is free)
  const unsigned int o = rank ? rank[i] : (unsigned int)i;
  pn4[o] = make_float4(p.x, p.y, p.z, (float)e.x);
  n2[o] = make_float2((float)e.y, (float)e.z);
}

}  // namespace

extern "C" {

int glim_amd_cloud_estimate_covariances(glim_amd_cloud* c, int k_neighbors) {
  if (!c) return GLIM_AMD_ERR_INVALID;
  if (!c->neighbors || c->k <= 0) return GLIM_AMD_ERR_STATE;
  if (k_neighbors <= 0 || k_neighbors > c->k) return GLIM_AMD_ERR_INVALID;
  glim_amd_ctx* ctx = c->ctx;
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  const size_t nn = (size_t)(c->n > 0 ? c->n : 1);
  if (!c->covA) GA_HIP(pool_malloc(&c->covA, nn * sizeof(float4)));
  if (!c->covB) GA_HIP(pool_malloc(&c->covB, nn * sizeof(float2)));
  if (!c->normals) GA_HIP(pool_malloc(&c->normals, nn * sizeof(float4)));
  // General factor streams built from earlier covariances are stale now, and factor plans hold their addresses: wait for asynchronous
  // launches that may still read them, then give the cloud a new identity so that every plan built from the old streams is rebuilt.
  quiesce_device(ctx->device, c->uid);
  c->uid = next_uid();
  global_mutation_epoch()++;
  if (c->gs0) { (void)pool_free(c->gs0); c->gs0 = nullptr; }
  if (c->gs1) { (void)pool_free(c->gs1); c->gs1 = nullptr; }
  if (c->gs2) { (void)pool_free(c->gs2); c->gs2 = nullptr; }
  if (c->gsn) { (void)pool_free(c->gsn); c->gsn = nullptr; }
  if (c->gbox) { (void)pool_free(c->gbox); c->gbox = nullptr; }
  if (!c->pn4) GA_HIP(pool_malloc(&c->pn4, nn * sizeof(float4)));
  if (!c->n2) GA_HIP(pool_malloc(&c->n2, nn * sizeof(float2)));
  if (c->n > 0) {
    const int n = (int)c->n;
    GA_TRY(cloud_curve_rank(c, ctx, ctx->stream()));
    if (c->pts64)
      covariance_kernel<true><<<(n + 255) / 256, 256, 0, ctx->stream()>>>(n, c->pts, c->pts64, c->neighbors, c->k, k_neighbors, c->covA, c->covB, c->normals,
                                                                          c->pn4, c->n2, c->curve_rank);
    else
      covariance_kernel<false><<<(n + 255) / 256, 256, 0, ctx->stream()>>>(n, c->pts, nullptr, c->neighbors, c->k, k_neighbors, c->covA, c->covB, c->normals,
                                                                           c->pn4, c->n2, c->curve_rank);
    GA_HIP(hipGetLastError());
    GA_HIP(hipStreamSynchronize(ctx->stream()));
  }
  c->has_covs = true;
  c->has_normals = true;
  c->plane_form = true;
  return GLIM_AMD_OK;
}

}  // extern "C"
