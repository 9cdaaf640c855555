// deskew.hip -- motion-undistortion of a scan fused with the device upload.
// Replaces glim::CloudDeskewing::deskew (src/glim/common/cloud_deskewing.cpp:11-53 constant velocity, :55-133 IMU poses) as
// called between preprocessing and covariance estimation (src/glim/odometry/odometry_estimation_imu.cpp:313-316,
// src/glim/mapping/sub_mapping.cpp:356-372), fused with the step both callers take next -- every deskewed point moved into the IMU frame,
// `pt = T_imu_lidar * pt` (odometry_estimation_imu.cpp:314-316, sub_mapping.cpp:368-370) -- and with PointCloudGPU::clone: one kernel reads
// the raw Vector4d points and writes the deskewed cloud (exact FP64 points for the covariance kernel + their FP32 image for the factor path),
// so the deskewed FP64 copy never exists on the host.
//
// The reference quantises time into a table (a new entry whenever a point is more than 0.1 ms after the last entry, :24-36,
// :72-84) and computes ONE rigid transform T_lidar0_lidar1 per entry; every point is moved by the transform of its entry.
// The table (a few hundred 3x4 matrices) and the per-point entry index are built on the host exactly like the reference does
// -- a serial pass over n time stamps plus <= ~1000 small pose compositions -- and the per-point work (the data-parallel
// part: n gathers of a 96-byte matrix + one FP64 3x4 transform + the FP32 pack) runs on the device.
#include <cmath>
#include <vector>

#include "internal.hpp"
#include "device_math.hpp"

using namespace glim_amd;

// every FP64 expression of the host-side table below is evaluated with separate roundings, like the oracle (-ffp-contract=off)
#pragma clang fp contract(off)

namespace {

struct Pose {
  double m[12];  // row-major 3x4 [R | t]
};

Pose compose(const Pose& A, const Pose& B) {
  Pose C;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) C.m[4 * r + c] = A.m[4 * r + 0] * B.m[c] + A.m[4 * r + 1] * B.m[4 + c] + A.m[4 * r + 2] * B.m[8 + c];
    C.m[4 * r + 3] = A.m[4 * r + 0] * B.m[3] + A.m[4 * r + 1] * B.m[7] + A.m[4 * r + 2] * B.m[11] + A.m[4 * r + 3];
  }
  return C;
}

Pose inverse(const Pose& A) {
  Pose I;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) I.m[4 * r + c] = A.m[4 * c + r];
  for (int r = 0; r < 3; r++) I.m[4 * r + 3] = -(I.m[4 * r + 0] * A.m[3] + I.m[4 * r + 1] * A.m[7] + I.m[4 * r + 2] * A.m[11]);
  return I;
}

// gtsam::Pose3::Expmap([omega; v]) as GTSAM 4.2 evaluates it (cloud_deskewing.cpp:43), in the operation order of the oracle's restatement
// (oracle/vgicp_oracle.c pose3_expmap_gtsam, itself bit-equal to the compiled reference over the stand-in Pose3.h): R = I + sin(theta) K +
// (1 - cos(theta)) K^2, K = hat(omega) / theta, 1 - cos(theta) = 2 sin^2(theta / 2); t = (omega x v - R (omega x v) + omega (omega . v)) / theta^2.
// The table has to carry the oracle's BITS: the deskewed FP64 points are compared bit for bit (tests/test_ref.py).
Pose se3_exp(const double* xi) {
  const double eps = 2.220446049250313e-16;
  const double w[3] = {xi[0], xi[1], xi[2]}, v[3] = {xi[3], xi[4], xi[5]};
  const double theta2 = (w[0] * w[0] + w[1] * w[1]) + w[2] * w[2];
  const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  double R[9];
  if (theta2 <= eps) {
    for (int i = 0; i < 9; i++) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + W[i];
  } else {
    const double theta = std::sqrt(theta2);
    const double sin_theta = std::sin(theta);
    const double s2 = std::sin(theta / 2.0);
    const double one_minus_cos = 2.0 * s2 * s2;
    double K[9], KK[9];
    for (int i = 0; i < 9; i++) K[i] = W[i] / theta;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) KK[3 * r + c] = (K[3 * r] * K[c] + K[3 * r + 1] * K[3 + c]) + K[3 * r + 2] * K[6 + c];
    for (int i = 0; i < 9; i++) R[i] = (((i % 4 == 0) ? 1.0 : 0.0) + K[i] * sin_theta) + KK[i] * one_minus_cos;
  }
  double t[3];
  if (theta2 > eps) {
    const double wv = (w[0] * v[0] + w[1] * v[1]) + w[2] * v[2];
    const double c[3] = {w[1] * v[2] - w[2] * v[1], w[2] * v[0] - w[0] * v[2], w[0] * v[1] - w[1] * v[0]};
    double Rc[3];
    for (int r = 0; r < 3; r++) Rc[r] = (R[3 * r] * c[0] + R[3 * r + 1] * c[1]) + R[3 * r + 2] * c[2];
    for (int r = 0; r < 3; r++) t[r] = ((c[r] - Rc[r]) + w[r] * wv) / theta2;
  } else {
    t[0] = v[0]; t[1] = v[1]; t[2] = v[2];
  }
  Pose T;
  for (int r = 0; r < 3; r++) {
    T.m[4 * r + 0] = R[3 * r + 0];
    T.m[4 * r + 1] = R[3 * r + 1];
    T.m[4 * r + 2] = R[3 * r + 2];
    T.m[4 * r + 3] = t[r];
  }
  return T;
}

// Eigen::Quaterniond(Matrix3d), ::slerp, ::toRotationMatrix (the routines cloud_deskewing.cpp:113-119 calls); q = (x, y, z, w)
void quat_from_rot(const Pose& T, double* q) {
  double m[3][3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) m[r][c] = T.m[4 * r + c];
  double t = m[0][0] + (m[1][1] + m[2][2]);  // Eigen's trace(): a0 + (a1 + a2)
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[2][1] - m[1][2]) * t;
    q[1] = (m[0][2] - m[2][0]) * t;
    q[2] = (m[1][0] - m[0][1]) * t;
  } else {
    int i = 0;
    if (m[1][1] > m[0][0]) i = 1;
    if (m[2][2] > m[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m[k][j] - m[j][k]) * t;
    q[j] = (m[j][i] + m[i][j]) * t;
    q[k] = (m[k][i] + m[i][k]) * t;
  }
}

void quat_slerp(const double* a, double t, const double* b, double* out) {
  const double one = 1.0 - 2.220446049250313e-16;
  const double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  const double absD = std::fabs(d);
  double s0, s1;
  if (absD >= one) {
    s0 = 1.0 - t;
    s1 = t;
  } else {
    const double theta = std::acos(absD), sinTheta = std::sin(theta);
    s0 = std::sin((1.0 - t) * theta) / sinTheta;
    s1 = std::sin(t * theta) / sinTheta;
  }
  if (d < 0.0) s1 = -s1;
  for (int i = 0; i < 4; i++) out[i] = s0 * a[i] + s1 * b[i];
}

void quat_to_rot(const double* q, double* R) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0 - (txx + tyy);
}

// Eigen::Isometry3d * Eigen::Vector4d as the oracle states it: three 4-term sums with separate roundings (no FMA contraction), w copied
__device__ __forceinline__ double4 apply_pose(const double* __restrict__ T, const double4 p) {
  double4 q;
  q.x = dadd(dadd(dadd(dmul(T[0], p.x), dmul(T[1], p.y)), dmul(T[2], p.z)), dmul(T[3], p.w));
  q.y = dadd(dadd(dadd(dmul(T[4], p.x), dmul(T[5], p.y)), dmul(T[6], p.z)), dmul(T[7], p.w));
  q.z = dadd(dadd(dadd(dmul(T[8], p.x), dmul(T[9], p.y)), dmul(T[10], p.z)), dmul(T[11], p.w));
  q.w = p.w;
  return q;
}

// one thread per point: q = T[entry(i)] p in FP64 (cloud_deskewing.cpp:47-51, :127-130) and -- IMU_FRAME -- q = T_imu_lidar q as a SECOND
// FP64 product with its own roundings, which is what both callers do to every deskewed point before they estimate covariances
// (odometry_estimation_imu.cpp:314-316 `pt = T_imu_lidar * pt`, sub_mapping.cpp:368-370).  The exact FP64 result is kept (pts64: the
// covariance kernel reads it, as the reference estimates from the FP64 deskewed points, :320 / :374) next to its FP32 image for the factor path.
template <bool IMU_FRAME>
__global__ __launch_bounds__(256) void deskew_pack_kernel(int64_t n, const double* __restrict__ points4, const int* __restrict__ entry,
                                                          const double* __restrict__ table, const double* __restrict__ T_imu_lidar,
                                                          float4* __restrict__ pts, double4* __restrict__ pts64) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double4 p = reinterpret_cast<const double4*>(points4)[i];
  double4 q = apply_pose(table + 12 * (size_t)entry[i], p);
  if (IMU_FRAME) q = apply_pose(T_imu_lidar, q);
  pts64[i] = q;
  pts[i] = make_float4((float)q.x, (float)q.y, (float)q.z, 1.0f);
}

// host: time table + one transform per entry (cloud_deskewing.cpp:22-45 / :70-124)
void build_deskew_table(int64_t n, const double* times, const double* T_imu_lidar12, int32_t n_imu, const double* imu_times, const double* imu_poses12,
                        double stamp, const double* linear_vel3, const double* angular_vel3, std::vector<int>& entry, std::vector<Pose>& TT) {
  const double time_eps = 1e-4;
  std::vector<double> table;
  entry.resize((size_t)n);
  for (int64_t i = 0; i < n; i++) {
    if (table.empty() || times[i] - table.back() > time_eps) table.push_back(times[i]);
    entry[(size_t)i] = (int)table.size() - 1;
  }
  Pose T_imu_lidar, T_lidar_imu;
  memcpy(T_imu_lidar.m, T_imu_lidar12, sizeof(T_imu_lidar.m));
  T_lidar_imu = inverse(T_imu_lidar);
  TT.resize(table.size());
  if (n_imu == 0) {
    const double zero[3] = {0, 0, 0};
    const double* lv = linear_vel3 ? linear_vel3 : zero;
    const double* av = angular_vel3 ? angular_vel3 : zero;
    for (size_t i = 0; i < table.size(); i++) {
      const double dt = table[i];
      const double xi[6] = {dt * av[0], dt * av[1], dt * av[2], dt * lv[0], dt * lv[1], dt * lv[2]};
      TT[i] = compose(compose(T_lidar_imu, inverse(se3_exp(xi))), T_imu_lidar);
    }
  } else {
    int cursor = 0;
    Pose T_imu0_world{};
    for (size_t i = 0; i < table.size(); i++) {
      const double time = stamp + table[i];
      while (cursor < n_imu - 1 && imu_times[cursor + 1] < time) cursor++;
      Pose L, Rp;
      memcpy(L.m, imu_poses12 + 12 * (size_t)cursor, sizeof(L.m));
      if (i == 0) T_imu0_world = inverse(L);
      Pose T_world_imu1 = L;
      if (cursor + 1 < n_imu) {
        memcpy(Rp.m, imu_poses12 + 12 * (size_t)(cursor + 1), sizeof(Rp.m));
        const double t0 = imu_times[cursor], t1 = imu_times[cursor + 1];
        const double p = std::max(0.0, std::min(1.0, (time - t0) / (t1 - t0)));
        double ql[4], qr[4], qs[4], R[9];
        quat_from_rot(L, ql);
        quat_from_rot(Rp, qr);
        quat_slerp(ql, p, qr, qs);
        quat_to_rot(qs, R);
        for (int r = 0; r < 3; r++) {
          for (int c = 0; c < 3; c++) T_world_imu1.m[4 * r + c] = R[3 * r + c];
          T_world_imu1.m[4 * r + 3] = (1.0 - p) * L.m[4 * r + 3] + p * Rp.m[4 * r + 3];
        }
      }
      TT[i] = compose(compose(T_lidar_imu, compose(T_imu0_world, T_world_imu1)), T_imu_lidar);
    }
  }
}

// device: (upload the raw points unless they are already resident,) entry indices and the table; transform + pack.
// Caller holds ctx->mu.
// carry: a preprocessed cloud whose neighbour lists the result takes over (copied on the device before the one synchronise of this call), or null
// imu_frame: 12 doubles (T_imu_lidar) to move the deskewed points on into the IMU frame, or null to leave them in the LiDAR frame
int run_deskew(glim_amd_ctx* ctx, int64_t n, const double* h_points4, const double* d_points4, const std::vector<int>& entry, const std::vector<Pose>& TT,
               const double* imu_frame, const glim_amd_cloud* carry, glim_amd_cloud** out) {
  glim_amd_cloud* c = new glim_amd_cloud();
  c->ctx = ctx;
  c->n = n;
  hipError_t e = pool_malloc(&c->pts, (size_t)(n > 0 ? n : 1) * sizeof(float4));
  if (e == hipSuccess) e = pool_malloc(&c->pts64, (size_t)(n > 0 ? n : 1) * sizeof(double4));
  if (e != hipSuccess) {
    set_hip_error(e, "pool_malloc(deskewed cloud)");
    delete c;
    return GLIM_AMD_ERR_HIP;
  }
  if (n > 0) {
    hipStream_t s = ctx->stream();
    DeviceTemp dp;
    if (!d_points4) {
      e = pool_malloc(&dp.p, (size_t)n * 4 * sizeof(double));
      if (e == hipSuccess) e = hipMemcpyAsync(dp.p, h_points4, (size_t)n * 4 * sizeof(double), hipMemcpyHostToDevice, s);
      d_points4 = (const double*)dp.p;
    }
    // table (+ the extrinsic) and per-point entries travel in ONE pinned block (two pageable copies are staged by the runtime one after the other)
    const size_t entry_bytes = ((size_t)n * sizeof(int) + 63) & ~(size_t)63, table_bytes = (TT.size() + 1) * sizeof(Pose);
    char* stage = nullptr;
    DeviceTemp dev;
    if (e == hipSuccess) e = pool_malloc(&dev.p, entry_bytes + table_bytes);
    if (e == hipSuccess) e = pinned_malloc(&stage, entry_bytes + table_bytes);
    if (e == hipSuccess) {
      memcpy(stage, entry.data(), (size_t)n * sizeof(int));
      memcpy(stage + entry_bytes, TT.data(), TT.size() * sizeof(Pose));
      if (imu_frame) memcpy(stage + entry_bytes + TT.size() * sizeof(Pose), imu_frame, sizeof(Pose));
      e = hipMemcpyAsync(dev.p, stage, entry_bytes + table_bytes, hipMemcpyHostToDevice, s);
    }
    if (e == hipSuccess) {
      const double* d_table = (const double*)((const char*)dev.p + entry_bytes);
      const unsigned int grid = (unsigned int)((n + 255) / 256);
      if (imu_frame) deskew_pack_kernel<true><<<grid, 256, 0, s>>>(n, d_points4, (const int*)dev.p, d_table, d_table + 12 * TT.size(), c->pts, c->pts64);
      else deskew_pack_kernel<false><<<grid, 256, 0, s>>>(n, d_points4, (const int*)dev.p, d_table, nullptr, c->pts, c->pts64);
      e = hipGetLastError();
    }
    if (e == hipSuccess && carry && carry->neighbors) {
      e = pool_malloc(&c->neighbors, (size_t)n * carry->k * sizeof(int32_t));
      if (e == hipSuccess) e = hipMemcpyAsync(c->neighbors, carry->neighbors, (size_t)n * carry->k * sizeof(int32_t), hipMemcpyDeviceToDevice, s);
      if (e == hipSuccess) c->k = carry->k;
    }
    if (e == hipSuccess && carry && carry->curve_rank) {
      // ... and so is the Hilbert rank of the raw points (motion correction moves a point by centimetres; the order only affects the speed of
      // the factor kernel's voxel look-ups): estimate_covariances does not have to sort the cloud a second time
      e = pool_malloc(&c->curve_rank, (size_t)n * sizeof(unsigned int));
      if (e == hipSuccess) e = hipMemcpyAsync(c->curve_rank, carry->curve_rank, (size_t)n * sizeof(unsigned int), hipMemcpyDeviceToDevice, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    else (void)hipStreamSynchronize(s);  // (an upload that did get enqueued must not outlive its staging block)
    if (stage) (void)pinned_free(stage);
    if (e != hipSuccess) {
      set_hip_error(e, "cloud deskew");
      glim_amd_cloud_destroy(c);
      return GLIM_AMD_ERR_HIP;
    }
  }
  *out = c;
  return GLIM_AMD_OK;
}

}  // namespace

extern "C" {

int glim_amd_cloud_create_exact(glim_amd_ctx* ctx, int64_t n, const double* points4, glim_amd_cloud** out) {
  if (!ctx || !out || n < 0 || (n > 0 && !points4)) return GLIM_AMD_ERR_INVALID;
  *out = nullptr;
  if (n > (int64_t)(1 << 28)) return GLIM_AMD_ERR_INVALID;
  // the deskewing path with ONE table entry, the identity: q = 1 * x + 0 * y + 0 * z + 0 is exact, so the cloud's FP64 points are the caller's
  const std::vector<int> entry((size_t)n, 0);
  std::vector<Pose> TT(1);
  memset(TT[0].m, 0, sizeof(TT[0].m));
  TT[0].m[0] = TT[0].m[5] = TT[0].m[10] = 1.0;
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  return run_deskew(ctx, n, points4, nullptr, entry, TT, nullptr, nullptr, out);
}

int glim_amd_cloud_create_deskewed(glim_amd_ctx* ctx, int64_t n, const double* points4, const double* times, const double* T_imu_lidar12,
                                   int32_t n_imu, const double* imu_times, const double* imu_poses12, double stamp, const double* linear_vel3,
                                   const double* angular_vel3, int32_t to_imu_frame, glim_amd_cloud** out) {
  if (!ctx || !out || n < 0 || !T_imu_lidar12 || (n > 0 && (!points4 || !times))) return GLIM_AMD_ERR_INVALID;
  if (n_imu < 0 || (n_imu > 0 && (!imu_times || !imu_poses12))) return GLIM_AMD_ERR_INVALID;
  *out = nullptr;
  if (n > (int64_t)(1 << 28)) return GLIM_AMD_ERR_INVALID;
  std::vector<int> entry;
  std::vector<Pose> TT;
  build_deskew_table(n, times, T_imu_lidar12, n_imu, imu_times, imu_poses12, stamp, linear_vel3, angular_vel3, entry, TT);
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  return run_deskew(ctx, n, points4, nullptr, entry, TT, to_imu_frame ? T_imu_lidar12 : nullptr, nullptr, out);
}

int glim_amd_debug_deskew_table(int64_t n, const double* times, const double* T_imu_lidar12, int32_t n_imu, const double* imu_times, const double* imu_poses12,
                                double stamp, const double* linear_vel3, const double* angular_vel3, int32_t* entry_out, double* table12_out, int32_t table_cap,
                                int32_t* table_size) {
  if (n < 0 || !T_imu_lidar12 || (n > 0 && !times) || !table_size) return GLIM_AMD_ERR_INVALID;
  if (n_imu < 0 || (n_imu > 0 && (!imu_times || !imu_poses12))) return GLIM_AMD_ERR_INVALID;
  std::vector<int> entry;
  std::vector<Pose> TT;
  build_deskew_table(n, times, T_imu_lidar12, n_imu, imu_times, imu_poses12, stamp, linear_vel3, angular_vel3, entry, TT);
  *table_size = (int32_t)TT.size();
  if (entry_out) memcpy(entry_out, entry.data(), (size_t)n * sizeof(int32_t));
  if (table12_out) {
    if ((int32_t)TT.size() > table_cap) return GLIM_AMD_ERR_INVALID;
    memcpy(table12_out, TT.data(), TT.size() * sizeof(Pose));
  }
  return GLIM_AMD_OK;
}

int glim_amd_cloud_deskew(const glim_amd_cloud* pre, const double* T_imu_lidar12, int32_t n_imu, const double* imu_times, const double* imu_poses12,
                          double stamp, const double* linear_vel3, const double* angular_vel3, int32_t to_imu_frame, glim_amd_cloud** out) {
  if (!pre || !out || !T_imu_lidar12) return GLIM_AMD_ERR_INVALID;
  if (n_imu < 0 || (n_imu > 0 && (!imu_times || !imu_poses12))) return GLIM_AMD_ERR_INVALID;
  *out = nullptr;
  if (!pre->pts64 || !pre->times || (int64_t)pre->h_times.size() != pre->n) return GLIM_AMD_ERR_STATE;  // not a preprocessed cloud
  glim_amd_ctx* ctx = pre->ctx;
  const int64_t n = pre->n;
  std::vector<int> entry;
  std::vector<Pose> TT;
  build_deskew_table(n, pre->h_times.data(), T_imu_lidar12, n_imu, imu_times, imu_poses12, stamp, linear_vel3, angular_vel3, entry, TT);
  std::lock_guard<std::mutex> lock(ctx->mu);
  GA_HIP(hipSetDevice(ctx->device));
  glim_amd_cloud* c = nullptr;
  // (the neighbour lists found on the raw scan are carried over: odometry_estimation_imu.cpp:320)
  GA_TRY(run_deskew(ctx, n, nullptr, reinterpret_cast<const double*>(pre->pts64), entry, TT, to_imu_frame ? T_imu_lidar12 : nullptr, pre, &c));
  *out = c;
  return GLIM_AMD_OK;
}

}  // extern "C"
