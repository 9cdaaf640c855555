// cloud_covariance_estimation_hip.cpp -- the HIP-backed twin of GLIM's src/glim/common/cloud_covariance_estimation.cpp.
//
// Same header (include/glim/common/cloud_covariance_estimation.hpp), same class, same members: a HIP build of libglim compiles THIS file in place
// of the reference's translation unit, and every call site -- odometry_estimation_imu.cpp:189,320, sub_mapping.cpp:374, the CPU odometry -- reaches
// the device kernel (covariance.hip, K1) without an edit.  Per call: the points go up as they are (FP64 Vector4d, kept exact), the neighbour lists the caller
// already holds are attached (glim_amd_cloud_set_neighbors), one kernel forms the population covariance of the first k_neighbors neighbours, its
// closed-form eigen-decomposition, the PLANE regularisation V diag(1e-3, 1, 1) V^T and the sensor-facing normal in FP64
// (cloud_covariance_estimation.cpp:76-101,181-196), and normals + covariances come back (FP32 storage on the device: the parity gate is 1e-5 relative,
// tests/test_ref.py measures 3e-8).  NONE / NORMALIZED_MIN_EIG / FROBENIUS are unreachable in the reference too (the constructor fixes PLANE, :20).
#include <glim/common/cloud_covariance_estimation.hpp>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include <glim_amd/gtsam_points_compat.hpp>

namespace glim {

namespace {

// One device round trip: upload, attach the caller's neighbour lists, estimate, download.  normals may be null (the covariance-only overloads).
void estimate_on_device(const std::vector<Eigen::Vector4d>& points, const std::vector<int>& neighbors, int k_neighbors, std::vector<Eigen::Vector4d>* normals,
                        std::vector<Eigen::Matrix4d>& covs) {
  static_assert(sizeof(Eigen::Vector4d) == 4 * sizeof(double) && sizeof(Eigen::Matrix4d) == 16 * sizeof(double), "dense fixed-size Eigen storage");
  static_assert(sizeof(int) == sizeof(std::int32_t), "neighbour indices are int32 on the device");
  const std::int64_t n = (std::int64_t)points.size();
  const int k_correspondences = (int)(neighbors.size() / points.size());
  if ((std::size_t)k_correspondences * points.size() != neighbors.size() || k_neighbors > k_correspondences || k_neighbors <= 0)
    throw std::runtime_error("CloudCovarianceEstimation: neighbors must hold N * m indices with k_neighbors <= m");
  glim_amd::Context ctx = glim_amd::StreamTempBufferRoundRobin::default_instance();
  glim_amd_cloud* cloud = nullptr;
  // (the exact FP64 points stay with the cloud: the callers pass deskewed IMU-frame points, which no FP32 image represents)
  glim_amd::check(glim_amd_cloud_create_exact(ctx->context(), n, reinterpret_cast<const double*>(points.data()), &cloud), "CloudCovarianceEstimation: upload");
  std::vector<float> cov33((std::size_t)n * 9), nrm3((std::size_t)n * 3);
  int rc = glim_amd_cloud_set_neighbors(cloud, k_correspondences, reinterpret_cast<const std::int32_t*>(neighbors.data()));
  if (rc == GLIM_AMD_OK) rc = glim_amd_cloud_estimate_covariances(cloud, k_neighbors);
  if (rc == GLIM_AMD_OK) rc = glim_amd_cloud_download(cloud, nullptr, cov33.data(), nrm3.data(), nullptr);
  (void)glim_amd_cloud_destroy(cloud);
  glim_amd::check(rc, "CloudCovarianceEstimation::estimate");
  covs.resize((std::size_t)n);
  if (normals) normals->resize((std::size_t)n);
  for (std::int64_t i = 0; i < n; i++) {
    double* c = reinterpret_cast<double*>(&covs[(std::size_t)i]);  // column-major Matrix4d; the 4th row / column stay zero (:96)
    std::memset(c, 0, 16 * sizeof(double));
    for (int r = 0; r < 3; r++)
      for (int col = 0; col < 3; col++) c[4 * col + r] = (double)cov33[9 * (std::size_t)i + 3 * r + col];
    if (normals) {
      double* v = reinterpret_cast<double*>(&(*normals)[(std::size_t)i]);
      v[0] = nrm3[3 * (std::size_t)i];
      v[1] = nrm3[3 * (std::size_t)i + 1];
      v[2] = nrm3[3 * (std::size_t)i + 2];
      v[3] = 0.0;
    }
  }
}

// cyclic Jacobi on a symmetric 3x3 (host; regularize() of ONE matrix is API surface, not a hot path): eigenvalues ascending, eigenvectors in columns
void eigen_sym3(const double A[3][3], double w[3], double V[3][3]) {
  double a[3][3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      a[r][c] = A[r][c];
      V[r][c] = r == c ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 64; sweep++) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (a[p][q] == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::abs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++) {
          const double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {
          const double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  int order[3] = {0, 1, 2};
  for (int i = 0; i < 2; i++)
    for (int j = i + 1; j < 3; j++)
      if (a[order[j]][order[j]] < a[order[i]][order[i]]) std::swap(order[i], order[j]);
  double Vs[3][3];
  for (int c = 0; c < 3; c++) {
    w[c] = a[order[c]][order[c]];
    for (int r = 0; r < 3; r++) Vs[r][c] = V[r][order[c]];
  }
  std::memcpy(V, Vs, sizeof(Vs));
}

}  // namespace

CloudCovarianceEstimation::CloudCovarianceEstimation(const int num_threads) : regularization_method(RegularizationMethod::PLANE), num_threads(num_threads) {}

CloudCovarianceEstimation::~CloudCovarianceEstimation() {}

void CloudCovarianceEstimation::estimate(const std::vector<Eigen::Vector4d>& points, const std::vector<int>& neighbors, std::vector<Eigen::Vector4d>& normals,
                                         std::vector<Eigen::Matrix4d>& covs) const {
  if (points.empty()) return;
  estimate(points, neighbors, (int)(neighbors.size() / points.size()), normals, covs);
}

void CloudCovarianceEstimation::estimate(const std::vector<Eigen::Vector4d>& points, const std::vector<int>& neighbors, const int k_neighbors,
                                         std::vector<Eigen::Vector4d>& normals, std::vector<Eigen::Matrix4d>& covs) const {
  if (points.empty()) return;
  estimate_on_device(points, neighbors, k_neighbors, &normals, covs);
}

// The covariance-only overloads divide by k - 1 instead of k (:153): a scale of the raw covariance, which the PLANE regularisation discards
// (only the eigenvectors survive) -- so they are the same device call.
std::vector<Eigen::Matrix4d> CloudCovarianceEstimation::estimate(const std::vector<Eigen::Vector4d>& points, const std::vector<int>& neighbors, const int k_neighbors) const {
  std::vector<Eigen::Matrix4d> covs;
  if (!points.empty()) estimate_on_device(points, neighbors, k_neighbors, nullptr, covs);
  return covs;
}

std::vector<Eigen::Matrix4d> CloudCovarianceEstimation::estimate(const std::vector<Eigen::Vector4d>& points, const std::vector<int>& neighbors) const {
  if (points.empty()) return std::vector<Eigen::Matrix4d>();
  return estimate(points, neighbors, (int)(neighbors.size() / points.size()));
}

// One matrix, on the host (the kernels regularise whole clouds; this member is public API only)
Eigen::Matrix4d CloudCovarianceEstimation::regularize(const Eigen::Matrix4d& cov, Eigen::Vector3d* eigenvalues, Eigen::Matrix3d* eigenvectors) const {
  if (regularization_method != RegularizationMethod::PLANE) return cov;
  const double* m = reinterpret_cast<const double*>(&cov);  // column-major
  double A[3][3], w[3], V[3][3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) A[r][c] = m[4 * c + r];
  eigen_sym3(A, w, V);
  if (eigenvalues) {
    double* e = reinterpret_cast<double*>(eigenvalues);
    for (int i = 0; i < 3; i++) e[i] = w[i];
  }
  if (eigenvectors) {
    double* e = reinterpret_cast<double*>(eigenvectors);
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) e[3 * c + r] = V[r][c];
  }
  const double values[3] = {1e-3, 1.0, 1.0};
  Eigen::Matrix4d out;
  double* o = reinterpret_cast<double*>(&out);
  std::memset(o, 0, 16 * sizeof(double));
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double s = 0.0;
      for (int k = 0; k < 3; k++) s += V[r][k] * values[k] * V[c][k];
      o[4 * c + r] = s;
    }
  return out;
}

}  // namespace glim
