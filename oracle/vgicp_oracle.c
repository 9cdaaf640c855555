/*
 * vgicp_oracle.c -- see vgicp_oracle.h.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (header explains).
 *
 * Build: gcc -O2 -std=c11 -fopenmp -ffp-contract=off -march=x86-64-v3 -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off + explicit fma() keeps the FP64 point transform bit-identical to the HIP kernels.
 */
#include "vgicp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------ */
/* helpers                                                                                          */
/* ------------------------------------------------------------------------------------------------ */

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static int clamp_threads(int t) {
  if (t <= 0) t = orc_max_threads();
  return t < 1 ? 1 : t;
}

int32_t orc_fast_floor(double x) {
  const int32_t i = (int32_t)x;
  return i - (x < (double)i);
}

void orc_voxel_coord(const double* p, double inv_res, int32_t* c) {
  c[0] = orc_fast_floor(p[0] * inv_res);
  c[1] = orc_fast_floor(p[1] * inv_res);
  c[2] = orc_fast_floor(p[2] * inv_res);
}

void orc_transform_point(const double* T, const double* p, double* q) {
  for (int r = 0; r < 3; r++) {
    q[r] = fma(T[4 * r + 0], p[0], fma(T[4 * r + 1], p[1], fma(T[4 * r + 2], p[2], T[4 * r + 3])));
  }
}

static void hat3(const double* a, double* H /* row-major 3x3 */) {
  H[0] = 0.0;   H[1] = -a[2]; H[2] = a[1];
  H[3] = a[2];  H[4] = 0.0;   H[5] = -a[0];
  H[6] = -a[1]; H[7] = a[0];  H[8] = 0.0;
}

static void mat3_mul(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0.0;
      for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j];
      C[3 * i + j] = s;
    }
}

/* SE(3) exponential, [omega; v] order (gtsam::Pose3::Expmap). */
void orc_se3_exp(const double* xi, double* T) {
  const double wx = xi[0], wy = xi[1], wz = xi[2];
  const double th2 = wx * wx + wy * wy + wz * wz;
  const double th = sqrt(th2);
  double W[9], W2[9];
  hat3(xi, W);
  mat3_mul(W, W, W2);
  double a, b, c; /* R = I + a W + b W^2 ; V = I + b W + c W^2 */
  if (th < 1e-8) {
    a = 1.0 - th2 / 6.0;
    b = 0.5 - th2 / 24.0;
    c = 1.0 / 6.0 - th2 / 120.0;
  } else {
    a = sin(th) / th;
    b = (1.0 - cos(th)) / th2;
    c = (th - sin(th)) / (th2 * th);
  }
  double R[9], V[9];
  for (int i = 0; i < 9; i++) {
    const double I = (i % 4 == 0) ? 1.0 : 0.0;
    R[i] = I + a * W[i] + b * W2[i];
    V[i] = I + b * W[i] + c * W2[i];
  }
  for (int r = 0; r < 3; r++) {
    T[4 * r + 0] = R[3 * r + 0];
    T[4 * r + 1] = R[3 * r + 1];
    T[4 * r + 2] = R[3 * r + 2];
    T[4 * r + 3] = V[3 * r + 0] * xi[3] + V[3 * r + 1] * xi[4] + V[3 * r + 2] * xi[5];
  }
}

void orc_pose_compose(const double* A, const double* B, double* C) {
  double out[12];
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) out[4 * r + c] = A[4 * r + 0] * B[c] + A[4 * r + 1] * B[4 + c] + A[4 * r + 2] * B[8 + c];
    out[4 * r + 3] = A[4 * r + 0] * B[3] + A[4 * r + 1] * B[7] + A[4 * r + 2] * B[11] + A[4 * r + 3];
  }
  memcpy(C, out, sizeof(out));
}

void orc_pose_inverse(const double* A, double* Ai) {
  double out[12];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) out[4 * r + c] = A[4 * c + r];
  for (int r = 0; r < 3; r++) out[4 * r + 3] = -(out[4 * r + 0] * A[3] + out[4 * r + 1] * A[7] + out[4 * r + 2] * A[11]);
  memcpy(Ai, out, sizeof(out));
}

/* ------------------------------------------------------------------------------------------------ */
/* kNN  -- src/glim/preprocess/cloud_preprocessor.cpp:190-221                                        */
/* ------------------------------------------------------------------------------------------------ */

/* squared distance, fixed evaluation order (dx*dx + dy*dy) + dz*dz, no fma (part of the parity contract
 * with the device kNN kernel: FP32-representable inputs make dx,dy,dz exact in FP64). */
static inline double sqdist3(const double* a, const double* b) {
  const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return (dx * dx + dy * dy) + dz * dz;
}

/* keep the k best (d, idx) pairs sorted ascending by (d, idx) */
static inline void knn_push(double* bd, int32_t* bi, int* cnt, int k, double d, int32_t idx) {
  int n = *cnt;
  if (n == k) {
    if (d > bd[k - 1] || (d == bd[k - 1] && idx > bi[k - 1])) return;
    n = k - 1;
  }
  int j = n;
  while (j > 0 && (bd[j - 1] > d || (bd[j - 1] == d && bi[j - 1] > idx))) {
    bd[j] = bd[j - 1];
    bi[j] = bi[j - 1];
    j--;
  }
  bd[j] = d;
  bi[j] = idx;
  *cnt = n + 1;
}

void orc_knn_bruteforce(const double* pts, int n, int k, int32_t* out, int num_threads) {
  num_threads = clamp_threads(num_threads);
#pragma omp parallel for num_threads(num_threads) schedule(guided, 8)
  for (int i = 0; i < n; i++) {
    double bd[64];
    int32_t bi[64];
    int cnt = 0;
    const int kk = k > 64 ? 64 : k;
    for (int j = 0; j < n; j++) knn_push(bd, bi, &cnt, kk, sqdist3(pts + 4 * i, pts + 4 * j), j);
    /* fewer than k points: the tail is 0 -- the reference pre-fills its scratch indices with i (:197) but copies only the FOUND ones
     * into the zero-initialised result (:193, :200): pinned by the compiled reference, tests/test_ref.py */
    for (int j = 0; j < k; j++) out[(size_t)i * k + j] = (j < cnt) ? bi[j] : 0;
  }
}

void orc_knn_grid(const double* pts, int n, int k, double cell, int32_t* out, int num_threads) {
  num_threads = clamp_threads(num_threads);
  if (n <= 0) return;
  if (k > 64) k = 64;
  double lo[3] = {pts[0], pts[1], pts[2]}, hi[3] = {pts[0], pts[1], pts[2]};
  for (int i = 1; i < n; i++)
    for (int a = 0; a < 3; a++) {
      if (pts[4 * i + a] < lo[a]) lo[a] = pts[4 * i + a];
      if (pts[4 * i + a] > hi[a]) hi[a] = pts[4 * i + a];
    }
  if (cell <= 0.0) {
    /* surface-like clouds: aim for ~2 points per cell assuming a 2-D manifold in the bounding box */
    double ext[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
    double area = ext[0] * ext[1] + ext[1] * ext[2] + ext[0] * ext[2];
    cell = sqrt(2.0 * (2.0 * area + 1e-12) / (double)n);
    if (!(cell > 1e-6)) cell = 1e-6;
  }
  int dim[3];
  for (;;) {
    double cells = 1.0;
    for (int a = 0; a < 3; a++) {
      dim[a] = (int)floor((hi[a] - lo[a]) / cell) + 1;
      cells *= (double)dim[a];
    }
    if (cells <= 64.0e6) break;
    cell *= 1.5;
  }
  const size_t ncell = (size_t)dim[0] * dim[1] * dim[2];
  int32_t* start = (int32_t*)calloc(ncell + 1, sizeof(int32_t));
  int32_t* cid = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  const double inv = 1.0 / cell;
  for (int i = 0; i < n; i++) {
    int c[3];
    for (int a = 0; a < 3; a++) {
      c[a] = (int)floor((pts[4 * i + a] - lo[a]) * inv);
      if (c[a] < 0) c[a] = 0;
      if (c[a] >= dim[a]) c[a] = dim[a] - 1;
    }
    cid[i] = (int32_t)(((size_t)c[2] * dim[1] + c[1]) * dim[0] + c[0]);
    start[cid[i] + 1]++;
  }
  for (size_t c = 0; c < ncell; c++) start[c + 1] += start[c];
  {
    int32_t* cur = (int32_t*)malloc(sizeof(int32_t) * ncell);
    memcpy(cur, start, sizeof(int32_t) * ncell);
    for (int i = 0; i < n; i++) order[cur[cid[i]]++] = i;
    free(cur);
  }
  const int maxring = dim[0] > dim[1] ? (dim[0] > dim[2] ? dim[0] : dim[2]) : (dim[1] > dim[2] ? dim[1] : dim[2]);
#pragma omp parallel for num_threads(num_threads) schedule(guided, 8)
  for (int i = 0; i < n; i++) {
    double bd[64];
    int32_t bi[64];
    int cnt = 0;
    const double* p = pts + 4 * i;
    int c[3];
    {
      int32_t id = cid[i];
      c[0] = id % dim[0];
      c[1] = (id / dim[0]) % dim[1];
      c[2] = id / (dim[0] * dim[1]);
    }
    for (int ring = 0; ring <= maxring; ring++) {
      /* every point outside the (2 ring - 1)^3 cube already scanned is at distance >= (ring-1)*cell + margin;
       * stop once the k-th best is provably inside the scanned region. */
      if (cnt == k && ring >= 1) {
        double margin = 1e300;
        for (int a = 0; a < 3; a++) {
          const double f = (p[a] - lo[a]) - (double)c[a] * cell; /* offset inside own cell */
          const double m0 = f, m1 = cell - f;
          if (m0 < margin) margin = m0;
          if (m1 < margin) margin = m1;
        }
        if (margin < 0.0) margin = 0.0;
        const double reach = (double)(ring - 1) * cell + margin;
        if (bd[k - 1] < reach * reach) break;
      }
      for (int dz = -ring; dz <= ring; dz++) {
        const int z = c[2] + dz;
        if (z < 0 || z >= dim[2]) continue;
        for (int dy = -ring; dy <= ring; dy++) {
          const int y = c[1] + dy;
          if (y < 0 || y >= dim[1]) continue;
          const int on_shell_yz = (abs(dz) == ring) || (abs(dy) == ring);
          for (int dx = -ring; dx <= ring; dx++) {
            if (!on_shell_yz && abs(dx) != ring) continue;
            const int x = c[0] + dx;
            if (x < 0 || x >= dim[0]) continue;
            const size_t cc = ((size_t)z * dim[1] + y) * dim[0] + x;
            for (int32_t s = start[cc]; s < start[cc + 1]; s++) {
              const int32_t j = order[s];
              knn_push(bd, bi, &cnt, k, sqdist3(p, pts + 4 * j), j);
            }
          }
        }
      }
    }
    for (int j = 0; j < k; j++) out[(size_t)i * k + j] = (j < cnt) ? bi[j] : 0; /* see orc_knn_bruteforce */
  }
  free(start);
  free(cid);
  free(order);
}

/* ------------------------------------------------------------------------------------------------ */
/* symmetric 3x3 eigen-solver: restatement of Eigen 3.4 SelfAdjointEigenSolver<Matrix3d>::computeDirect */
/* (third-party dependency of the reference, not in /root/reference; used at                         */
/* src/glim/common/cloud_covariance_estimation.cpp:182-183).  m(r,c) below is symmetric.            */
/* ------------------------------------------------------------------------------------------------ */

static void eig3_roots(const double m[3][3], double roots[3]) {
  const double s_inv3 = 1.0 / 3.0;
  const double s_sqrt3 = sqrt(3.0);
  const double c0 = m[0][0] * m[1][1] * m[2][2] + 2.0 * m[1][0] * m[2][0] * m[2][1] - m[0][0] * m[2][1] * m[2][1] -
                    m[1][1] * m[2][0] * m[2][0] - m[2][2] * m[1][0] * m[1][0];
  const double c1 = m[0][0] * m[1][1] - m[1][0] * m[1][0] + m[0][0] * m[2][2] - m[2][0] * m[2][0] + m[1][1] * m[2][2] -
                    m[2][1] * m[2][1];
  const double c2 = m[0][0] + m[1][1] + m[2][2];
  const double c2_over_3 = c2 * s_inv3;
  double a_over_3 = (c2 * c2_over_3 - c1) * s_inv3;
  if (a_over_3 < 0.0) a_over_3 = 0.0;
  const double half_b = 0.5 * (c0 + c2_over_3 * (2.0 * c2_over_3 * c2_over_3 - c1));
  double q = a_over_3 * a_over_3 * a_over_3 - half_b * half_b;
  if (q < 0.0) q = 0.0;
  const double rho = sqrt(a_over_3);
  const double theta = atan2(sqrt(q), half_b) * s_inv3;
  const double cos_theta = cos(theta);
  const double sin_theta = sin(theta);
  roots[0] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  roots[1] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  roots[2] = c2_over_3 + 2.0 * rho * cos_theta;
}

static void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

/* kernel (null vector) of a rank-2 symmetric matrix; `rep` receives the representative column. */
static void eig3_extract_kernel(const double mat[3][3], double* res, double* rep) {
  int i0 = 0;
  double best = fabs(mat[0][0]);
  for (int i = 1; i < 3; i++)
    if (fabs(mat[i][i]) > best) {
      best = fabs(mat[i][i]);
      i0 = i;
    }
  double col0[3], col1[3], col2[3];
  for (int r = 0; r < 3; r++) {
    col0[r] = mat[r][i0];
    col1[r] = mat[r][(i0 + 1) % 3];
    col2[r] = mat[r][(i0 + 2) % 3];
  }
  rep[0] = col0[0];
  rep[1] = col0[1];
  rep[2] = col0[2];
  double c0[3], c1[3];
  cross3(col0, col1, c0);
  cross3(col0, col2, c1);
  const double n0 = c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2];
  const double n1 = c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2];
  if (n0 > n1) {
    const double s = sqrt(n0);
    res[0] = c0[0] / s;
    res[1] = c0[1] / s;
    res[2] = c0[2] / s;
  } else {
    const double s = sqrt(n1);
    res[0] = c1[0] / s;
    res[1] = c1[1] / s;
    res[2] = c1[2] / s;
  }
}

void orc_eigen3_direct(const double* m9, double* evals, double* evecs /* column-major */) {
  const double eps = 2.220446049250313e-16;
  double sm[3][3];
  /* lower-triangular view mirrored (selfadjointView<Lower>) */
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) sm[r][c] = (r >= c) ? m9[3 * r + c] : m9[3 * c + r];
  const double shift = (sm[0][0] + (sm[1][1] + sm[2][2])) / 3.0; /* mat.trace(): Eigen's unrolled scalar reduction of 3 terms is a0 + (a1 + a2) */
  for (int i = 0; i < 3; i++) sm[i][i] -= shift;
  double scale = 0.0;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)
      if (fabs(sm[r][c]) > scale) scale = fabs(sm[r][c]);
  if (scale > 0.0)
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) sm[r][c] /= scale;

  double ev[3];
  eig3_roots(sm, ev);

  double V[3][3]; /* V[c] = eigenvector c */
  if ((ev[2] - ev[0]) <= eps) {
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) V[c][r] = (r == c) ? 1.0 : 0.0;
  } else {
    double tmp[3][3];
    memcpy(tmp, sm, sizeof(tmp));
    double d0 = ev[2] - ev[1];
    const double d1 = ev[1] - ev[0];
    int k = 0, l = 2;
    if (d0 > d1) {
      k = 2;
      l = 0;
      d0 = d1;
    }
    for (int i = 0; i < 3; i++) tmp[i][i] -= ev[k];
    eig3_extract_kernel(tmp, V[k], V[l]);
    if (d0 <= 2.0 * eps * d1) {
      const double dot = V[k][0] * V[l][0] + V[k][1] * V[l][1] + V[k][2] * V[l][2];
      for (int r = 0; r < 3; r++) V[l][r] -= dot * V[l][r];
      const double nn = sqrt(V[l][0] * V[l][0] + V[l][1] * V[l][1] + V[l][2] * V[l][2]);
      for (int r = 0; r < 3; r++) V[l][r] /= nn;
    } else {
      memcpy(tmp, sm, sizeof(tmp));
      for (int i = 0; i < 3; i++) tmp[i][i] -= ev[l];
      double dummy[3];
      eig3_extract_kernel(tmp, V[l], dummy);
    }
    double c1[3];
    cross3(V[2], V[0], c1);
    const double nn = sqrt(c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2]);
    for (int r = 0; r < 3; r++) V[1][r] = c1[r] / nn;
  }
  for (int i = 0; i < 3; i++) evals[i] = ev[i] * scale + shift;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) evecs[3 * c + r] = V[c][r];
}

/* ------------------------------------------------------------------------------------------------ */
/* covariance + normal -- src/glim/common/cloud_covariance_estimation.cpp:43-122 and :175-196 (PLANE) */
/* ------------------------------------------------------------------------------------------------ */

int orc_covariance_estimate(const double* pts, int n, const int32_t* nbrs, int k_corr, int k_nbr, double* normals,
                            double* covs, int num_threads) {
  if (n <= 0) return 0;
  if (k_nbr > k_corr || k_nbr <= 0) return -1;
  num_threads = clamp_threads(num_threads);
#pragma omp parallel for num_threads(num_threads) schedule(guided, 8)
  for (int i = 0; i < n; i++) {
    /* :81-89  sum of neighbour points and of p p^T (the reference precomputes pt_cross[i] = p p^T, :57-75;
     * the products are the same FP64 values either way) */
    double s[4] = {0, 0, 0, 0};
    double S[4][4];
    memset(S, 0, sizeof(S));
    const int32_t* row = nbrs + (size_t)k_corr * i;
    for (int j = 0; j < k_nbr; j++) {
      const double* p = pts + 4 * (size_t)row[j];
      for (int a = 0; a < 4; a++) {
        s[a] += p[a];
        for (int b = 0; b < 4; b++) S[a][b] += p[a] * p[b];
      }
    }
    /* :91-92  mean and population covariance (divide by k, not k-1) */
    double mean[4], cov[3][3];
    for (int a = 0; a < 4; a++) mean[a] = s[a] / (double)k_nbr;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) cov[a][b] = (S[a][b] - mean[a] * s[b]) / (double)k_nbr;
    /* :181-196  PLANE regularisation: eigenvalues -> (1e-3, 1, 1), recompose V diag V^T */
    double evals[3], V[9];
    orc_eigen3_direct(&cov[0][0], evals, V);
    const double vals[3] = {1e-3, 1.0, 1.0};
    double* C = covs + 16 * (size_t)i;
    memset(C, 0, sizeof(double) * 16);
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) {
        double acc = 0.0;
        for (int e = 0; e < 3; e++) acc += (V[3 * e + r] * vals[e]) * V[3 * e + c];
        C[4 * c + r] = acc; /* column-major 4x4; (3,3) stays 0  (:96) */
      }
    /* :98-101  normal = eigenvector of the smallest eigenvalue, flipped to face the sensor origin */
    double* nrm = normals + 4 * (size_t)i;
    nrm[0] = V[0];
    nrm[1] = V[1];
    nrm[2] = V[2];
    nrm[3] = 0.0;
    const double* p = pts + 4 * (size_t)i;
    if (p[0] * nrm[0] + p[1] * nrm[1] + p[2] * nrm[2] + p[3] * nrm[3] > 0.0) {
      nrm[0] = -nrm[0];
      nrm[1] = -nrm[1];
      nrm[2] = -nrm[2];
      nrm[3] = -nrm[3];
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* Gaussian voxel map (CPU semantics)                                                               */
/* ------------------------------------------------------------------------------------------------ */

typedef struct {
  int32_t coord[3];
  int32_t num_points;
  int finalized;
  int lru; /* insert() counter of the last insert that touched the voxel (IncrementalVoxelMap VoxelInfo::lru, upstream-recall) */
  double mean[4];
  double cov[16];
} orc_voxel;

struct orc_voxelmap {
  double resolution;
  double inv_resolution;
  int num_voxels, cap_voxels;
  orc_voxel* voxels;
  size_t table_size; /* power of two */
  int32_t* table;    /* voxel index or -1 */
  /* least-recently-used eviction of the incremental map (orc_voxelmap_set_lru; 0 = never evict, the default here) */
  int lru_horizon, lru_clear_cycle, lru_counter;
};

/* XOR-prime spatial hash (gtsam_points util/vector3i_hash.hpp, upstream-recall; informative only). */
static inline size_t coord_hash(const int32_t* c) {
  return (size_t)(((int64_t)c[0] * 73856093) ^ ((int64_t)c[1] * 19349669) ^ ((int64_t)c[2] * 83492791));
}

orc_voxelmap* orc_voxelmap_create(double resolution) {
  orc_voxelmap* m = (orc_voxelmap*)calloc(1, sizeof(orc_voxelmap));
  m->resolution = resolution;
  m->inv_resolution = 1.0 / resolution;
  m->cap_voxels = 1024;
  m->voxels = (orc_voxel*)malloc(sizeof(orc_voxel) * (size_t)m->cap_voxels);
  m->table_size = 4096;
  m->table = (int32_t*)malloc(sizeof(int32_t) * m->table_size);
  memset(m->table, 0xff, sizeof(int32_t) * m->table_size);
  return m;
}

void orc_voxelmap_destroy(orc_voxelmap* m) {
  if (!m) return;
  free(m->voxels);
  free(m->table);
  free(m);
}

/* (!) upstream-recall: gtsam_points::IncrementalVoxelMap<GaussianVoxel> (ann/impl/incremental_voxelmap_impl.hpp), which GaussianVoxelMapCPU is.
 * insert(): every point's voxel gets `lru = lru_counter`; after the points, `if ((++lru_counter) % lru_clear_cycle == 0)` every voxel with
 * `lru + lru_horizon < lru_counter` is removed (std::remove_if: the survivors keep their order, so voxel indices are renumbered) and the
 * coordinate index is rebuilt; then every voxel is finalised.  Upstream's defaults are lru_horizon = 10, lru_clear_cycle = 10; GLIM's CPU
 * odometry sets the horizon to lru_thresh = 100 (src/glim/odometry/odometry_estimation_cpu.cpp:63-68, config_odometry_cpu.json:24) and inserts
 * one downsampled frame per update_target (:177-191).  horizon <= 0 switches eviction off (the restatement's default: the GPU maps have none). */
void orc_voxelmap_set_lru(orc_voxelmap* m, int horizon, int clear_cycle) {
  m->lru_horizon = horizon;
  m->lru_clear_cycle = clear_cycle > 0 ? clear_cycle : 10;
}
int orc_voxelmap_lru_counter(const orc_voxelmap* m) { return m->lru_counter; }

int orc_voxelmap_num_voxels(const orc_voxelmap* m) { return m->num_voxels; }
double orc_voxelmap_resolution(const orc_voxelmap* m) { return m->resolution; }

int orc_voxelmap_lookup(const orc_voxelmap* m, const int32_t* c) {
  size_t h = coord_hash(c) & (m->table_size - 1);
  for (;;) {
    const int32_t v = m->table[h];
    if (v < 0) return -1;
    const int32_t* vc = m->voxels[v].coord;
    if (vc[0] == c[0] && vc[1] == c[1] && vc[2] == c[2]) return v;
    h = (h + 1) & (m->table_size - 1);
  }
}

static void voxelmap_rehash(orc_voxelmap* m, size_t new_size) {
  free(m->table);
  m->table_size = new_size;
  m->table = (int32_t*)malloc(sizeof(int32_t) * new_size);
  memset(m->table, 0xff, sizeof(int32_t) * new_size);
  for (int v = 0; v < m->num_voxels; v++) {
    size_t h = coord_hash(m->voxels[v].coord) & (new_size - 1);
    while (m->table[h] >= 0) h = (h + 1) & (new_size - 1);
    m->table[h] = v;
  }
}

static int voxelmap_get_or_create(orc_voxelmap* m, const int32_t* c) {
  size_t h = coord_hash(c) & (m->table_size - 1);
  for (;;) {
    const int32_t v = m->table[h];
    if (v < 0) break;
    const int32_t* vc = m->voxels[v].coord;
    if (vc[0] == c[0] && vc[1] == c[1] && vc[2] == c[2]) return v;
    h = (h + 1) & (m->table_size - 1);
  }
  if (m->num_voxels == m->cap_voxels) {
    m->cap_voxels *= 2;
    m->voxels = (orc_voxel*)realloc(m->voxels, sizeof(orc_voxel) * (size_t)m->cap_voxels);
  }
  const int v = m->num_voxels++;
  orc_voxel* vx = &m->voxels[v];
  memset(vx, 0, sizeof(*vx));
  vx->coord[0] = c[0];
  vx->coord[1] = c[1];
  vx->coord[2] = c[2];
  m->table[h] = v;
  if ((size_t)m->num_voxels * 2 > m->table_size) voxelmap_rehash(m, m->table_size * 2);
  return v;
}

void orc_voxelmap_insert(orc_voxelmap* m, const double* pts, const double* covs, int n) {
  /* GaussianVoxel::add (upstream-recall): a finalized voxel is re-opened (mean *= n, cov *= n) before
   * accumulating; num_points++, mean += p, cov += C. */
  for (int i = 0; i < n; i++) {
    int32_t c[3];
    orc_voxel_coord(pts + 4 * (size_t)i, m->inv_resolution, c);
    const int vid = voxelmap_get_or_create(m, c); /* may realloc m->voxels: index first, then address */
    orc_voxel* vx = &m->voxels[vid];
    if (vx->finalized) {
      vx->finalized = 0;
      for (int a = 0; a < 4; a++) vx->mean[a] *= (double)vx->num_points;
      for (int a = 0; a < 16; a++) vx->cov[a] *= (double)vx->num_points;
    }
    vx->lru = m->lru_counter;
    vx->num_points++;
    for (int a = 0; a < 4; a++) vx->mean[a] += pts[4 * (size_t)i + a];
    for (int a = 0; a < 16; a++) vx->cov[a] += covs[16 * (size_t)i + a];
  }
  m->lru_counter++;
  if (m->lru_horizon > 0 && m->lru_counter % m->lru_clear_cycle == 0) {
    int kept = 0;
    for (int v = 0; v < m->num_voxels; v++) {
      if (m->voxels[v].lru + m->lru_horizon < m->lru_counter) continue; /* least recently used: dropped */
      if (kept != v) m->voxels[kept] = m->voxels[v];
      kept++;
    }
    if (kept != m->num_voxels) {
      m->num_voxels = kept;
      voxelmap_rehash(m, m->table_size);
    }
  }
  /* GaussianVoxel::finalize: mean /= n, cov /= n */
  for (int v = 0; v < m->num_voxels; v++) {
    orc_voxel* vx = &m->voxels[v];
    if (vx->finalized) continue;
    for (int a = 0; a < 4; a++) vx->mean[a] /= (double)vx->num_points;
    for (int a = 0; a < 16; a++) vx->cov[a] /= (double)vx->num_points;
    vx->finalized = 1;
  }
}

void orc_voxelmap_get(const orc_voxelmap* m, int i, int32_t* coord, int32_t* num_points, double* mean, double* cov) {
  const orc_voxel* vx = &m->voxels[i];
  if (coord) memcpy(coord, vx->coord, sizeof(vx->coord));
  if (num_points) *num_points = vx->num_points;
  if (mean) memcpy(mean, vx->mean, sizeof(vx->mean));
  if (cov) memcpy(cov, vx->cov, sizeof(vx->cov));
}

void orc_voxelmap_round_to_f32(orc_voxelmap* m) {
  for (int v = 0; v < m->num_voxels; v++) {
    orc_voxel* vx = &m->voxels[v];
    for (int a = 0; a < 4; a++) vx->mean[a] = (double)(float)vx->mean[a];
    for (int a = 0; a < 16; a++) vx->cov[a] = (double)(float)vx->cov[a];
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* VGICP factor                                                                                      */
/* ------------------------------------------------------------------------------------------------ */

/* inverse of a symmetric positive-definite 3x3 via cofactors (Eigen's fixed-size 3x3 inverse is the same
 * cofactor/determinant form).  The reference inverts the 4x4 [S 0; 0 1]; its 3x3 block is S^-1. */
static void inv3(const double* S, double* M) {
  const double a = S[0], b = S[1], c = S[2], d = S[3], e = S[4], f = S[5], g = S[6], h = S[7], i = S[8];
  const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const double det = a * A + b * B + c * C;
  const double id = 1.0 / det;
  M[0] = A * id;
  M[1] = -(b * i - c * h) * id;
  M[2] = (b * f - c * e) * id;
  M[3] = B * id;
  M[4] = (a * i - c * g) * id;
  M[5] = -(a * f - c * d) * id;
  M[6] = C * id;
  M[7] = -(a * h - b * g) * id;
  M[8] = (a * e - b * d) * id;
}

/* Mahalanobis matrix of one correspondence: (C_B + R C_A R^T)^-1 on the 3x3 block. */
static void fused_mahalanobis(const double* covB16, const double* covA16, const double* T, double* M) {
  double R[9], CA[9], RC[9], S[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      R[3 * r + c] = T[4 * r + c];
      CA[3 * r + c] = covA16[4 * c + r];
    }
  mat3_mul(R, CA, RC);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double s = 0.0;
      for (int k = 0; k < 3; k++) s += RC[3 * r + k] * R[3 * c + k];
      S[3 * r + c] = covB16[4 * c + r] + s;
    }
  inv3(S, M);
}

typedef struct {
  double H_tt[36], H_ss[36], H_ts[36], b_t[6], b_s[6], err;
  int64_t inliers;
} acc_t;

/* accumulate one matched point (evaluate(), upstream-recall) */
static void accumulate_point(acc_t* A, const double* T, const double* p, const double* q, const double* muB, const double* M,
                             int need_H) {
  double r[3] = {muB[0] - q[0], muB[1] - q[1], muB[2] - q[2]};
  double Mr[3];
  for (int a = 0; a < 3; a++) Mr[a] = M[3 * a + 0] * r[0] + M[3 * a + 1] * r[1] + M[3 * a + 2] * r[2];
  A->err += r[0] * Mr[0] + r[1] * Mr[1] + r[2] * Mr[2];
  A->inliers++;
  if (!need_H) return;
  /* J_t = [ -hat(q) | I ],  J_s = [ R hat(p) | -R ]   (3 x 6) */
  double Jt[18], Js[18], Hq[9], Hp[9], R[9], RHp[9];
  hat3(q, Hq);
  hat3(p, Hp);
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) R[3 * a + b] = T[4 * a + b];
  mat3_mul(R, Hp, RHp);
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      Jt[6 * a + b] = -Hq[3 * a + b];
      Jt[6 * a + 3 + b] = (a == b) ? 1.0 : 0.0;
      Js[6 * a + b] = RHp[3 * a + b];
      Js[6 * a + 3 + b] = -R[3 * a + b];
    }
  /* JtM = J_t^T M (6x3), JsM = J_s^T M */
  double JtM[18], JsM[18];
  for (int a = 0; a < 6; a++)
    for (int b = 0; b < 3; b++) {
      double s0 = 0.0, s1 = 0.0;
      for (int k = 0; k < 3; k++) {
        s0 += Jt[6 * k + a] * M[3 * k + b];
        s1 += Js[6 * k + a] * M[3 * k + b];
      }
      JtM[3 * a + b] = s0;
      JsM[3 * a + b] = s1;
    }
  for (int a = 0; a < 6; a++) {
    for (int b = 0; b < 6; b++) {
      double tt = 0.0, ss = 0.0, ts = 0.0;
      for (int k = 0; k < 3; k++) {
        tt += JtM[3 * a + k] * Jt[6 * k + b];
        ss += JsM[3 * a + k] * Js[6 * k + b];
        ts += JtM[3 * a + k] * Js[6 * k + b];
      }
      A->H_tt[6 * a + b] += tt;
      A->H_ss[6 * a + b] += ss;
      A->H_ts[6 * a + b] += ts;
    }
    A->b_t[a] += JtM[3 * a + 0] * r[0] + JtM[3 * a + 1] * r[1] + JtM[3 * a + 2] * r[2];
    A->b_s[a] += JsM[3 * a + 0] * r[0] + JsM[3 * a + 1] * r[1] + JsM[3 * a + 2] * r[2];
  }
}

static void acc_add(acc_t* dst, const acc_t* src) {
  for (int i = 0; i < 36; i++) {
    dst->H_tt[i] += src->H_tt[i];
    dst->H_ss[i] += src->H_ss[i];
    dst->H_ts[i] += src->H_ts[i];
  }
  for (int i = 0; i < 6; i++) {
    dst->b_t[i] += src->b_t[i];
    dst->b_s[i] += src->b_s[i];
  }
  dst->err += src->err;
  dst->inliers += src->inliers;
}

/* Surface validation of IntegratedVGICPFactorGPU::set_enable_surface_validation(true) (src/glim/odometry/odometry_estimation_gpu.cpp:145,162).
 * (!) The upstream predicate lives in gtsam_points (not in /root/reference) and is UNVERIFIED; this oracle states the one DESIGN.md section 4.8
 * documents, in FP64: a correspondence is dropped when the transformed source normal faces away from the target-frame origin,
 *     s_i = (R n_i) . q_i > 0,   q_i = delta p_i,  R = delta's rotation,
 * (the covariance estimator orients n so that p . n <= 0, cloud_covariance_estimation.cpp:98-101).  `force` (optional, n entries) overrides
 * the decision per point: 0 reject, 1 accept, anything else = the predicate -- so a test can hand the FP32 device's decisions for the few
 * points whose |s| is below FP32 resolution and still compare sums.  s_out (optional): s_i of every point. */
typedef struct {
  const double* normals4;
  const int8_t* force;
  double* s_out;
} sv_args;

/* shared driver: correspondences at T_lin, residuals at T_eval */
static void vgicp_run_sv(const orc_voxelmap* map, const double* pts, const double* covs, int n, const double* T_lin,
                         const double* T_eval, int num_threads, int need_H, acc_t* total, int32_t* corr, const sv_args* sv) {
  num_threads = clamp_threads(num_threads);
  acc_t* parts = (acc_t*)calloc((size_t)num_threads, sizeof(acc_t));
#pragma omp parallel num_threads(num_threads)
  {
    /* thread-local accumulator on the thread's own stack, handed over once at the end: the heap array of per-thread sums packs neighbouring
     * threads' hot fields (the tail of one acc_t, the head of the next) into one cache line, and with every point updating both the line
     * bounced between the cores -- two threads were SLOWER than one.  Same additions in the same order. */
    acc_t A_local;
    memset(&A_local, 0, sizeof(A_local));
    acc_t* A = &A_local;
#ifdef _OPENMP
    const int tid = omp_get_thread_num();
#else
    const int tid = 0;
#endif
#pragma omp for schedule(guided, 8) nowait
    for (int i = 0; i < n; i++) {
      const double* p = pts + 4 * (size_t)i;
      double q_lin[3];
      orc_transform_point(T_lin, p, q_lin);
      int32_t c[3];
      orc_voxel_coord(q_lin, map->inv_resolution, c);
      int v = orc_voxelmap_lookup(map, c);
      if (sv) {
        const double* nn = sv->normals4 + 4 * (size_t)i;
        double rn[3];
        for (int r = 0; r < 3; r++) rn[r] = (T_lin[4 * r + 0] * nn[0] + T_lin[4 * r + 1] * nn[1]) + T_lin[4 * r + 2] * nn[2];
        const double s = (rn[0] * q_lin[0] + rn[1] * q_lin[1]) + rn[2] * q_lin[2];
        if (sv->s_out) sv->s_out[i] = s;
        int reject = s > 0.0;
        if (sv->force && (sv->force[i] == 0 || sv->force[i] == 1)) reject = !sv->force[i];
        if (reject) v = -1;
      }
      if (corr) {
        corr[4 * (size_t)i + 0] = c[0];
        corr[4 * (size_t)i + 1] = c[1];
        corr[4 * (size_t)i + 2] = c[2];
        corr[4 * (size_t)i + 3] = v;
      }
      if (v < 0) continue;
      const orc_voxel* vx = &map->voxels[v];
      double M[9];
      fused_mahalanobis(vx->cov, covs + 16 * (size_t)i, T_lin, M);
      double q[3];
      if (T_eval == T_lin) {
        q[0] = q_lin[0];
        q[1] = q_lin[1];
        q[2] = q_lin[2];
      } else {
        orc_transform_point(T_eval, p, q);
      }
      accumulate_point(A, T_eval, p, q, vx->mean, M, need_H);
    }
    parts[tid] = A_local;
  }
  memset(total, 0, sizeof(*total));
  for (int t = 0; t < num_threads; t++) acc_add(total, &parts[t]);
  free(parts);
}

static void vgicp_run(const orc_voxelmap* map, const double* pts, const double* covs, int n, const double* T_lin,
                      const double* T_eval, int num_threads, int need_H, acc_t* total, int32_t* corr) {
  vgicp_run_sv(map, pts, covs, n, T_lin, T_eval, num_threads, need_H, total, corr, NULL);
}

static void acc_to_out(const acc_t* total, orc_linearized6* out) {
  out->num_inliers = total->inliers;
  out->error = ORC_ERROR_SCALE * total->err;
  memcpy(out->H_tt, total->H_tt, sizeof(total->H_tt));
  memcpy(out->H_ss, total->H_ss, sizeof(total->H_ss));
  memcpy(out->H_ts, total->H_ts, sizeof(total->H_ts));
  memcpy(out->b_t, total->b_t, sizeof(total->b_t));
  memcpy(out->b_s, total->b_s, sizeof(total->b_s));
}

int orc_vgicp_linearize_sv(const orc_voxelmap* target, const double* pts, const double* covs, const double* normals4, int n, const double* delta,
                           const int8_t* force, int num_threads, orc_linearized6* out, int32_t* corr, double* s_out) {
  if (!target || !out || !normals4) return -1;
  const sv_args sv = {normals4, force, s_out};
  acc_t total;
  vgicp_run_sv(target, pts, covs, n, delta, delta, num_threads, 1, &total, corr, &sv);
  acc_to_out(&total, out);
  return 0;
}

double orc_vgicp_error_frozen_sv(const orc_voxelmap* target, const double* pts, const double* covs, const double* normals4, int n,
                                 const double* delta_lin, const double* delta_eval, const int8_t* force, int num_threads, int64_t* num_inliers) {
  const sv_args sv = {normals4, force, NULL};
  acc_t total;
  vgicp_run_sv(target, pts, covs, n, delta_lin, delta_eval, num_threads, 0, &total, NULL, &sv);
  if (num_inliers) *num_inliers = total.inliers;
  return ORC_ERROR_SCALE * total.err;
}

int orc_vgicp_linearize(const orc_voxelmap* target, const double* pts, const double* covs, int n, const double* delta,
                        int num_threads, orc_linearized6* out, int32_t* corr) {
  if (!target || !out) return -1;
  acc_t total;
  vgicp_run(target, pts, covs, n, delta, delta, num_threads, 1, &total, corr);
  out->num_inliers = total.inliers;
  out->error = ORC_ERROR_SCALE * total.err;
  memcpy(out->H_tt, total.H_tt, sizeof(total.H_tt));
  memcpy(out->H_ss, total.H_ss, sizeof(total.H_ss));
  memcpy(out->H_ts, total.H_ts, sizeof(total.H_ts));
  memcpy(out->b_t, total.b_t, sizeof(total.b_t));
  memcpy(out->b_s, total.b_s, sizeof(total.b_s));
  return 0;
}

double orc_vgicp_error(const orc_voxelmap* target, const double* pts, const double* covs, int n, const double* delta,
                       int num_threads, int64_t* num_inliers) {
  acc_t total;
  vgicp_run(target, pts, covs, n, delta, delta, num_threads, 0, &total, NULL);
  if (num_inliers) *num_inliers = total.inliers;
  return ORC_ERROR_SCALE * total.err;
}

/* ------------------------------------------------------------------------------------------------ */
/* GICP factor: gtsam_points::IntegratedGICPFactor as constructed at          */
/* src/glim/mapping/sub_mapping.cpp:202, global_mapping.cpp:400, global_mapping_pose_graph.cpp:393  */
/* ------------------------------------------------------------------------------------------------ */
/* (!) upstream-recall (src/gtsam_points/factors/impl/integrated_gicp_factor_impl.hpp): update_correspondences() finds, for every
 * source point, the nearest target point of q = delta p with a kd-tree (k = 1) and drops it when the squared distance exceeds
 * max_correspondence_distance^2; the Mahalanobis matrix is (C_B + R C_A R^T)^-1 with the TARGET POINT's covariance; evaluate()
 * is the VGICP expression with mu_B = the matched target point.  Oracle rules where upstream is implementation defined:
 * exact nearest neighbour by FP64 (dx^2 + dy^2) + dz^2 without contraction, ties to the smaller target index; a match is kept
 * when d^2 <= max^2.  corr (optional): n_s matched target indices or -1. */
static void gicp_run(const double* tpts, const double* tcovs, int nt, const double* pts, const double* covs, int n, const double* T,
                     double max_sq, int num_threads, int need_H, acc_t* total, int32_t* corr) {
  num_threads = clamp_threads(num_threads);
  acc_t* parts = (acc_t*)calloc((size_t)num_threads, sizeof(acc_t));
#pragma omp parallel num_threads(num_threads)
  {
    /* thread-local accumulator on the thread's own stack, handed over once at the end: the heap array of per-thread sums packs neighbouring
     * threads' hot fields (the tail of one acc_t, the head of the next) into one cache line, and with every point updating both the line
     * bounced between the cores -- two threads were SLOWER than one.  Same additions in the same order. */
    acc_t A_local;
    memset(&A_local, 0, sizeof(A_local));
    acc_t* A = &A_local;
#ifdef _OPENMP
    const int tid = omp_get_thread_num();
#else
    const int tid = 0;
#endif
#pragma omp for schedule(guided, 8) nowait
    for (int i = 0; i < n; i++) {
      const double* p = pts + 4 * (size_t)i;
      double q[3];
      orc_transform_point(T, p, q);
      int best = -1;
      double best_d = INFINITY;
      for (int j = 0; j < nt; j++) {
        const double d = sqdist3(q, tpts + 4 * (size_t)j);
        if (d < best_d) {
          best_d = d;
          best = j;
        }
      }
      if (best >= 0 && !(best_d <= max_sq)) best = -1;
      if (corr) corr[i] = best;
      if (best < 0) continue;
      double M[9];
      fused_mahalanobis(tcovs + 16 * (size_t)best, covs + 16 * (size_t)i, T, M);
      accumulate_point(A, T, p, q, tpts + 4 * (size_t)best, M, need_H);
    }
    parts[tid] = A_local;
  }
  memset(total, 0, sizeof(*total));
  for (int t = 0; t < num_threads; t++) acc_add(total, &parts[t]);
  free(parts);
}

int orc_gicp_linearize(const double* target_points4, const double* target_covs16, int nt, const double* src_points4, const double* src_covs16, int n,
                       const double* delta, double max_correspondence_distance, int num_threads, orc_linearized6* out, int32_t* corr) {
  if (!out) return -1;
  acc_t total;
  gicp_run(target_points4, target_covs16, nt, src_points4, src_covs16, n, delta, max_correspondence_distance * max_correspondence_distance, num_threads, 1,
           &total, corr);
  out->num_inliers = total.inliers;
  out->error = ORC_ERROR_SCALE * total.err;
  memcpy(out->H_tt, total.H_tt, sizeof(total.H_tt));
  memcpy(out->H_ss, total.H_ss, sizeof(total.H_ss));
  memcpy(out->H_ts, total.H_ts, sizeof(total.H_ts));
  memcpy(out->b_t, total.b_t, sizeof(total.b_t));
  memcpy(out->b_s, total.b_s, sizeof(total.b_s));
  return 0;
}

double orc_gicp_error(const double* target_points4, const double* target_covs16, int nt, const double* src_points4, const double* src_covs16, int n,
                      const double* delta, double max_correspondence_distance, int num_threads, int64_t* num_inliers) {
  acc_t total;
  gicp_run(target_points4, target_covs16, nt, src_points4, src_covs16, n, delta, max_correspondence_distance * max_correspondence_distance, num_threads, 0,
           &total, NULL);
  if (num_inliers) *num_inliers = total.inliers;
  return ORC_ERROR_SCALE * total.err;
}

double orc_vgicp_error_frozen(const orc_voxelmap* target, const double* pts, const double* covs, int n, const double* delta_lin,
                              const double* delta_eval, int num_threads, int64_t* num_inliers) {
  acc_t total;
  vgicp_run(target, pts, covs, n, delta_lin, delta_eval, num_threads, 0, &total, NULL);
  if (num_inliers) *num_inliers = total.inliers;
  return ORC_ERROR_SCALE * total.err;
}

double orc_overlap(const orc_voxelmap* const* targets, const double* deltas, int num_targets, const double* pts, int n,
                   int num_threads) {
  if (n <= 0) return 0.0;
  num_threads = clamp_threads(num_threads);
  long hits = 0;
#pragma omp parallel for num_threads(num_threads) schedule(guided, 8) reduction(+ : hits)
  for (int i = 0; i < n; i++) {
    for (int t = 0; t < num_targets; t++) {
      double q[3];
      orc_transform_point(deltas + 12 * (size_t)t, pts + 4 * (size_t)i, q);
      int32_t c[3];
      orc_voxel_coord(q, targets[t]->inv_resolution, c);
      if (orc_voxelmap_lookup(targets[t], c) >= 0) {
        hits++;
        break;
      }
    }
  }
  return (double)hits / (double)n;
}

/* ------------------------------------------------------------------------------------------------ */
/* one optimisation step                                                                             */
/* ------------------------------------------------------------------------------------------------ */

int orc_solve6(const double* H, const double* b, double lambda, double* x) {
  double L[36];
  memset(L, 0, sizeof(L));
  for (int i = 0; i < 6; i++) {
    for (int j = 0; j <= i; j++) {
      double s = 0.5 * (H[6 * i + j] + H[6 * j + i]) + ((i == j) ? lambda : 0.0);
      for (int k = 0; k < j; k++) s -= L[6 * i + k] * L[6 * j + k];
      if (i == j) {
        if (!(s > 0.0)) return -1;
        L[6 * i + i] = sqrt(s);
      } else {
        L[6 * i + j] = s / L[6 * j + j];
      }
    }
  }
  double y[6];
  for (int i = 0; i < 6; i++) {
    double s = -b[i];
    for (int k = 0; k < i; k++) s -= L[6 * i + k] * y[k];
    y[i] = s / L[6 * i + i];
  }
  for (int i = 5; i >= 0; i--) {
    double s = y[i];
    for (int k = i + 1; k < 6; k++) s -= L[6 * k + i] * x[k];
    x[i] = s / L[6 * i + i];
  }
  return 0;
}

int orc_gn_align(const orc_voxelmap* target, const double* pts, const double* covs, int n, double* T, int max_iters,
                 double lambda, int num_threads, double* deltas_out) {
  int it = 0;
  for (; it < max_iters; it++) {
    orc_linearized6 L;
    orc_vgicp_linearize(target, pts, covs, n, T, num_threads, &L, NULL);
    double d[6];
    if (orc_solve6(L.H_ss, L.b_s, lambda, d) != 0) break;
    if (deltas_out) memcpy(deltas_out + 6 * it, d, sizeof(d));
    double E[12];
    orc_se3_exp(d, E);
    orc_pose_compose(T, E, T);
    const double dt = sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    const double dr = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (dt < 1e-3 && dr < 1e-3 * 3.14159265358979323846 / 180.0) {
      it++;
      break;
    }
  }
  return it;
}

/* ------------------------------------------------------------------------------------------------ */
/* deskewing -- src/glim/common/cloud_deskewing.cpp:11-53 (constant velocity) and :55-133 (IMU poses)  */
/* (the step between preprocessing and covariance estimation,                  */
/*  src/glim/odometry/odometry_estimation_imu.cpp:313-316)                                           */
/* ------------------------------------------------------------------------------------------------ */

/* time table: a new entry whenever times[i] - table.back() > 1e-4; time_indices[i] = current last entry (:24-36, :72-84) */
static int build_time_table(const double* times, int n, double** table_out, int32_t* indices) {
  const double time_eps = 1e-4;
  double* table = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  int m = 0;
  for (int i = 0; i < n; i++) {
    if (m == 0 || times[i] - table[m - 1] > time_eps) table[m++] = times[i];
    indices[i] = m - 1;
  }
  *table_out = table;
  return m;
}

/* Eigen::Quaterniond(Matrix3d) -- Eigen 3.4 quaternionbase_assign_impl<Other,3,3> (w,x,y,z in q[3],q[0..2]) */
static void quat_from_rot(const double* T12, double* q /* x y z w */) {
  double m[3][3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) m[r][c] = T12[4 * r + c];
  double t = m[0][0] + (m[1][1] + m[2][2]); /* mat.trace(), see orc_eigen3_direct */
  if (t > 0.0) {
    t = sqrt(t + 1.0);
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
    t = sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m[k][j] - m[j][k]) * t;
    q[j] = (m[j][i] + m[i][j]) * t;
    q[k] = (m[k][i] + m[i][k]) * t;
  }
}

/* Eigen::Quaterniond::slerp (Eigen 3.4) */
static void quat_slerp(const double* a, double t, const double* b, double* out) {
  const double one = 1.0 - 2.220446049250313e-16;
  const double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  const double absD = fabs(d);
  double scale0, scale1;
  if (absD >= one) {
    scale0 = 1.0 - t;
    scale1 = t;
  } else {
    const double theta = acos(absD);
    const double sinTheta = sin(theta);
    scale0 = sin((1.0 - t) * theta) / sinTheta;
    scale1 = sin(t * theta) / sinTheta;
  }
  if (d < 0.0) scale1 = -scale1;
  for (int i = 0; i < 4; i++) out[i] = scale0 * a[i] + scale1 * b[i];
}

/* Eigen::Quaterniond::toRotationMatrix */
static void quat_to_rot(const double* q, double* R /* row-major 3x3 */) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

static void apply_pose(const double* T, const double* p4, double* out4) {
  for (int r = 0; r < 3; r++) out4[r] = ((T[4 * r + 0] * p4[0] + T[4 * r + 1] * p4[1]) + T[4 * r + 2] * p4[2]) + T[4 * r + 3] * p4[3];
  out4[3] = p4[3];
}

/* gtsam::Pose3::Expmap as GTSAM 4.2 evaluates it (cloud_deskewing.cpp:43 calls it): R = so3::ExpmapFunctor(omega).expmap() = I + sin(theta) K +
 * (1 - cos(theta)) K^2 with K = hat(omega) / theta and 1 - cos(theta) = 2 sin^2(theta / 2) (first order I + hat(omega) when theta^2 <= eps);
 * t = (omega x v - R (omega x v) + omega (omega . v)) / theta^2, or v when theta^2 <= eps.  Same operation order as the stand-in
 * oracle/ref_standin/gtsam/geometry/Pose3.h, so that orc_deskew_constvel == the compiled reference (oracle/_ref) bit for bit. */
static void pose3_expmap_gtsam(const double* xi, double* T) {
  const double eps = 2.220446049250313e-16;
  const double w[3] = {xi[0], xi[1], xi[2]}, v[3] = {xi[3], xi[4], xi[5]};
  const double theta2 = (w[0] * w[0] + w[1] * w[1]) + w[2] * w[2];
  double W[9], R[9];
  hat3(w, W);
  if (theta2 <= eps) {
    for (int i = 0; i < 9; i++) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + W[i];
  } else {
    const double theta = sqrt(theta2);
    const double sin_theta = sin(theta);
    const double s2 = sin(theta / 2.0);
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
    double c[3], Rc[3];
    cross3(w, v, c);
    for (int r = 0; r < 3; r++) Rc[r] = (R[3 * r] * c[0] + R[3 * r + 1] * c[1]) + R[3 * r + 2] * c[2];
    for (int r = 0; r < 3; r++) t[r] = ((c[r] - Rc[r]) + w[r] * wv) / theta2;
  } else {
    t[0] = v[0]; t[1] = v[1]; t[2] = v[2];
  }
  for (int r = 0; r < 3; r++) {
    T[4 * r + 0] = R[3 * r + 0];
    T[4 * r + 1] = R[3 * r + 1];
    T[4 * r + 2] = R[3 * r + 2];
    T[4 * r + 3] = t[r];
  }
}

int orc_deskew_constvel(const double* T_imu_lidar, const double* linear_vel, const double* angular_vel, const double* times,
                        const double* points4, int n, double* out4) {
  if (n <= 0) return 0;
  int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  double* table = NULL;
  const int m = build_time_table(times, n, &table, idx);
  double T_lidar_imu[12];
  orc_pose_inverse(T_imu_lidar, T_lidar_imu);
  double* TT = (double*)malloc(sizeof(double) * 12 * (size_t)m);
  for (int i = 0; i < m; i++) {
    const double dt = table[i];
    const double xi[6] = {dt * angular_vel[0], dt * angular_vel[1], dt * angular_vel[2], dt * linear_vel[0], dt * linear_vel[1], dt * linear_vel[2]};
    double T_imu1_imu0[12], inv[12], tmp[12];
    pose3_expmap_gtsam(xi, T_imu1_imu0);          /* :43 */
    orc_pose_inverse(T_imu1_imu0, inv);
    orc_pose_compose(T_lidar_imu, inv, tmp);      /* :44  T_lidar_imu * T_imu1_imu0^-1 * T_imu_lidar */
    orc_pose_compose(tmp, T_imu_lidar, TT + 12 * (size_t)i);
  }
  for (int i = 0; i < n; i++) apply_pose(TT + 12 * (size_t)idx[i], points4 + 4 * (size_t)i, out4 + 4 * (size_t)i); /* :47-51 */
  free(TT);
  free(table);
  free(idx);
  return 0;
}

int orc_deskew_imu(const double* T_imu_lidar, const double* imu_times, const double* imu_poses12, int n_imu, double stamp,
                   const double* times, const double* points4, int n, double* out4) {
  if (n <= 0) return 0;
  if (n_imu <= 0) { /* :66-68 */
    const double zero[3] = {0, 0, 0};
    return orc_deskew_constvel(T_imu_lidar, zero, zero, times, points4, n, out4);
  }
  int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  double* table = NULL;
  const int m = build_time_table(times, n, &table, idx);
  double T_lidar_imu[12];
  orc_pose_inverse(T_imu_lidar, T_lidar_imu);
  double* TT = (double*)malloc(sizeof(double) * 12 * (size_t)m);
  int cursor = 0;
  double T_imu0_world[12];
  for (int i = 0; i < m; i++) {
    const double time = stamp + table[i];
    while (cursor < n_imu - 1 && imu_times[cursor + 1] < time) cursor++; /* :95-97 */
    if (i == 0) orc_pose_inverse(imu_poses12 + 12 * (size_t)cursor, T_imu0_world); /* :99-102 */
    double T_world_imu1[12];
    if (cursor + 1 >= n_imu) {
      memcpy(T_world_imu1, imu_poses12 + 12 * (size_t)cursor, sizeof(double) * 12); /* :105-106 */
    } else {
      const double t0 = imu_times[cursor], t1 = imu_times[cursor + 1];
      double p = (time - t0) / (t1 - t0);
      p = p < 1.0 ? p : 1.0; /* std::max(0, std::min(1, .)) :111 */
      p = p > 0.0 ? p : 0.0;
      const double* L = imu_poses12 + 12 * (size_t)cursor;
      const double* Rr = imu_poses12 + 12 * (size_t)(cursor + 1);
      double ql[4], qr[4], qs[4], R[9];
      quat_from_rot(L, ql);
      quat_from_rot(Rr, qr);
      quat_slerp(ql, p, qr, qs);
      quat_to_rot(qs, R);
      for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T_world_imu1[4 * r + c] = R[3 * r + c];
        T_world_imu1[4 * r + 3] = (1.0 - p) * L[4 * r + 3] + p * Rr[4 * r + 3]; /* :118 */
      }
    }
    double T_imu0_imu1[12], tmp[12];
    orc_pose_compose(T_imu0_world, T_world_imu1, T_imu0_imu1);  /* :122 */
    orc_pose_compose(T_lidar_imu, T_imu0_imu1, tmp);            /* :123 */
    orc_pose_compose(tmp, T_imu_lidar, TT + 12 * (size_t)i);
  }
  for (int i = 0; i < n; i++) apply_pose(TT + 12 * (size_t)idx[i], points4 + 4 * (size_t)i, out4 + 4 * (size_t)i); /* :127-130 */
  free(TT);
  free(table);
  free(idx);
  return 0;
}

/* `pt = T_imu_lidar * pt` for every deskewed point (src/glim/odometry/odometry_estimation_imu.cpp:314-316, src/glim/mapping/sub_mapping.cpp:368-370):
 * Eigen::Isometry3d * Eigen::Vector4d, the same product CloudDeskewing::deskew applies (apply_pose). */
int orc_transform_points(const double* T12, const double* points4, int n, double* out4) {
  for (int i = 0; i < n; i++) apply_pose(T12, points4 + 4 * (size_t)i, out4 + 4 * (size_t)i);
  return 0;
}
