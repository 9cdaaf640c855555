/* preprocess_oracle.c -- CPU restatement of GLIM's scan preprocessing.
 *
 * TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see vgicp_oracle.h).  Restates
 *   glim::CloudPreprocessor::preprocess_impl      /root/reference/src/glim/preprocess/cloud_preprocessor.cpp:92-188
 * and the three routines of the un-vendored dependency koide3/gtsam_points (>= 1.2.2, CMakeLists.txt:28) it calls:
 *   gtsam_points::voxelgrid_sampling   (cloud_preprocessor.cpp:108)   upstream src/gtsam_points/types/point_cloud_cpu_funcs.cpp
 *   gtsam_points::randomgrid_sampling  (cloud_preprocessor.cpp:106)   same file
 *   gtsam_points::remove_outliers      (cloud_preprocessor.cpp:162)   same file
 * Their published algorithms are restated from the upstream sources as recalled; every point that could not be checked
 * against the real library in this container is marked (!):
 *   (!) voxel key = 3 x 21-bit coordinates fast_floor(p / res) + 2^20, x lowest; points with a non-finite coordinate or a
 *       coordinate outside the 21-bit range are dropped.
 *   (!) both samplers sort the points by voxel key with an UNSTABLE parallel quick sort, so upstream the order of the
 *       points inside a voxel (hence the last bits of a voxel's sum) is implementation defined.  Oracle rule: ascending
 *       (key, original index).
 *   (!) voxelgrid_sampling averages in blocks of 1024 sorted entries, so a voxel that straddles a block boundary yields one
 *       output point per block; `voxelgrid_block_size` (1024; 0 = never split) reproduces that.  Upstream appends the blocks
 *       in thread-completion order; oracle rule: ascending key (the caller re-sorts by time anyway).
 *   (!) randomgrid_sampling keeps at most points_per_voxel = ceil(rate N / num_voxels) points of every voxel (all of them if
 *       the voxel holds fewer), then, if more than floor(1.2 rate N) points survived, a uniform sample of exactly that many,
 *       and returns them in ascending original index; rate >= 0.99 returns the cloud unchanged.  Upstream draws with
 *       std::sample from a std::mt19937 (per 1024-entry block); its stream cannot be reproduced by a data-parallel
 *       implementation, so parity with upstream is statistical only.  Oracle rule: a counter-based generator
 *       h = splitmix64(seed, original index); inside a voxel the points with the smallest (h >> 32, index) are kept, the
 *       global cap keeps the smallest (h & 0xffffffff, index).  The per-block split is not reproduced for this sampler.
 *   (!) remove_outliers: d_i = mean Euclidean distance to the k nearest neighbours (query included), threshold
 *       mean(d) + std_mul * sqrt(mean(d^2) - mean(d)^2), keep d_i < threshold.
 * Arithmetic that decides an inequality is written in one fixed order without contraction so that the HIP path can
 * reproduce it bit for bit. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "vgicp_oracle.h"

uint64_t orc_sample_hash(uint64_t seed, uint64_t index) {
  uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ull; /* splitmix64 */
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

#define ORC_INVALID_KEY 0xFFFFFFFFFFFFFFFFull

/* (!) voxel key; returns ORC_INVALID_KEY for a point that has no voxel */
uint64_t orc_sampling_key(const double* p4, double inv_res) {
  uint64_t key = 0;
  for (int a = 0; a < 3; a++) {
    const double t = p4[a] * inv_res;
    if (!(t >= -1048576.0 && t < 1048576.0)) return ORC_INVALID_KEY; /* also catches NaN / inf */
    const int64_t c = (int64_t)orc_fast_floor(t) + 1048576;
    key |= ((uint64_t)c & 0x1FFFFFull) << (21 * a);
  }
  if (!isfinite(p4[3])) return ORC_INVALID_KEY; /* points[i].array().isFinite().all() */
  return key;
}

typedef struct {
  uint64_t key;
  uint64_t aux;
  int32_t idx;
} orc_sort_entry;

static int cmp_entry(const void* a, const void* b) {
  const orc_sort_entry* x = (const orc_sort_entry*)a;
  const orc_sort_entry* y = (const orc_sort_entry*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  if (x->aux != y->aux) return x->aux < y->aux ? -1 : 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}

static orc_sort_entry* sorted_by_key(const double* points4, int n, double res, uint64_t seed, int with_hash, int* n_valid) {
  orc_sort_entry* e = (orc_sort_entry*)malloc(sizeof(orc_sort_entry) * (size_t)(n > 0 ? n : 1));
  const double inv_res = 1.0 / res;
  for (int i = 0; i < n; i++) {
    e[i].key = orc_sampling_key(points4 + 4 * (size_t)i, inv_res);
    e[i].aux = with_hash ? (orc_sample_hash(seed, (uint64_t)i) >> 32) : 0;
    e[i].idx = i;
  }
  qsort(e, (size_t)n, sizeof(orc_sort_entry), cmp_entry);
  int v = n;
  while (v > 0 && e[v - 1].key == ORC_INVALID_KEY) v--;
  *n_valid = v;
  return e;
}

int orc_voxelgrid_sampling(const double* points4, const double* times, const double* intensities, int n, double resolution, int block_size,
                           double* out_points4, double* out_times, double* out_intensities) {
  int nv = 0, m = 0;
  orc_sort_entry* e = sorted_by_key(points4, n, resolution, 0, 0, &nv);
  int pos = 0;
  while (pos < nv) {
    double s[4] = {0, 0, 0, 0}, st = 0.0, si = 0.0;
    int end = pos;
    do { /* one run: equal keys, not crossing a multiple of block_size */
      const double* p = points4 + 4 * (size_t)e[end].idx;
      for (int a = 0; a < 4; a++) s[a] += p[a];
      if (times) st += times[e[end].idx];
      if (intensities) si += intensities[e[end].idx];
      end++;
    } while (end < nv && e[end].key == e[pos].key && !(block_size > 0 && end % block_size == 0));
    for (int a = 0; a < 4; a++) out_points4[4 * (size_t)m + a] = s[a] / s[3];
    if (times) out_times[m] = st / s[3];
    if (intensities) out_intensities[m] = si / s[3];
    m++;
    pos = end;
  }
  free(e);
  return m;
}

static int cmp_u64_idx(const void* a, const void* b) { return cmp_entry(a, b); }

int orc_randomgrid_sampling(const double* points4, int n, double resolution, double rate, uint64_t seed, int32_t* out_indices) {
  if (rate >= 0.99) {
    for (int i = 0; i < n; i++) out_indices[i] = i;
    return n;
  }
  int nv = 0;
  orc_sort_entry* e = sorted_by_key(points4, n, resolution, seed, 1, &nv);
  int64_t num_voxels = 0;
  for (int i = 0; i < nv; i++) num_voxels += (i == 0 || e[i].key != e[i - 1].key);
  if (num_voxels == 0) {
    free(e);
    return 0;
  }
  const int64_t ppv = (int64_t)ceil((rate * (double)n) / (double)num_voxels);
  const int64_t max_num_points = (int64_t)((double)n * rate * 1.2);
  orc_sort_entry* sel = (orc_sort_entry*)malloc(sizeof(orc_sort_entry) * (size_t)(nv > 0 ? nv : 1));
  int64_t m = 0;
  for (int i = 0; i < nv; i++) {
    /* rank inside the voxel < ppv  <=>  the entry ppv places earlier belongs to another voxel */
    if (i < ppv || e[i - ppv].key != e[i].key) {
      sel[m].key = orc_sample_hash(seed, (uint64_t)e[i].idx) & 0xFFFFFFFFull;
      sel[m].aux = 0;
      sel[m].idx = e[i].idx;
      m++;
    }
  }
  if (m > max_num_points) {
    qsort(sel, (size_t)m, sizeof(orc_sort_entry), cmp_u64_idx);
    m = max_num_points;
  }
  for (int64_t i = 0; i < m; i++) {
    sel[i].key = 0;
    sel[i].aux = 0;
  }
  qsort(sel, (size_t)m, sizeof(orc_sort_entry), cmp_u64_idx); /* ascending original index */
  for (int64_t i = 0; i < m; i++) out_indices[i] = sel[i].idx;
  free(sel);
  free(e);
  return (int)m;
}

int orc_find_inliers(const double* points4, int n, int k, double std_mul, int32_t* out_indices, int num_threads) {
  if (n == 0) return 0;
  int32_t* nb = (int32_t*)malloc(sizeof(int32_t) * (size_t)n * (size_t)k);
  double* d = (double*)malloc(sizeof(double) * (size_t)n);
  orc_knn_grid(points4, n, k, 0.0, nb, num_threads);
  double sum = 0.0, sum_sq = 0.0;
  for (int i = 0; i < n; i++) {
    double s = 0.0;
    for (int j = 0; j < k; j++) {
      const double* p = points4 + 4 * (size_t)i;
      const double* q = points4 + 4 * (size_t)nb[(size_t)i * k + j];
      const double dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
      s += sqrt((dx * dx + dy * dy) + dz * dz);
    }
    d[i] = s / (double)k;
    sum += d[i];
    sum_sq += d[i] * d[i];
  }
  const double mean = sum / (double)n;
  const double var = sum_sq / (double)n - mean * mean;
  const double thresh = mean + std_mul * sqrt(var > 0.0 ? var : 0.0);
  int m = 0;
  for (int i = 0; i < n; i++)
    if (d[i] < thresh) out_indices[m++] = i;
  free(nb);
  free(d);
  return m;
}

typedef struct {
  double t;
  int32_t idx;
} time_entry;

static int cmp_time(const void* a, const void* b) {
  const time_entry* x = (const time_entry*)a;
  const time_entry* y = (const time_entry*)b;
  if (x->t != y->t) return x->t < y->t ? -1 : 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}

/* distance / finite / cropbox predicate on one point (cloud_preprocessor.cpp:122-128, :146-160 -- the reference applies the
 * cropbox after the time sort; a per-point predicate commutes with the sort) */
int orc_preprocess_keep(const double* p, const orc_preprocess_params* prm) {
  const int finite = isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]) && isfinite(p[3]);
  const double d2 = (p[0] * p[0] + p[2] * p[2]) + p[1] * p[1]; /* squaredNorm of (x y z 0) as Eigen's packet reduction adds it */
  const double near2 = prm->distance_near_thresh * prm->distance_near_thresh, far2 = prm->distance_far_thresh * prm->distance_far_thresh;
  if (!(d2 > near2 && d2 < far2 && finite)) return 0;
  if (prm->enable_cropbox_filter) {
    double q[3] = {p[0], p[1], p[2]};
    if (prm->crop_bbox_frame_imu) {
      const double* T = prm->T_imu_lidar;
      for (int r = 0; r < 3; r++) q[r] = ((T[4 * r] * p[0] + T[4 * r + 1] * p[1]) + T[4 * r + 2] * p[2]) + T[4 * r + 3];
    }
    int inside = 1;
    for (int a = 0; a < 3; a++) inside &= (q[a] >= prm->crop_bbox_min[a]) && (q[a] <= prm->crop_bbox_max[a]);
    if (inside) return 0;
  }
  return 1;
}

int orc_preprocess(const double* points4, const double* times, const double* intensities, int n, const orc_preprocess_params* prm,
                   double* out_points4, double* out_times, double* out_intensities, int32_t* out_neighbors, int num_threads) {
  /* ---- downsampling (cloud_preprocessor.cpp:103-109) ---- */
  double* P = (double*)malloc(sizeof(double) * 4 * (size_t)(n > 0 ? n : 1));
  double* T = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  double* I = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  int m;
  if (prm->use_random_grid_downsampling) {
    const double rate = prm->downsample_target > 0 ? (double)prm->downsample_target / (double)n : prm->downsample_rate;
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    m = n > 0 ? orc_randomgrid_sampling(points4, n, prm->downsample_resolution, rate, prm->seed, idx) : 0;
    for (int i = 0; i < m; i++) {
      memcpy(P + 4 * (size_t)i, points4 + 4 * (size_t)idx[i], 4 * sizeof(double));
      T[i] = times[idx[i]];
      I[i] = intensities ? intensities[idx[i]] : 0.0;
    }
    free(idx);
  } else {
    m = orc_voxelgrid_sampling(points4, times, intensities, n, prm->downsample_resolution, prm->voxelgrid_block_size, P, T, I);
  }
  /* ---- distance filter (:117-128), sort by time (:134-136), global shutter (:138-140), cropbox (:143-160) ---- */
  time_entry* te = (time_entry*)malloc(sizeof(time_entry) * (size_t)(m > 0 ? m : 1));
  int f = 0;
  for (int i = 0; i < m; i++) {
    if (orc_preprocess_keep(P + 4 * (size_t)i, prm)) {
      te[f].t = T[i];
      te[f].idx = i;
      f++;
    }
  }
  qsort(te, (size_t)f, sizeof(time_entry), cmp_time); /* std::sort is unstable upstream; oracle rule: (time, position) */
  double* Q = (double*)malloc(sizeof(double) * 4 * (size_t)(f > 0 ? f : 1));
  double* QT = (double*)malloc(sizeof(double) * (size_t)(f > 0 ? f : 1));
  double* QI = (double*)malloc(sizeof(double) * (size_t)(f > 0 ? f : 1));
  for (int i = 0; i < f; i++) {
    memcpy(Q + 4 * (size_t)i, P + 4 * (size_t)te[i].idx, 4 * sizeof(double));
    QT[i] = prm->global_shutter ? 0.0 : T[te[i].idx];
    QI[i] = I[te[i].idx];
  }
  /* ---- outlier removal (:162-164) ---- */
  int out_n = f;
  if (prm->enable_outlier_removal && f > 0) {
    int32_t* keep = (int32_t*)malloc(sizeof(int32_t) * (size_t)f);
    out_n = orc_find_inliers(Q, f, prm->outlier_removal_k, prm->outlier_std_mul_factor, keep, num_threads);
    for (int i = 0; i < out_n; i++) { /* keep[] ascending: in-place compaction is safe */
      memmove(Q + 4 * (size_t)i, Q + 4 * (size_t)keep[i], 4 * sizeof(double));
      QT[i] = QT[keep[i]];
      QI[i] = QI[keep[i]];
    }
    free(keep);
  }
  memcpy(out_points4, Q, sizeof(double) * 4 * (size_t)out_n);
  memcpy(out_times, QT, sizeof(double) * (size_t)out_n);
  if (out_intensities) memcpy(out_intensities, QI, sizeof(double) * (size_t)out_n);
  /* ---- kNN for the covariances (:183-184) ---- */
  if (out_neighbors && out_n > 0) orc_knn_grid(Q, out_n, prm->k_correspondences, 0.0, out_neighbors, num_threads);
  free(P); free(T); free(I); free(te); free(Q); free(QT); free(QI);
  return out_n;
}

/* ---- submap merge: gtsam_points::merge_frames as called at src/glim/mapping/sub_mapping.cpp:480-497 ----
 * (!) restated from the upstream algorithm as recalled (src/gtsam_points/types/point_cloud_cpu_funcs.cpp): every frame is
 *     moved into the submap origin -- p' = T p, C' = T C T^T with the 4x4 matrices (top-left block R C R^T) -- the frames are
 *     concatenated in order and voxelgrid_sampling averages points AND covariances per voxel (the covariance of a voxel is the
 *     plain mean of its members' covariances).
 * (!) the 4th argument (submap_target_num_points, config_sub_mapping_gpu.json:54) is taken to mean what the in-tree sibling
 *     sub_mapping_passthrough.cpp:149-151 does with the parameter of the same name: if the merged cloud is larger, a uniform
 *     random sample of floor(size * (target / size)) points in their original order (gtsam_points::random_sampling).  Oracle
 *     rule for the draw: the points with the smallest (orc_sample_hash(seed, index) >> 32, index).
 * Oracle rules where upstream is implementation defined: R C R^T is evaluated as (R C) R^T, rows by columns, left to right,
 * no contraction, and only its upper triangle is used (the product is symmetric up to rounding); order inside a voxel =
 * (frame, index). */
int orc_merge_frames(int num_frames, const double* poses12, const double* const* points4, const double* const* covs16, const int* sizes,
                     double resolution, int block_size, int target_num_points, uint64_t seed, double* out_points4, double* out_covs16) {
  int total = 0;
  for (int f = 0; f < num_frames; f++) total += sizes[f];
  double* P = (double*)malloc(sizeof(double) * 4 * (size_t)(total > 0 ? total : 1));
  double* C = (double*)malloc(sizeof(double) * 6 * (size_t)(total > 0 ? total : 1)); /* c00 c01 c02 c11 c12 c22 */
  int at = 0;
  for (int f = 0; f < num_frames; f++) {
    const double* T = poses12 + 12 * (size_t)f;
    for (int i = 0; i < sizes[f]; i++, at++) {
      const double* p = points4[f] + 4 * (size_t)i;
      const double* c = covs16[f] + 16 * (size_t)i; /* column-major Matrix4d; symmetric: (r, c) = c[4 * c + r] */
      for (int r = 0; r < 3; r++) P[4 * (size_t)at + r] = ((T[4 * r] * p[0] + T[4 * r + 1] * p[1]) + T[4 * r + 2] * p[2]) + T[4 * r + 3] * p[3];
      P[4 * (size_t)at + 3] = p[3];
      double RC[3][3];
      for (int r = 0; r < 3; r++)
        for (int k = 0; k < 3; k++) RC[r][k] = (T[4 * r] * c[4 * k + 0] + T[4 * r + 1] * c[4 * k + 1]) + T[4 * r + 2] * c[4 * k + 2];
      int o = 0;
      for (int r = 0; r < 3; r++)
        for (int k = r; k < 3; k++) C[6 * (size_t)at + o++] = (RC[r][0] * T[4 * k] + RC[r][1] * T[4 * k + 1]) + RC[r][2] * T[4 * k + 2];
    }
  }
  int nv = 0, m = 0;
  orc_sort_entry* e = sorted_by_key(P, total, resolution, 0, 0, &nv);
  double* MP = (double*)malloc(sizeof(double) * 4 * (size_t)(nv > 0 ? nv : 1));
  double* MC = (double*)malloc(sizeof(double) * 6 * (size_t)(nv > 0 ? nv : 1));
  int pos = 0;
  while (pos < nv) {
    double s[4] = {0, 0, 0, 0}, sc[6] = {0, 0, 0, 0, 0, 0};
    int end = pos;
    do {
      for (int a = 0; a < 4; a++) s[a] += P[4 * (size_t)e[end].idx + a];
      for (int a = 0; a < 6; a++) sc[a] += C[6 * (size_t)e[end].idx + a];
      end++;
    } while (end < nv && e[end].key == e[pos].key && !(block_size > 0 && end % block_size == 0));
    for (int a = 0; a < 4; a++) MP[4 * (size_t)m + a] = s[a] / s[3];
    for (int a = 0; a < 6; a++) MC[6 * (size_t)m + a] = sc[a] / s[3];
    m++;
    pos = end;
  }
  free(e);
  /* (!) final random sampling to the target size, original (voxel) order kept */
  int* keep = (int*)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
  int out_n = m;
  if (target_num_points > 0 && m > target_num_points) {
    const double rate = (double)target_num_points / (double)m;
    out_n = (int)((double)m * rate);
    orc_sort_entry* h = (orc_sort_entry*)malloc(sizeof(orc_sort_entry) * (size_t)m);
    for (int i = 0; i < m; i++) {
      h[i].key = orc_sample_hash(seed, (uint64_t)i) >> 32;
      h[i].aux = 0;
      h[i].idx = i;
    }
    qsort(h, (size_t)m, sizeof(orc_sort_entry), cmp_entry);
    for (int i = 0; i < out_n; i++) {
      h[i].key = 0;
    }
    qsort(h, (size_t)out_n, sizeof(orc_sort_entry), cmp_entry);
    for (int i = 0; i < out_n; i++) keep[i] = h[i].idx;
    free(h);
  } else {
    for (int i = 0; i < m; i++) keep[i] = i;
  }
  for (int i = 0; i < out_n; i++) {
    const int j = keep[i];
    memcpy(out_points4 + 4 * (size_t)i, MP + 4 * (size_t)j, 4 * sizeof(double));
    const double* c = MC + 6 * (size_t)j;
    double* o = out_covs16 + 16 * (size_t)i;
    memset(o, 0, 16 * sizeof(double));
    o[0] = c[0]; o[4] = o[1] = c[1]; o[8] = o[2] = c[2]; o[5] = c[3]; o[9] = o[6] = c[4]; o[10] = c[5];
  }
  free(keep); free(MP); free(MC); free(P); free(C);
  return out_n;
}

/* ---- adaptive voxel resolution: gtsam_points::median_distance(frame, 256) as called at
 * src/glim/odometry/odometry_estimation_gpu.cpp:90-93 and src/glim/mapping/global_mapping.cpp:238-241 ----
 * (!) upstream-recall: every step-th point (step = n / max_scan_count, at least 1), Euclidean norm of xyz, the element at
 * index size / 2 of the sorted list (std::nth_element).  The call sites then blend the base resolution:
 * p = clamp((d - dmin) / (dmax - dmin), 0, 1), resolution = r0 + p (rmax - r0). */
static int cmp_double(const void* a, const void* b) {
  const double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}
double orc_median_distance(const double* points4, int n, int max_scan_count) {
  if (n <= 0) return 0.0;
  const int step = n < max_scan_count ? 1 : n / max_scan_count;
  double* d = (double*)malloc(sizeof(double) * (size_t)(n / step + 1));
  int m = 0;
  for (int i = 0; i < n; i += step) {
    const double* p = points4 + 4 * (size_t)i;
    d[m++] = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
  }
  qsort(d, (size_t)m, sizeof(double), cmp_double);
  const double med = d[m / 2];
  free(d);
  return med;
}
double orc_adaptive_resolution(double dist_median, double r0, double rmax, double dmin, double dmax) {
  double p = (dist_median - dmin) / (dmax - dmin);
  p = p < 0.0 ? 0.0 : (p > 1.0 ? 1.0 : p);
  return r0 + p * (rmax - r0);
}
