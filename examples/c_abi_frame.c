/* c_abi_frame.c -- the per-frame path of GLIM's GPU odometry through the plain C ABI (include/glim_amd.h), C99, no C++:
 *   upload two scans (PointCloudGPU::clone) -> kNN -> covariances -> Gaussian voxel map of the first -> one unary VGICP factor of the
 *   second against it -> linearize -> print the Gauss-Newton step.
 * The calls mirror src/glim/odometry/odometry_estimation_gpu.cpp:96 (clone), :103-104 (voxel map), :161 (unary factor), :383-386 (set).
 * Build:  gcc -std=c99 -Iinclude examples/c_abi_frame.c -Lglim_amd -lglim_amd -Wl,-rpath,$PWD/glim_amd -lm -o c_abi_frame
 * Without a HIP device the program reports that and exits with 0 (tests/test_abi_cpu.py compiles and runs it that way). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "glim_amd.h"

#define CHECK(call)                                                                                          \
  do {                                                                                                       \
    const int rc_ = (call);                                                                                  \
    if (rc_ != GLIM_AMD_OK) {                                                                                \
      fprintf(stderr, "%s failed: %s (%s)\n", #call, glim_amd_error_string(rc_), glim_amd_last_hip_error()); \
      return 1;                                                                                              \
    }                                                                                                        \
  } while (0)

/* a room corner (floor + two walls) seen from (ox, oy): n points, Vector4d layout (x y z 1) */
static double* make_scan(int n, double ox, double oy, unsigned seed) {
  double* p = (double*)malloc(sizeof(double) * 4 * (size_t)n);
  int i;
  srand(seed);
  for (i = 0; i < n; i++) {
    const double a = rand() / (double)RAND_MAX, b = rand() / (double)RAND_MAX;
    double x, y, z;
    if (i % 3 == 0) { x = 8.0 * a; y = 6.0 * b; z = -1.5; }
    else if (i % 3 == 1) { x = 8.0; y = 6.0 * a; z = -1.5 + 3.0 * b; }
    else { x = 8.0 * a; y = 6.0; z = -1.5 + 3.0 * b; }
    p[4 * i + 0] = (double)(float)(x - ox);
    p[4 * i + 1] = (double)(float)(y - oy);
    p[4 * i + 2] = (double)(float)z;
    p[4 * i + 3] = 1.0;
  }
  return p;
}

int main(void) {
  const int n = 20000, k = 10;
  glim_amd_ctx* ctx = NULL;
  glim_amd_cloud *target = NULL, *source = NULL;
  glim_amd_voxelmap* map = NULL;
  glim_amd_factor_set* set = NULL;
  glim_amd_linearized6 lin;
  /* T_target_source, row-major 3x4: the second scan was taken 0.3 m / 0.1 m away; start the factor 5 cm off */
  double T[12] = {1, 0, 0, 0.25, 0, 1, 0, 0.10, 0, 0, 1, 0.0};
  double *pa, *pb, overlap = 0.0;
  int32_t num_voxels = 0;
  int i;

  printf("glim_amd ABI version %d, %d HIP device(s)\n", glim_amd_version(), glim_amd_device_count());
  if (glim_amd_device_count() < 1) {
    printf("no HIP device: nothing to run (the library has no CPU fallback)\n");
    return 0;
  }
  pa = make_scan(n, 1.0, 1.0, 1u);
  pb = make_scan(n, 1.3, 1.1, 2u);
  CHECK(glim_amd_ctx_create(0, 1, NULL, &ctx));
  CHECK(glim_amd_cloud_create(ctx, n, pa, NULL, NULL, &target));
  CHECK(glim_amd_cloud_create(ctx, n, pb, NULL, NULL, &source));
  CHECK(glim_amd_cloud_find_neighbors(target, k, NULL));
  CHECK(glim_amd_cloud_find_neighbors(source, k, NULL));
  CHECK(glim_amd_cloud_estimate_covariances(target, k));
  CHECK(glim_amd_cloud_estimate_covariances(source, k));
  CHECK(glim_amd_voxelmap_create(ctx, 0.5, 8192 * 2, 10, 1e-3, &map));
  CHECK(glim_amd_voxelmap_insert(map, target));
  CHECK(glim_amd_voxelmap_info(map, &num_voxels, NULL, NULL, NULL));
  CHECK(glim_amd_factor_set_create(ctx, &set));
  CHECK(glim_amd_factor_set_add(set, map, source, 0u /* unary, no surface validation */, NULL));
  CHECK(glim_amd_factor_set_linearize(set, T, &lin));
  CHECK(glim_amd_overlap(ctx, 1, (const glim_amd_voxelmap* const*)&map, T, source, &overlap));
  printf("%d voxels, %lld / %d inliers (overlap %.3f), error %.4f\n", (int)num_voxels, (long long)lin.num_inliers, n, overlap, lin.error);
  printf("gradient b_s = [");
  for (i = 0; i < 6; i++) printf("%s%.3f", i ? ", " : "", lin.b_s[i]);
  printf("],  H_ss diagonal = [");
  for (i = 0; i < 6; i++) printf("%s%.1f", i ? ", " : "", lin.H_ss[7 * i]);
  printf("]\n");
  /* children first, then the context (a context refuses to die before its children: GLIM_AMD_ERR_STATE) */
  CHECK(glim_amd_factor_set_destroy(set));
  CHECK(glim_amd_voxelmap_destroy(map));
  CHECK(glim_amd_cloud_destroy(source));
  CHECK(glim_amd_cloud_destroy(target));
  CHECK(glim_amd_ctx_destroy(ctx));
  free(pa);
  free(pb);
  return lin.num_inliers > n / 2 ? 0 : 1;
}
