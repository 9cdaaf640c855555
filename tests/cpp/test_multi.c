/* test_multi.c -- the single-process multi-device cost evaluation through the plain C ABI (include/glim_amd.h: glim_amd_multi_*), run with
 * every visible device (one on the GPU box of this project: the RCCL all-gather is then a one-rank collective, still issued).  Checks the
 * sharded evaluation against a plain single-context factor set: identical records, factor by factor.
 * Built and run by tests/test_multi_gpu.py:  gcc -std=c99 test_multi.c -lglim_amd -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/glim_amd.h"
#include "../../include/glim_amd_diag.h" /* glim_amd_multi_profile: a timing loop, not part of the drop-in ABI */

#define REQUIRE(cond)                                                          \
  do {                                                                         \
    if (!(cond)) {                                                             \
      fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, glim_amd_last_hip_error()); \
      return 1;                                                                \
    }                                                                          \
  } while (0)

static unsigned long long rng_state = 88172645463325252ull;
static double urand(void) {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (double)(rng_state >> 11) / 9007199254740992.0;
}

/* three walls of a room corner seen from (ox, oy, yaw); FP32-representable coordinates */
static void make_scan(int n, double ox, double oy, double yaw, float* xyz) {
  const double c = cos(yaw), s = sin(yaw);
  for (int i = 0; i < n; i++) {
    double x, y, z;
    const int wall = i % 3;
    if (wall == 0) { x = 8.0 * urand(); y = 6.0 * urand(); z = -1.5 + 0.004 * (urand() - 0.5); }
    else if (wall == 1) { x = 8.0 + 0.004 * (urand() - 0.5); y = 6.0 * urand(); z = -1.5 + 3.0 * urand(); }
    else { x = 8.0 * urand(); y = 6.0 + 0.004 * (urand() - 0.5); z = -1.5 + 3.0 * urand(); }
    const double wx = x - ox, wy = y - oy;
    xyz[3 * i + 0] = (float)(c * wx + s * wy);
    xyz[3 * i + 1] = (float)(-s * wx + c * wy);
    xyz[3 * i + 2] = (float)z;
  }
}

static void rel_pose(double xa, double ya, double wa, double xb, double yb, double wb, double* T) { /* T_a_b, row-major 3x4 */
  const double dw = wb - wa, ca = cos(wa), sa = sin(wa);
  const double dx = xb - xa, dy = yb - ya;
  const double t0 = ca * dx + sa * dy, t1 = -sa * dx + ca * dy;
  const double R[12] = {cos(dw), -sin(dw), 0, t0, sin(dw), cos(dw), 0, t1, 0, 0, 1, 0};
  memcpy(T, R, sizeof(R));
}

int main(int argc, char** argv) {
  const int ndev = glim_amd_device_count();
  REQUIRE(ndev >= 1);
  int32_t devices[16];
  /* `test_multi virtual N` (with GLIM_AMD_DIAG=multi_virtual=1 in the environment): device 0 listed N times -- the N > 1 path of multi.hip (threads,
   * host barrier, per-device shards and pieces, the exchange) on a one-GPU box; the exchange is then the same-device stand-in, not RCCL */
  const int virtual_n = (argc > 2 && strcmp(argv[1], "virtual") == 0) ? atoi(argv[2]) : 0;
  const int use = virtual_n > 0 ? (virtual_n < 16 ? virtual_n : 16) : (ndev < 16 ? ndev : 16);
  for (int i = 0; i < use; i++) devices[i] = virtual_n > 0 ? 0 : i;
  glim_amd_multi* multi = NULL;
  REQUIRE(glim_amd_multi_create(devices, use, &multi) == GLIM_AMD_OK);
  int32_t nd = 0, rccl = 0;
  REQUIRE(glim_amd_multi_info(multi, &nd, &rccl, NULL) == GLIM_AMD_OK && nd == use);
  printf("devices %d, rccl %d\n", nd, rccl);
  if (virtual_n > 1) REQUIRE(rccl == 0);
  else REQUIRE(rccl == 1); /* librccl ships with the ROCm image: the collective path must be the one that runs */

  enum { S = 6, N = 20000 };
  const double px[S] = {1.0, 1.5, 2.1, 2.4, 3.0, 3.3}, py[S] = {1.0, 1.2, 1.1, 1.6, 1.9, 2.4}, pw[S] = {0.0, 0.05, 0.1, 0.12, 0.2, 0.25};
  glim_amd_ctx* ctx = NULL; /* the unsharded reference: one context on device 0 */
  REQUIRE(glim_amd_ctx_create(0, 1, NULL, &ctx) == GLIM_AMD_OK);
  glim_amd_cloud* clouds[S];
  glim_amd_voxelmap* maps[S];
  int32_t cid[S], mid[S];
  float* xyz = (float*)malloc(sizeof(float) * 3 * N);
  for (int s = 0; s < S; s++) {
    const int n = N - 1000 * s; /* different sizes: the shards are cost-balanced, not count-balanced */
    make_scan(n, px[s], py[s], pw[s], xyz);
    REQUIRE(glim_amd_multi_add_cloud_f32(multi, n, xyz, NULL, NULL, &cid[s]) == GLIM_AMD_OK);
    REQUIRE(glim_amd_multi_cloud_estimate_covariances(multi, cid[s], 10) == GLIM_AMD_OK);
    REQUIRE(glim_amd_multi_add_voxelmap(multi, cid[s], 0.5, &mid[s]) == GLIM_AMD_OK);
    REQUIRE(glim_amd_cloud_create_f32(ctx, n, xyz, NULL, NULL, &clouds[s]) == GLIM_AMD_OK);
    REQUIRE(glim_amd_cloud_find_neighbors(clouds[s], 10, NULL) == GLIM_AMD_OK);
    REQUIRE(glim_amd_cloud_estimate_covariances(clouds[s], 10) == GLIM_AMD_OK);
    REQUIRE(glim_amd_voxelmap_create(ctx, 0.5, 16384, 10, 1e-3, &maps[s]) == GLIM_AMD_OK);
    REQUIRE(glim_amd_voxelmap_insert(maps[s], clouds[s]) == GLIM_AMD_OK);
  }
  free(xyz);

  enum { F = S * (S - 1) / 2 };
  int32_t tmap[F], scloud[F];
  uint32_t flags[F];
  double T[12 * F];
  glim_amd_factor_set* set = NULL;
  REQUIRE(glim_amd_factor_set_create(ctx, &set) == GLIM_AMD_OK);
  int f = 0;
  for (int i = 0; i < S; i++)
    for (int j = i + 1; j < S; j++, f++) {
      tmap[f] = mid[i];
      scloud[f] = cid[j];
      flags[f] = (f % 3 == 0) ? 0u : GLIM_AMD_FACTOR_BINARY;
      rel_pose(px[i], py[i], pw[i], px[j] + 0.01, py[j] - 0.02, pw[j] + 0.003, T + 12 * f);
      REQUIRE(glim_amd_factor_set_add(set, maps[i], clouds[j], flags[f], NULL) == GLIM_AMD_OK);
    }
  REQUIRE(glim_amd_multi_set_factors(multi, F, tmap, scloud, flags) == GLIM_AMD_OK);
  int64_t bounds[17];
  REQUIRE(glim_amd_multi_shard(multi, bounds) == GLIM_AMD_OK && bounds[0] == 0 && bounds[use] == F);

  glim_amd_linearized6* got = (glim_amd_linearized6*)calloc(F, sizeof(glim_amd_linearized6));
  glim_amd_linearized6* want = (glim_amd_linearized6*)calloc(F, sizeof(glim_amd_linearized6));
  double total = 0.0, total_want = 0.0;
  for (int rep = 0; rep < 3; rep++) { /* repeated evaluations reuse the plan and the communicators */
    REQUIRE(glim_amd_multi_linearize(multi, T, got, &total) == GLIM_AMD_OK);
    REQUIRE(glim_amd_factor_set_linearize(set, T, want) == GLIM_AMD_OK);
    total_want = 0.0;
    for (int k = 0; k < F; k++) {
      REQUIRE(got[k].num_inliers == want[k].num_inliers && got[k].num_inliers > 1000);
      /* one device: the sharded set IS the whole set, so the plan (blocks per factor) and hence every FP32 partial sum is identical;
       * several devices: each set holds fewer factors, the block partition differs, sums agree to FP32 summation-order level */
      if (use == 1) {
        REQUIRE(memcmp(&got[k], &want[k], sizeof(glim_amd_linearized6)) == 0);
      } else {
        double worst = 0.0, scale = 1e-300;
        for (int q = 0; q < 36; q++) {
          worst = fmax(worst, fabs(got[k].H_ss[q] - want[k].H_ss[q]));
          scale = fmax(scale, fabs(want[k].H_ss[q]));
        }
        REQUIRE(worst <= 1e-4 * scale);
      }
      total_want += want[k].error;
    }
    REQUIRE(fabs(total - total_want) <= 1e-9 * fabs(total_want));
  }
  float ms = 0.f;
  REQUIRE(glim_amd_multi_profile(multi, T, 10, &ms) == GLIM_AMD_OK && ms > 0.f);
  printf("%d factors over %d device(s): %.3f ms per cost evaluation, total error %.6f\n", F, use, ms, total);
  /* an empty factor list and a second list on the same handle */
  REQUIRE(glim_amd_multi_set_factors(multi, 0, NULL, NULL, NULL) == GLIM_AMD_OK);
  REQUIRE(glim_amd_multi_linearize(multi, NULL, NULL, &total) == GLIM_AMD_OK && total == 0.0);
  REQUIRE(glim_amd_multi_set_factors(multi, 2, tmap, scloud, NULL) == GLIM_AMD_OK);
  REQUIRE(glim_amd_multi_linearize(multi, T, got, NULL) == GLIM_AMD_OK && got[1].num_inliers == want[1].num_inliers);
  free(got);
  free(want);
  REQUIRE(glim_amd_factor_set_destroy(set) == GLIM_AMD_OK);
  for (int s = 0; s < S; s++) {
    glim_amd_voxelmap_destroy(maps[s]);
    glim_amd_cloud_destroy(clouds[s]);
  }
  REQUIRE(glim_amd_ctx_destroy(ctx) == GLIM_AMD_OK);
  REQUIRE(glim_amd_multi_destroy(multi) == GLIM_AMD_OK);
  printf("test_multi OK\n");
  return 0;
}
