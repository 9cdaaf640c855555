// VALU issue-cost microbenchmark for gfx950: cycles per wave64 instruction per SIMD for the instruction classes the VGICP kernel uses.
// Each test runs ITER x 32 independent instructions (8 accumulators x 4) per wave with W waves per SIMD resident; reports
// cycles / instruction / SIMD (s_memtime ticks = shader cycles).  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 2000

#define BODY8(INS)  INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)

template <int KIND>
__global__ void bench(unsigned long long* out, float seed) {
  float a[8], b = seed, c = seed * 0.5f;
  double d[8], db = seed, dc = seed * 0.25;
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f p[8], pb = {seed, seed}, pc = {seed * 0.5f, seed};
  int n[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { a[i] = seed + i; d[i] = seed + i; p[i] = v2f{seed + i, seed - i}; n[i] = (int)seed + i; }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (KIND == 0) {
#define I(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
        BODY8(I)
#undef I
      } else if (KIND == 1) {
#define I(k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k]) : "v"(pb), "v"(pc));
        BODY8(I)
#undef I
      } else if (KIND == 2) {
#define I(k) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[k]) : "v"(db), "v"(dc));
        BODY8(I)
#undef I
      } else if (KIND == 3) {
#define I(k) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
        BODY8(I)
#undef I
      } else if (KIND == 4) {
#define I(k) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[k]) : "v"(db));
        BODY8(I)
#undef I
      } else if (KIND == 5) {
#define I(k) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(n[k]) : "v"(n[(k + 1) & 7]));
        BODY8(I)
#undef I
      } else if (KIND == 6) {
#define I(k) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[k]) : "v"(d[k]));
        BODY8(I)
#undef I
      } else if (KIND == 7) {
#define I(k) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[k]) : "v"(b) : );
        BODY8(I)
#undef I
      } else if (KIND == 8) {
#define I(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
        BODY8(I)
#undef I
      } else if (KIND == 9) {
#define I(k) asm volatile("v_floor_f64 %0, %0" : "+v"(d[k]));
        BODY8(I)
#undef I
      } else if (KIND == 10) {
#define I(k) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[k]) : "v"(d[k]));
        BODY8(I)
#undef I
      } else if (KIND == 11) {
#define I(k) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(n[k]) : "v"(n[(k + 1) & 7]));
        BODY8(I)
#undef I
      } else if (KIND == 12) {
#define I(k) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(n[k]) : "v"(n[(k + 1) & 7]));
        BODY8(I)
#undef I
      } else if (KIND == 13) {
#define I(k) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[k]) : "v"(pb));
        BODY8(I)
#undef I
      } else if (KIND == 14) {
#define I(k) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[k]) : "v"(pb));
        BODY8(I)
#undef I
      } else if (KIND == 15) {
#define I(k) asm volatile("v_fma_f32 %0, %1, s4, %0" : "+v"(a[k]) : "v"(b) : "s4");
        BODY8(I)
#undef I
      } else if (KIND == 16) {
#define I(k) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(d[k]) : "v"(db));
        BODY8(I)
#undef I
      } else if (KIND == 17) {
#define I(k) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(a[k]));
        BODY8(I)
#undef I
      } else if (KIND == 18) {
#define I(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");
        BODY8(I)
#undef I
      } else if (KIND == 19) {
#define I(k) asm volatile("v_lshl_add_u32 %0, %1, 3, %0" : "+v"(n[k]) : "v"(n[(k + 1) & 7]));
        BODY8(I)
#undef I
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) s += a[i] + (float)d[i] + p[i].x + p[i].y + (float)n[i];
  if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
  if (s == 12345.678f) out[blockIdx.x * 2 + 1] = 1;  // keep results alive
}

template <int KIND>
void run(const char* name, int cus, unsigned long long* d_out) {
  printf("%-22s", name);
  for (int wps : {1, 2, 4, 8}) {  // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
    const int blocks = cus * wps;
    hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 2);
    hipMemcpy(h.data(), d_out, blocks * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double avg = 0;
    for (int b = 0; b < blocks; b++) avg += (double)h[2 * b];
    avg /= blocks;
    // each wave issued ITER*32 instructions; wps waves share a SIMD: cycles per instruction per SIMD = ticks / (ITER*32*wps)
    printf("  W=%d: %6.2f", wps, avg / ((double)ITER * 32 * wps));
  }
  printf("\n");
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  unsigned long long* d_out;
  hipMalloc(&d_out, cus * 8 * 2 * sizeof(unsigned long long));
  printf("%s, %d CUs; s_memtime ticks per wave64 instruction per SIMD (lower = faster)\n", prop.gcnArchName, cus);
  run<0>("v_fma_f32", cus, d_out);
  run<15>("v_fma_f32 (sgpr src)", cus, d_out);
  run<3>("v_mul_f32", cus, d_out);
  run<1>("v_pk_fma_f32", cus, d_out);
  run<13>("v_pk_mul_f32", cus, d_out);
  run<14>("v_pk_add_f32", cus, d_out);
  run<2>("v_fma_f64", cus, d_out);
  run<16>("v_mul_f64", cus, d_out);
  run<4>("v_add_f64", cus, d_out);
  run<9>("v_floor_f64", cus, d_out);
  run<6>("v_cvt_f32_f64", cus, d_out);
  run<17>("v_cvt_f64_f32", cus, d_out);
  run<10>("v_cvt_i32_f64", cus, d_out);
  run<5>("v_mul_lo_u32", cus, d_out);
  run<11>("v_mul_u32_u24", cus, d_out);
  run<12>("v_xor_b32", cus, d_out);
  run<19>("v_lshl_add_u32", cus, d_out);
  run<7>("v_cndmask_b32", cus, d_out);
  run<18>("v_cmp_lt_f32", cus, d_out);
  run<8>("v_rcp_f32", cus, d_out);
  return 0;
}
