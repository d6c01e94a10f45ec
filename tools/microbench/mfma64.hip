// mfma64.hip -- issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (tuning aid for the class-batched M2L, not part of the product).
// One wavefront per SIMD (256-thread workgroups, 1 per CU) and 2 per SIMD; NACC independent accumulators per wavefront.
// build: hipcc --offload-arch=gfx950 -O3 mfma64.hip -o mfma64 ; run: ./mfma64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void spin(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  if (s == 1.2345e-300) out[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void spin_valu(double* out, int iters, double a0, double b0) {
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = i;
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  if (s == 1.2345e-300) out[0] = s;
}
int run_valu(int wgs_per_cu) {
  double* out; CK(hipMalloc(&out, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 200000, grid = 256 * wgs_per_cu;
  spin_valu<16><<<grid, 256>>>(out, 100, 1.0, 2.0);
  CK(hipEventRecord(e0));
  spin_valu<16><<<grid, 256>>>(out, iters, 1.0, 2.0);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("v_fma_f64, 16 chains, %d wavefronts/SIMD: %.1f TFLOP/s\n", wgs_per_cu, (double)grid * 256 * iters * 16 * 2 / ms / 1e9);
  return 0;
}

template <int NACC>
int run(int wgs_per_cu) {
  double* out; CK(hipMalloc(&out, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000, grid = 256 * wgs_per_cu;
  spin<NACC><<<grid, 256>>>(out, 100, 1.0, 2.0);
  CK(hipEventRecord(e0));
  spin<NACC><<<grid, 256>>>(out, iters, 1.0, 2.0);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double n = (double)grid * 4 * iters * NACC;
  printf("NACC %d, %d wavefronts/SIMD: %.1f TFLOP/s, %.1f cycles per MFMA per SIMD at 2.4 GHz\n", NACC, wgs_per_cu,
         n * 2048 / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * NACC * wgs_per_cu));
  return 0;
}
int main() { return run_valu(1) || run_valu(2) || run_valu(4) || run<1>(1) || run<8>(1) || run<8>(2) || run<8>(3) || run<8>(4) || run<4>(8) || run<2>(8); }
