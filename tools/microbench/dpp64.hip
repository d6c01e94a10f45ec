// dpp64.hip -- does `v_fmac_f64_dpp ... row_newbcast:k` (a VGPR lane of each 16-lane row broadcast as the multiplier) issue at
// the rate of a plain v_fma_f64 with an SGPR operand?  Used to decide how the M2L rotation kernel gets its wave-uniform
// constants.   hipcc --offload-arch=gfx950 -O3 -o dpp64 dpp64.hip && ./dpp64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define FM(k) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #k " row_mask:0xf bank_mask:0xf" : "+v"(a[(k) & 7]) : "v"(cv), "v"(d[(k) & 3]))

__global__ __launch_bounds__(64) void dpp_kernel(const double* __restrict__ tab, const double* __restrict__ x, double* __restrict__ y, int iters) {
  const int lane = threadIdx.x;
  double cv = tab[lane & 15];
  double d[4] = {x[lane], x[lane + 64], x[lane + 128], x[lane + 192]};
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    FM(0); FM(1); FM(2); FM(3); FM(4); FM(5); FM(6); FM(7); FM(8); FM(9); FM(10); FM(11); FM(12); FM(13); FM(14); FM(15);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  y[blockIdx.x * 64 + lane] = s;
}

__global__ __launch_bounds__(64) void sgpr_kernel(const double* __restrict__ tab, const double* __restrict__ x, double* __restrict__ y, int iters) {
  const int lane = threadIdx.x;
  double c[16];
  for (int i = 0; i < 16; ++i) c[i] = tab[i];                 // uniform: scalar loads, SGPR operands
  double d[4] = {x[lane], x[lane + 64], x[lane + 128], x[lane + 192]};
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      a[k & 7] = fma(c[k], d[k & 3], a[k & 7]);
      asm volatile("" : "+v"(a[k & 7]));
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  y[blockIdx.x * 64 + lane] = s;
}

int main() {
  const int blocks = 256 * 8 * 4, iters = 2000;
  std::vector<double> tab(16), x(256), y(blocks * 64), y2(blocks * 64);
  for (int i = 0; i < 16; ++i) tab[i] = 1.0 + 0.01 * i;
  for (int i = 0; i < 256; ++i) x[i] = 1e-3 * (i + 1);
  double *dt, *dx, *dy;
  hipMalloc(&dt, 16 * 8); hipMalloc(&dx, 256 * 8); hipMalloc(&dy, blocks * 64 * 8);
  hipMemcpy(dt, tab.data(), 16 * 8, hipMemcpyHostToDevice); hipMemcpy(dx, x.data(), 256 * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(dpp_kernel, dim3(blocks), dim3(64), 0, 0, dt, dx, dy, iters);
      else hipLaunchKernelGGL(sgpr_kernel, dim3(blocks), dim3(64), 0, 0, dt, dx, dy, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 16 * iters * 64.0 * blocks;
    hipMemcpy(which == 0 ? y.data() : y2.data(), dy, blocks * 64 * 8, hipMemcpyDeviceToHost);
    std::printf("%s: %.3f ms, %.1f TFLOP/s\n", which == 0 ? "v_fmac_f64_dpp row_newbcast" : "v_fma_f64 SGPR operand   ", ms, flops / ms / 1e9);
  }
  double maxdiff = 0;
  for (int i = 0; i < blocks * 64; ++i) maxdiff = std::fmax(maxdiff, std::fabs(y[i] - y2[i]) / std::fabs(y2[i]));
  std::printf("max relative difference between the two: %.2e (lane 0: %.17g vs %.17g)\n", maxdiff, y[0], y2[0]);
  return 0;
}
