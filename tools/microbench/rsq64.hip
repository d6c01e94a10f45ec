// rsq64.hip -- issue rate of v_rsq_f64 against v_fma_f64 on gfx950: the ceiling of the matrix-free near field (mf_sweep,
// kernels_near.hip), which spends one reciprocal square root + a Newton step per quadrature point and panel pair.
//   hipcc --offload-arch=gfx950 -O3 -o rsq64 rsq64.hip && ./rsq64      (numbers: profiles/r04m_rsq64_issue_rate.txt)
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>   // 0: 8 independent v_rsq_f64 per iteration; 1: 8 independent v_fma_f64; 2: rsq + the Newton step of mf_sweep (rsq, mul, fma, fma, mul)
__global__ __launch_bounds__(256) void k(const double* __restrict__ x, double* __restrict__ y, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = x[(t + i) & 255] + 1.0 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[i]));
      else if (MODE == 1) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a[i]));
      else {
        double r;
        asm volatile("v_rsq_f64 %0, %1" : "=v"(r) : "v"(a[i]));
        const double h = 0.5 * a[i] * r;                 // one Newton step: r (1.5 - 0.5 a r^2)
        r = fma(-h, r, 1.5) * r;
        a[i] = fma(r, 1e-9, a[i]);
      }
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  y[t] = s;
}

int main() {
  const int blocks = 256 * 8, iters = 4000;
  double *dx, *dy;
  hipMalloc(&dx, 256 * 8); hipMalloc(&dy, (size_t)blocks * 256 * 8);
  hipMemset(dx, 0, 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"v_rsq_f64", "v_fma_f64", "rsq + Newton step (5 FP64 instructions)"};
  for (int mode = 0; mode < 3; ++mode) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, dx, dy, iters);
      else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, dx, dy, iters);
      else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, dx, dy, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double ops = 8.0 * iters * 256.0 * blocks;
    std::printf("%-42s %.3f ms  %.2f T per-lane operations/s%s\n", names[mode], ms, ops / ms / 1e9,
                mode == 2 ? "  (= reciprocal square roots with their Newton step per second)" : "");
  }
  return 0;
}
