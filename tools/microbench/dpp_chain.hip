// dpp_chain.hip -- at ONE wavefront per SIMD, how fast do v_fmac_f64_dpp instructions issue when consecutive ones feed the same
// accumulator (1), alternate between two (2: the sa / sb of the M2L rotation kernel), or rotate over 4 or 8?
//   hipcc --offload-arch=gfx950 -O3 -o dpp_chain dpp_chain.hip && ./dpp_chain
#include <hip/hip_runtime.h>
#include <cstdio>

#define FM(k, acc) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #k " row_mask:0xf bank_mask:0xf" : "+v"(a[(acc)]) : "v"(cv), "v"(d[(k) & 3]))

template <int NACC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void chain_kernel(const double* __restrict__ tab, const double* __restrict__ x, double* __restrict__ y, int iters) {
  __shared__ double pad[4800];                         // 38 KB: one workgroup per SIMD's share of LDS as well
  const int lane = threadIdx.x;
  pad[lane] = x[lane];
  double cv = tab[lane & 15];
  double d[4] = {x[lane], x[lane + 64], x[lane + 128], x[lane + 192]};
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    FM(0, 0 % NACC); FM(1, 1 % NACC); FM(2, 2 % NACC); FM(3, 3 % NACC); FM(4, 4 % NACC); FM(5, 5 % NACC); FM(6, 6 % NACC); FM(7, 7 % NACC);
    FM(8, 8 % NACC); FM(9, 9 % NACC); FM(10, 10 % NACC); FM(11, 11 % NACC); FM(12, 12 % NACC); FM(13, 13 % NACC); FM(14, 14 % NACC); FM(15, 15 % NACC);
  }
  double s = pad[(lane + 1) & 63];
  for (int i = 0; i < 8; ++i) s += a[i];
  y[blockIdx.x * 64 + lane] = s;
}

// the same with plain v_fma_f64 (VGPR operands, no DPP) and with an SGPR multiplier
template <int NACC, int MODE, int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void plain_kernel(const double* __restrict__ tab, const double* __restrict__ x, double* __restrict__ y, int iters) {
  __shared__ double pad[WAVES == 1 ? 4800 : 2400];
  const int lane = threadIdx.x;
  pad[lane] = x[lane];
  double cv = tab[lane & 15];
  const double cs = tab[3];
  double d[4] = {x[lane], x[lane + 64], x[lane + 128], x[lane + 192]};
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (MODE == 0) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[k % NACC]) : "v"(cv), "v"(d[k & 3]));
      else if (MODE == 1) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[k % NACC]) : "s"(cs), "v"(d[k & 3]));
      else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[k % NACC]) : "v"(cv), "v"(d[k & 3]));
    }
  }
  double s = pad[(lane + 1) & 63];
  for (int i = 0; i < 8; ++i) s += a[i];
  y[blockIdx.x * 64 + lane] = s;
}
template <int NACC, int MODE, int WAVES>
void run2(const double* dt, const double* dx, double* dy, int blocks, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((plain_kernel<NACC, MODE, WAVES>), dim3(blocks), dim3(64), 0, 0, dt, dx, dy, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double fmas = 16.0 * iters * (blocks / 1024.0);
  std::printf("%s, %d accumulators, %d wavefront(s) per SIMD: %.3f ms, %.2f ns per FMA per SIMD (%.1f TFLOP/s)\n",
              MODE == 0 ? "v_fmac_f64 vgpr" : MODE == 1 ? "v_fmac_f64 sgpr" : "v_fmac_f64_dpp ", NACC, WAVES, ms, ms * 1e6 / fmas, 2.0 * 16 * iters * 64.0 * blocks / ms / 1e9);
}

template <int NACC>
void run(const double* dt, const double* dx, double* dy, int blocks, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain_kernel<NACC>, dim3(blocks), dim3(64), 0, 0, dt, dx, dy, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double fmas = 16.0 * iters * (blocks / 1024.0);      // per SIMD
  std::printf("%d accumulator(s): %.3f ms, %.2f ns per FMA per SIMD (%.1f TFLOP/s)\n", NACC, ms, ms * 1e6 / fmas, 2.0 * 16 * iters * 64.0 * blocks / ms / 1e9);
}

int main() {
  const int blocks = 1024 * 4, iters = 20000;
  double tab[16], x[256];
  for (int i = 0; i < 16; ++i) tab[i] = 1.0 + 1e-9 * i;
  for (int i = 0; i < 256; ++i) x[i] = 1e-30 * (i + 1);
  double *dt, *dx, *dy;
  hipMalloc(&dt, sizeof(tab)); hipMalloc(&dx, sizeof(x)); hipMalloc(&dy, blocks * 64 * 8);
  hipMemcpy(dt, tab, sizeof(tab), hipMemcpyHostToDevice); hipMemcpy(dx, x, sizeof(x), hipMemcpyHostToDevice);
  run<1>(dt, dx, dy, blocks, iters); run<2>(dt, dx, dy, blocks, iters); run<4>(dt, dx, dy, blocks, iters); run<8>(dt, dx, dy, blocks, iters);
  run2<1, 0, 1>(dt, dx, dy, blocks, iters); run2<2, 0, 1>(dt, dx, dy, blocks, iters); run2<4, 0, 1>(dt, dx, dy, blocks, iters); run2<8, 0, 1>(dt, dx, dy, blocks, iters);
  run2<2, 1, 1>(dt, dx, dy, blocks, iters); run2<4, 1, 1>(dt, dx, dy, blocks, iters);
  run2<2, 2, 2>(dt, dx, dy, blocks, iters); run2<4, 2, 2>(dt, dx, dy, blocks, iters); run2<2, 0, 2>(dt, dx, dy, blocks, iters); run2<4, 0, 2>(dt, dx, dy, blocks, iters);
  return 0;
}
