// issue_mix.hip -- at ONE wavefront per SIMD an FP64 FMA issues every ~6.5 cycles, not every 4 (dpp_chain.hip).  Is the gap behind
// an FMA an issue slot that another kind of instruction of the same wavefront can take?  Sixteen v_fmac_f64_dpp (four accumulators)
// per iteration, alone and with sixteen other instructions between or behind them.
//   hipcc --offload-arch=gfx950 -O3 -o issue_mix issue_mix.hip && ./issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>

#define FMA(k) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[(k) & 3]) : "v"(cv), "v"(d[(k) & 3]))
#define MOV32(k) asm volatile("v_mov_b32 %0, %1" : "=v"(t[(k) & 7]) : "v"(u[(k) & 3]))
#define MOV64(k) asm volatile("v_mov_b64 %0, 0" : "=v"(z[(k) & 3]))
#define ACCR(k) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t[(k) & 7]) : "a"(g[(k) & 3]))
#define MUL64(k) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(z[(k) & 3]) : "v"(d[(k) & 3]), "v"(d[((k) + 1) & 3]))
#define ADD64(k) asm volatile("v_add_f64 %0, %1, %2" : "=v"(z[(k) & 3]) : "v"(d[(k) & 3]), "v"(d[((k) + 1) & 3]))
#define SNOP(k) asm volatile("s_nop 0")
#define SALU(k) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc))

template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void mix_kernel(const double* __restrict__ tab, const double* __restrict__ x, double* __restrict__ y, int iters) {
  __shared__ double pad[4800];                         // 38 KB: one workgroup per SIMD's share of LDS as well
  const int lane = threadIdx.x;
  pad[lane] = x[lane];
  double cv = tab[lane & 15];
  double d[4] = {x[lane], x[lane + 64], x[lane + 128], x[lane + 192]};
  double a[4] = {0, 0, 0, 0}, z[4] = {0, 0, 0, 0};
  int t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, u[4] = {lane, lane + 1, lane + 2, lane + 3}, g[4];
  unsigned sc = 0;
  for (int i = 0; i < 4; ++i) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(g[i]) : "v"(u[i]));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (MODE != 5 && MODE != 8) FMA(k);
      if (MODE == 1) MOV32(k);
      if (MODE == 2) MOV64(k);
      if (MODE == 3) ACCR(k);
      if (MODE == 6 || MODE == 8) MUL64(k);
      if (MODE == 7) { MOV32(2 * k); MOV32(2 * k + 1); }
      if (MODE == 9) SNOP(k);
      if (MODE == 10) SALU(k);
      if (MODE == 11) ADD64(k);
      if (MODE == 5) MOV32(k);
    }
    if (MODE == 4) {
#pragma unroll
      for (int k = 0; k < 16; ++k) MOV32(k);
    }
  }
  double s = pad[(lane + 1) & 63] + sc;
  for (int i = 0; i < 4; ++i) s += a[i] + z[i];
  for (int i = 0; i < 8; ++i) s += t[i];
  y[blockIdx.x * 64 + lane] = s;
}

template <int MODE>
void run(const char* what, const double* dt, const double* dx, double* dy, int blocks, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mix_kernel<MODE>, dim3(blocks), dim3(64), 0, 0, dt, dx, dy, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double groups = 16.0 * iters * (blocks / 1024.0);      // per SIMD
  std::printf("%-64s %.3f ms, %.2f ns per slot of the sixteen\n", what, ms, ms * 1e6 / groups);
}

int main() {
  const int blocks = 1024 * 4, iters = 20000;
  double tab[16], x[256];
  for (int i = 0; i < 16; ++i) tab[i] = 1.0 + 1e-9 * i;
  for (int i = 0; i < 256; ++i) x[i] = 1e-30 * (i + 1);
  double *dt, *dx, *dy;
  hipMalloc(&dt, sizeof(tab)); hipMalloc(&dx, sizeof(x)); hipMalloc(&dy, blocks * 64 * 8);
  hipMemcpy(dt, tab, sizeof(tab), hipMemcpyHostToDevice); hipMemcpy(dx, x, sizeof(x), hipMemcpyHostToDevice);
  run<0>("16 FMA", dt, dx, dy, blocks, iters);
  run<5>("16 v_mov_b32", dt, dx, dy, blocks, iters);
  run<8>("16 v_mul_f64", dt, dx, dy, blocks, iters);
  run<1>("16 x (FMA, v_mov_b32)", dt, dx, dy, blocks, iters);
  run<7>("16 x (FMA, v_mov_b32, v_mov_b32)", dt, dx, dy, blocks, iters);
  run<4>("16 FMA then 16 v_mov_b32", dt, dx, dy, blocks, iters);
  run<2>("16 x (FMA, v_mov_b64 0)", dt, dx, dy, blocks, iters);
  run<3>("16 x (FMA, v_accvgpr_read_b32)", dt, dx, dy, blocks, iters);
  run<6>("16 x (FMA, v_mul_f64)", dt, dx, dy, blocks, iters);
  run<11>("16 x (FMA, v_add_f64)", dt, dx, dy, blocks, iters);
  run<9>("16 x (FMA, s_nop 0)", dt, dx, dy, blocks, iters);
  run<10>("16 x (FMA, s_add_u32)", dt, dx, dy, blocks, iters);
  return 0;
}
