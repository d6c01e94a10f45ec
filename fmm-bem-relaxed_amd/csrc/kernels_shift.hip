// kernels_shift.hip -- M2M and L2L of the tree passes, one (parent, child) pair per WAVEFRONT, lane = output coefficient.
// Reference: LaplaceSpherical::M2M (kernel/LaplaceSpherical.hpp:245-285), ::L2L (:378-411).  Structure, tables and the reason
// for this mapping: shift_lanes.hpp.  The arithmetic of every output is, operation by operation, that of the one-pair-per-lane
// rotation kernels (kernels_m2l_rot.hip, FMMBEM_ROT_OP = 1, 2) -- tests/test_gpu_parity.py compares the two bit for bit -- so a
// plan, or a shard of one, may take either kernel level by level without moving a bit.
//
// A wavefront walks units (M2M: a parent with its <= 8 children; L2L: a child) with the stride of the grid.  Per pair:
//   load    lane = coefficient (n, m) of the source expansion, turned by e^{i m beta}                       -> LDS
//   stage 0 row (n, m) of conj(X): two chains of FMAs over the degree's inputs (LDS), turned by e^{i m alpha}   -> LDS
//   stage 1 row of X^T, scaled by rho^-n (L2L: rho^n)                                                       -> LDS
//   stage 2 axial row (j, k): sum over the degrees n of order k (both parts with one constant), scaled by rho^j -> LDS
//   stage 3 conj(X), turned by e^{-i m alpha}                                                               -> LDS
//   stage 4 X^T, turned by e^{-i m beta}                                                                    -> registers
//   M2M: the children of a parent are added in the chain order of the one-pair kernel, (c0 + c4) + (c1 + c5) ..., and the
//   parent's M written once;  L2L: added into the child's L.
// The constants of a lane's rows (about 30 doubles and as many LDS offsets at p = 10) are loaded once per wavefront.
#include "device_launch.hpp"
#include "shift_lanes.hpp"

#include <type_traits>

namespace fmmbem {

namespace {

constexpr int kWave = 64;
constexpr int kSlWaves = 4;                           // wavefronts per workgroup, each on its own units and its own LDS slice

__device__ __forceinline__ void sl_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int P, int OP>
__global__ __launch_bounds__(kSlWaves * kWave) void shift_lanes_kernel(const DevicePlan d, const ShiftLaneWork w) {
  constexpr int S = sl_S(P), R = sl_rounds(P), LR = sl_rot_len(P), LX = sl_axial_len(P);
  constexpr int XS = 2 * S + 2;                       // a[S] then b[S]
  __shared__ double xbuf[kSlWaves][2][XS];
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  double* X0 = xbuf[wv][0];
  double* X1 = xbuf[wv][1];
  const int slot = d.act[blockIdx.y];
  const size_t box_stride = (size_t)d.nslots * d.s_max;
  const double2* src_base = (OP == kRotL2L ? d.L : d.M) + (size_t)slot * d.s_max;

  // ---- this lane's rows: constants and operand offsets, once ----
  double rc[R][2][2][LR], xc[R][LX];
  int rs[R][2][2][LR], xs[R][LX], rn[R], rm[R];
  bool rok[R];
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const int row = lane + kWave * q;
    rok[q] = row < S;
    const int rr = rok[q] ? row : 0;
    int n = 0;
    while ((n + 1) * (n + 2) / 2 <= rr) ++n;
    rn[q] = n; rm[q] = rr - n * (n + 1) / 2;
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < LR; ++t) {
          const size_t e = ((size_t)(v * 2 + c) * LR + t) * S + rr;
          rc[q][v][c][t] = rok[q] ? w.rot_c[e] : 0.0;
          rs[q][v][c][t] = rok[q] ? w.rot_s[e] : 0;
        }
#pragma unroll
    for (int t = 0; t < LX; ++t) {
      xc[q][t] = rok[q] ? w.ax_c[(size_t)t * S + rr] : 0.0;
      xs[q][t] = rok[q] ? w.ax_s[(size_t)t * S + rr] : 0;
    }
  }

  const int PM = w.p_max;
  // a rotation stage: out row = two FMA chains over the inputs in LDS (ascending input order, as the one-pair kernel's sa / sb)
  auto rotate = [&](int v, const double* in, double (&ya)[R], double (&yb)[R]) {
#pragma unroll
    for (int q = 0; q < R; ++q) {
      double sa = 0, sb = 0;
#pragma unroll
      for (int t = 0; t < LR; ++t) sa = fma(rc[q][v][0][t], in[rs[q][v][0][t]], sa);
#pragma unroll
      for (int t = 0; t < LR; ++t) sb = fma(rc[q][v][1][t], in[rs[q][v][1][t]], sb);
      ya[q] = sa; yb[q] = sb;
    }
  };
  // v[n, m] *= e^{i m g}: (c, s) = the class table's (cos m g, sin m g); m = 0 is left alone, as z_rotation does
  auto zrot = [&](const double* cs, bool neg, double (&ya)[R], double (&yb)[R]) {
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const double cm = cs[rm[q]], s0 = cs[PM + rm[q]], sm = neg ? -s0 : s0;
      const double x = ya[q], y = yb[q];
      const double na = fma(x, cm, -(y * sm)), nb = fma(x, sm, y * cm);
      ya[q] = rm[q] ? na : x; yb[q] = rm[q] ? nb : y;
    }
  };
  auto put = [&](double* out, const double (&ya)[R], const double (&yb)[R]) {
#pragma unroll
    for (int q = 0; q < R; ++q)
      if (rok[q]) { out[lane + kWave * q] = ya[q]; out[S + lane + kWave * q] = yb[q]; }
    sl_sync();
  };

  // the shifted expansion of one pair, left in (ya, yb): row (n, m) of the result in the lane that owns it
  auto shift_pair = [&](int pi, double (&ya)[R], double (&yb)[R]) {
    const double2* Msrc = src_base + (size_t)w.src[pi] * box_stride;
    const double* ct = w.class_tab + (size_t)w.cls[pi] * w.class_stride;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const double2 v = rok[q] ? Msrc[lane + kWave * q] : double2{0, 0};
      ya[q] = v.x; yb[q] = v.y;
    }
    zrot(ct, false, ya, yb);                          // e^{i m beta}
    sl_sync();                                        // the previous pair's readers are through with X0
    put(X0, ya, yb);
    rotate(0, X0, ya, yb);
    zrot(ct + 2 * PM, false, ya, yb);                 // e^{i m alpha}
    put(X1, ya, yb);
    rotate(1, X1, ya, yb);
#pragma unroll
    for (int q = 0; q < R; ++q) { const double r = ct[4 * PM + rn[q]]; ya[q] *= r; yb[q] *= r; }   // rho^-n (L2L: rho^n); b[n, 0] is zero
    put(X0, ya, yb);
#pragma unroll
    for (int q = 0; q < R; ++q) {                     // axial shift: order k = the row's m, degrees n ascending, one constant for both parts
      double s1 = 0, s2 = 0;
#pragma unroll
      for (int t = 0; t < LX; ++t) {
        s1 = fma(xc[q][t], X0[xs[q][t]], s1);
        s2 = fma(xc[q][t], X0[S + xs[q][t]], s2);
      }
      const double r = ct[5 * PM + rn[q]];          // rho^j (L2L: rho^-j)
      ya[q] = s1 * r; yb[q] = s2 * r;
    }
    put(X1, ya, yb);
    rotate(0, X1, ya, yb);
    zrot(ct + 2 * PM, true, ya, yb);                  // e^{-i m alpha}
    put(X0, ya, yb);
    rotate(1, X0, ya, yb);
    zrot(ct, true, ya, yb);                           // e^{-i m beta}
  };

  const int wave_id = blockIdx.x * kSlWaves + wv, nwaves = gridDim.x * kSlWaves;
  for (int u = wave_id; u < w.n_units; u += nwaves) {
    double ya[R], yb[R];
    if constexpr (OP == kRotM2M) {
      // parent = ((c0 + c4) + (c1 + c5)) + ((c2 + c6) + (c3 + c7)), absent children zero: the one-pair kernel's order
      double pa[4][R], pb[4][R];
#pragma unroll
      for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int q = 0; q < R; ++q) { pa[h][q] = 0; pb[h][q] = 0; }
      const int c0 = w.unit_ptr[u], c1 = w.unit_ptr[u + 1];
      for (int pi = c0; pi < c1; ++pi) {
        shift_pair(pi, ya, yb);
        const int k = pi - c0;
#pragma unroll
        for (int h = 0; h < 4; ++h)
          if ((k & 3) == h) {
#pragma unroll
            for (int q = 0; q < R; ++q) {
              // first half of the chain pair (k < 4): v + 0 comes later as (v_k + v_{k+4}); keep v and add the partner (or zero) to it
              if (k < 4) { pa[h][q] = ya[q]; pb[h][q] = yb[q]; } else { pa[h][q] += ya[q]; pb[h][q] += yb[q]; }
            }
          }
      }
      const int nch = c1 - c0;
      double2* out = d.M + ((size_t)w.tgt[c0] * d.nslots + slot) * d.s_max;
#pragma unroll
      for (int q = 0; q < R; ++q) {
        // children u >= 4 that do not exist are the zero the one-pair kernel adds: v + 0.0
#pragma unroll
        for (int h = 0; h < 4; ++h)
          if (nch <= h + 4) { pa[h][q] += 0.0; pb[h][q] += 0.0; }
        const double sx = (pa[0][q] + pa[1][q]) + (pa[2][q] + pa[3][q]);
        const double sy = (pb[0][q] + pb[1][q]) + (pb[2][q] + pb[3][q]);
        if (rok[q]) out[lane + kWave * q] = double2{sx, sy};
      }
    } else {
      shift_pair(u, ya, yb);
      double2* own = d.L + ((size_t)w.tgt[u] * d.nslots + slot) * d.s_max;
#pragma unroll
      for (int q = 0; q < R; ++q)
        if (rok[q]) { double2 v = own[lane + kWave * q]; v.x += ya[q]; v.y += yb[q]; own[lane + kWave * q] = v; }
    }
  }
}

template <int OP>
hipError_t launch(const DevicePlan& d, const ShiftLaneWork& w, int p, hipStream_t s) {
  if (w.n_units <= 0) return hipSuccess;
  const int wgs = (w.n_units + kSlWaves - 1) / kSlWaves;
  const dim3 grid(wgs < 256 * 16 ? wgs : 256 * 16, d.n_act), block(kSlWaves * kWave);
#define SL_CASE(PP) case PP: hipLaunchKernelGGL((shift_lanes_kernel<PP, OP>), grid, block, 0, s, d, w); break;
  switch (p) {
    SL_CASE(1) SL_CASE(2) SL_CASE(3) SL_CASE(4) SL_CASE(5) SL_CASE(6) SL_CASE(7) SL_CASE(8) SL_CASE(9) SL_CASE(10) SL_CASE(11) SL_CASE(12)
    default: return hipErrorInvalidValue;
  }
#undef SL_CASE
  return hipGetLastError();
}

}  // namespace

bool shift_lanes_supported(int p) { return p >= 1 && p <= kShiftLanesPmax; }
hipError_t launch_m2m_lanes(const DevicePlan& d, const ShiftLaneWork& w, int p, hipStream_t s) { return launch<kRotM2M>(d, w, p, s); }
hipError_t launch_l2l_lanes(const DevicePlan& d, const ShiftLaneWork& w, int p, hipStream_t s) { return launch<kRotL2L>(d, w, p, s); }

}  // namespace fmmbem
