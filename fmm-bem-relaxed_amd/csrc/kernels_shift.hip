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

// GROUP (M2M only): a WORKGROUP takes a parent and its four wavefronts the parent's children -- wavefront h the children h and
// h + 4, i.e. chain h of the sum -- for the levels near the root and for a shard's levels, where there are fewer parents than
// wavefronts on the chip and a wavefront that walks a parent's children one after the other is what one waits for (12 us per
// launch for eight children against 5 for the L2L of the same level).  Same operations, same order, same bits.
#ifndef FMMBEM_SL_OCC
#define FMMBEM_SL_OCC(P) 0                            // wavefronts per SIMD asked of the compiler; 0 = its own choice (2 at p = 7 ... 10)
#endif
template <int P, int OP, bool GROUP>
__global__ __launch_bounds__(kSlWaves * kWave) __attribute__((amdgpu_waves_per_eu(FMMBEM_SL_OCC(P) ? FMMBEM_SL_OCC(P) : 1, FMMBEM_SL_OCC(P) ? FMMBEM_SL_OCC(P) : 8)))
void shift_lanes_kernel(const DevicePlan d, const ShiftLaneWork w) {
  constexpr int S = sl_S(P), R = sl_rounds(P), LR = sl_rot_len(P), LX = sl_axial_len(P);
  constexpr int XS = 2 * S + 2;                       // a[S] then b[S]
  __shared__ double xbuf[kSlWaves][2][XS];
  __shared__ double part[GROUP ? 2 : 1][kSlWaves][GROUP ? XS : 1];   // GROUP: the four chain sums of a parent, double-buffered
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
  auto put = [&](double* out, const double (&ya)[R], const double (&yb)[R]) {
#pragma unroll
    for (int q = 0; q < R; ++q)
      if (rok[q]) { out[lane + kWave * q] = ya[q]; out[S + lane + kWave * q] = yb[q]; }
    sl_sync();
  };

  // What a pair brings from memory, fetched one pair AHEAD (a wavefront walks consecutive pairs: two dependent round trips --
  // index, then expansion and class values -- would otherwise stand in front of every pair's ~1 500 cycles of work; measured
  // 0.161 ms against 0.081 for the one-pair kernel's M2M at N = 1M, p = 10 without it): the source expansion and the six class
  // values of each of the lane's rows.
  struct Operands { double ma[R], mb[R], cb[R], sb[R], ca[R], sa[R], r1[R], r2[R], oa[R], ob[R]; };   // oa, ob: L2L, the child's own L
  auto fetch = [&](int src_box, int cls, int tgt_box, Operands& o) {
    const double2* Msrc = src_base + (size_t)src_box * box_stride;
    if constexpr (OP == kRotL2L) {
      const double2* own = d.L + ((size_t)tgt_box * d.nslots + slot) * d.s_max;
#pragma unroll
      for (int q = 0; q < R; ++q) { const double2 v = rok[q] ? own[lane + kWave * q] : double2{0, 0}; o.oa[q] = v.x; o.ob[q] = v.y; }
    }
    const double* ct = w.class_tab + (size_t)cls * w.class_stride;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const double2 v = rok[q] ? Msrc[lane + kWave * q] : double2{0, 0};
      o.ma[q] = v.x; o.mb[q] = v.y;
      o.cb[q] = ct[rm[q]]; o.sb[q] = ct[PM + rm[q]]; o.ca[q] = ct[2 * PM + rm[q]]; o.sa[q] = ct[3 * PM + rm[q]];
      o.r1[q] = ct[4 * PM + rn[q]]; o.r2[q] = ct[5 * PM + rn[q]];
    }
  };
  // v[n, m] *= e^{+-i m g}; m = 0 is left alone, as z_rotation does
  auto turn = [&](const double (&c)[R], const double (&sn)[R], bool neg, double (&ya)[R], double (&yb)[R]) {
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const double cm = c[q], sm = neg ? -sn[q] : sn[q];
      const double x = ya[q], y = yb[q];
      const double na = fma(x, cm, -(y * sm)), nb = fma(x, sm, y * cm);
      ya[q] = rm[q] ? na : x; yb[q] = rm[q] ? nb : y;
    }
  };
  // the shifted expansion of one pair, left in (ya, yb): row (n, m) of the result in the lane that owns it
  auto shift_pair = [&](const Operands& o, double (&ya)[R], double (&yb)[R]) {
#pragma unroll
    for (int q = 0; q < R; ++q) { ya[q] = o.ma[q]; yb[q] = o.mb[q]; }
    turn(o.cb, o.sb, false, ya, yb);                  // e^{i m beta}
    sl_sync();                                        // the previous pair's readers are through with X0
    put(X0, ya, yb);
    rotate(0, X0, ya, yb);
    turn(o.ca, o.sa, false, ya, yb);                  // e^{i m alpha}
    put(X1, ya, yb);
    rotate(1, X1, ya, yb);
#pragma unroll
    for (int q = 0; q < R; ++q) { ya[q] *= o.r1[q]; yb[q] *= o.r1[q]; }   // rho^-n (L2L: rho^n); b[n, 0] is zero
    put(X0, ya, yb);
#pragma unroll
    for (int q = 0; q < R; ++q) {                     // axial shift: order k = the row's m, degrees n ascending, one constant for both parts
      double s1 = 0, s2 = 0;
#pragma unroll
      for (int t = 0; t < LX; ++t) {
        s1 = fma(xc[q][t], X0[xs[q][t]], s1);
        s2 = fma(xc[q][t], X0[S + xs[q][t]], s2);
      }
      ya[q] = s1 * o.r2[q]; yb[q] = s2 * o.r2[q];     // rho^j (L2L: rho^-j)
    }
    put(X1, ya, yb);
    rotate(0, X1, ya, yb);
    turn(o.ca, o.sa, true, ya, yb);                   // e^{-i m alpha}
    put(X0, ya, yb);
    rotate(1, X0, ya, yb);
    turn(o.cb, o.sb, true, ya, yb);                   // e^{-i m beta}
  };

  if constexpr (GROUP) {
    // a workgroup takes a contiguous run of parents; wavefront h of it the children h and h + 4 of each
    const int h = __builtin_amdgcn_readfirstlane(wv);
    const int g0 = (int)((long long)w.n_units * blockIdx.x / gridDim.x), g1 = (int)((long long)w.n_units * (blockIdx.x + 1) / gridDim.x);
    Operands cur, sec;
    int c0 = g0 < g1 ? w.unit_ptr[g0] : 0, c1 = g0 < g1 ? w.unit_ptr[g0 + 1] : 0;
    if (g0 < g1 && c0 + h < c1) fetch(w.src[c0 + h], w.cls[c0 + h], 0, cur);
    int buf = 0;
    for (int u = g0; u < g1; ++u, buf ^= 1) {
      const int nch = c1 - c0, tgt_box = w.tgt[c0];
      const Operands o = cur;
      const bool two = h + 4 < nch;
      if (two) fetch(w.src[c0 + h + 4], w.cls[c0 + h + 4], 0, sec);
      // the next parent's child h: on its way while this parent's are computed
      const int n0 = c1, n1 = u + 1 < g1 ? w.unit_ptr[u + 2] : c1;
      if (u + 1 < g1 && n0 + h < n1) fetch(w.src[n0 + h], w.cls[n0 + h], 0, cur);
      double ya[R], yb[R], sa_[R], sb_[R];
#pragma unroll
      for (int q = 0; q < R; ++q) { sa_[q] = 0; sb_[q] = 0; }
      if (h < nch) {
        shift_pair(o, ya, yb);
#pragma unroll
        for (int q = 0; q < R; ++q) { sa_[q] = ya[q]; sb_[q] = yb[q]; }
      }
      if (two) {
        shift_pair(sec, ya, yb);
#pragma unroll
        for (int q = 0; q < R; ++q) { sa_[q] += ya[q]; sb_[q] += yb[q]; }
      } else {
#pragma unroll
        for (int q = 0; q < R; ++q) { sa_[q] += 0.0; sb_[q] += 0.0; }      // the absent child h + 4: v + 0.0, as the one-pair kernel adds it
      }
#pragma unroll
      for (int q = 0; q < R; ++q)
        if (rok[q]) { part[buf][h][lane + kWave * q] = sa_[q]; part[buf][h][S + lane + kWave * q] = sb_[q]; }
      __syncthreads();
      if (h == 0) {
        double2* out = d.M + ((size_t)tgt_box * d.nslots + slot) * d.s_max;
#pragma unroll
        for (int q = 0; q < R; ++q)
          if (rok[q]) {
            const int r = lane + kWave * q;
            const double sx = (part[buf][0][r] + part[buf][1][r]) + (part[buf][2][r] + part[buf][3][r]);
            const double sy = (part[buf][0][S + r] + part[buf][1][S + r]) + (part[buf][2][S + r] + part[buf][3][S + r]);
            out[r] = double2{sx, sy};
          }
      }
      c0 = n0; c1 = n1;
    }
    return;
  }
  // a wavefront takes a contiguous run of units, i.e. of pairs: [pb, pe)
  const int wave_id = __builtin_amdgcn_readfirstlane(blockIdx.x * kSlWaves + wv), nwaves = gridDim.x * kSlWaves;
  const int u0 = (int)((long long)w.n_units * wave_id / nwaves), u1 = (int)((long long)w.n_units * (wave_id + 1) / nwaves);
  if (u0 >= u1) return;
  const int pb = OP == kRotM2M ? w.unit_ptr[u0] : u0, pe = OP == kRotM2M ? w.unit_ptr[u1] : u1;
  Operands cur, nxt;
  int tgt = w.tgt[pb], ntgt = tgt, n2src = 0, n2cls = 0, n2tgt = 0;
  fetch(w.src[pb], w.cls[pb], tgt, cur);
  if (pb + 1 < pe) { ntgt = w.tgt[pb + 1]; fetch(w.src[pb + 1], w.cls[pb + 1], ntgt, nxt); }
  if (pb + 2 < pe) { n2src = w.src[pb + 2]; n2cls = w.cls[pb + 2]; n2tgt = w.tgt[pb + 2]; }
  // M2M: parent = ((c0 + c4) + (c1 + c5)) + ((c2 + c6) + (c3 + c7)), absent children zero: the one-pair kernel's order
  double pa[4][R], pb_[4][R];
  int k = 0;                                          // position of the current pair among its parent's children
#pragma unroll
  for (int h = 0; h < 4; ++h)
#pragma unroll
    for (int q = 0; q < R; ++q) { pa[h][q] = 0; pb_[h][q] = 0; }
  for (int pi = pb; pi < pe; ++pi) {
    double ya[R], yb[R];
    const Operands o = cur;
    const int this_tgt = tgt;
    const bool more = pi + 1 < pe;
    const bool last_child = !more || ntgt != tgt;
    // hand over: the pair after this one arrived while the previous pair was computed; ask for the one after that
    cur = nxt; tgt = ntgt;
    if (pi + 2 < pe) { ntgt = n2tgt; fetch(n2src, n2cls, n2tgt, nxt); }
    if (pi + 3 < pe) { n2src = w.src[pi + 3]; n2cls = w.cls[pi + 3]; n2tgt = w.tgt[pi + 3]; }
    shift_pair(o, ya, yb);
    if constexpr (OP == kRotM2M) {
#pragma unroll
      for (int h = 0; h < 4; ++h)
        if ((k & 3) == h) {
#pragma unroll
          for (int q = 0; q < R; ++q) {
            if (k < 4) { pa[h][q] = ya[q]; pb_[h][q] = yb[q]; } else { pa[h][q] += ya[q]; pb_[h][q] += yb[q]; }
          }
        }
      ++k;
      if (last_child) {
        double2* out = d.M + ((size_t)this_tgt * d.nslots + slot) * d.s_max;
#pragma unroll
        for (int q = 0; q < R; ++q) {
          // children h + 4 that do not exist are the zero the one-pair kernel adds: v + 0.0
#pragma unroll
          for (int h = 0; h < 4; ++h)
            if (k <= h + 4) { pa[h][q] += 0.0; pb_[h][q] += 0.0; }
          const double sx = (pa[0][q] + pa[1][q]) + (pa[2][q] + pa[3][q]);
          const double sy = (pb_[0][q] + pb_[1][q]) + (pb_[2][q] + pb_[3][q]);
          if (rok[q]) out[lane + kWave * q] = double2{sx, sy};
#pragma unroll
          for (int h = 0; h < 4; ++h) { pa[h][q] = 0; pb_[h][q] = 0; }
        }
        k = 0;
      }
    } else {
      // (a child is the target of ONE pair: what was fetched ahead is still its L)
      double2* own = d.L + ((size_t)this_tgt * d.nslots + slot) * d.s_max;
#pragma unroll
      for (int q = 0; q < R; ++q)
        if (rok[q]) own[lane + kWave * q] = double2{o.oa[q] + ya[q], o.ob[q] + yb[q]};
    }
  }
}

template <int OP>
hipError_t launch(const DevicePlan& d, const ShiftLaneWork& w, int p, hipStream_t s) {
  if (w.n_units <= 0) return hipSuccess;
  // what is resident at once (two wavefronts per SIMD at p = 10), each wavefront on a contiguous run of units
  const int wgs = (w.n_units + kSlWaves - 1) / kSlWaves;
  const dim3 block(kSlWaves * kWave);
  constexpr int kGroupMax = 1024;                       // parents up to which a workgroup per parent wins
  if (OP == kRotM2M && w.n_units <= kGroupMax) {
    const dim3 grid(w.n_units < 256 * 2 ? w.n_units : 256 * 2, d.n_act);
#define SL_CASE(PP) case PP: hipLaunchKernelGGL((shift_lanes_kernel<PP, kRotM2M, true>), grid, block, 0, s, d, w); break;
    switch (p) {
      SL_CASE(1) SL_CASE(2) SL_CASE(3) SL_CASE(4) SL_CASE(5) SL_CASE(6) SL_CASE(7) SL_CASE(8) SL_CASE(9) SL_CASE(10) SL_CASE(11) SL_CASE(12)
      default: return hipErrorInvalidValue;
    }
#undef SL_CASE
    return hipGetLastError();
  }
  const dim3 grid(wgs < 256 * 2 ? wgs : 256 * 2, d.n_act);
#define SL_CASE(PP) case PP: hipLaunchKernelGGL((shift_lanes_kernel<PP, OP, false>), grid, block, 0, s, d, w); break;
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
