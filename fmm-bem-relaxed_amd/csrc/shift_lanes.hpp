// shift_lanes.hpp -- M2M / L2L of the tree passes with one (parent, child) pair on a whole WAVEFRONT: lane = output coefficient.
// Reference: LaplaceSpherical::M2M (kernel/LaplaceSpherical.hpp:245-285) and ::L2L (:378-411), one call per box pair of
// M2M_list / L2L_list (executor/EvalInteractionLazySparse.hpp:185-190, 209-214).
//
// The operator is the rotation / axial shift / rotation-back factorisation of m2l_rot.hpp (section "The same factorisation serves
// the two shifts"), the constants are build_rot_stream's, and every output coefficient is formed by the SAME sequence of
// floating-point operations as in the one-pair-per-lane kernels (kernels_m2l_rot.hip with FMMBEM_ROT_OP = 1, 2): a chain of FMAs
// over ascending input order per rotation row, over ascending degree per axial row, the z rotations and powers of rho from the
// same recurrences.  Only the mapping differs: there a lane runs all ~1 500 FMAs of a pair out of its registers (a pass takes
// ~20 us however few pairs a level or a shard holds); here the 64 lanes of a wavefront share one pair, a lane owns one output
// row per stage (~10 FMAs), the pair's coefficients go from stage to stage through LDS, and the constants a lane needs for its
// rows stay in its registers across the pairs the wavefront walks.  Same bits, a few hundred instructions per pair.
//
// This header: the row / chain structure (constexpr, shared by host and kernel) and the host-side tables.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "m2l_rot.hpp"

namespace fmmbem {

constexpr int kShiftLanesPmax = 12;                 // instantiated orders (the orders the one-pair kernels exist for)
constexpr int sl_S(int P) { return P * (P + 1) / 2; }
constexpr int sl_rounds(int P) { return (sl_S(P) + 63) / 64; }   // rows per lane: row r = lane + 64 * round
constexpr int sl_row_n(int r) { int n = 0; while ((n + 1) * (n + 2) / 2 <= r) ++n; return n; }
constexpr int sl_row_m(int r) { return r - sl_row_n(r) * (sl_row_n(r) + 1) / 2; }

// rotation row (n, m): the live entries mp = 0..n in ascending order split into the chain of the real part (routing phase even)
// and of the imaginary part (odd) -- fixed_rotation of kernels_m2l_rot.hip, "sa" and "sb"
constexpr int sl_chain_len(int n, int m, int chain) {
  int c = 0;
  for (int mp = 0; mp <= n; ++mp)
    if (rot_live(n, m, mp) && (rot_kk(n, m, mp) & 1) == chain) ++c;
  return c;
}
constexpr int sl_rot_len(int P) {                   // longest chain of any row: every lane runs that many terms (padded with zeros)
  int best = 1;
  for (int n = 1; n < P; ++n)
    for (int m = 0; m <= n; ++m)
      for (int c = 0; c < 2; ++c) best = sl_chain_len(n, m, c) > best ? sl_chain_len(n, m, c) : best;
  return best;
}
constexpr int sl_axial_len(int P) { return P; }     // longest axial row (j = P - 1, k = 0 for M2M; j = k = 0 for L2L)

// per class (translation vector of a parent-child pair): what the z rotations and the scalings of a pair multiply by, from the
// class record (1/rho, cos a, sin a, cos b, sin b, rho) by the recurrences of z_rotation / the scale loops of the one-pair kernel
//   [0*P + m] cos(m beta)   [1*P + m] sin(m beta)   [2*P + m] cos(m alpha)   [3*P + m] sin(m alpha)     m = 0: (1, 0), unused
//   [4*P + n] first scaling (M2M rho^-n, L2L rho^n)      [5*P + j] second scaling (M2M rho^j, L2L rho^-j)
constexpr int sl_class_doubles(int Pmax) { return 6 * Pmax; }

inline void sl_class_table(const double rec[8], int Pmax, int op, double* out) {
  auto zrot = [&](double c1, double s1, double* c, double* s) {
    c[0] = 1.0; s[0] = 0.0;
    double cm = c1, sm = s1;
    for (int m = 1; m < Pmax; ++m) {
      c[m] = cm; s[m] = sm;
      const double c2 = std::fma(cm, c1, -(sm * s1)), s2 = std::fma(sm, c1, cm * s1);
      cm = c2; sm = s2;
    }
  };
  zrot(rec[3], rec[4], out, out + Pmax);
  zrot(rec[1], rec[2], out + 2 * Pmax, out + 3 * Pmax);
  const double b1 = op == kRotL2L ? rec[5] : rec[0], b2 = op == kRotM2M ? rec[5] : rec[0];
  double r = b1;
  out[4 * Pmax] = 1.0;
  for (int n = 1; n < Pmax; ++n) { out[4 * Pmax + n] = r; r *= b1; }
  r = 1.0;
  out[5 * Pmax] = 1.0;
  for (int j = 1; j < Pmax; ++j) { r *= b2; out[5 * Pmax + j] = r; }
}

// per order and operator: what lane `row % 64` needs for row `row`, [term][row] so that a wavefront reads a term's 64 rows at once
//   rot_c / rot_s   [variant 2 (conj(X), X^T)][chain 2][sl_rot_len][S]   constant, LDS index of its operand (a[i]: i, b[i]: S + i)
//   ax_c / ax_s     [sl_axial_len][S]                                     T[j, n, k], index of a[n, k] (b[n, k] is S further)
struct ShiftLaneTables {
  std::vector<double> rot_c, ax_c;
  std::vector<int32_t> rot_s, ax_s;
};

inline void build_shift_lane_tables(int P, int op, ShiftLaneTables& t) {
  std::vector<double> stream;
  build_rot_stream(P, stream, op);                   // the one-pair kernel's constants, signs folded
  const int S = sl_S(P), LR = sl_rot_len(P), LX = sl_axial_len(P);
  t.rot_c.assign((size_t)2 * 2 * LR * S, 0.0);
  t.rot_s.assign((size_t)2 * 2 * LR * S, 0);
  t.ax_c.assign((size_t)LX * S, 0.0);
  t.ax_s.assign((size_t)LX * S, 0);
  auto idx = [](int n, int m) { return n * (n + 1) / 2 + m; };
  for (int v = 0; v < 2; ++v) {
    // degree 0 is the identity in the one-pair kernel: real part = 1.0 * a[0] (exact), imaginary part stays the zero it is
    t.rot_c[((size_t)(v * 2 + 0) * LR + 0) * S + 0] = 1.0;
    for (int n = 1; n < P; ++n)
      for (int m = 0; m <= n; ++m) {
        int at[2] = {0, 0};
        for (int mp = 0; mp <= n; ++mp) {
          if (!rot_live(n, m, mp)) continue;
          const int chain = rot_kk(n, m, mp) & 1;
          const bool even = ((n + m) & 1) == 0;
          const size_t e = ((size_t)(v * 2 + chain) * LR + at[chain]++) * S + idx(n, m);
          t.rot_c[e] = stream[(size_t)rot_stage_base(P, v, op) + rot_index(n, m, mp)];
          t.rot_s[e] = (mp == 0 || even) ? idx(n, mp) : S + idx(n, mp);
        }
      }
  }
  for (int k = 0; k < P; ++k)
    for (int j = k; j < P; ++j) {
      int at = 0;
      for (int n = axial_row_begin(P, op, k, j); n < axial_row_end(P, op, k, j); ++n, ++at) {
        const size_t e = (size_t)at * S + idx(j, k);
        t.ax_c[e] = stream[(size_t)rot_stage_base(P, 2, op) + axial_index(P, op, k, j, n)];
        t.ax_s[e] = idx(n, k);
      }
    }
}

// One launch = one tree level.  M2M: unit u = a parent, its children are pairs [unit_ptr[u], unit_ptr[u+1]) of (src = child,
// cls, tgt = parent), at most eight, in box order.  L2L: unit = pair u of (src = parent, cls, tgt = child), unit_ptr unused.
struct ShiftLaneWork {
  const int *src = nullptr, *cls = nullptr, *tgt = nullptr, *unit_ptr = nullptr;
  int n_units = 0;
  const double* class_tab = nullptr;                // [class][sl_class_doubles(p_max)]
  int class_stride = 0, p_max = 0;
  const double *rot_c = nullptr, *ax_c = nullptr;   // this order's tables
  const int32_t *rot_s = nullptr, *ax_s = nullptr;
};

}  // namespace fmmbem
