// Host emulation of the one-pair-per-wavefront shift kernel (csrc/kernels_shift.hip): the same tables (csrc/shift_lanes.hpp
// build_shift_lane_tables, sl_class_table), the same five stages, one "lane" per output row -- so that the tables and the stage
// order are checked on the CPU against the oracle's M2M / L2L (tests/test_rot_operators.py), without a GPU.
// usage: shift_lanes_emulate <P> <op: 1 M2M, 2 L2L> <tx> <ty> <tz>   stdin: S complex coefficients "re im"   stdout: the shifted ones
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "shift_lanes.hpp"

using namespace fmmbem;

int main(int argc, char** argv) {
  const int P = std::atoi(argv[1]), op = std::atoi(argv[2]);
  const double tr[3] = {std::atof(argv[3]), std::atof(argv[4]), std::atof(argv[5])};
  const int S = sl_S(P), LR = sl_rot_len(P), LX = sl_axial_len(P);
  // class record as plan.hip rot_record (cart2sph of the reference, kernel/LaplaceSpherical.hpp:528-541)
  const double kEps = 1e-12;
  double rec[8] = {0};
  const double rho = std::sqrt(tr[0] * tr[0] + tr[1] * tr[1] + tr[2] * tr[2]) + kEps, ca = tr[2] / rho;
  rec[0] = 1.0 / rho; rec[1] = ca; rec[2] = std::sqrt((1.0 - ca) * (1.0 + ca)); rec[5] = rho;
  if (std::fabs(tr[0]) + std::fabs(tr[1]) < kEps) { rec[3] = 1; rec[4] = 0; }
  else if (std::fabs(tr[0]) < kEps) { rec[3] = 0; rec[4] = tr[1] > 0 ? 1.0 : -1.0; }
  else { const double h = 1.0 / std::sqrt(tr[0] * tr[0] + tr[1] * tr[1]); rec[3] = tr[0] * h; rec[4] = tr[1] * h; }
  std::vector<double> ct(sl_class_doubles(P));
  sl_class_table(rec, P, op, ct.data());
  ShiftLaneTables t;
  build_shift_lane_tables(P, op, t);
  std::vector<double> X(2 * S), Y(2 * S);
  for (int i = 0; i < S; ++i)
    if (std::scanf("%lf %lf", &X[i], &X[S + i]) != 2) return 1;
  auto turn = [&](const double* c, const double* s, bool neg) {
    for (int r = 0; r < S; ++r) {
      const int m = sl_row_m(r);
      if (!m) continue;
      const double cm = c[m], sm = neg ? -s[m] : s[m], x = X[r], y = X[S + r];
      X[r] = std::fma(x, cm, -(y * sm)); X[S + r] = std::fma(x, sm, y * cm);
    }
  };
  auto rotate = [&](int v) {
    for (int r = 0; r < S; ++r)
      for (int c = 0; c < 2; ++c) {
        double acc = 0;
        for (int k = 0; k < LR; ++k) {
          const size_t e = ((size_t)(v * 2 + c) * LR + k) * S + r;
          acc = std::fma(t.rot_c[e], X[t.rot_s[e]], acc);
        }
        Y[c * S + r] = acc;
      }
    X = Y;
  };
  turn(ct.data(), ct.data() + P, false);
  rotate(0);
  turn(ct.data() + 2 * P, ct.data() + 3 * P, false);
  rotate(1);
  for (int r = 0; r < S; ++r) { X[r] *= ct[4 * P + sl_row_n(r)]; X[S + r] *= ct[4 * P + sl_row_n(r)]; }
  for (int r = 0; r < S; ++r) {
    double s1 = 0, s2 = 0;
    for (int k = 0; k < LX; ++k) {
      const size_t e = (size_t)k * S + r;
      s1 = std::fma(t.ax_c[e], X[t.ax_s[e]], s1);
      s2 = std::fma(t.ax_c[e], X[S + t.ax_s[e]], s2);
    }
    Y[r] = s1 * ct[5 * P + sl_row_n(r)]; Y[S + r] = s2 * ct[5 * P + sl_row_n(r)];
  }
  X = Y;
  rotate(0);
  turn(ct.data() + 2 * P, ct.data() + 3 * P, true);
  rotate(1);
  turn(ct.data(), ct.data() + P, true);
  for (int i = 0; i < S; ++i) std::printf("%.17g %.17g\n", X[i], X[S + i]);
  return 0;
}
