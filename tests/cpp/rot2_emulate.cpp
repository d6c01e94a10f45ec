// CPU emulation of the two forms of the rotation M2L program, constant by constant in the order the kernels consume them
// (csrc/m2l_rot.hpp build_rot_stream / build_rot2_stream): the one-pair-per-lane form of kernels_m2l_rot.hip and the split
// form of kernels_m2l_rot2.hip (a pair on an E lane with the even degrees and an O lane with the odd ones, one exchange in the
// axial translation).  Prints, per order p, the largest relative difference between the two over random multipoles and random
// translation vectors.  Plain g++, no device.   usage: rot2_emulate [trials]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../fmm-bem-relaxed_amd/csrc/m2l_rot.hpp"

using namespace fmmbem;

static int idx_of(int n, int m) { return n * (n + 1) / 2 + m; }

struct Geo { double inv_rho, ca, sa, cb, sb, rho; };

// ---- the form of kernels_m2l_rot.hip ----
static void plain_program(int P, int op, const std::vector<double>& st, const Geo& g, std::vector<double>& a, std::vector<double>& b) {
  auto zrot = [&](double c1, double s1) {
    double cm = c1, sm = s1;
    for (int m = 1; m < P; ++m) {
      for (int n = m; n < P; ++n) {
        const double x = a[idx_of(n, m)], y = b[idx_of(n, m)];
        a[idx_of(n, m)] = std::fma(x, cm, -(y * sm));
        b[idx_of(n, m)] = std::fma(x, sm, y * cm);
      }
      const double c2 = std::fma(cm, c1, -(sm * s1)), s2 = std::fma(sm, c1, cm * s1);
      cm = c2; sm = s2;
    }
  };
  auto fixed = [&](int stage) {
    for (int n = 1; n < P; ++n) {
      std::vector<double> na(n + 1), nb(n + 1);
      for (int m = 0; m <= n; ++m) {
        double sa = 0, sb = 0;
        for (int mp = 0; mp <= n; ++mp) {
          if (!rot_live(n, m, mp)) continue;
          const double c = st[(size_t)rot_stage_base(P, stage, op) + rot_index(n, m, mp)];
          const bool even = ((n + m) & 1) == 0;
          const double src = (mp == 0 || even) ? a[idx_of(n, mp)] : b[idx_of(n, mp)];
          if ((rot_kk(n, m, mp) & 1) == 0) sa = std::fma(c, src, sa); else sb = std::fma(c, src, sb);
        }
        na[m] = sa; nb[m] = sb;
      }
      for (int m = 0; m <= n; ++m) { a[idx_of(n, m)] = na[m]; b[idx_of(n, m)] = nb[m]; }
    }
  };
  zrot(g.cb, g.sb); fixed(0); zrot(g.ca, g.sa); fixed(1);
  { const double base = op == kRotL2L ? g.rho : g.inv_rho;          // M2L, M2M: rho^-n;  L2L: rho^n  (kernels_m2l_rot.hip)
    double r = base; for (int n = 1; n < P; ++n) { for (int m = 0; m <= n; ++m) { a[idx_of(n, m)] *= r; if (m) b[idx_of(n, m)] *= r; } r *= base; } }
  for (int k = 0; k < P; ++k) {
    std::vector<double> la(P), lb(P);
    for (int j = k; j < P; ++j) {
      double s1 = 0, s2 = 0;
      for (int n = axial_row_begin(P, op, k, j); n < axial_row_end(P, op, k, j); ++n) {
        const double c = st[(size_t)rot_stage_base(P, 2, op) + axial_index(P, op, k, j, n)];
        s1 = std::fma(c, a[idx_of(n, k)], s1);
        if (k) s2 = std::fma(c, b[idx_of(n, k)], s2);
      }
      la[j] = s1; lb[j] = s2;
    }
    for (int j = k; j < P; ++j) { a[idx_of(j, k)] = la[j]; b[idx_of(j, k)] = lb[j]; }
  }
  { const double base = op == kRotM2M ? g.rho : g.inv_rho;          // M2L: rho^-(j+1);  M2M: rho^j;  L2L: rho^-j
    double r = op == kRotM2L ? base : 1.0;
    for (int j = 0; j < P; ++j) { for (int k = 0; k <= j; ++k) { a[idx_of(j, k)] *= r; if (k) b[idx_of(j, k)] *= r; } r *= base; } }
  fixed(3); zrot(g.ca, -g.sa); fixed(4); zrot(g.cb, -g.sb);
}

// ---- the split form: lane[0] = E (even degrees), lane[1] = O (odd degrees); slots (q, t) ----
static void split_program(int P, int op, const std::vector<double>& st2, const Geo& g, std::vector<double>& a, std::vector<double>& b) {
  const int Q = rot2_pairs(P), NS = rot2_nslots(P);
  std::vector<double> A[2] = {std::vector<double>(NS, 0.0), std::vector<double>(NS, 0.0)}, B[2] = {std::vector<double>(NS, 0.0), std::vector<double>(NS, 0.0)};
  auto cst = [&](int e, int par) { return st2[(size_t)(e / kRotGroup) * 2 * kRotGroup + (size_t)par * kRotGroup + e % kRotGroup]; };
  auto coeff = [&](int par, int q, int t, int& n, int& m) { n = 2 * q + par; m = par ? t : t - 1; return n < P && m >= 0 && m <= n; };
  for (int par = 0; par < 2; ++par)
    for (int q = 0; q < Q; ++q)
      for (int t = 0; t <= 2 * q + 1; ++t) {
        int n, m;
        if (coeff(par, q, t, n, m)) { A[par][rot2_sidx(q, t)] = a[idx_of(n, m)]; B[par][rot2_sidx(q, t)] = b[idx_of(n, m)]; }
      }
  auto zrot = [&](double c1, double s1) {
    for (int par = 0; par < 2; ++par) {
      double cm = par ? c1 : 1.0, sm = par ? s1 : 0.0;               // slot t turns by e^{i t g} (O) / e^{i (t-1) g} (E)
      for (int t = 1; t < 2 * Q; ++t) {
        for (int q = rot2_qmin(t); q < Q; ++q) {
          const int s = rot2_sidx(q, t);
          const double x = A[par][s], y = B[par][s];
          A[par][s] = std::fma(x, cm, -(y * sm));
          B[par][s] = std::fma(x, sm, y * cm);
        }
        const double c2 = std::fma(cm, c1, -(sm * s1)), s2 = std::fma(sm, c1, cm * s1);
        cm = c2; sm = s2;
      }
    }
  };
  auto fixed = [&](int stage) {
    for (int par = 0; par < 2; ++par)
      for (int q = 0; q < Q; ++q) {
        const int n = 2 * q + 1;                                      // the skeleton: the odd degree's program, both lanes
        std::vector<double> na(n + 1), nb(n + 1);
        for (int m = 0; m <= n; ++m) {
          double sa = 0, sb = 0;
          for (int mp = 0; mp <= n; ++mp) {
            if (!rot_live(n, m, mp)) continue;
            const double c = cst(rot2_stage_base(P, stage) + rot2_rot_index(q, m, mp), par);
            const bool even = ((n + m) & 1) == 0;
            const double src = (mp == 0 || even) ? A[par][rot2_sidx(q, mp)] : B[par][rot2_sidx(q, mp)];
            if ((rot_kk(n, m, mp) & 1) == 0) sa = std::fma(c, src, sa); else sb = std::fma(c, src, sb);
          }
          na[m] = sa; nb[m] = sb;
        }
        for (int m = 0; m <= n; ++m) { A[par][rot2_sidx(q, m)] = na[m]; B[par][rot2_sidx(q, m)] = nb[m]; }
      }
  };
  auto scale = [&](double first_e, double first_o, double step) {     // per degree pair times step
    for (int par = 0; par < 2; ++par) {
      double r = par ? first_o : first_e;
      for (int q = 0; q < Q; ++q) {
        for (int t = 0; t <= 2 * q + 1; ++t) { A[par][rot2_sidx(q, t)] *= r; B[par][rot2_sidx(q, t)] *= r; }
        r *= step;
      }
    }
  };
  const double pre = op == kRotL2L ? g.rho : g.inv_rho, post = op == kRotM2M ? g.rho : g.inv_rho;
  zrot(g.cb, g.sb); fixed(0); zrot(g.ca, g.sa); fixed(1);
  scale(1.0, pre, pre * pre);                                         // pre^n: E degree 2q, O degree 2q+1
  {
    std::vector<double> pendA(Q, 0.0), pendB(Q, 0.0);                 // E lane: what O sent at the previous step, for slot t
    for (int t = 0; t < 2 * Q; ++t) {
      const int q0 = rot2_qmin(t);
      std::vector<double> ownA[2], ownB[2], othA[2], othB[2];
      for (int par = 0; par < 2; ++par) {
        ownA[par].assign(Q, 0.0); ownB[par].assign(Q, 0.0); othA[par].assign(Q, 0.0); othB[par].assign(Q, 0.0);
        for (int qo = rot2_qmin_other(t); qo < Q; ++qo)
          for (int other = qo >= q0 ? 0 : 1; other < 2; ++other) {
            double s1 = 0, s2 = 0;
            for (int qi = q0; qi < Q; ++qi) {
              const double c = cst(rot2_stage_base(P, 2) + rot2_axial_index(P, t, qo, other, qi), par);
              s1 = std::fma(c, A[par][rot2_sidx(qi, t)], s1);
              if (t) s2 = std::fma(c, B[par][rot2_sidx(qi, t)], s2);
            }
            (other ? othA : ownA)[par][qo] = s1; (other ? othB : ownB)[par][qo] = s2;
          }
      }
      for (int par = 0; par < 2; ++par)
        for (int qo = q0; qo < Q; ++qo) { A[par][rot2_sidx(qo, t)] = ownA[par][qo]; B[par][rot2_sidx(qo, t)] = ownB[par][qo]; }
      // the exchange: O receives E's "other" sums -> O's slot (qo, t - 1); E receives O's -> E's slot (qo, t + 1), one step later
      if (t >= 1)
        for (int qo = rot2_qmin_other(t); qo < Q; ++qo) { A[1][rot2_sidx(qo, t - 1)] += othA[0][qo]; B[1][rot2_sidx(qo, t - 1)] += othB[0][qo]; }
      for (int qo = q0; qo < Q; ++qo) { A[0][rot2_sidx(qo, t)] += pendA[qo]; B[0][rot2_sidx(qo, t)] += pendB[qo]; }
      for (int qo = 0; qo < Q; ++qo) { pendA[qo] = qo >= q0 ? othA[1][qo] : 0.0; pendB[qo] = qo >= q0 ? othB[1][qo] : 0.0; }
      // (a slot (qo, t + 1) that does not exist -- t + 1 > 2 qo + 1 -- receives a zero: its constants are zero)
      for (int qo = q0; qo < Q; ++qo) if (t + 1 > 2 * qo + 1) { pendA[qo] = 0; pendB[qo] = 0; }
    }
  }
  if (op == kRotM2L) scale(post, post * post, post * post);           // rho^-(j+1)
  else scale(1.0, post, post * post);                                 // M2M rho^j, L2L rho^-j
  fixed(3); zrot(g.ca, -g.sa); fixed(4); zrot(g.cb, -g.sb);
  for (int par = 0; par < 2; ++par)
    for (int q = 0; q < Q; ++q)
      for (int t = 0; t <= 2 * q + 1; ++t) {
        int n, m;
        if (coeff(par, q, t, n, m)) { a[idx_of(n, m)] = A[par][rot2_sidx(q, t)]; b[idx_of(n, m)] = B[par][rot2_sidx(q, t)]; }
      }
}

int main(int argc, char** argv) {
  const int trials = argc > 1 ? std::atoi(argv[1]) : 20;
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  double worst_all = 0;
  const char* names[3] = {"M2L", "M2M", "L2L"};
  for (int op = 0; op < 3; ++op)
  for (int P = 1; P <= kRotPmax; ++P) {
    std::vector<double> st, st2;
    build_rot_stream(P, st, op);
    build_rot2_stream(P, st2, op);
    const int S = P * (P + 1) / 2;
    double worst = 0;
    for (int it = 0; it < trials; ++it) {
      std::vector<double> a(S), b(S);
      for (int n = 0; n < P; ++n)
        for (int m = 0; m <= n; ++m) { a[idx_of(n, m)] = U(rng); b[idx_of(n, m)] = m ? U(rng) : 0.0; }
      double v[3] = {U(rng) * 3, U(rng) * 3, U(rng) * 3};
      if (it == 0) { v[0] = 0; v[1] = 0; v[2] = 2.5; }                  // along the axis
      const double rho = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + 1e-12, h = std::hypot(v[0], v[1]);
      Geo g;
      g.rho = rho;
      g.inv_rho = 1.0 / rho; g.ca = v[2] / rho; g.sa = std::sqrt((1 - g.ca) * (1 + g.ca));
      g.cb = h > 0 ? v[0] / h : 1.0; g.sb = h > 0 ? v[1] / h : 0.0;
      std::vector<double> a1 = a, b1 = b, a2 = a, b2 = b;
      plain_program(P, op, st, g, a1, b1);
      split_program(P, op, st2, g, a2, b2);
      double num = 0, den = 0;
      for (int i = 0; i < S; ++i) { num = std::max(num, std::max(std::fabs(a1[i] - a2[i]), std::fabs(b1[i] - b2[i]))); den = std::max(den, std::max(std::fabs(a1[i]), std::fabs(b1[i]))); }
      worst = std::max(worst, num / den);
    }
    std::printf("%s p=%d stream %d / %d constants  max rel diff %.3e\n", names[op], P, rot_stream_len(P, op), rot2_stream_len(P), worst);
    worst_all = std::max(worst_all, worst);
  }
  return worst_all < 1e-12 ? 0 : 1;
}
