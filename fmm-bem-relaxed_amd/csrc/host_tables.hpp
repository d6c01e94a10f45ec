// host_tables.hpp -- host-side tabulation shared by plan.hip and ops.hip: the EPS-scaled normalisation tables of the reference's
// spherical harmonics, cart2sph, the harmonics themselves, and the class record of the rotation kernels.
#pragma once
#include <cmath>
#include <complex>
#include <vector>

#include "host_plan.hpp"

namespace fmmbem {
namespace tables {

constexpr int kTabN = 2 * kPmax;        // harmonic degrees tabulated: n < 32
constexpr double kEps = 1e-12;          // kernel/LaplaceSpherical.hpp:30

// Anm / prefactor of LaplaceSpherical::precompute (kernel/LaplaceSpherical.hpp:87-104), for all degrees
// any p <= 16 needs.  The index n^2+n+m does not depend on P, so one table serves every order.
struct HarmonicTables {
  std::vector<double> A, invA, pref;
  HarmonicTables() : A(kTabN * kTabN), invA(kTabN * kTabN), pref(kTabN * kTabN) {
    for (int n = 0; n < kTabN; ++n)
      for (int m = -n; m <= n; ++m) {
        const int nm = n * n + n + m, am = std::abs(m);
        double fnmm = kEps, fnpm = kEps, fnma = 1.0, fnpa = 1.0;
        for (int i = 1; i <= n - m; ++i) fnmm *= i;
        for (int i = 1; i <= n + m; ++i) fnpm *= i;
        for (int i = 1; i <= n - am; ++i) fnma *= i;
        for (int i = 1; i <= n + am; ++i) fnpa *= i;
        pref[nm] = std::sqrt(fnma / fnpa);
        A[nm] = ((n & 1) ? -1.0 : 1.0) / std::sqrt(fnmm * fnpm);
        invA[nm] = 1.0 / A[nm];
      }
  }
};

struct SphHost { double rho, alpha, beta; };
// kernel/LaplaceSpherical.hpp:528-541
inline SphHost cart2sph_host(const double d[3]) {
  SphHost s;
  s.rho = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) + kEps;
  s.alpha = std::acos(d[2] / s.rho);
  if (std::fabs(d[0]) + std::fabs(d[1]) < kEps) s.beta = 0;
  else if (std::fabs(d[0]) < kEps) s.beta = d[1] / std::fabs(d[1]) * M_PI * 0.5;
  else if (d[0] > 0) s.beta = std::atan(d[1] / d[0]);
  else s.beta = std::atan(d[1] / d[0]) + M_PI;
  return s;
}

using cplx = std::complex<double>;

// Harmonics for orders m >= 0, degrees n < N.  regular: rho^n P_n^m pref e^{i m beta}
// (evalMultipole, :455-488); otherwise rho^{-n-1} ... (evalLocal, :491-524).  out[n(n+1)/2+m].
inline void harmonics(const HarmonicTables& T, bool regular, double rho, double alpha, double beta, int N,
               std::vector<cplx>& out) {
  out.assign((size_t)N * (N + 1) / 2, cplx(0, 0));
  const double x = std::cos(alpha), y = std::sin(alpha);
  double fact = 1, pn = 1, rhom = regular ? 1.0 : 1.0 / rho;
  for (int m = 0; m < N; ++m) {
    const cplx eim = std::exp(cplx(0, 1) * double(m * beta));
    double p = pn;
    out[(size_t)m * (m + 1) / 2 + m] = rhom * p * T.pref[m * m + 2 * m] * eim;
    double p1 = p;
    p = x * (2 * m + 1) * p1;
    if (regular) rhom *= rho; else rhom /= rho;
    double rhon = rhom;
    for (int n = m + 1; n < N; ++n) {
      out[(size_t)n * (n + 1) / 2 + m] = rhon * p * T.pref[n * n + n + m] * eim;
      const double p2 = p1;
      p1 = p;
      p = (x * (2 * n + 1) * p1 - (n + m) * p2) / (n - m + 1);
      if (regular) rhon *= rho; else rhon /= rho;
    }
    pn = -pn * fact * y;
    fact += 2;
  }
}

inline cplx i_pow(int q) {
  switch (q & 3) { case 0: return {1, 0}; case 1: return {0, 1}; case 2: return {-1, 0}; default: return {0, -1}; }
}


// class record of the rotation kernels: 1/rho, cos alpha, sin alpha, cos beta, sin beta, rho of a translation vector --
// cart2sph of the reference (kernel/LaplaceSpherical.hpp:528-541): rho = |d| + EPS, alpha = acos(z / rho), and the azimuth
// branches; kept as cosines and sines
inline void rot_record(const double tr[3], double* o) {
  const double rho = std::sqrt(tr[0] * tr[0] + tr[1] * tr[1] + tr[2] * tr[2]) + kEps;
  const double ca = tr[2] / rho;
  o[0] = 1.0 / rho; o[1] = ca; o[2] = std::sqrt((1.0 - ca) * (1.0 + ca)); o[5] = rho; o[6] = o[7] = 0.0;
  if (std::fabs(tr[0]) + std::fabs(tr[1]) < kEps) { o[3] = 1; o[4] = 0; }
  else if (std::fabs(tr[0]) < kEps) { o[3] = 0; o[4] = tr[1] > 0 ? 1.0 : -1.0; }
  else { const double h = 1.0 / std::sqrt(tr[0] * tr[0] + tr[1] * tr[1]); o[3] = tr[0] * h; o[4] = tr[1] * h; }
}

}  // namespace tables
}  // namespace fmmbem
