// relaxed_gmres.hpp -- test-side solver for tests/cpp/laplace_bem_sequence.cpp when the reference's own
// examples/BEM/{GMRES,SolverOptions,Preconditioner}.hpp are not on the include path (the GPU box has no reference tree).
// Same call shapes as the reference so that the driver source is identical in both builds:
//   SolverOptions{residual, max_iters, restart, max_p, variable_p}.predict_p(eps)      SolverOptions.hpp:11-39
//   Preconditioners::Identity / Diagonal<T>(K, first, last)                            Preconditioner.hpp:8-42
//   GMRES(MV, x, b, opts[, M])   restarted GMRES, modified Gram-Schmidt + Givens,      GMRES.hpp:119-252
//                                 p = max(1, predict_p(|resid|)) set before every matvec
// Written from the algorithm's description (Saad & Schultz), scalar unknowns only.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

struct SolverOptions {
  double residual = 1e-5;
  int max_iters = 500, restart = 500;
  unsigned max_p = 16, p_min = 5;
  bool variable_p = true;
  SolverOptions() = default;
  SolverOptions(double r, int iters, unsigned p) : residual(r), max_iters(iters), restart(50), max_p(p), variable_p(false) {}
  unsigned predict_p(double eps) const {
    if (!variable_p) return max_p;
    const double nu = std::min(residual / std::min(eps, 1.), 1.);
    return std::min((unsigned)std::ceil(-std::log2(nu)), max_p);
  }
};

namespace Preconditioners {
struct Identity {
  template <class V> void operator()(const V& x, V& y) const { y = x; }
};
template <class T>
class Diagonal {
  std::vector<T> recip_;
 public:
  template <class Kernel, class It>
  Diagonal(Kernel& K, It first, It last) {
    for (; first != last; ++first) recip_.push_back(1. / K(*first, *first));
  }
  template <class V> void operator()(const V& x, V& y) const {
    for (size_t i = 0; i < x.size(); ++i) y[i] = recip_[i] * x[i];
  }
};
}  // namespace Preconditioners

template <class Matvec, class PC>
void GMRES(Matvec& MV, std::vector<double>& x, std::vector<double>& b, const SolverOptions& opts, const PC& M) {
  const int R = opts.restart, n = (int)x.size();
  std::vector<std::vector<double>> V(R + 1, std::vector<double>(n)), H(R + 1, std::vector<double>(R, 0.));
  std::vector<double> s(R + 1), cs(R), sn(R), w, z(n);
  auto nrm = [](const std::vector<double>& v) { double t = 0; for (double a : v) t += a * a; return std::sqrt(t); };
  const double normb = nrm(b);
  int it = 0, i = 0;
  double resid = 0;
  do {
    w = MV.execute(x);
    for (int k = 0; k < n; ++k) w[k] -= b[k];
    const double beta = nrm(w);
    for (int k = 0; k < n; ++k) V[0][k] = -w[k] / beta;
    s[0] = beta;
    i = -1;
    resid = s[0] / normb;
    do {
      ++i; ++it;
      const int p = (int)std::max(1u, opts.predict_p(std::fabs(resid)));
      MV.kernel().set_p(p);
      M(V[i], z);
      w = MV.execute(z);
      for (int k = 0; k <= i; ++k) {
        double h = 0;
        for (int j = 0; j < n; ++j) h += w[j] * V[k][j];
        H[k][i] = h;
        for (int j = 0; j < n; ++j) w[j] -= h * V[k][j];
      }
      H[i + 1][i] = nrm(w);
      for (int j = 0; j < n; ++j) V[i + 1][j] = w[j] / H[i + 1][i];
      auto rot = [](double& dx, double& dy, double c, double sN) { const double t = c * dx + sN * dy; dy = -sN * dx + c * dy; dx = t; };
      for (int k = 0; k < i; ++k) rot(H[k][i], H[k + 1][i], cs[k], sn[k]);
      const double dx = H[i][i], dy = H[i + 1][i];
      if (dy == 0.) { cs[i] = 1; sn[i] = 0; }
      else if (std::fabs(dy) > std::fabs(dx)) { const double t = dx / dy; sn[i] = 1 / std::sqrt(1 + t * t); cs[i] = t * sn[i]; }
      else { const double t = dy / dx; cs[i] = 1 / std::sqrt(1 + t * t); sn[i] = t * cs[i]; }
      rot(H[i][i], H[i + 1][i], cs[i], sn[i]);
      rot(s[i], s[i + 1], cs[i], sn[i]);
      resid = s[i + 1] / normb;
      if (std::fabs(resid) < opts.residual) break;
      std::printf("it: %03d, res: %.3e, fmm_req_p: %01d\n", it, std::fabs(resid), p);
    } while (i + 1 < R && i + 1 <= opts.max_iters && std::fabs(resid) > opts.residual);
    for (int j = i; j >= 0; --j) {
      s[j] /= H[j][j];
      for (int k = j - 1; k >= 0; --k) s[k] -= H[k][j] * s[j];
    }
    for (int j = 0; j <= i; ++j) {
      M(V[j], z);
      for (int k = 0; k < n; ++k) x[k] += s[j] * z[k];
    }
  } while (std::fabs(resid) > opts.residual && it < opts.max_iters);
  std::printf("Final residual: %.4e, after %d iterations\n", std::fabs(resid), it);
}
template <class Matvec>
void GMRES(Matvec& MV, std::vector<double>& x, std::vector<double>& b, const SolverOptions& opts) {
  GMRES(MV, x, b, opts, Preconditioners::Identity());
}
