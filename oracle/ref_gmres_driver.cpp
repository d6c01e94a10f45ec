// ref_gmres_driver.cpp -- CPU ORACLE tooling (test infrastructure, NOT product code; build container only).
//
// Runs the REFERENCE's own relaxed GMRES -- /root/reference/examples/BEM/GMRES.hpp with the SolverOptions.hpp,
// Preconditioner.hpp and BLAS.hpp it includes, compiled where they lie, unmodified -- around a Matvec whose execute() is
// the oracle's restatement of the FMM matvec (liboracle.so).  Those four headers need only the standard library; the one
// name they mention without defining, Vec<N,T> (GMRES.hpp:27-34, BLAS.hpp:52,79,139,146: overloads for Vec-valued
// vectors that a scalar solve never instantiates), is DECLARED below and never defined.  What this pins: the caller side
// of the path -- predict_p, the order chosen before every matvec, the Arnoldi/Givens
// arithmetic, restart and stopping rules -- by the reference's code itself instead of digits copied from a log.
// It does not pin the matvec: that stays the oracle.
//
// The run mirrors examples/LaplaceBEM.cpp:163-291 (first-kind problem on UnitSphere(r), identity preconditioner):
//   b = A_{flipped BC} * 1 at p = max_p (:218-232), x0 = 0, GMRES(plan, x, b, solver_options) (:281-285)
// usage: ref_gmres <recursions> <max_p> <tol> <out.json>     (GMRES's own printf lines go to stdout)
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <std::size_t N, typename T> class Vec;     // declared only, see above

#include "GMRES.hpp"                                 // the reference's file (-I /root/reference/examples/BEM)

extern "C" {                                         // oracle/fmm_oracle.h is C-only; the entry points used here
struct orc_ctx;
long orc_unit_sphere(int recursions, double* verts_out);
orc_ctx* orc_create(int n, const double* verts, const uint8_t* bc, int K, double theta, unsigned ncrit);
void orc_destroy(orc_ctx* c);
int orc_matvec(orc_ctx* c, int P, const double* x, double* y, int flags, double stage_s[8]);
}

// The Matvec concept of GMRES.hpp:149-150,169,175,196,201: charge_type / result_type, kernel().set_p(p), execute(x)
struct OracleMatvec {
  typedef double charge_type;
  typedef double result_type;
  struct Kernel {
    int P;
    std::vector<int>* log;
    void set_p(int p) { P = p; log->push_back(p); }
  };
  orc_ctx* ctx;
  Kernel K;
  std::vector<int> p_set, p_used;
  OracleMatvec(orc_ctx* c, int p) : ctx(c), K{p, &p_set} {}
  Kernel& kernel() { return K; }
  std::vector<double> execute(const std::vector<double>& x) {
    std::vector<double> y(x.size());
    double stage[8];
    if (orc_matvec(ctx, K.P, x.data(), y.data(), 0, stage) != 0) { std::fprintf(stderr, "orc_matvec failed\n"); std::exit(3); }
    p_used.push_back(K.P);
    return y;
  }
};

int main(int argc, char** argv) {
  if (argc < 5) { std::fprintf(stderr, "usage: ref_gmres <recursions> <max_p> <tol> <out.json>\n"); return 1; }
  const int r = std::atoi(argv[1]), max_p = std::atoi(argv[2]);
  const double tol = std::atof(argv[3]);
  const long n = orc_unit_sphere(r, nullptr);
  std::vector<double> v(9 * (size_t)n);
  orc_unit_sphere(r, v.data());
  std::vector<uint8_t> bc0(n, 0), bc1(n, 1);
  orc_ctx* A = orc_create((int)n, v.data(), bc0.data(), 3, 0.5, 64);
  orc_ctx* R = orc_create((int)n, v.data(), bc1.data(), 3, 0.5, 64);
  if (!A || !R) return 2;

  SolverOptions solver_options;                      // LaplaceBEM.cpp:82, 127-131, 161-163
  solver_options.residual = tol;
  solver_options.max_p = max_p;
  solver_options.max_iters = 500;
  solver_options.restart = 500;

  std::vector<double> charges(n, 1.), x(n, 0.), b;
  {
    OracleMatvec rhs(R, max_p);
    b = rhs.execute(charges);
  }
  OracleMatvec plan(A, max_p);
  GMRES(plan, x, b, solver_options);

  double e = 0, sum = 0, nrm = 0, bn = 0;
  for (double xi : x) { e += (xi - 1.) * (xi - 1.); sum += xi; nrm += xi * xi; }
  for (double bi : b) bn += bi * bi;
  // true residual of the returned x at the full order
  plan.kernel().P = max_p;
  std::vector<double> ax = plan.execute(x);
  double rr = 0;
  for (long i = 0; i < n; ++i) rr += (ax[i] - b[i]) * (ax[i] - b[i]);
  std::FILE* f = std::fopen(argv[4], "w");
  if (!f) return 4;
  std::fprintf(f, "{\n \"recursions\": %d, \"n\": %ld, \"max_p\": %d, \"tol\": %.3e,\n", r, n, max_p, tol);
  std::fprintf(f, " \"p_set\": [");
  for (size_t i = 0; i < plan.p_set.size(); ++i) std::fprintf(f, "%s%d", i ? ", " : "", plan.p_set[i]);
  std::fprintf(f, "],\n \"matvecs\": %zu,\n", plan.p_used.size() - 1);
  std::fprintf(f, " \"norm_b\": %.17g, \"solution_sum\": %.17g, \"solution_norm\": %.17g,\n", std::sqrt(bn), sum, std::sqrt(nrm));
  std::fprintf(f, " \"relative_error_vs_sigma_1\": %.17g, \"true_residual_at_max_p\": %.17g,\n", std::sqrt(e / (double)n), std::sqrt(rr / bn));
  std::fprintf(f, " \"x_head\": [");
  for (int i = 0; i < 8; ++i) std::fprintf(f, "%s%.17g", i ? ", " : "", x[i]);
  std::fprintf(f, "]\n}\n");
  std::fclose(f);
  orc_destroy(A);
  orc_destroy(R);
  return 0;
}
