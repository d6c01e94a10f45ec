// FMM_plan.hpp -- header-only C++ adapter that presents the reference's plan/operator surface on top of
// the C ABI of include/fmmbem.h, so that the reference's solver code compiles against it unchanged:
//
//     FMM_plan<LaplaceSphericalBEM> plan(K, panels, opts);        // include/FMM_plan.hpp:34-43
//     plan.kernel().set_p(p);                                     // examples/BEM/GMRES.hpp:196
//     std::vector<double> w = plan.execute(z);                    // include/FMM_plan.hpp:75-90
//     plan.options();                                             // include/FMM_plan.hpp:94-96
//
// Same names, same argument meaning.  Differences, all on the error path: no exit()/printf -- failures
// throw fmmbem::Error carrying the C status code; copying a plan is deleted (the reference's copy is
// unsafe, FMM_plan.hpp:110).  There is no CPU fallback: without a HIP device the constructor throws.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../fmmbem.h"

namespace fmmbem {

struct Error : std::runtime_error {
  int status;
  Error(int s, const std::string& what) : std::runtime_error(what), status(s) {}
};
inline void check(int status) {
  if (status != FMMBEM_OK) throw Error(status, std::string(fmmbem_status_string(status)) + ": " + fmmbem_last_error());
}

}  // namespace fmmbem

// include/FMMOptions.hpp:9-60 -- the fields and setters the hot path reads
class FMMOptions {
 public:
  bool lazy_evaluation = true, local_evaluation = false, sparse_local = false, block_diagonal = false;
  enum EvalType { FMM, TREECODE };
  EvalType evaluator = FMM;
  double theta = 0.5;
  unsigned NCRIT_ = 64;
  bool reference_l2l = false;   // not in the reference: apply only the L2L edges its lazy evaluator queues (fmmbem_l2l_rule)
  void set_mac_theta(double t) { theta = t; }
  void set_max_per_box(unsigned n) { NCRIT_ = n; }
  unsigned max_per_box() const { return NCRIT_; }
  // executor/make_executor.hpp:24-60: lazy_evaluation wins, then local_evaluation, then block_diagonal; the
  // non-lazy upward/interact/downward evaluators compute the same operator as the lazy ones
  int c_evaluator() const {
    if (evaluator != FMM) throw fmmbem::Error(FMMBEM_ERR_UNSUPPORTED, "the treecode evaluator is not built");
    if (lazy_evaluation) return FMMBEM_EVAL_FMM;
    return local_evaluation ? FMMBEM_EVAL_LOCAL : block_diagonal ? FMMBEM_EVAL_BLOCK_DIAGONAL : FMMBEM_EVAL_FMM;
  }
};

// kernel/LaplaceSphericalBEM.hpp:14-140 -- the part of the kernel object the plan boundary uses:
// panel type, charge/result types, p and K.
class LaplaceSphericalBEM {
 public:
  typedef std::array<double, 3> point_type;
  typedef double charge_type;
  typedef double result_type;
  struct Panel {                                    // LaplaceSphericalBEM.hpp:38-118
    typedef enum { POTENTIAL, NORMAL_DERIV } BoundaryType;
    std::array<point_type, 3> vertices;
    BoundaryType BC = POTENTIAL;
    Panel() = default;
    Panel(point_type p0, point_type p1, point_type p2) : vertices{{p0, p1, p2}} {}
    void switch_BC() { BC = BC == POTENTIAL ? NORMAL_DERIV : POTENTIAL; }
  };
  typedef Panel source_type;
  typedef Panel target_type;
  unsigned K;
  explicit LaplaceSphericalBEM(int p = 5, unsigned k = 3) : K(k), P(p) {}
  void set_p(int p) { P = p; }                      // LaplaceSphericalBEM.hpp:137-140
  int p() const { return P; }

 private:
  int P;
};

// kernel/StokesSphericalBEM.hpp:9-141 -- panel type, Vec<3,double> charges/results, p, K, K_fine, mu.
// Only VELOCITY panels are accepted by the library (the reference's traction far field is not reproducible).
class StokesSphericalBEM {
 public:
  typedef std::array<double, 3> point_type;
  typedef std::array<double, 3> charge_type;
  typedef std::array<double, 3> result_type;
  struct Panel {
    typedef enum { VELOCITY, TRACTION } BoundaryType;
    std::array<point_type, 3> vertices;
    BoundaryType BC = VELOCITY;
    Panel() = default;
    Panel(point_type p0, point_type p1, point_type p2) : vertices{{p0, p1, p2}} {}
    void switch_BC() { BC = BC == VELOCITY ? TRACTION : VELOCITY; }
  };
  typedef Panel source_type;
  typedef Panel target_type;
  unsigned K, K_fine = 25;
  double Mu;
  explicit StokesSphericalBEM(int p = 5, unsigned k = 3, double mu = 1e-3) : K(k), Mu(mu), P(p) {}
  void set_p(int p) { P = p; }
  void set_Kfine(unsigned k) { K_fine = k; }
  int p() const { return P; }

 private:
  int P;
};

template <class Kernel>
class FMM_plan;

template <>
class FMM_plan<LaplaceSphericalBEM> {
 public:
  typedef LaplaceSphericalBEM kernel_type;
  typedef kernel_type::point_type point_type;
  typedef kernel_type::source_type source_type;
  typedef kernel_type::target_type target_type;
  typedef kernel_type::charge_type charge_type;
  typedef kernel_type::result_type result_type;

  // p_max: largest order later set through kernel().set_p(); defaults to the kernel's current p
  FMM_plan(const kernel_type& k, const std::vector<source_type>& source, FMMOptions& opts, int p_max = 0,
           int device = 0)
      : K(k), opts_(opts), n_(source.size()) {
    std::vector<double> v(9 * n_);
    std::vector<uint8_t> bc(n_);
    for (size_t i = 0; i < n_; ++i) {
      for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) v[9 * i + 3 * a + c] = source[i].vertices[a][c];
      bc[i] = source[i].BC == source_type::NORMAL_DERIV ? FMMBEM_BC_NORMAL_DERIV : FMMBEM_BC_POTENTIAL;
    }
    fmmbem_options o;
    fmmbem_options_default(&o);
    o.p_max = p_max > 0 ? p_max : K.p();
    o.quad_k = (int)K.K;
    o.theta = opts.theta;
    o.ncrit = opts.NCRIT_;
    o.sparse_local = opts.sparse_local ? 1 : 0;   // examples/LaplaceBEM.cpp:81 sets it; FMMOptions defaults to false
    o.device = device;
    o.evaluator = opts.c_evaluator();
    o.l2l_rule = opts.reference_l2l ? FMMBEM_L2L_REFERENCE : FMMBEM_L2L_COMPLETE;
    fmmbem::check(fmmbem_plan_create(&o, n_, v.data(), bc.data(), &plan_));
  }
  ~FMM_plan() { fmmbem_plan_destroy(plan_); }
  FMM_plan(const FMM_plan&) = delete;
  FMM_plan& operator=(const FMM_plan&) = delete;

  kernel_type& kernel() { return K; }
  const kernel_type& kernel() const { return K; }
  FMMOptions& options() { return opts_; }

  std::vector<result_type> execute(const std::vector<charge_type>& charges) {
    if (charges.size() != n_) throw fmmbem::Error(FMMBEM_ERR_INVALID, "charges.size() != number of panels");
    std::vector<result_type> results(charges.size());
    fmmbem::check(fmmbem_plan_execute(plan_, K.p(), charges.data(), results.data()));
    return results;
  }
  // preconditioner-style operator()(x, y) (examples/BEM/Preconditioner.hpp:11-15)
  void operator()(const std::vector<charge_type>& x, std::vector<result_type>& y) { y = execute(x); }

  fmmbem_plan* handle() { return plan_; }

 private:
  kernel_type K;
  FMMOptions opts_;
  size_t n_;
  fmmbem_plan* plan_ = nullptr;
};

template <>
class FMM_plan<StokesSphericalBEM> {
 public:
  typedef StokesSphericalBEM kernel_type;
  typedef kernel_type::source_type source_type;
  typedef kernel_type::charge_type charge_type;
  typedef kernel_type::result_type result_type;

  FMM_plan(const kernel_type& k, const std::vector<source_type>& source, FMMOptions& opts, int p_max = 0, int device = 0)
      : K(k), opts_(opts), n_(source.size()) {
    std::vector<double> v(9 * n_);
    std::vector<uint8_t> bc(n_);
    for (size_t i = 0; i < n_; ++i) {
      for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) v[9 * i + 3 * a + c] = source[i].vertices[a][c];
      bc[i] = source[i].BC == source_type::TRACTION;
    }
    fmmbem_options o;
    fmmbem_options_default(&o);
    o.kernel = FMMBEM_KERNEL_STOKES_BEM;
    o.p_max = p_max > 0 ? p_max : K.p();
    o.quad_k = (int)K.K;
    o.quad_k_fine = (int)K.K_fine;
    o.mu = K.Mu;
    o.theta = opts.theta;
    o.ncrit = opts.NCRIT_;
    o.sparse_local = 1;
    o.evaluator = opts.c_evaluator();
    o.l2l_rule = opts.reference_l2l ? FMMBEM_L2L_REFERENCE : FMMBEM_L2L_COMPLETE;
    o.device = device;
    fmmbem::check(fmmbem_plan_create(&o, n_, v.data(), bc.data(), &plan_));
  }
  ~FMM_plan() { fmmbem_plan_destroy(plan_); }
  FMM_plan(const FMM_plan&) = delete;
  FMM_plan& operator=(const FMM_plan&) = delete;

  kernel_type& kernel() { return K; }
  FMMOptions& options() { return opts_; }

  std::vector<result_type> execute(const std::vector<charge_type>& charges) {
    if (charges.size() != n_) throw fmmbem::Error(FMMBEM_ERR_INVALID, "charges.size() != number of panels");
    std::vector<result_type> results(charges.size());
    // std::array<double,3> is three contiguous doubles: the vectors are the N x 3 arrays the C ABI expects
    fmmbem::check(fmmbem_plan_execute(plan_, K.p(), charges.data()->data(), results.data()->data()));
    return results;
  }

 private:
  kernel_type K;
  FMMOptions opts_;
  size_t n_;
  fmmbem_plan* plan_ = nullptr;
};
