// FMM_plan.hpp -- header-only C++ adapter that presents the reference's plan/kernel surface on top of the C ABI of
// include/fmmbem.h, so that the reference's solver and driver code compiles against it unchanged:
//
//     FMM_plan<LaplaceSphericalBEM> plan(K, panels, opts);        // include/FMM_plan.hpp:34-43
//     plan.kernel().set_p(p);                                     // examples/BEM/GMRES.hpp:196
//     std::vector<double> w = plan.execute(z);                    // include/FMM_plan.hpp:75-90
//     plan.options();                                             // include/FMM_plan.hpp:94-96
//     Preconditioners::Diagonal<double> M(K, plan.source_begin(), plan.source_end());   // examples/LaplaceBEM.cpp:241-244
//
// What is mirrored, with the reference's names: Vec / Mat3 (fmmbem/Vec.hpp), FMMOptions, the two BEM kernel classes with
// the KernelSkeleton member typedefs (kernel/KernelSkeleton.hpp:38-60: dimension, point_type, source_type, target_type,
// charge_type, kernel_value_type, result_type, multipole_type, local_type), their Panel (center, normal, vertices,
// quad_points, Area, BC, switch_BC, cast to point_type), operator()(target, source), set_p, and FMM_plan's typedefs,
// kernel(), options(), execute(), source_begin()/source_end() (FMM_plan.hpp:19-28, 66-107).
// Differences, all on the error path: no exit()/printf -- failures throw fmmbem::Error carrying the C status code; copying
// a plan is deleted (the reference's copy is unsafe, FMM_plan.hpp:110).  There is no CPU fallback: every number comes from
// the device, and without a HIP device the constructor throws.
#pragma once
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "../fmmbem.h"
#include "Vec.hpp"

namespace fmmbem {

struct Error : std::runtime_error {
  int status;
  Error(int s, const std::string& what) : std::runtime_error(what), status(s) {}
};
inline void check(int status) {
  if (status != FMMBEM_OK) throw Error(status, std::string(fmmbem_status_string(status)) + ": " + fmmbem_last_error());
}

// The rule the panels of the next kernel are built with: examples/BEM/BEMConfig.hpp:7-75 keeps it in a process-wide
// singleton that every kernel constructor overwrites (:48-53); same behaviour, one inline variable.
inline unsigned& config_K() { static unsigned k = 3; return k; }

// Panel of kernel/LaplaceSphericalBEM.hpp:38-118 and kernel/StokesSphericalBEM.hpp:33-121 (the two are the same struct
// up to the names of the boundary flags).
template <int FLAG0, int FLAG1>
struct PanelT {
  typedef Vec<3, double> point_type;
  typedef enum { BC0 = FLAG0, BC1 = FLAG1 } BoundaryType;
  point_type center, normal;
  std::vector<point_type> vertices, quad_points;
  double Area = 0;
  BoundaryType BC = BC0;
  // K(s, s) of this panel when the plan handed it out through source_begin(): FMM_plan already holds the self
  // interactions in the assembled near matrix, and Preconditioners::Diagonal asks for exactly those
  double self_value = std::numeric_limits<double>::quiet_NaN();

  PanelT() = default;
  PanelT(point_type p0, point_type p1, point_type p2) : vertices{p0, p1, p2} {
    center = (p0 + p1 + p2) / 3;
    const point_type L0 = p2 - p0, L1 = p1 - p0;
    const point_type c(L0[1] * L1[2] - L0[2] * L1[1], -(L0[0] * L1[2] - L0[2] * L1[0]), L0[0] * L1[1] - L0[1] * L1[0]);
    Area = 0.5 * norm(c);
    normal = c / 2 / Area;
    double pts[FMMBEM_MAX_QUAD][3], w[FMMBEM_MAX_QUAD];
    int k = 0;
    check(fmmbem_quadrature((int)config_K(), &pts[0][0], w, &k));
    quad_points.resize(k);
    for (int i = 0; i < k; ++i) quad_points[i] = p0 * pts[i][0] + p1 * pts[i][1] + p2 * pts[i][2];
  }
  operator point_type() const { return center; }
  void switch_BC() { BC = BC == BC0 ? BC1 : BC0; }
};

}  // namespace fmmbem

// include/FMMOptions.hpp:9-106 -- every field, setter and getter of the reference's class, and get_options(argc, argv)
class FMMOptions {
 public:
  bool lazy_evaluation = true, local_evaluation = false, sparse_local = false, block_diagonal = false;
  enum EvalType { FMM, TREECODE };
  EvalType evaluator = FMM;
  // FMMOptions.hpp:19-31: accept a pair of boxes iff |c1 - c2|^2 > ((r1 + r2) / theta)^2, r = half the box side.  The plan
  // applies this rule on the host (csrc/host_plan.cpp, mac_accepts); the functor is here for callers that use it themselves
  struct DefaultMAC {
    double theta_;
    DefaultMAC(double theta) : theta_(theta) {}
    template <typename BOX>
    bool operator()(const BOX& b1, const BOX& b2) const {
      const double r0_normSq = normSq(b1.center() - b2.center());
      const double rhs = (b1.radius() + b2.radius()) / theta_;
      return r0_normSq > rhs * rhs;
    }
  };
  DefaultMAC MAC_ = DefaultMAC(0.5);
  unsigned NCRIT_ = 64;
  bool printTree = false;             // FMMOptions.hpp:38,67-68,100-101: set by -printtree, read by nothing in the reference either
  // Not in the reference.  Its lazy evaluator never queues L2L into a child that was an M2L target earlier in the
  // traversal than its parent (EvalInteractionLazySparse.hpp:199-237); on the meshes its generators produce no such child
  // exists and the two rules are one list, on clustered meshes the reference's list loses far field.
  // false (default): every child of a box that holds L;  true: exactly the reference's list.
  bool reference_l2l = false;
  void set_mac_theta(double t) { MAC_ = DefaultMAC(t); }
  DefaultMAC MAC() { return MAC_; }
  void set_max_per_box(unsigned n) { NCRIT_ = n; }
  unsigned max_per_box() const { return NCRIT_; }
  void print_tree(bool v) { printTree = v; }
  bool print_tree() const { return printTree; }
  // executor/make_executor.hpp:24-60: lazy_evaluation wins, then local_evaluation, then block_diagonal; the
  // non-lazy upward/interact/downward evaluators compute the same operator as the lazy ones
  int c_evaluator() const {
    if (evaluator != FMM) throw fmmbem::Error(FMMBEM_ERR_UNSUPPORTED, "the treecode evaluator is not built");
    if (lazy_evaluation) return FMMBEM_EVAL_FMM;
    return local_evaluation ? FMMBEM_EVAL_LOCAL : block_diagonal ? FMMBEM_EVAL_BLOCK_DIAGONAL : FMMBEM_EVAL_FMM;
  }
};

// FMMOptions.hpp:74-106: -theta <t>, -eval FMM|TREE, -lazy_eval, -ncrit <n>, -printtree; everything else is left to the
// caller's own parser (the drivers scan argv again for their flags).  `inline`: the reference defines it in the header
// without it, which is why its headers admit one translation unit only.
inline FMMOptions get_options(int argc, char** argv) {
  FMMOptions opts = FMMOptions();
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "-theta" && i + 1 < argc) opts.set_mac_theta(std::atof(argv[++i]));
    else if (a == "-eval" && i + 1 < argc) {
      const std::string e = argv[++i];
      if (e == "FMM") opts.evaluator = FMMOptions::FMM;
      else if (e == "TREE") opts.evaluator = FMMOptions::TREECODE;
      else std::printf("[W]: Unknown evaluator type: \"%s\"\n", e.c_str());
    } else if (a == "-lazy_eval") opts.lazy_evaluation = true;
    else if (a == "-ncrit" && i + 1 < argc) opts.set_max_per_box((unsigned)std::atoi(argv[++i]));
    else if (a == "-printtree") opts.print_tree(true);
  }
  return opts;
}

// kernel/LaplaceSphericalBEM.hpp:14-140 (+ the typedefs it inherits from kernel/LaplaceSpherical.hpp:33-50)
class LaplaceSphericalBEM {
 public:
  typedef double real;
  typedef std::complex<real> complex;
  static constexpr unsigned dimension = 3;
  typedef Vec<dimension, real> point_type;
  struct Panel : fmmbem::PanelT<0, 1> {
    static constexpr BoundaryType POTENTIAL = BC0, NORMAL_DERIV = BC1;
    using fmmbem::PanelT<0, 1>::PanelT;
  };
  typedef Panel source_type;
  typedef Panel target_type;
  typedef Panel panel_type;
  typedef real charge_type;
  typedef double kernel_value_type;
  typedef double result_type;
  typedef std::vector<std::vector<complex>> multipole_type;      // two expansions per box (G, dG/dn), :123-126
  typedef std::vector<std::vector<complex>> local_type;
  unsigned K;
  LaplaceSphericalBEM() : LaplaceSphericalBEM(5, 3) {}
  explicit LaplaceSphericalBEM(int p, unsigned k = 3) : K(k), P(p) { fmmbem::config_K() = k; }
  void set_p(int p) { P = p; }                      // LaplaceSphericalBEM.hpp:137-140
  int p() const { return P; }
  int device = 0;                                   // where operator() evaluates

  // one near-matrix entry, int_source G or dG/dn at the target centroid (:273-297); the target's BC picks the integrand
  kernel_value_type operator()(const target_type& t, const source_type& s) const {
    if (&t == &s && t.self_value == t.self_value) return t.self_value;
    fmmbem_options o;
    fmmbem_options_default(&o);
    o.quad_k = (int)K;
    o.device = device;
    double tv[9], sv[9], out = 0;
    for (int a = 0; a < 3; ++a)
      for (int c = 0; c < 3; ++c) { tv[3 * a + c] = t.vertices[a][c]; sv[3 * a + c] = s.vertices[a][c]; }
    const uint8_t bc = t.BC == Panel::NORMAL_DERIV ? FMMBEM_BC_NORMAL_DERIV : FMMBEM_BC_POTENTIAL;
    fmmbem::check(fmmbem_kernel_entries(&o, 1, tv, &bc, sv, &out));
    return out;
  }

 protected:
  int P;
};

// kernel/StokesSphericalBEM.hpp:9-141 -- Vec<3,double> charges/results, Mat3 kernel values, p, K, K_fine, mu.
class StokesSphericalBEM {
 public:
  typedef double real;
  typedef std::complex<real> complex;
  static constexpr unsigned dimension = 3;
  typedef Vec<dimension, real> point_type;
  struct Panel : fmmbem::PanelT<0, 1> {
    static constexpr BoundaryType VELOCITY = BC0, TRACTION = BC1;
    using fmmbem::PanelT<0, 1>::PanelT;
  };
  typedef Panel source_type;
  typedef Panel target_type;
  typedef Panel panel_type;
  typedef Vec<dimension, real> charge_type;
  typedef Mat3<real> kernel_value_type;
  typedef Vec<dimension, real> result_type;
  typedef std::vector<std::vector<std::vector<complex>>> multipole_type;   // M[2][4] per box, :143-153
  typedef std::vector<std::vector<std::vector<complex>>> local_type;
  unsigned K, K_fine = 25;
  double Mu;
  StokesSphericalBEM() : StokesSphericalBEM(5, 3, 1e-3) {}
  StokesSphericalBEM(int p, unsigned k) : StokesSphericalBEM(p, k, 1e-3) {}
  StokesSphericalBEM(int p, unsigned k, double mu) : K(k), Mu(mu), P(p) { fmmbem::config_K() = k; }
  void set_p(int p) { P = p; }
  void set_Kfine(unsigned k) { K_fine = k; }
  int p() const { return P; }
  int device = 0;

  // the 3x3 block at the target centroid (:377-389): VELOCITY target (1/2mu) int (I/r + d d^T/r^3), TRACTION target
  // -3 int (d.n) d d^T / r^5 (self: 2 pi I)
  kernel_value_type operator()(const target_type& t, const source_type& s) const {
    fmmbem_options o;
    fmmbem_options_default(&o);
    o.kernel = FMMBEM_KERNEL_STOKES_BEM;
    o.quad_k = (int)K;
    o.quad_k_fine = (int)K_fine;
    o.mu = Mu;
    o.device = device;
    double tv[9], sv[9];
    for (int a = 0; a < 3; ++a)
      for (int c = 0; c < 3; ++c) { tv[3 * a + c] = t.vertices[a][c]; sv[3 * a + c] = s.vertices[a][c]; }
    const uint8_t bc = t.BC == Panel::TRACTION;
    kernel_value_type m;
    fmmbem::check(fmmbem_kernel_entries(&o, 1, tv, &bc, sv, m.vals_));
    return m;
  }

 protected:
  int P;
};

namespace fmmbem {

template <class Kernel> struct KernelBinding;
template <> struct KernelBinding<LaplaceSphericalBEM> {
  static void fill(const LaplaceSphericalBEM& K, const FMMOptions& opts, fmmbem_options& o) {
    o.kernel = FMMBEM_KERNEL_LAPLACE_BEM;
    o.quad_k = (int)K.K;
    o.sparse_local = opts.sparse_local ? 1 : 0;     // examples/LaplaceBEM.cpp:81 sets it; FMMOptions defaults to false
  }
  static const double* in(const std::vector<double>& v) { return v.data(); }
  static double* out(std::vector<double>& v) { return v.data(); }
};
template <> struct KernelBinding<StokesSphericalBEM> {
  static void fill(const StokesSphericalBEM& K, const FMMOptions& opts, fmmbem_options& o) {
    o.kernel = FMMBEM_KERNEL_STOKES_BEM;
    o.quad_k = (int)K.K;
    o.quad_k_fine = (int)K.K_fine;
    o.mu = K.Mu;
    o.sparse_local = opts.sparse_local ? 1 : 0;     // examples/StokesBEM.cpp:147 sets it, -disable_sparse (:197) clears it
  }
  // Vec<3,double> is three contiguous doubles: the vectors are the N x 3 arrays the C ABI expects
  static const double* in(const std::vector<Vec<3, double>>& v) { return v.data()->data(); }
  static double* out(std::vector<Vec<3, double>>& v) { return v.data()->data(); }
};

// FMM_plan<Kernel> of include/FMM_plan.hpp:16-128 for the two kernels the library stands in for
template <class Kernel>
class PlanAdapter {
 public:
  typedef Kernel kernel_type;
  typedef typename kernel_type::point_type point_type;
  typedef typename kernel_type::source_type source_type;
  typedef typename kernel_type::target_type target_type;
  typedef typename kernel_type::charge_type charge_type;
  typedef typename kernel_type::result_type result_type;
  typedef typename std::vector<source_type>::const_iterator body_source_iterator;

  // p_max: the order the plan's expansions, tables and stored P2M moments are sized for.  The reference accepts any p at any
  // time (LaplaceSpherical::set_p resizes its tables, kernel/LaplaceSpherical.hpp:119-128); so does this class: the default
  // is the kernel's order at construction -- what the drivers pass as SolverOptions::max_p too (LaplaceBEM.cpp:160-168) --
  // and an execute() at a higher order re-creates the plan for it, once (sizing every plan for the largest pre-compiled
  // order instead would cost 2.5x the memory at p = 10 and refuse Stokes TRACTION plans outright).
  PlanAdapter(const kernel_type& k, const std::vector<source_type>& source, FMMOptions& opts, int p_max = 0, int device = 0)
      : K(k), opts_(opts), sources_(source), n_(source.size()), device_(device) {
    K.device = device;
    create(p_max > 0 ? p_max : K.p());
  }
  ~PlanAdapter() { fmmbem_plan_destroy(plan_); }
  PlanAdapter(const PlanAdapter&) = delete;
  PlanAdapter& operator=(const PlanAdapter&) = delete;

  kernel_type& kernel() { return K; }
  const kernel_type& kernel() const { return K; }
  FMMOptions& options() { return opts_; }

  std::vector<result_type> execute(const std::vector<charge_type>& charges) {
    if (charges.size() != n_) throw Error(FMMBEM_ERR_INVALID, "charges.size() != number of panels");
    std::vector<result_type> results(charges.size());
    if (K.p() > p_max_) create(K.p());                  // set_p above what the plan was sized for: grow, as set_p does
    check(fmmbem_plan_execute(plan_, K.p(), KernelBinding<Kernel>::in(charges), KernelBinding<Kernel>::out(results)));
    return results;
  }
  // preconditioner-style operator()(x, y) (examples/BEM/Preconditioner.hpp:11-15)
  void operator()(const std::vector<charge_type>& x, std::vector<result_type>& y) { y = execute(x); }

  // The plan's copy of the sources in TREE order (ExecutorSingleTree.hpp:145, 196-225; FMM_plan.hpp:100-107).  The
  // panels carry their self interaction (fmmbem_plan_get_diagonal) so that K(*it, *it) costs nothing.
  body_source_iterator source_begin() { tree_sources(); return tree_.begin(); }
  body_source_iterator source_end() { tree_sources(); return tree_.end(); }

  fmmbem_plan* handle() { return plan_; }
  int p_max() const { return p_max_; }

 private:
  void create(int p_max) {
    std::vector<double> v(9 * n_);
    std::vector<uint8_t> bc(n_);
    for (size_t i = 0; i < n_; ++i) {
      for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) v[9 * i + 3 * a + c] = sources_[i].vertices[a][c];
      bc[i] = sources_[i].BC == source_type::BC1;
    }
    fmmbem_options o;
    fmmbem_options_default(&o);
    KernelBinding<Kernel>::fill(K, opts_, o);
    o.p_max = p_max;
    o.theta = opts_.MAC().theta_;
    o.ncrit = opts_.NCRIT_;
    o.device = device_;
    o.evaluator = opts_.c_evaluator();
    o.l2l_rule = opts_.reference_l2l ? FMMBEM_L2L_REFERENCE : FMMBEM_L2L_COMPLETE;
    sparse_ = o.sparse_local != 0;
    fmmbem_plan* fresh = nullptr;
    check(fmmbem_plan_create(&o, n_, v.data(), bc.data(), &fresh));
    if (plan_) fmmbem_plan_destroy(plan_);
    plan_ = fresh;
    p_max_ = p_max;
  }

  void tree_sources() {
    if (!tree_.empty() || n_ == 0) return;
    std::vector<uint32_t> perm(n_);
    check(fmmbem_plan_get_perm(plan_, perm.data()));
    std::vector<double> diag;
    if (sparse_ && std::is_same<charge_type, double>::value) {
      diag.resize(n_);
      check(fmmbem_plan_get_diagonal(plan_, diag.data()));
    }
    tree_.reserve(n_);
    for (size_t i = 0; i < n_; ++i) {
      tree_.push_back(sources_[perm[i]]);
      if (!diag.empty()) tree_.back().self_value = diag[perm[i]];
    }
  }

  kernel_type K;
  FMMOptions opts_;
  std::vector<source_type> sources_, tree_;
  size_t n_;
  int device_ = 0, p_max_ = 0;
  bool sparse_ = false;
  fmmbem_plan* plan_ = nullptr;
};

}  // namespace fmmbem

template <class Kernel>
class FMM_plan;

template <>
class FMM_plan<LaplaceSphericalBEM> : public fmmbem::PlanAdapter<LaplaceSphericalBEM> {
 public:
  using fmmbem::PlanAdapter<LaplaceSphericalBEM>::PlanAdapter;
};

template <>
class FMM_plan<StokesSphericalBEM> : public fmmbem::PlanAdapter<StokesSphericalBEM> {
 public:
  using fmmbem::PlanAdapter<StokesSphericalBEM>::PlanAdapter;
};
