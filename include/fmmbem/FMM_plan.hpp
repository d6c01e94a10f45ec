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
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <memory>
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

// The translation operators one at a time (kernel/KernelSkeleton.hpp:62-212): the kernel classes below forward P2M, M2M, M2L,
// L2L, L2P to fmmbem_ops_* (the matvec's own device kernels on a two-box plan, csrc/ops.hip) with the reference's argument
// lists, so that ExpansionTraits<Kernel>::is_valid_fmm (include/KernelTraits.hpp:188-194) holds for them and code that drives
// single operators (tests/single_level.cpp) compiles.  The handle is made at the first call and shared by copies of the kernel.
class SingleOperators {
 public:
  typedef std::complex<double> complex;
  typedef std::vector<complex> expansion;
  fmmbem_ops* handle(int kernel, unsigned K, double mu, int device) const {
    if (!h_ || h_->kernel != kernel || h_->K != K || h_->mu != mu || h_->device != device) {
      fmmbem_options o;
      fmmbem_options_default(&o);
      o.kernel = kernel; o.p_max = 16; o.quad_k = (int)K; o.mu = mu; o.device = device;
      fmmbem_ops* raw = nullptr;
      check(fmmbem_ops_create(&o, &raw));
      h_ = std::make_shared<Handle>(raw, kernel, K, mu, device);
    }
    return h_->ops;
  }
  // [slot][S] complex <-> the C ABI's (re, im) pairs
  static std::vector<double> pack(const std::vector<const expansion*>& e, int P) {
    const size_t S = (size_t)P * (P + 1) / 2;
    std::vector<double> out(2 * S * e.size());
    for (size_t s = 0; s < e.size(); ++s) {
      if (e[s]->size() != S) throw Error(FMMBEM_ERR_INVALID, "an expansion of another order than the kernel's p (init_multipole / init_local size it)");
      for (size_t i = 0; i < S; ++i) { out[2 * (s * S + i)] = (*e[s])[i].real(); out[2 * (s * S + i) + 1] = (*e[s])[i].imag(); }
    }
    return out;
  }
  static void unpack(const std::vector<double>& in, const std::vector<expansion*>& e, int P) {
    const size_t S = (size_t)P * (P + 1) / 2;
    for (size_t s = 0; s < e.size(); ++s)
      for (size_t i = 0; i < S; ++i) (*e[s])[i] = complex(in[2 * (s * S + i)], in[2 * (s * S + i) + 1]);
  }
  template <class Fn>
  void shift(Fn fn, fmmbem_ops* ops, int P, const std::vector<const expansion*>& src, const std::vector<expansion*>& tgt,
             const Vec<3, double>& translation) const {
    if (src.size() != tgt.size()) throw Error(FMMBEM_ERR_INVALID, "source and target hold different numbers of expansions");
    const std::vector<double> a = pack(src, P);
    std::vector<double> b = pack(std::vector<const expansion*>(tgt.begin(), tgt.end()), P);
    const double t[3] = {translation[0], translation[1], translation[2]};
    check(fn(ops, P, (int)src.size(), a.data(), b.data(), t));
    unpack(b, tgt, P);
  }
  template <class Panel>
  static void vertices_of(const Panel& p, double v[9]) {
    if (p.vertices.size() != 3) throw Error(FMMBEM_ERR_INVALID, "a panel without vertices");
    for (int a = 0; a < 3; ++a)
      for (int c = 0; c < 3; ++c) v[3 * a + c] = p.vertices[a][c];
  }

 private:
  struct Handle {
    fmmbem_ops* ops; int kernel; unsigned K; double mu; int device;
    Handle(fmmbem_ops* o, int k, unsigned q, double m, int d) : ops(o), kernel(k), K(q), mu(m), device(d) {}
    ~Handle() { fmmbem_ops_destroy(ops); }
    Handle(const Handle&) = delete;
    Handle& operator=(const Handle&) = delete;
  };
  mutable std::shared_ptr<Handle> h_;
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
  // Not in the reference: share of the near-field pairs kept as a matrix (1: the reference's assembled matrix); below 1 the target
  // leaves with the most rows keep none and are recomputed every matvec beside the streamed rest (fmmbem.h near_stream_fraction):
  // fewer HBM bytes and a smaller footprint at the same operator, last bits differ.  Stokes plans (measured optimum 0.5-0.55).
  double near_stream_fraction = 1.0;
  // Not in the reference: the devices ONE plan runs on (fmmbem.h fmmbem_options.n_devices): more than one entry shards the target
  // leaves over them inside the plan; vectors handed to the device entry points live on the first.  Empty: the constructor's
  // `device` argument (or the environment's FMMBEM_DEVICES list, which is how the reference's unmodified drivers get there).
  std::vector<int> devices;
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

  // ---- the single operators of kernel/LaplaceSphericalBEM.hpp:143-156, 307-476 (M2P, the treecode's, is not offered) ----
  void init_multipole(multipole_type& M, const point_type&, unsigned) const { M.assign(2, std::vector<complex>((size_t)P * (P + 1) / 2, 0.)); }
  void init_local(local_type& L, const point_type&, unsigned) const { L.assign(2, std::vector<complex>((size_t)P * (P + 1) / 2, 0.)); }
  // M[0] += the G moments of a POTENTIAL source, M[1] += the dG/dn moments of a NORMAL_DERIV source (:323-344)
  void P2M(const source_type& source, const charge_type& charge, const point_type& center, multipole_type& M) const {
    double v[9];
    fmmbem::SingleOperators::vertices_of(source, v);
    const uint8_t bc = source.BC == Panel::NORMAL_DERIV ? FMMBEM_BC_NORMAL_DERIV : FMMBEM_BC_POTENTIAL;
    const double c[3] = {center[0], center[1], center[2]};
    std::vector<double> m = fmmbem::SingleOperators::pack(cptrs(M), P);
    fmmbem::check(fmmbem_ops_p2m(ops_.handle(FMMBEM_KERNEL_LAPLACE_BEM, K, 1.0, device), P, 1, v, &bc, &charge, c, m.data()));
    fmmbem::SingleOperators::unpack(m, ptrs(M), P);
  }
  // the vectorised forms (kernel/LaplaceSpherical.hpp:214-232; ExpansionTraits::has_vector_P2M / has_vector_L2P): one device call
  // for all the sources of a box / all the targets of a box
  template <typename SourceIter, typename ChargeIter>
  void P2M(SourceIter first, SourceIter last, ChargeIter c_first, const point_type& center, multipole_type& M) const {
    std::vector<double> v, q;
    std::vector<uint8_t> bc;
    for (; first != last; ++first, ++c_first) {
      double one[9];
      fmmbem::SingleOperators::vertices_of(*first, one);
      v.insert(v.end(), one, one + 9);
      bc.push_back((*first).BC == Panel::NORMAL_DERIV ? FMMBEM_BC_NORMAL_DERIV : FMMBEM_BC_POTENTIAL);
      q.push_back(*c_first);
    }
    const double c[3] = {center[0], center[1], center[2]};
    std::vector<double> m = fmmbem::SingleOperators::pack(cptrs(M), P);
    fmmbem::check(fmmbem_ops_p2m(ops_.handle(FMMBEM_KERNEL_LAPLACE_BEM, K, 1.0, device), P, bc.size(), v.data(), bc.data(), q.data(), c, m.data()));
    fmmbem::SingleOperators::unpack(m, ptrs(M), P);
  }
  template <typename TargetIter, typename ResultIter>
  void L2P(const local_type& L, const point_type& center, TargetIter first, TargetIter last, ResultIter r_first) const {
    std::vector<double> v;
    std::vector<uint8_t> bc;
    for (TargetIter it = first; it != last; ++it) {
      double one[9];
      fmmbem::SingleOperators::vertices_of(*it, one);
      v.insert(v.end(), one, one + 9);
      bc.push_back((*it).BC == Panel::NORMAL_DERIV ? FMMBEM_BC_NORMAL_DERIV : FMMBEM_BC_POTENTIAL);
    }
    const double c[3] = {center[0], center[1], center[2]};
    const std::vector<double> l = fmmbem::SingleOperators::pack(cptrs(L), P);
    std::vector<double> r(bc.size(), 0.0);
    fmmbem::check(fmmbem_ops_l2p(ops_.handle(FMMBEM_KERNEL_LAPLACE_BEM, K, 1.0, device), P, l.data(), c, bc.size(), v.data(), bc.data(), r.data()));
    for (size_t i = 0; i < r.size(); ++i, ++r_first) *r_first += r[i];
  }
  void M2M(const multipole_type& source, multipole_type& target, const point_type& translation) const {
    ops_.shift(fmmbem_ops_m2m, ops_.handle(FMMBEM_KERNEL_LAPLACE_BEM, K, 1.0, device), P, cptrs(source), ptrs(target), translation);
  }
  void M2L(const multipole_type& source, local_type& target, const point_type& translation) const {
    ops_.shift(fmmbem_ops_m2l, ops_.handle(FMMBEM_KERNEL_LAPLACE_BEM, K, 1.0, device), P, cptrs(source), ptrs(target), translation);
  }
  void L2L(const local_type& source, local_type& target, const point_type& translation) const {
    ops_.shift(fmmbem_ops_l2l, ops_.handle(FMMBEM_KERNEL_LAPLACE_BEM, K, 1.0, device), P, cptrs(source), ptrs(target), translation);
  }
  // result += r0 at a POTENTIAL target, -= r1 at a NORMAL_DERIV target (:448-476)
  void L2P(const local_type& L, const point_type& center, const target_type& target, result_type& result) const {
    double v[9];
    fmmbem::SingleOperators::vertices_of(target, v);
    const uint8_t bc = target.BC == Panel::NORMAL_DERIV ? FMMBEM_BC_NORMAL_DERIV : FMMBEM_BC_POTENTIAL;
    const double c[3] = {center[0], center[1], center[2]};
    const std::vector<double> l = fmmbem::SingleOperators::pack(cptrs(L), P);
    fmmbem::check(fmmbem_ops_l2p(ops_.handle(FMMBEM_KERNEL_LAPLACE_BEM, K, 1.0, device), P, l.data(), c, 1, v, &bc, &result));
  }

 protected:
  int P;
  fmmbem::SingleOperators ops_;
  static std::vector<const std::vector<complex>*> cptrs(const multipole_type& E) {
    if (E.size() != 2) throw fmmbem::Error(FMMBEM_ERR_INVALID, "a LaplaceSphericalBEM expansion holds two coefficient vectors (init_multipole / init_local)");
    return {&E[0], &E[1]};
  }
  static std::vector<std::vector<complex>*> ptrs(multipole_type& E) {
    if (E.size() != 2) throw fmmbem::Error(FMMBEM_ERR_INVALID, "a LaplaceSphericalBEM expansion holds two coefficient vectors (init_multipole / init_local)");
    return {&E[0], &E[1]};
  }
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
  // kernel/StokesSphericalBEM.hpp:12, 420-464: running totals of the strengths every P2M call has seen, summed without
  // synchronisation across OpenMP threads and over all matvecs of a run; examples/StokesBEM.cpp:370-373 prints them as
  // "Totals".  A diagnostic of the reference's host loop with no counterpart on the device: present so that the driver
  // compiles, left at zero.
  mutable complex stokeslet_str[4] = {0., 0., 0., 0.}, stresslet_str[4] = {0., 0., 0., 0.};
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

  // ---- the single operators of kernel/StokesSphericalBEM.hpp:143-153, 391-530.  M[0][0..3]: the stokeslet group that VELOCITY
  // sources feed and VELOCITY targets read; M[1][0..3] is carried through M2M / M2L / L2L like the reference carries it, but
  // P2M of a TRACTION source and L2P at a TRACTION target throw FMMBEM_ERR_UNSUPPORTED (include/fmmbem.h: the plans' far field
  // for those rows is not the reference's stresslet moments) ----
  void init_multipole(multipole_type& M, const point_type&, unsigned) const {
    M.assign(2, std::vector<std::vector<complex>>(4, std::vector<complex>((size_t)P * (P + 1) / 2, 0.)));
  }
  void init_local(local_type& L, const point_type&, unsigned) const {
    L.assign(2, std::vector<std::vector<complex>>(4, std::vector<complex>((size_t)P * (P + 1) / 2, 0.)));
  }
  void P2M(const source_type& source, const charge_type& charge, const point_type& center, multipole_type& M) const {
    double v[9];
    fmmbem::SingleOperators::vertices_of(source, v);
    const uint8_t bc = source.BC == Panel::TRACTION;
    const double c[3] = {center[0], center[1], center[2]}, f[3] = {charge[0], charge[1], charge[2]};
    std::vector<double> m = fmmbem::SingleOperators::pack(cptrs(M, 1), P);
    fmmbem::check(fmmbem_ops_p2m(ops_.handle(FMMBEM_KERNEL_STOKES_BEM, K, Mu, device), P, 1, v, &bc, f, c, m.data()));
    fmmbem::SingleOperators::unpack(m, ptrs(M, 1), P);
  }
  template <typename SourceIter, typename ChargeIter>
  void P2M(SourceIter first, SourceIter last, ChargeIter c_first, const point_type& center, multipole_type& M) const {
    std::vector<double> v, q;
    std::vector<uint8_t> bc;
    for (; first != last; ++first, ++c_first) {
      double one[9];
      fmmbem::SingleOperators::vertices_of(*first, one);
      v.insert(v.end(), one, one + 9);
      bc.push_back((*first).BC == Panel::TRACTION);
      for (int k = 0; k < 3; ++k) q.push_back((*c_first)[k]);
    }
    const double c[3] = {center[0], center[1], center[2]};
    std::vector<double> m = fmmbem::SingleOperators::pack(cptrs(M, 1), P);
    fmmbem::check(fmmbem_ops_p2m(ops_.handle(FMMBEM_KERNEL_STOKES_BEM, K, Mu, device), P, bc.size(), v.data(), bc.data(), q.data(), c, m.data()));
    fmmbem::SingleOperators::unpack(m, ptrs(M, 1), P);
  }
  template <typename TargetIter, typename ResultIter>
  void L2P(const local_type& L, const point_type& center, TargetIter first, TargetIter last, ResultIter r_first) const {
    std::vector<double> v;
    std::vector<uint8_t> bc;
    for (TargetIter it = first; it != last; ++it) {
      double one[9];
      fmmbem::SingleOperators::vertices_of(*it, one);
      v.insert(v.end(), one, one + 9);
      bc.push_back((*it).BC == Panel::TRACTION);
    }
    const double c[3] = {center[0], center[1], center[2]};
    const std::vector<double> l = fmmbem::SingleOperators::pack(cptrs(L, 1), P);
    std::vector<double> r(3 * bc.size(), 0.0);
    fmmbem::check(fmmbem_ops_l2p(ops_.handle(FMMBEM_KERNEL_STOKES_BEM, K, Mu, device), P, l.data(), c, bc.size(), v.data(), bc.data(), r.data()));
    for (size_t i = 0; i < bc.size(); ++i, ++r_first)
      for (int k = 0; k < 3; ++k) (*r_first)[k] += r[3 * i + k];
  }
  void M2M(const multipole_type& source, multipole_type& target, const point_type& translation) const {
    ops_.shift(fmmbem_ops_m2m, ops_.handle(FMMBEM_KERNEL_STOKES_BEM, K, Mu, device), P, cptrs(source, 2), ptrs(target, 2), translation);
  }
  void M2L(const multipole_type& source, local_type& target, const point_type& translation) const {
    ops_.shift(fmmbem_ops_m2l, ops_.handle(FMMBEM_KERNEL_STOKES_BEM, K, Mu, device), P, cptrs(source, 2), ptrs(target, 2), translation);
  }
  void L2L(const local_type& source, local_type& target, const point_type& translation) const {
    ops_.shift(fmmbem_ops_l2l, ops_.handle(FMMBEM_KERNEL_STOKES_BEM, K, Mu, device), P, cptrs(source, 2), ptrs(target, 2), translation);
  }
  void L2P(const local_type& L, const point_type& center, const target_type& target, result_type& result) const {
    double v[9], r[3] = {result[0], result[1], result[2]};
    fmmbem::SingleOperators::vertices_of(target, v);
    const uint8_t bc = target.BC == Panel::TRACTION;
    const double c[3] = {center[0], center[1], center[2]};
    const std::vector<double> l = fmmbem::SingleOperators::pack(cptrs(L, 1), P);
    fmmbem::check(fmmbem_ops_l2p(ops_.handle(FMMBEM_KERNEL_STOKES_BEM, K, Mu, device), P, l.data(), c, 1, v, &bc, r));
    for (int k = 0; k < 3; ++k) result[k] = r[k];
  }

 protected:
  int P;
  fmmbem::SingleOperators ops_;
  // the first `groups` groups of an M[2][4] as a list of coefficient vectors
  static std::vector<const std::vector<complex>*> cptrs(const multipole_type& E, size_t groups) {
    if (E.size() != 2 || E[0].size() != 4 || E[1].size() != 4) throw fmmbem::Error(FMMBEM_ERR_INVALID, "a StokesSphericalBEM expansion is M[2][4] (init_multipole / init_local)");
    std::vector<const std::vector<complex>*> out;
    for (size_t g = 0; g < groups; ++g) for (size_t e = 0; e < 4; ++e) out.push_back(&E[g][e]);
    return out;
  }
  static std::vector<std::vector<complex>*> ptrs(multipole_type& E, size_t groups) {
    if (E.size() != 2 || E[0].size() != 4 || E[1].size() != 4) throw fmmbem::Error(FMMBEM_ERR_INVALID, "a StokesSphericalBEM expansion is M[2][4] (init_multipole / init_local)");
    std::vector<std::vector<complex>*> out;
    for (size_t g = 0; g < groups; ++g) for (size_t e = 0; e < 4; ++e) out.push_back(&E[g][e]);
    return out;
  }
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
  // size the plan for orders up to p now (what execute() does lazily when set_p asks for more)
  void reserve_order(int p) { if (p > p_max_) create(p); }

 private:
  void create(int p_max) {
    std::vector<double> v(9 * n_);
    std::vector<uint8_t> bc(n_);
    for (size_t i = 0; i < n_; ++i) {
      // a default-constructed Panel has no vertices (the reference's MshReader.hpp leaves such panels behind when a file numbers
      // other elements in front of its triangles): say so instead of reading through an empty vector
      if (sources_[i].vertices.size() < 3)
        throw std::invalid_argument("fmmbem: source " + std::to_string(i) + " has " + std::to_string(sources_[i].vertices.size()) + " vertices, a panel needs 3 (default-constructed Panel?)");
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
    if (opts_.devices.size() > 1) {
      o.n_devices = (int32_t)std::min<size_t>(opts_.devices.size(), 8);
      for (int i = 0; i < o.n_devices; ++i) o.devices[i] = opts_.devices[(size_t)i];
      o.device = o.devices[0];
    }
    o.evaluator = opts_.c_evaluator();
    o.l2l_rule = opts_.reference_l2l ? FMMBEM_L2L_REFERENCE : FMMBEM_L2L_COMPLETE;
    o.near_stream_fraction = opts_.near_stream_fraction;
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

// ---------------------------------------------------------------------------------------------------------------------
// The relaxed solvers with the reference's argument lists, resident on the device (fmmbem_gmres, include/fmmbem.h):
//
//     fmmbem::GMRES(plan, x, b, solver_options);          // examples/BEM/GMRES.hpp:119-128;  GMRES_Stokes.hpp the same on Vec<3,double>
//     fmmbem::GMRES(plan, x, b, solver_options, M);       // :131-141, M = Preconditioners::Identity / Diagonal<T>, fmmbem::InnerSolverPC
//     fmmbem::FGMRES(plan, x, b, solver_options[, M]);    // :254-274
//
// The reference's own GMRES.hpp keeps working against this header (every matvec then moves x and y across PCIe and the
// Arnoldi vectors live on the host: INTEGRATION.md gives the cost); these functions are what replaces it when the solve
// should run where the matvec runs: x and b cross once.  `SolverOptions` is whatever struct the caller uses -- the
// reference's (examples/BEM/SolverOptions.hpp:11-39) or one with the same members: residual, max_iters, restart, max_p,
// variable_p, and, where present, p_min and relax_type.  The order rule follows the kernel as the reference's two headers
// do: LaplaceSphericalBEM p = max(1, predict_p) (GMRES.hpp:195), StokesSphericalBEM p = max(p_min, predict_p - 1)
// (GMRES_Stokes.hpp:229), FGMRES :324 / GMRES_Stokes.hpp:373.  With fmmbem::solver_output() (default on, as
// GMRESContext::output) the lines the reference prints -- "it: 003, res: 1.234e-05, fmm_req_p: 4", "Final residual: ..." --
// are printed once the solve returns.
// ---------------------------------------------------------------------------------------------------------------------
namespace fmmbem {

inline bool& solver_output() { static bool on = true; return on; }

struct SolveReport {
  int iterations = 0;
  double residual = 0, seconds = 0;
  std::vector<int> p;               // order of every inner iteration
  std::vector<double> resid;        // |r| / |b| after it
};

namespace detail {
template <class O> auto relax_of(const O& o, int) -> decltype((int)o.relax_type) { return (int)o.relax_type; }
template <class O> int relax_of(const O&, long) { return FMMBEM_RELAX_BOURAS; }
template <class O> auto pmin_of(const O& o, int) -> decltype((int)o.p_min) { return (int)o.p_min; }
template <class O> int pmin_of(const O&, long) { return 5; }

template <class Kernel> struct OrderRules;
template <> struct OrderRules<LaplaceSphericalBEM> { static constexpr int gmres = FMMBEM_ORDER_GMRES, fgmres = FMMBEM_ORDER_FGMRES; };
template <> struct OrderRules<StokesSphericalBEM> { static constexpr int gmres = FMMBEM_ORDER_GMRES_STOKES, fgmres = FMMBEM_ORDER_FGMRES_STOKES; };

template <class Kernel, class Options>
fmmbem_solver_options c_solver_options(const Options& o, bool flexible, int initial_p) {
  fmmbem_solver_options c;
  fmmbem_solver_options_default(&c);
  c.residual = o.residual; c.max_iters = o.max_iters; c.restart = o.restart; c.max_p = (int)o.max_p;
  c.p_min = pmin_of(o, 0); c.variable_p = o.variable_p ? 1 : 0; c.relax_type = relax_of(o, 0);
  c.order_rule = flexible ? OrderRules<Kernel>::fgmres : OrderRules<Kernel>::gmres;
  c.flexible = flexible ? 1 : 0;
  c.initial_p = initial_p;
  return c;
}
}  // namespace detail

// Preconditioners::LocalInnerSolver (examples/BEM/LocalPC.hpp:26-59) and Preconditioners::BlockDiagonal
// (BlockDiagonalPC.hpp:16-60): a few GMRES steps on the near-field-only / leaf-diagonal operator of the same panels.  As a
// functor (x, y) it works with any solver, the reference's FGMRES included; fmmbem::GMRES / FGMRES recognise it and run the
// inner solves on the device as well.
template <class Kernel>
class InnerSolverPC {
 public:
  typedef typename Kernel::charge_type value_type;
  enum Kind { LOCAL, BLOCK_DIAGONAL };
  InnerSolverPC(const Kernel& k, const std::vector<typename Kernel::source_type>& sources, Kind kind = LOCAL, int device = 0)
      : opts_(make_options(kind)), plan_(k, sources, opts_, 0, device) {
    fmmbem_solver_options_default(&inner_);       // LocalPC.hpp:52-54 on top of SolverOptions()
    inner_.residual = 1e-1; inner_.variable_p = 0; inner_.max_iters = 1;
    inner_.max_p = plan_.p_max();
  }
  void operator()(const std::vector<value_type>& x, std::vector<value_type>& y) {
    y.assign(x.size(), value_type(0.));
    check(fmmbem_gmres(plan_.handle(), &inner_, KernelBinding<Kernel>::out(y), KernelBinding<Kernel>::in(x), nullptr, nullptr));
  }
  fmmbem_plan* handle() { return plan_.handle(); }
  const fmmbem_solver_options& inner_options() const { return inner_; }

 private:
  static FMMOptions make_options(Kind kind) {     // LocalPC.hpp:7-16, BlockDiagonalPC.hpp:53-63
    FMMOptions o;
    o.lazy_evaluation = false;
    o.local_evaluation = kind == LOCAL;
    o.block_diagonal = kind == BLOCK_DIAGONAL;
    o.sparse_local = true;
    o.set_mac_theta(0.5);
    return o;
  }
  FMMOptions opts_;
  PlanAdapter<Kernel> plan_;
  fmmbem_solver_options inner_;
};

namespace detail {

// A preconditioner functor of the reference's shape, M(x, y), reduced to what the device solver takes.  Identity and
// Preconditioners::Diagonal are linear and diagonal: two probes recover the reciprocals and prove the shape; anything else
// is refused (an inner-solver preconditioner goes in as fmmbem::InnerSolverPC).
template <class Kernel, class PC>
void describe(PC& M, size_t n, fmmbem_preconditioner& pc, std::vector<double>& recip) {
  typedef typename Kernel::charge_type value_type;
  const size_t dof = sizeof(value_type) / sizeof(double);
  std::vector<value_type> one(n), u(n), y1(n), y2(n);
  double* pu = KernelBinding<Kernel>::out(u);
  double* p1 = KernelBinding<Kernel>::out(one);
  for (size_t i = 0; i < n * dof; ++i) { p1[i] = 1.0; pu[i] = 1.0 + 0.25 * double((i * 2654435761u >> 7) & 3u); }
  M(one, y1);
  M(u, y2);
  const double* r = KernelBinding<Kernel>::in(y1);
  const double* q = KernelBinding<Kernel>::in(y2);
  bool identity = true;
  for (size_t i = 0; i < n * dof; ++i) {
    if (!(std::fabs(q[i] - r[i] * pu[i]) <= 1e-12 * std::fabs(r[i] * pu[i])))
      throw Error(FMMBEM_ERR_UNSUPPORTED, "fmmbem::GMRES: the preconditioner is not diagonal (use fmmbem::InnerSolverPC for inner-solver preconditioners)");
    identity = identity && r[i] == 1.0;
  }
  pc.kind = identity ? FMMBEM_PC_IDENTITY : FMMBEM_PC_DIAGONAL;
  if (!identity) { recip.assign(r, r + n * dof); pc.reciprocals = recip.data(); }
}
template <class Kernel>
void describe(InnerSolverPC<Kernel>& M, size_t, fmmbem_preconditioner& pc, std::vector<double>&) {
  pc.kind = FMMBEM_PC_INNER_PLAN;
  pc.inner_plan = M.handle();
  pc.inner = M.inner_options();
}

template <class Kernel, class Options>
SolveReport solve(PlanAdapter<Kernel>& MV, std::vector<typename Kernel::charge_type>& x, std::vector<typename Kernel::result_type>& b,
                  const Options& opts, const fmmbem_preconditioner* pc, bool flexible) {
  if (x.size() != b.size()) throw Error(FMMBEM_ERR_INVALID, "x.size() != b.size()");
  MV.reserve_order((int)opts.max_p);                       // set_p above the plan's size grows it once, as execute() does
  fmmbem_solver_options c = c_solver_options<Kernel>(opts, flexible, MV.kernel().p());
  SolveReport rep;
  const int cap = opts.max_iters + opts.restart + 2;
  rep.p.assign((size_t)cap, 0);
  rep.resid.assign((size_t)cap, 0.0);
  fmmbem_solver_log log;
  log.capacity = cap; log.p = rep.p.data(); log.resid = rep.resid.data();
  check(fmmbem_gmres(MV.handle(), &c, KernelBinding<Kernel>::out(x), KernelBinding<Kernel>::in(b), pc, &log));
  rep.iterations = log.iterations; rep.residual = log.residual; rep.seconds = log.seconds;
  rep.p.resize((size_t)std::min(log.iterations, cap));
  rep.resid.resize(rep.p.size());
  if (!rep.p.empty()) MV.kernel().set_p(rep.p.back());     // the kernel object ends at the last order set (GMRES.hpp:196)
  if (solver_output()) {
    for (size_t k = 0; k < rep.p.size(); ++k)
      if (!(rep.resid[k] < opts.residual)) std::printf("it: %03d, res: %.3e, fmm_req_p: %01d\n", (int)k + 1, rep.resid[k], rep.p[k]);
    std::printf("Final residual: %.4e, after %d iterations\n", rep.residual, rep.iterations);
  }
  return rep;
}
}  // namespace detail

// In a nested namespace reached through a using-directive: `fmmbem::GMRES(...)` finds the functions, argument-dependent
// lookup does not (using-directives of an associated namespace are ignored) -- so an unqualified `GMRES(plan, ...)` in code
// that also includes the reference's GMRES.hpp keeps meaning the reference's.
namespace device_solvers {
template <class Kernel, class Options>
SolveReport GMRES(PlanAdapter<Kernel>& MV, std::vector<typename Kernel::charge_type>& x, std::vector<typename Kernel::result_type>& b,
                  const Options& opts) {
  return detail::solve(MV, x, b, opts, nullptr, false);
}
template <class Kernel, class Options, class PC>
SolveReport GMRES(PlanAdapter<Kernel>& MV, std::vector<typename Kernel::charge_type>& x, std::vector<typename Kernel::result_type>& b,
                  const Options& opts, PC&& M) {
  fmmbem_preconditioner pc = {};
  std::vector<double> recip;
  detail::describe<Kernel>(M, x.size(), pc, recip);
  return detail::solve(MV, x, b, opts, &pc, false);
}
template <class Kernel, class Options>
SolveReport FGMRES(PlanAdapter<Kernel>& MV, std::vector<typename Kernel::charge_type>& x, std::vector<typename Kernel::result_type>& b,
                   const Options& opts) {
  return detail::solve(MV, x, b, opts, nullptr, true);
}
template <class Kernel, class Options, class PC>
SolveReport FGMRES(PlanAdapter<Kernel>& MV, std::vector<typename Kernel::charge_type>& x, std::vector<typename Kernel::result_type>& b,
                   const Options& opts, PC&& M) {
  fmmbem_preconditioner pc = {};
  std::vector<double> recip;
  detail::describe<Kernel>(M, x.size(), pc, recip);
  return detail::solve(MV, x, b, opts, &pc, true);
}
}  // namespace device_solvers
using namespace device_solvers;

}  // namespace fmmbem
