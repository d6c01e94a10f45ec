// The kernel classes of include/fmmbem/FMM_plan.hpp under the reference's kernel contract (kernel/KernelSkeleton.hpp:62-212):
// the chain of the reference's tests/single_level.cpp -- INITM, P2M, M2M, M2L, INITL, L2L, L2P -- with the BEM kernels, checked
// against Kernel::operator() summed over the sources (what include/Direct.hpp computes).
// With -DUSE_REFERENCE_TRAITS -I$(REFERENCE)/include the REFERENCE's own KernelTraits.hpp / executor/INITM.hpp / INITL.hpp are
// read in place, unmodified: ExpansionTraits<Kernel>::is_valid_fmm (include/KernelTraits.hpp:188-194) must hold, and the
// initialisers go through the reference's dispatchers.
#include <cmath>
#include <cstdio>
#include <vector>

#include "fmmbem/FMM_plan.hpp"

#ifdef USE_REFERENCE_TRAITS
#include "KernelTraits.hpp"
#include "executor/INITL.hpp"
#include "executor/INITM.hpp"
static_assert(ExpansionTraits<LaplaceSphericalBEM>::is_valid_fmm, "LaplaceSphericalBEM: P2M, M2M, M2L, L2L, L2P and operator() with the reference's signatures");
static_assert(ExpansionTraits<StokesSphericalBEM>::is_valid_fmm, "StokesSphericalBEM: the same");
static_assert(ExpansionTraits<LaplaceSphericalBEM>::has_init_multipole && ExpansionTraits<LaplaceSphericalBEM>::has_init_local, "initialisers");
static_assert(ExpansionTraits<StokesSphericalBEM>::has_init_multipole && ExpansionTraits<StokesSphericalBEM>::has_init_local, "initialisers");
static_assert(!ExpansionTraits<LaplaceSphericalBEM>::has_M2P, "the treecode's operator is not offered");
static_assert(ExpansionTraits<LaplaceSphericalBEM>::has_vector_P2M && ExpansionTraits<LaplaceSphericalBEM>::has_vector_L2P, "the vectorised forms");
static_assert(ExpansionTraits<StokesSphericalBEM>::has_vector_P2M && ExpansionTraits<StokesSphericalBEM>::has_vector_L2P, "the vectorised forms");
#endif

template <class Kernel>
static void init_m(const Kernel& K, typename Kernel::multipole_type& M, const typename Kernel::point_type& ext) {
#ifdef USE_REFERENCE_TRAITS
  INITM::eval(K, M, ext, 1u);
#else
  K.init_multipole(M, ext, 1u);
#endif
}
template <class Kernel>
static void init_l(const Kernel& K, typename Kernel::local_type& L, const typename Kernel::point_type& ext) {
#ifdef USE_REFERENCE_TRAITS
  INITL::eval(K, L, ext, 1u);
#else
  K.init_local(L, ext, 1u);
#endif
}

static double lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / (1u << 24) * 2 - 1; }

template <class Kernel>
static std::vector<typename Kernel::source_type> panels_near(const typename Kernel::point_type& c, int n, unsigned& seed) {
  typedef typename Kernel::point_type point;
  std::vector<typename Kernel::source_type> out;
  for (int i = 0; i < n; ++i) {
    const point o = c + point(lcg(seed), lcg(seed), lcg(seed)) * 0.06;
    out.emplace_back(o + point(lcg(seed), lcg(seed), lcg(seed)) * 0.015, o + point(lcg(seed), lcg(seed), lcg(seed)) * 0.015,
                     o + point(lcg(seed), lcg(seed), lcg(seed)) * 0.015);
  }
  return out;
}

static double mag(double v) { return std::fabs(v); }
static double mag(const Vec<3, double>& v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
static double charge_of(double, unsigned& s) { return lcg(s); }
static Vec<3, double> charge_of(const Vec<3, double>&, unsigned& s) { return Vec<3, double>(lcg(s), lcg(s), lcg(s)); }

template <class Kernel>
static double chain(const Kernel& K) {
  typedef typename Kernel::point_type point;
  typedef typename Kernel::charge_type charge;
  typedef typename Kernel::result_type result;
  unsigned seed = 12345;
  const point c_src_child(0.125, 0.125, 0.125), c_src(0.25, 0.25, 0.25), c_tgt(2.25, 0.25, 0.25), c_tgt_child(2.125, 0.375, 0.125);
  const point ext(0.5, 0.5, 0.5);
  const auto src = panels_near<Kernel>(c_src_child, 5, seed);
  const auto tgt = panels_near<Kernel>(c_tgt_child, 6, seed);
  std::vector<charge> q;
  for (size_t j = 0; j < src.size(); ++j) q.push_back(charge_of(charge(), seed));
  typename Kernel::multipole_type M1, M2;
  typename Kernel::local_type L1, L2;
  init_m(K, M1, ext); init_m(K, M2, ext); init_l(K, L1, ext); init_l(K, L2, ext);
  for (size_t j = 0; j < src.size(); ++j) K.P2M(src[j], q[j], c_src_child, M1);
  K.M2M(M1, M2, c_src - c_src_child);
  K.M2L(M2, L1, c_tgt - c_src);
  K.L2L(L1, L2, c_tgt_child - c_tgt);
  // the vectorised forms: all sources in one P2M, all targets in one L2P -- the same numbers as one at a time
  typename Kernel::multipole_type M1v;
  init_m(K, M1v, ext);
  K.P2M(src.begin(), src.end(), q.begin(), c_src_child, M1v);
  std::vector<result> rv(tgt.size(), result());
  K.L2P(L2, c_tgt_child, tgt.begin(), tgt.end(), rv.begin());
  double worst = 0, scale = 0, vec_diff = 0;
  for (size_t i = 0; i < tgt.size(); ++i) {
    result r = result(), want = result();
    K.L2P(L2, c_tgt_child, tgt[i], r);
    for (size_t j = 0; j < src.size(); ++j) want += K(tgt[i], src[j]) * q[j];
    worst = std::max(worst, mag(r - want));
    scale = std::max(scale, mag(want));
    vec_diff = std::max(vec_diff, mag(r - rv[i]));
  }
  if (vec_diff > 1e-14 * scale) return 1.0;            // the vector L2P disagrees with the single one
  {
    typename Kernel::multipole_type Mp1, Mp2;
    init_m(K, Mp1, ext); init_m(K, Mp2, ext);
    K.M2M(M1, Mp1, c_src - c_src_child);
    K.M2M(M1v, Mp2, c_src - c_src_child);
    typename Kernel::local_type La, Lb;
    init_l(K, La, ext); init_l(K, Lb, ext);
    K.M2L(Mp1, La, c_tgt_child - c_src); K.M2L(Mp2, Lb, c_tgt_child - c_src);
    result ra = result(), rb = result();
    K.L2P(La, c_tgt_child, tgt[0], ra); K.L2P(Lb, c_tgt_child, tgt[0], rb);
    if (mag(ra - rb) > 1e-12 * scale) return 1.0;        // the vector P2M disagrees with the sum of single ones
  }
  return worst / scale;
}

int main() {
  try {
    const double el = chain(LaplaceSphericalBEM(12, 3));
    const double es = chain(StokesSphericalBEM(12, 3, 1e-3));
#ifdef USE_REFERENCE_TRAITS
    std::printf("traits is_valid_fmm %d %d\n", (int)ExpansionTraits<LaplaceSphericalBEM>::is_valid_fmm, (int)ExpansionTraits<StokesSphericalBEM>::is_valid_fmm);
#endif
    std::printf("laplace chain rel_err %.3e\nstokes chain rel_err %.3e\n", el, es);
    // an expansion sized for another order is refused
    LaplaceSphericalBEM K(6, 3);
    LaplaceSphericalBEM::multipole_type M, N;
    K.init_multipole(M, Vec<3, double>(1, 1, 1), 0u);
    K.set_p(4);
    K.init_multipole(N, Vec<3, double>(1, 1, 1), 0u);
    bool refused = false;
    try { K.M2M(M, N, Vec<3, double>(0.5, 0, 0)); } catch (const fmmbem::Error& e) { refused = e.status == FMMBEM_ERR_INVALID; }
    // a TRACTION source has no P2M here
    StokesSphericalBEM KS(4, 3, 1e-3);
    StokesSphericalBEM::multipole_type MS;
    KS.init_multipole(MS, Vec<3, double>(1, 1, 1), 0u);
    StokesSphericalBEM::Panel p(Vec<3, double>(0, 0, 0), Vec<3, double>(0.1, 0, 0), Vec<3, double>(0, 0.1, 0));
    p.switch_BC();
    bool unsupported = false;
    try { KS.P2M(p, Vec<3, double>(1, 0, 0), Vec<3, double>(0, 0, 0), MS); } catch (const fmmbem::Error& e) { unsupported = e.status == FMMBEM_ERR_UNSUPPORTED; }
    std::printf("refused %d unsupported %d\n", (int)refused, (int)unsupported);
    return el < 2e-6 && es < 2e-6 && refused && unsupported ? 0 : 1;
  } catch (const fmmbem::Error& e) {
    std::printf("%s\n", e.what());
    return e.status == FMMBEM_ERR_NO_DEVICE ? 2 : 3;
  }
}
