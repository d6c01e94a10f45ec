// Direct.hpp under the reference's file name: Direct::matvec (include/Direct.hpp:232-302) for the adapter's kernel classes -- the
// O(N M) sum r_i += K(t_i, s_j) c_j the drivers use for their exterior-point check (examples/LaplaceBEM.cpp:356-366) and that
// examples/BEM/DirectMatvec.hpp wraps.  Every K(t, s) is evaluated on the device, a target row at a time
// (fmmbem_kernel_entries); there is no host arithmetic of the kernels in this repository.
#pragma once
#include <cassert>
#include <algorithm>
#include <vector>

#include "../FMM_plan.hpp"

class Direct {
  template <class Panel>
  static void flat(const Panel& p, double* v) {
    for (int a = 0; a < 3; ++a)
      for (int c = 0; c < 3; ++c) v[3 * a + c] = p.vertices[a][c];
  }
  static fmmbem_options options_of(const LaplaceSphericalBEM& K) {
    fmmbem_options o;
    fmmbem_options_default(&o);
    o.kernel = FMMBEM_KERNEL_LAPLACE_BEM; o.quad_k = (int)K.K; o.device = K.device;
    return o;
  }
  static fmmbem_options options_of(const StokesSphericalBEM& K) {
    fmmbem_options o;
    fmmbem_options_default(&o);
    o.kernel = FMMBEM_KERNEL_STOKES_BEM; o.quad_k = (int)K.K; o.quad_k_fine = (int)K.K_fine; o.mu = K.Mu; o.device = K.device;
    return o;
  }
  static void add(double& r, const double* k, const double& c) { r += k[0] * c; }
  static void add(Vec<3, double>& r, const double* k, const Vec<3, double>& c) {
    for (int a = 0; a < 3; ++a) r[a] += k[3 * a] * c[0] + k[3 * a + 1] * c[1] + k[3 * a + 2] * c[2];
  }

 public:
  /** Asymmetric matvec (Direct.hpp:236-247): r_i += sum_j K(t_i, s_j) c_j */
  template <typename Kernel, typename SourceIter, typename ChargeIter, typename TargetIter, typename ResultIter>
  static void matvec(const Kernel& K, SourceIter s_first, SourceIter s_last, ChargeIter c_first, TargetIter t_first, TargetIter t_last,
                     ResultIter r_first) {
    const size_t ns = (size_t)(s_last - s_first);
    if (!ns) return;
    constexpr size_t kv = sizeof(typename Kernel::kernel_value_type) / sizeof(double);
    // fmmbem_kernel_entries takes independent (target, source) pairs: a batch of targets against all sources per call, sized so
    // that a call moves ~64 MB (the drivers hand this a few exterior points; DirectMatvec-style use with targets = sources is
    // O(N^2) pairs through PCIe either way -- then in N^2 / batch calls instead of N)
    const size_t batch = std::max<size_t>(1, std::min<size_t>((size_t)(t_last - t_first), ((size_t)1 << 19) / ns + 1));
    std::vector<double> s1(9 * ns), sv(9 * ns * batch), tv(9 * ns * batch), out(kv * ns * batch);
    std::vector<uint8_t> tbc(ns * batch);
    size_t j = 0;
    for (SourceIter s = s_first; s != s_last; ++s, ++j) flat(*s, &s1[9 * j]);
    for (size_t b = 0; b < batch; ++b) std::copy(s1.begin(), s1.end(), sv.begin() + (std::ptrdiff_t)(9 * ns * b));
    const fmmbem_options o = options_of(K);
    while (t_first != t_last) {
      size_t nb = 0;
      TargetIter t = t_first;
      for (; t != t_last && nb < batch; ++t, ++nb) {
        double one[9];
        flat(*t, one);
        for (size_t q = 0; q < ns; ++q) { std::copy(one, one + 9, &tv[9 * (nb * ns + q)]); tbc[nb * ns + q] = t->BC == Kernel::source_type::BC1; }
      }
      fmmbem::check(fmmbem_kernel_entries(&o, ns * nb, tv.data(), tbc.data(), sv.data(), out.data()));
      for (size_t b = 0; b < nb; ++b, ++t_first, ++r_first) {
        ChargeIter c = c_first;
        for (size_t q = 0; q < ns; ++q, ++c) add(*r_first, &out[kv * (b * ns + q)], *c);
      }
    }
  }
  /** Convenience function for std::vector (Direct.hpp:276-289) */
  template <typename Kernel>
  static void matvec(const Kernel& K, const std::vector<typename Kernel::source_type>& s, const std::vector<typename Kernel::charge_type>& c,
                     const std::vector<typename Kernel::target_type>& t, std::vector<typename Kernel::result_type>& r) {
    assert(s.size() == c.size());
    assert(t.size() == r.size());
    matvec(K, s.begin(), s.end(), c.begin(), t.begin(), t.end(), r.begin());
  }
  /** Symmetric form on one set (Direct.hpp:291-302): the same sum with targets = sources */
  template <typename Kernel>
  static void matvec(const Kernel& K, const std::vector<typename Kernel::source_type>& p, const std::vector<typename Kernel::charge_type>& c,
                     std::vector<typename Kernel::result_type>& r) {
    assert(p.size() == c.size());
    assert(p.size() == r.size());
    matvec(K, p.begin(), p.end(), c.begin(), p.begin(), p.end(), r.begin());
  }
};
