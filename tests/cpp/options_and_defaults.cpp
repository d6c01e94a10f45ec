// The rest of the reference's options surface and the constructor defaults, as the reference's drivers use them:
//   FMMOptions opts = get_options(argc, argv);                      examples/LaplaceBEM.cpp:70, include/FMMOptions.hpp:74-106
//   opts.MAC().theta_, opts.MAC()(b1, b2), opts.print_tree()        include/FMMOptions.hpp:19-31, 52-68
//   FMM_plan<K> plan(K, panels, opts);                              no p_max: sized for K's order, grows when set_p asks for more
//   StokesBEM.cpp:266-270: switch_BC() on every panel, then FMM_plan<StokesSphericalBEM>(K, panels, opts) with the defaults
// Prints "options ..." first (no device needed), then "<tag> <n> <p> <p_max of the plan> <sum> <first> <last>" lines.
// usage: options_and_defaults <recursions> [reference-style flags]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fmmbem/FMM_plan.hpp"

struct Box {                       // what DefaultMAC needs of a box: center() and radius()
  Vec<3, double> c;
  double r;
  Vec<3, double> center() const { return c; }
  double radius() const { return r; }
};

int main(int argc, char** argv) {
  const int r = argc > 1 ? std::atoi(argv[1]) : 4;
  FMMOptions opts = get_options(argc, argv);
  opts.sparse_local = true;
  const Box a{Vec<3, double>(0., 0., 0.), 0.5}, b{Vec<3, double>(2.1, 0., 0.), 0.5}, c{Vec<3, double>(1.9, 0., 0.), 0.5};
  std::printf("options %.17g %u %d %d %d %d %d\n", opts.MAC().theta_, opts.max_per_box(), (int)opts.print_tree(), (int)opts.lazy_evaluation,
              (int)(opts.evaluator == FMMOptions::FMM), (int)opts.MAC()(a, b), (int)opts.MAC()(a, c));
  size_t n = 0;
  fmmbem::check(fmmbem_mesh_unit_sphere(r, nullptr, &n));
  std::vector<double> v(9 * n);
  fmmbem::check(fmmbem_mesh_unit_sphere(r, v.data(), &n));
  try {
    {
      typedef LaplaceSphericalBEM::Panel Panel;
      typedef LaplaceSphericalBEM::point_type P3;
      std::vector<Panel> panels;
      for (size_t i = 0; i < n; ++i)
        panels.emplace_back(P3{v[9 * i], v[9 * i + 1], v[9 * i + 2]}, P3{v[9 * i + 3], v[9 * i + 4], v[9 * i + 5]}, P3{v[9 * i + 6], v[9 * i + 7], v[9 * i + 8]});
      LaplaceSphericalBEM K(6, 3);
      FMM_plan<LaplaceSphericalBEM> plan(K, panels, opts);        // defaults
      std::vector<double> charges(n, 1.0);
      for (int p : {6, 4, 9, 6}) {                                // 9 > the order the plan was built for: the plan grows
        plan.kernel().set_p(p);
        std::vector<double> res = plan.execute(charges);
        double sum = 0;
        for (double x : res) sum += x;
        std::printf("laplace %zu %d %d %.17g %.17g %.17g\n", n, p, plan.p_max(), sum, res[0], res[n - 1]);
      }
    }
    {
      typedef StokesSphericalBEM::Panel SPanel;
      typedef StokesSphericalBEM::point_type P3;
      std::vector<SPanel> spanels;
      for (size_t i = 0; i < n; ++i)
        spanels.emplace_back(P3{v[9 * i], v[9 * i + 1], v[9 * i + 2]}, P3{v[9 * i + 3], v[9 * i + 4], v[9 * i + 5]}, P3{v[9 * i + 6], v[9 * i + 7], v[9 * i + 8]});
      StokesSphericalBEM KS(8, 3, 1e-3);
      KS.set_Kfine(19);
      for (auto& pnl : spanels) pnl.switch_BC();                  // StokesBEM.cpp:266-268
      FMM_plan<StokesSphericalBEM> rhs_plan(KS, spanels, opts);   // :269, the defaults
      std::vector<StokesSphericalBEM::charge_type> f(n, StokesSphericalBEM::charge_type{1., 0., 0.});
      std::vector<StokesSphericalBEM::result_type> bvec = rhs_plan.execute(f);
      double mean = 0, off = 0;
      for (const auto& q : bvec) { mean += q[0]; off += std::fabs(q[1]) + std::fabs(q[2]); }
      std::printf("traction %zu %d %d %.17g %.17g\n", n, KS.p(), rhs_plan.p_max(), mean / n / (4 * M_PI), off / n);
    }
  } catch (const fmmbem::Error& e) {
    std::printf("error %d %s\n", e.status, e.what());
    return 2;
  }
  return 0;
}
