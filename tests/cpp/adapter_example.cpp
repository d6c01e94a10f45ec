// Mirrors how the reference's own programs drive a plan (tests/scaling.cpp:41-54,
// examples/LaplaceBEM.cpp:203-232): build panels, make a plan, execute, relax p, execute again.
// Prints "<n> <p> <sum of results> <result[0]> <result[n-1]>" per execute for the Python test to compare
// against the oracle.  usage: adapter_example <recursions>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fmmbem/FMM_plan.hpp"

int main(int argc, char** argv) {
  const int r = argc > 1 ? std::atoi(argv[1]) : 4;
  size_t n = 0;
  fmmbem::check(fmmbem_mesh_unit_sphere(r, nullptr, &n));
  std::vector<double> v(9 * n);
  fmmbem::check(fmmbem_mesh_unit_sphere(r, v.data(), &n));
  typedef LaplaceSphericalBEM::Panel Panel;
  std::vector<Panel> panels;
  for (size_t i = 0; i < n; ++i)
    panels.emplace_back(LaplaceSphericalBEM::point_type{v[9 * i], v[9 * i + 1], v[9 * i + 2]}, LaplaceSphericalBEM::point_type{v[9 * i + 3], v[9 * i + 4], v[9 * i + 5]},
                        LaplaceSphericalBEM::point_type{v[9 * i + 6], v[9 * i + 7], v[9 * i + 8]});
  FMMOptions opts;
  opts.sparse_local = true;
  LaplaceSphericalBEM K(10, 3);
  try {
    FMM_plan<LaplaceSphericalBEM> plan(K, panels, opts, 12);
    std::vector<double> charges(n, 1.0);
    for (int p : {12, 10, 5}) {
      plan.kernel().set_p(p);
      std::vector<double> res = plan.execute(charges);
      double sum = 0;
      for (double x : res) sum += x;
      std::printf("%zu %d %.17g %.17g %.17g\n", n, p, sum, res[0], res[n - 1]);
    }
    // examples/StokesBEM.cpp:113-141 -- same mesh as Stokes velocity panels, charges (1,0,0)
    typedef StokesSphericalBEM::Panel SPanel;
    std::vector<SPanel> spanels;
    for (size_t i = 0; i < n; ++i)
      spanels.emplace_back(StokesSphericalBEM::point_type{v[9 * i], v[9 * i + 1], v[9 * i + 2]}, StokesSphericalBEM::point_type{v[9 * i + 3], v[9 * i + 4], v[9 * i + 5]},
                           StokesSphericalBEM::point_type{v[9 * i + 6], v[9 * i + 7], v[9 * i + 8]});
    StokesSphericalBEM KS(8, 3, 1e-3);
    KS.set_Kfine(19);
    FMM_plan<StokesSphericalBEM> splan(KS, spanels, opts, 8);
    std::vector<StokesSphericalBEM::charge_type> f(n, StokesSphericalBEM::charge_type{1., 0., 0.});
    std::vector<StokesSphericalBEM::result_type> u = splan.execute(f);
    double s3[3] = {0, 0, 0};
    for (const auto& r : u)
      for (int c = 0; c < 3; ++c) s3[c] += r[c];
    std::printf("stokes %zu %d %.17g %.17g %.17g\n", n, KS.p(), s3[0], s3[1], s3[2]);
    // the driver's right-hand-side step (examples/StokesBEM.cpp:266-278): every panel switched to TRACTION, the plan run on
    // u = (1,0,0); the double layer of a closed surface gives 4 pi u
    for (auto& pnl : spanels) pnl.switch_BC();
    {
      FMM_plan<StokesSphericalBEM> rhs_plan(KS, spanels, opts, 8);
      std::vector<StokesSphericalBEM::result_type> b = rhs_plan.execute(f);
      double mean = 0, off = 0;
      for (const auto& r : b) { mean += r[0]; off += std::fabs(r[1]) + std::fabs(r[2]); }
      std::printf("traction accepted %.17g %.17g\n", mean / n / (4 * M_PI), off / n);
    }
    // ... at any order up to 16 (13 ... 16 through the double-sum M2L): 4 pi u again, closer
    {
      StokesSphericalBEM K14(14, 3, 1e-3);
      K14.set_Kfine(19);
      FMM_plan<StokesSphericalBEM> rhs14(K14, spanels, opts, 14);
      std::vector<StokesSphericalBEM::result_type> b = rhs14.execute(f);
      double mean = 0, off = 0;
      for (const auto& r : b) { mean += r[0]; off += std::fabs(r[1]) + std::fabs(r[2]); }
      std::printf("traction14 accepted %.17g %.17g\n", mean / n / (4 * M_PI), off / n);
    }
  } catch (const fmmbem::Error& e) {
    std::printf("error %d %s\n", e.status, e.what());
    return 2;
  }
  return 0;
}
