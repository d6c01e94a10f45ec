// Mirrors how the reference's own programs drive a plan (tests/scaling.cpp:41-54,
// examples/LaplaceBEM.cpp:203-232): build panels, make a plan, execute, relax p, execute again.
// Prints "<n> <p> <sum of results> <result[0]> <result[n-1]>" per execute for the Python test to compare
// against the oracle.  usage: adapter_example <recursions>
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
  } catch (const fmmbem::Error& e) {
    std::printf("error %d %s\n", e.status, e.what());
    return 2;
  }
  return 0;
}
