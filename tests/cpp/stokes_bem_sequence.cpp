// The call sequence of examples/StokesBEM.cpp:216-218, 255-310 against the adapter header, compiled with the REFERENCE's
// examples/BEM/GMRES_Stokes.hpp (unmodified, -I<reference>/examples/BEM): Vec<3,double> charges and results through
// FMM_plan<StokesSphericalBEM>::execute, kernel().set_p(), VecToArray / ArrayToVec on the adapter's Vec.
// Build-container check only (the GPU box has no reference tree); tests/cpp/adapter_example.cpp executes the Stokes plan.
// usage: stokes_bem_sequence <recursions> <p> <tol>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fmmbem/FMM_plan.hpp"
#include "GMRES_Stokes.hpp"

int main(int argc, char** argv) {
  const int recursions = argc > 1 ? std::atoi(argv[1]) : 3;
  const int p = argc > 2 ? std::atoi(argv[2]) : 8;
  const double tol = argc > 3 ? std::atof(argv[3]) : 1e-5;
  typedef StokesSphericalBEM kernel_type;
  typedef kernel_type::point_type point_type;
  typedef kernel_type::source_type source_type;
  typedef kernel_type::charge_type charge_type;
  typedef kernel_type::result_type result_type;
  static_assert(std::is_same<kernel_type::kernel_value_type, Mat3<double>>::value, "KernelSkeleton::kernel_value_type");
  try {
    FMMOptions opts;
    opts.sparse_local = true;
    SolverOptions solver_options;
    solver_options.residual = tol;
    solver_options.max_p = p;
    kernel_type K(p, 4, 1e-3);                                   // StokesBEM.cpp:216
    K.set_Kfine(19);                                             // :218
    size_t n = 0;
    fmmbem::check(fmmbem_mesh_unit_sphere(recursions, nullptr, &n));
    std::vector<double> v(9 * n);
    fmmbem::check(fmmbem_mesh_unit_sphere(recursions, v.data(), &n));
    std::vector<source_type> panels;
    for (size_t i = 0; i < n; ++i)
      panels.push_back(source_type(point_type(v[9 * i], v[9 * i + 1], v[9 * i + 2]), point_type(v[9 * i + 3], v[9 * i + 4], v[9 * i + 5]),
                                   point_type(v[9 * i + 6], v[9 * i + 7], v[9 * i + 8])));
    std::vector<charge_type> x(panels.size(), charge_type(1.));  // :260 -- ONE argument: the zero vector
    std::vector<result_type> b(panels.size(), result_type(4 * M_PI, 0., 0.));   // :273-276
    FMM_plan<kernel_type> plan = FMM_plan<kernel_type>(K, panels, opts);       // :285
    GMRES(plan, x, b, solver_options);                           // :306-308
    result_type sum(0.);
    for (auto& xi : x) sum += xi;
    std::printf("traction sum: %.12e %.12e %.12e\n", sum[0], sum[1], sum[2]);
    const kernel_type::kernel_value_type self = K(panels[0], panels[0]);
    std::printf("self block: %.12e %.12e %.12e\n", self(0, 0), self(0, 1), self(2, 2));
  } catch (const fmmbem::Error& err) {
    std::printf("error %d %s\n", err.status, err.what());
    return 2;
  }
  return 0;
}
