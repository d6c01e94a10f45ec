// The call sequence of the reference's driver, examples/LaplaceBEM.cpp:163-291 and :348-351, against the adapter header:
// kernel, panels, plan, right-hand side through a second plan with flipped boundary flags, the diagonal preconditioner
// built from plan.source_begin()/source_end(), relaxed GMRES, error against sigma = 1.
// Built two ways (tests/test_cpp_adapter.py):
//   * -DUSE_REFERENCE_SOLVER -I<reference>/examples/BEM : GMRES.hpp, SolverOptions.hpp, Preconditioner.hpp and BLAS.hpp
//     are the REFERENCE's files, unmodified -- the proof that the adapter is a drop-in for what they touch
//     (build container only: the GPU box has no reference tree);
//   * otherwise tests/cpp/relaxed_gmres.hpp supplies the same names.
//   * -DUSE_DEVICE_SOLVER (with either of the above): ONE more line, `#define GMRES fmmbem::GMRES`, and the same call sites
//     run the relaxed solve resident on the device (include/fmmbem/FMM_plan.hpp, fmmbem_gmres): the Arnoldi vectors never
//     leave HBM.  Without it the solver above the plan is host code and every matvec crosses PCIe twice.
// usage: laplace_bem_sequence <recursions> <p> <tol> <pc: 0 identity, 1 diagonal> [spheres = 1, centres 3 apart] [restart]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fmmbem/FMM_plan.hpp"
#ifdef USE_REFERENCE_SOLVER
#include "GMRES.hpp"
#else
#include "relaxed_gmres.hpp"
#endif
#ifdef USE_DEVICE_SOLVER
#define GMRES fmmbem::GMRES
#endif

int main(int argc, char** argv) {
  const int recursions = argc > 1 ? std::atoi(argv[1]) : 4;
  const int p = argc > 2 ? std::atoi(argv[2]) : 12;
  const double tol = argc > 3 ? std::atof(argv[3]) : 1e-5;
  const int pc = argc > 4 ? std::atoi(argv[4]) : 0;
  const int spheres = argc > 5 ? std::atoi(argv[5]) : 1;
  const int restart = argc > 6 ? std::atoi(argv[6]) : 0;

  typedef LaplaceSphericalBEM kernel_type;
  typedef kernel_type::point_type point_type;
  typedef kernel_type::source_type source_type;
  typedef kernel_type::charge_type charge_type;
  typedef kernel_type::result_type result_type;
  static_assert(kernel_type::dimension == 3, "KernelSkeleton::dimension");
  static_assert(std::is_same<kernel_type::kernel_value_type, double>::value, "KernelSkeleton::kernel_value_type");
  static_assert(std::is_same<kernel_type::target_type, source_type>::value, "one panel type");
  typedef kernel_type::multipole_type multipole_type;
  typedef kernel_type::local_type local_type;
  (void)sizeof(multipole_type); (void)sizeof(local_type);

  try {
    FMMOptions opts;
    opts.sparse_local = true;                                    // LaplaceBEM.cpp:81
    opts.set_mac_theta(0.5);
    SolverOptions solver_options;
    solver_options.residual = tol;
    solver_options.max_p = p;
    solver_options.restart = solver_options.max_iters;           // :162-163
    if (restart > 0) solver_options.max_iters = solver_options.restart = restart;   // config 5: max_iters = restart = 50

    kernel_type K(p, 3);                                         // :168
    size_t n = 0;
    fmmbem::check(fmmbem_mesh_unit_sphere(recursions, nullptr, &n));
    std::vector<double> v(9 * n);
    fmmbem::check(fmmbem_mesh_unit_sphere(recursions, v.data(), &n));
    std::vector<source_type> panels;                             // Triangulation::UnitSphere, :185
    for (int body = 0; body < spheres; ++body)                   // config 3 / 5: disjoint spheres, centres 3 apart on x
      for (size_t i = 0; i < n; ++i) {
        const double dx = 3.0 * body;
        panels.push_back(source_type(point_type(v[9 * i] + dx, v[9 * i + 1], v[9 * i + 2]), point_type(v[9 * i + 3] + dx, v[9 * i + 4], v[9 * i + 5]),
                                     point_type(v[9 * i + 6] + dx, v[9 * i + 7], v[9 * i + 8])));
      }
    std::vector<charge_type> charges(panels.size(), 1.);         // :203

    FMM_plan<kernel_type> plan = FMM_plan<kernel_type>(K, panels, opts);   // :209

    std::vector<charge_type> x(panels.size(), 0.);
    std::vector<result_type> b(panels.size(), 0.);
    {                                                            // :218-232
      for (auto& it : panels) it.switch_BC();
      FMM_plan<kernel_type> rhs_plan = FMM_plan<kernel_type>(K, panels, opts);
      b = rhs_plan.execute(charges);
      for (auto& it : panels) it.switch_BC();
    }

    Preconditioners::Diagonal<charge_type> M(K, plan.source_begin(), plan.source_end());   // :241-244
    const auto t0 = std::chrono::steady_clock::now();
    if (pc == 0) GMRES(plan, x, b, solver_options);              // :281-285
    else GMRES(plan, x, b, solver_options, M);
    std::printf("solve seconds: %.6f\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());

    double e = 0., e2 = 0.;                                      // :348-351
    for (auto xi : x) { e += (xi - 1.) * (xi - 1.); e2 += 1.; }
    std::printf("relative error: %.6e\n", std::sqrt(e / e2));
    double sum = 0;
    for (auto xi : x) sum += xi;
    std::printf("solution sum: %.15e\n", sum);
    // K(s, s) through the kernel object on a panel that does not come from the plan: the device evaluates it
    std::printf("self entry: %.15e %.15e\n", K(panels[0], panels[0]), K(*plan.source_begin(), *plan.source_begin()));
    std::printf("pair entry: %.15e\n", K(panels[0], panels[1]));
  } catch (const fmmbem::Error& err) {
    std::printf("error %d %s\n", err.status, err.what());
    return 2;
  }
  return 0;
}
