// The device-resident solvers of include/fmmbem/FMM_plan.hpp beyond the Laplace GMRES of laplace_bem_sequence.cpp:
//   stokes   fmmbem::GMRES on FMM_plan<StokesSphericalBEM> (Vec<3,double> unknowns; order rule of GMRES_Stokes.hpp:229) --
//            the call of examples/StokesBEM.cpp:306-308 on a velocity-BC sphere;
//   fgmres   fmmbem::FGMRES with fmmbem::InnerSolverPC (LOCAL / BLOCK_DIAGONAL) -- examples/LaplaceBEM.cpp:303-311,
//            StokesBEM.cpp:313-320; and the same preconditioner as a plain functor (x, y).
// SolverOptions comes from tests/cpp/relaxed_gmres.hpp (the reference's members); tests/test_cpp_adapter.py compares the
// printed schedule and sums with solver.py on the same plans.
// usage: device_solver_sequence <recursions> <p> <tol>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fmmbem/FMM_plan.hpp"
#include "relaxed_gmres.hpp"

template <class Kernel>
std::vector<typename Kernel::source_type> sphere(int recursions) {
  typedef typename Kernel::point_type point_type;
  size_t n = 0;
  fmmbem::check(fmmbem_mesh_unit_sphere(recursions, nullptr, &n));
  std::vector<double> v(9 * n);
  fmmbem::check(fmmbem_mesh_unit_sphere(recursions, v.data(), &n));
  std::vector<typename Kernel::source_type> panels;
  for (size_t i = 0; i < n; ++i)
    panels.push_back(typename Kernel::source_type(point_type(v[9 * i], v[9 * i + 1], v[9 * i + 2]), point_type(v[9 * i + 3], v[9 * i + 4], v[9 * i + 5]),
                                                  point_type(v[9 * i + 6], v[9 * i + 7], v[9 * i + 8])));
  return panels;
}

int main(int argc, char** argv) {
  const int recursions = argc > 1 ? std::atoi(argv[1]) : 4;
  const int p = argc > 2 ? std::atoi(argv[2]) : 8;
  const double tol = argc > 3 ? std::atof(argv[3]) : 1e-5;
  try {
    FMMOptions opts;
    opts.sparse_local = true;
    SolverOptions so;
    so.residual = tol;
    so.max_p = p;
    so.max_iters = so.restart = 100;
    {                                                            // ---- Stokes, velocity BC: u = (1, 0, 0) on the sphere
      typedef StokesSphericalBEM kernel_type;
      kernel_type K(p, 4, 1e-3);
      K.set_Kfine(19);
      auto panels = sphere<kernel_type>(recursions);
      std::vector<kernel_type::charge_type> x(panels.size(), kernel_type::charge_type(1.));   // ONE argument: the zero vector
      std::vector<kernel_type::result_type> b(panels.size(), kernel_type::result_type(1., 0., 0.));
      FMM_plan<kernel_type> plan(K, panels, opts);
      std::printf("stokes begin\n");
      const fmmbem::SolveReport rep = fmmbem::GMRES(plan, x, b, so);
      kernel_type::result_type sum(0.);
      for (auto& xi : x) sum += xi;
      std::printf("stokes sum: %.12e %.12e %.12e iterations %d kernel_p %d\n", sum[0], sum[1], sum[2], rep.iterations, plan.kernel().p());
    }
    {                                                            // ---- Laplace first kind, FGMRES + inner-solver preconditioners
      typedef LaplaceSphericalBEM kernel_type;
      kernel_type K(p, 3);
      auto panels = sphere<kernel_type>(recursions);
      std::vector<double> ones(panels.size(), 1.), b;
      FMM_plan<kernel_type> plan(K, panels, opts);
      {
        for (auto& it : panels) it.switch_BC();
        FMM_plan<kernel_type> rhs_plan(K, panels, opts);
        b = rhs_plan.execute(ones);
        for (auto& it : panels) it.switch_BC();
      }
      for (int kind = 0; kind < 2; ++kind) {
        fmmbem::InnerSolverPC<kernel_type> M(K, panels, kind == 0 ? fmmbem::InnerSolverPC<kernel_type>::LOCAL
                                                                  : fmmbem::InnerSolverPC<kernel_type>::BLOCK_DIAGONAL);
        std::vector<double> x(panels.size(), 0.);
        plan.kernel().set_p(p);
        std::printf("fgmres %d begin\n", kind);
        const fmmbem::SolveReport rep = fmmbem::FGMRES(plan, x, b, so, M);
        double sum = 0;
        for (double xi : x) sum += xi;
        std::printf("fgmres %d sum: %.12e iterations %d\n", kind, sum, rep.iterations);
        std::vector<double> z;                                   // the same preconditioner as a functor
        M(b, z);
        double zs = 0;
        for (double zi : z) zs += zi;
        std::printf("functor %d sum: %.12e\n", kind, zs);
      }
      // a preconditioner that is not diagonal is refused, not silently mis-applied
      struct Shift { void operator()(const std::vector<double>& a, std::vector<double>& y) const { y = a; std::rotate(y.begin(), y.begin() + 1, y.end()); } } shift;
      std::vector<double> x(panels.size(), 0.);
      try {
        fmmbem::GMRES(plan, x, b, so, shift);
        std::printf("shift accepted\n");
      } catch (const fmmbem::Error& e) {
        std::printf("shift refused %d\n", e.status);
      }
    }
  } catch (const fmmbem::Error& err) {
    std::printf("error %d %s\n", err.status, err.what());
    return 2;
  }
  return 0;
}
