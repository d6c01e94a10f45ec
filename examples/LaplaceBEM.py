#!/usr/bin/env python3
"""The reference's potential-problem driver, examples/LaplaceBEM.cpp, on the MI355X library: same command-line flags
(:47-66, :98-160), same steps (sphere or gmsh mesh, right-hand side from the plan with flipped boundary conditions
:218-232, GMRES / FGMRES with relaxed p and the Identity / Diagonal / Local / block-diagonal preconditioners :276-316),
same report lines (timing, potential at the exterior point (3,3,3) :346-369, relative error against the analytic density
:372).  All matvecs run in libfmmbem_hip.so; this file is host glue.

    python examples/LaplaceBEM.py -recursions 6 -p 12 -theta 0.5
"""
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fmm_bem_relaxed_amd as fb  # noqa: E402


def print_help_and_exit():
    print("serialBEM : FMM-BEM for Potential problems\n\nUsage: LaplaceBEM.py <options>\n\nFMM/Treecode Options:\n"
          "-theta <double> : Set MAC theta for treecode evaluators\n"
          "-eval {FMM,TREE} : Choose either FMM or treecode evaluator (only FMM is built)\n"
          "-ncrit <int> : Maximum # of particles per Octree box\n\nProblem & Solver Options:\n"
          "-p <double> : Number of terms in the Multipole / Local expansions\n"
          "-k {1,3,4,7} : Number of Gauss integration points used per panel\n"
          "-recursions <int> : number of recursive subdivisions to create a sphere - # panels = 2*4^recursions, default = 4\n"
          "-second_kind : enable 'second-kind' option to solve second-kind integral equations\n"
          "-fixed_p : enable 'non-relaxed' option\n"
          "-solver_tol <double> : Set the solver tolerance, default = 1e-5\n"
          "-max_iters <int>, -gmres, -fgmres, -local, -diagonal, -mesh <file.msh>\n-help : print this message")
    sys.exit(0)


class _Logged:
    """Prints the reference's per-iteration line (examples/BEM/GMRES.hpp:219-220) from the solver's log."""

    def __init__(self):
        self.rows = []

    def append(self, row):
        it, p, res = row
        self.rows.append(row)
        print("it: %03d, res: %.3e, fmm_req_p: %01d" % (it, res, p))


def exterior_potential(v, bc_flip, k, density, point):
    """Direct::matvec of the driver (:346-364) for ONE exterior target: far-regime Gauss quadrature of G or dG/dn over
    every panel (kernel/LaplaceSphericalBEM.hpp:199-205, 246-258; the point is far from every panel)."""
    pts, w = fb.quadrature(k)
    q = np.einsum("qa,nac->nqc", pts, v)                               # quadrature points (N, K, 3)
    e0, e1 = v[:, 2] - v[:, 0], v[:, 1] - v[:, 0]
    c = np.cross(e0, e1)
    area = 0.5 * np.linalg.norm(c, axis=1)
    normal = c / (2 * area)[:, None]
    dx = q - np.asarray(point)[None, None, :]
    r = np.linalg.norm(dx, axis=2)
    if bc_flip:                                                        # target POTENTIAL-side kernel: G
        kern = (w[None, :] * area[:, None] / r).sum(axis=1)
    else:
        kern = (w[None, :] * area[:, None] * np.einsum("nqc,nc->nq", dx, normal) / r ** 3).sum(axis=1)
    return float(kern @ density)


def main(argv):
    print("\nLaplaceBEM on a sphere")
    if len(argv) == 1:
        print_help_and_exit()
    theta, ncrit, p, k, recursions = 0.5, 64, 5, 3, 4
    second_kind, mesh = False, None
    so = fb.SolverOptions()
    max_iterations, solver, pc = 500, "gmres", "identity"
    print("parameters : \n============ ")
    i = 1
    while i < len(argv):
        a = argv[i]
        if a == "-theta":
            i += 1; theta = float(argv[i]); print("theta = %s" % argv[i])
        elif a == "-recursions":
            i += 1; recursions = int(argv[i]); print("N = %i" % (2 * 4 ** recursions))
        elif a == "-eval":
            i += 1
        elif a == "-ncrit":
            i += 1; ncrit = int(argv[i]); print("ncrit = %s" % argv[i])
        elif a == "-printtree":
            pass
        elif a == "-p":
            i += 1; p = int(argv[i]); so.max_p = p; print("max-p = %i" % p)
        elif a == "-k":
            i += 1; k = int(argv[i])
        elif a == "-second_kind":
            second_kind = True; print("second-kind = True")
        elif a == "-fixed_p":
            so.variable_p = False; print("relaxed = False")
        elif a == "-solver_tol":
            i += 1; so.residual = float(argv[i]); print("solver_tol = %.2e" % so.residual)
        elif a == "-max_iters":
            i += 1; max_iterations = int(argv[i])
        elif a == "-gmres":
            solver = "gmres"
        elif a == "-fgmres":
            solver = "fgmres"
        elif a == "-local":
            solver, pc = "fgmres", "local"
        elif a == "-diagonal":
            pc = "diagonal"
        elif a == "-help":
            print_help_and_exit()
        elif a == "-mesh":
            i += 1; mesh = argv[i]
        else:
            print('[W]: Unknown command line arg: "%s"' % a)
            print_help_and_exit()
        i += 1
    print("============")
    so.max_iters = so.restart = max_iterations
    so.max_p = p

    if mesh:
        print("reading mesh from: %s" % mesh)
        v = fb.read_msh(mesh)
    else:
        v = fb.unit_sphere(recursions)
    n = len(v)
    opts = fb.FMMOptions()
    opts.set_mac_theta(theta)
    opts.set_max_per_box(ncrit)
    bc = np.full(n, 1 if second_kind else 0, dtype=np.uint8)           # panels switch_BC() for the second kind (:190-191)
    K = fb.LaplaceSphericalBEM(p, k)
    plan = fb.FMM_plan(K, v, opts, bc=bc, p_max=p)
    dev = torch.device("cuda", 0)
    charges = torch.ones(n, dtype=torch.float64, device=dev)

    tic = time.time()
    t2 = time.time()
    print("Flipping BC: %g" % (time.time() - t2))
    t2 = time.time()
    rhs_plan = fb.FMM_plan(fb.LaplaceSphericalBEM(p, k), v, opts, bc=1 - bc, p_max=p)
    print("Creating plan: %g" % (time.time() - t2))
    t2 = time.time()
    b = rhs_plan.execute_torch(charges)
    torch.cuda.synchronize()
    print("Executing plan: %g" % (time.time() - t2))
    rhs_plan.close()
    setup_time = time.time() - tic

    tic = time.time()
    x = torch.zeros(n, dtype=torch.float64, device=dev)
    print("2nd-kind equation being solved" if second_kind else "1st-kind equation being solved")
    log = _Logged()
    if solver == "gmres" and pc == "identity":
        print("Solver: GMRES\nPreconditioner: Identity")
        x, it, res = fb.gmres(plan, x, b, so, log=log)
    elif solver == "gmres" and pc == "diagonal":
        print("Solver: GMRES\nPreconditioner: Diagonal")
        x, it, res = fb.gmres(plan, x, b, so, M=fb.Diagonal(plan, dev), log=log)
    elif solver == "fgmres" and pc == "identity":
        print("Solver: FMRES\nPreconditioner: Identity")
        x, it, res = fb.fgmres(plan, x, b, so, lambda z: z, log=log)
    elif solver == "fgmres" and pc == "diagonal":
        print("Solver: FGMRES\nPreconditioner: Block Diagonal")
        x, it, res = fb.fgmres(plan, x, b, so, fb.BlockDiagonal(fb, fb.LaplaceSphericalBEM(p, k), v, bc=bc), log=log)
    else:
        print("Solver: FGMRES\nPreconditioner: Local solve")
        x, it, res = fb.fgmres(plan, x, b, so, fb.LocalInnerSolver(fb, fb.LaplaceSphericalBEM(p, k), v, bc=bc), log=log)
    torch.cuda.synchronize()
    print("Final residual: %.4e, after %d iterations" % (res, it))
    solve_time = time.time() - tic

    print("\nTIMING:\n\tsetup : %.4es\n\tsolve : %.4es" % (setup_time, solve_time))
    xs = x.cpu().numpy()
    # potential at an exterior point from the computed density and the prescribed data (:346-369)
    out = (3.0, 3.0, 3.0)
    # the driver's exterior target is a default (POTENTIAL) panel, then switch_BC(): G with x, dG/dn with the charges,
    # whichever equation was solved (:355-361)
    r2 = exterior_potential(v, True, k, xs, out)
    r1 = exterior_potential(v, False, k, np.ones(n), out)
    exact = 1.0 / math.sqrt(27.0)
    outside = (r2 - r1) / 4 / math.pi
    print("external phi: %.5g, exact: %.5g, error: %.4e" % (outside, exact, abs(outside - exact) / abs(exact)))
    e = float(((xs - 1.0) ** 2).sum())
    print("relative error: %.3e" % math.sqrt(e / n))
    return it, res, math.sqrt(e / n), abs(outside - exact) / abs(exact)


if __name__ == "__main__":
    main(sys.argv)
