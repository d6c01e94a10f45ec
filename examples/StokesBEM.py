#!/usr/bin/env python3
"""The reference's Stokes driver, examples/StokesBEM.cpp, on the MI355X library: same command-line flags (:80-97,
:147-212), same steps -- unit sphere / red blood cell / mesh, unit velocity (1,0,0) on every panel, the right-hand side
the driver ends up using (it computes A_traction * u by FMM, reports how far that is from 4 pi, and then OVERWRITES it with
the analytic (4 pi, 0, 0), :262-277; the reference's own traction far field is wrong, SURVEY.md section 8a -- here the
double-layer far field of csrc/kernels_far.hip evaluates that step, up to order 12), the first-
kind stokeslet system solved by the relaxed GMRES of GMRES_Stokes.hpp (:307-330), and the same report: timing, drag
against 6 pi mu, area and pointwise traction errors (:337-372).  All matvecs run in libfmmbem_hip.so.

    python examples/StokesBEM.py -recursions 4 -p 10
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
    print("StokesBEM : FMM-BEM for Stokes problems\n\nUsage: StokesBEM.py <options>\n\nOptions:\n"
          "-theta <double> : Set MAC theta for treecode evaluators\n"
          "-eval {FMM,TREE} : Choose either FMM or treecode evaluator (only FMM is built)\n"
          "-p <double> : Number of terms in the Multipole / Local expansions\n"
          "-k {1,3,4,7} : Number of Gauss integration points used per panel\n"
          "-ncrit <int> : Maximum # of particles per Octree box\n"
          "-recursions <int> : number of recursive subdivisions to create a sphere - # panels = 2*4^recursions\n"
          "-rbc <int> : number of recursive subdivisions to create a red blood cell - # panels = 2*4^recursions\n"
          "-cells <int> : number of red blood cells to generate\n"
          "-fixed_p : Disable relaxation\n"
          "-pmin <int>, -mu <double>, -kfine <int>, -solver_tol <double>, -mesh <file.msh>, -vert <f> -face <f>,\n"
          "-fgmres, -diagonal, -local\n-help : print this message")
    sys.exit(0)


class _Logged(list):
    """The per-iteration line of GMRES_Stokes.hpp:253."""

    def append(self, row):
        super().append(row)
        print("it: %03d, res: %.3e, fmm_req_p: %01d" % (row[0], row[2], row[1]))


def main(argv):
    if len(argv) == 1:
        print_help_and_exit()
    recursions, p, k, kfine, cells, mu, p_min = 4, 8, 4, 19, 1, 1e-3, 5
    theta, ncrit = 0.5, 64
    mesh = vert = face = None
    rbc = False
    so = fb.SolverOptions()
    solver, pc = "gmres", "identity"
    i = 1
    while i < len(argv):
        a = argv[i]
        if a == "-p":
            i += 1; p = int(argv[i])
        elif a == "-pmin":
            i += 1; p_min = int(argv[i])
        elif a == "-k":
            i += 1; k = int(argv[i])
        elif a == "-mu":
            i += 1; mu = float(argv[i])
        elif a == "-recursions":
            i += 1; recursions = int(argv[i])
        elif a == "-rbc":
            i += 1; recursions = int(argv[i]); rbc = True
        elif a == "-cells":
            i += 1; cells = int(argv[i])
        elif a == "-fixed_p":
            so.variable_p = False
        elif a == "-solver_tol":
            i += 1; so.residual = float(argv[i])
        elif a == "-mesh":
            i += 1; mesh = argv[i]
        elif a == "-fgmres":
            solver = "fgmres"
        elif a in ("-diagonal", "-diag"):
            solver, pc = "fgmres", "diagonal"
        elif a == "-local":
            solver, pc = "fgmres", "local"
        elif a == "-help":
            print_help_and_exit()
        elif a == "-kfine":
            i += 1; kfine = int(argv[i])
        elif a == "-vert":
            i += 1; vert = argv[i]
        elif a == "-face":
            i += 1; face = argv[i]
        elif a == "-theta":                                # get_options(), include/FMMOptions.hpp
            i += 1; theta = float(argv[i])
        elif a == "-ncrit":
            i += 1; ncrit = int(argv[i])
        elif a == "-eval":
            i += 1
        elif a == "-disable_sparse":
            raise SystemExit("-disable_sparse: the Stokes near field is only built in assembled form")
        i += 1                                             # unknown arguments are ignored, as the reference does (:207-211)
    so.max_p, so.p_min, so.max_iters, so.restart = p, p_min, 100, 100

    if vert or face:
        print("reading mesh from %s, %s" % (vert, face))
        v = fb.read_vert_face(vert, face)
    elif mesh:
        print("reading mesh from %s" % mesh)
        v = fb.read_msh(mesh)
    elif rbc:
        v = fb.red_blood_cell(recursions) if cells <= 1 else fb.red_blood_cells(recursions, cells)
    else:
        v = fb.unit_sphere(recursions)
    n = len(v)
    opts = fb.FMMOptions()
    opts.set_mac_theta(theta)
    opts.set_max_per_box(ncrit)
    dev = torch.device("cuda", 0)

    print("generating RHS")
    tic = time.time()
    def kernel():
        K = fb.StokesSphericalBEM(p, k, mu)
        K.set_Kfine(kfine)
        return K

    # the reference runs the traction-BC plan on u = (1,0,0) here (switch_BC on every panel), prints the summed relative
    # distance of the result from 4 pi, and replaces it by the analytic value (:266-278)
    b = torch.zeros((n, 3), dtype=torch.float64, device=dev)
    if p <= 12:
        rhs_plan = fb.FMM_plan(kernel(), v, opts, p_max=p, bc=np.ones(n, dtype=np.uint8))
        u = torch.zeros((n, 3), dtype=torch.float64, device=dev)
        u[:, 0] = 1.0
        bt = rhs_plan.execute_torch(u.reshape(-1)).reshape(n, 3)
        print("rhs error: %.4e" % float(((bt[:, 0] - 4 * math.pi).abs() / 4 / math.pi).sum()))
        rhs_plan.close()
    else:
        print("rhs error: not evaluated (the far field of TRACTION targets is built for p <= 12)")
    b[:, 0] = 4 * math.pi
    print("done")
    setup_time = time.time() - tic

    plan = fb.FMM_plan(kernel(), v, opts, p_max=p)
    # x(panels.size(), charge_type(1.)): Vec<3,double> with ONE argument is the zero vector
    x = torch.zeros(3 * n, dtype=torch.float64, device=dev)
    log = _Logged()
    tic = time.time()
    if solver == "gmres":
        print("Solver: GMRES, Preconditioner: Identity")
        x, it, res = fb.gmres(plan, x, b.reshape(-1), so, log=log, stokes=True)
    elif pc == "identity":
        print("Solver: FGMRES, Preconditioner: Identity")
        x, it, res = fb.fgmres(plan, x, b.reshape(-1), so, lambda z: z, log=log, stokes=True)
    elif pc == "diagonal":
        print("Solver: FGMRES, Preconditioner: Block-Diagonal")
        x, it, res = fb.fgmres(plan, x, b.reshape(-1), so, fb.BlockDiagonal(fb, kernel(), v), log=log, stokes=True)
    else:
        print("Solver: FGMRES, Preconditioner: Local Solve")
        x, it, res = fb.fgmres(plan, x, b.reshape(-1), so, fb.LocalInnerSolver(fb, kernel(), v), log=log, stokes=True)
    torch.cuda.synchronize()
    solve_time = time.time() - tic
    print("\nTIMING:\n\tsetup : %.4es\n\tsolve : %.4es\n" % (setup_time, solve_time))

    t = x.reshape(n, 3).cpu().numpy()
    e0, e1 = v[:, 2] - v[:, 0], v[:, 1] - v[:, 0]
    area = 0.5 * np.linalg.norm(np.cross(e0, e1), axis=1)
    fx, fy, fz = (t * area[:, None]).sum(axis=0)
    t_exact = 1.5 * mu
    print("t_exact : %.4g" % t_exact)
    approx = t[:, 0] * area                                # the driver compares t_x * Area with t_exact (:349-351)
    e, e2 = float(((approx - t_exact) ** 2).sum()), n * t_exact * t_exact
    analytical, analytical_area = 6 * math.pi * mu, 4 * math.pi
    print("\n\nFx: %.5f, analytical: %.4g" % (fx, analytical))
    print("Fy: %.4g, Fz: %.4g" % (fy, fz))
    drag_error = abs(analytical - fx) / abs(analytical)
    print("error on a sphere: %.5e" % drag_error)
    print("\n\n\n\tdrag error per panel : %.5e, average panel area: %.5e" % (drag_error / n, area.sum() / n))
    print("\tArea error : %.5e" % (abs(area.sum() - analytical_area) / analytical_area))
    print("\n\nPOINTWISE ERRORS\n\terror: %.3e" % math.sqrt(e / e2))
    return it, res, drag_error, [row[1] for row in log]


if __name__ == "__main__":
    main(sys.argv)
