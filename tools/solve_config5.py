#!/usr/bin/env python3
"""SURVEY.md section 8d config 5 at full size: the first-kind Laplace BEM solve of examples/LaplaceBEM.cpp:168-291 on
the config-3 input (two disjoint UnitSphere(9), N = 1 048 576), GMRES with the Bouras-Fraysse relaxation of p
(max_p = 12, restart 50, tol 1e-5) and, for comparison, the same solve at fixed p = 12.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fmm_bem_relaxed_amd as fb  # noqa: E402


def main():
    tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-5
    r = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    v = np.concatenate([fb.unit_sphere(r, center=(3.0 * i, 0.0, 0.0)) for i in range(2)])
    n = len(v)
    t0 = time.time()
    plan = fb.FMM_plan(fb.LaplaceSphericalBEM(12, 3), v, p_max=12)
    rhs = fb.FMM_plan(fb.LaplaceSphericalBEM(12, 3), v, bc=np.ones(n, dtype=np.uint8), p_max=12)
    build_s = time.time() - t0
    b = rhs.execute_torch(torch.ones(n, dtype=torch.float64, device="cuda"))       # LaplaceBEM.cpp:218-232
    rhs.close()
    out = {"n_panels": n, "tol": tol, "plans_build_s": build_s}
    # one untimed solve first: the Krylov basis allocation (51 x N doubles) and first launches are one-time costs
    fb.gmres(plan, torch.zeros(n, dtype=torch.float64, device="cuda"), b,
             fb.SolverOptions(residual=tol, max_iters=50, restart=50, max_p=12, variable_p=True))
    for name, variable in (("relaxed", True), ("fixed_p12", False)):
        so = fb.SolverOptions(residual=tol, max_iters=50, restart=50, max_p=12, variable_p=variable)
        x = torch.zeros(n, dtype=torch.float64, device="cuda")
        log = []
        torch.cuda.synchronize()
        t0 = time.time()
        x, it, res = fb.gmres(plan, x, b, so, log=log)
        torch.cuda.synchronize()
        dt = time.time() - t0
        out[name] = {"iterations": it, "residual": res, "solve_s": dt, "p_schedule": [p for _, p, _ in log]}
        if name == "relaxed":
            x_relaxed = x.clone()
        else:
            out["rel_diff_relaxed_vs_fixed"] = float(torch.linalg.vector_norm(x - x_relaxed) / torch.linalg.vector_norm(x))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
