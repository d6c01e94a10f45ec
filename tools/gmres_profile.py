#!/usr/bin/env python3
"""Where a relaxed GMRES iteration spends its time: wall time per execute() (synchronised) vs the rest of the iteration."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fmm_bem_relaxed_amd as fb  # noqa: E402


class Timed:
    def __init__(self, plan):
        self.plan, self.t, self.p = plan, [], []

    def kernel(self):
        return self.plan.kernel()

    def execute_torch(self, x):
        torch.cuda.synchronize()
        t0 = time.time()
        y = self.plan.execute_torch(x)
        torch.cuda.synchronize()
        self.t.append(time.time() - t0)
        self.p.append(self.plan.kernel().P)
        return y


def main():
    r = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    v = np.concatenate([fb.unit_sphere(r, center=(3.0 * i, 0.0, 0.0)) for i in range(2)])
    n = len(v)
    plan = fb.FMM_plan(fb.LaplaceSphericalBEM(12, 3), v, p_max=12)
    rhs = fb.FMM_plan(fb.LaplaceSphericalBEM(12, 3), v, bc=np.ones(n, dtype=np.uint8), p_max=12)
    b = rhs.execute_torch(torch.ones(n, dtype=torch.float64, device="cuda"))
    rhs.close()
    for rep in range(2):
        for variable in (True, False):
            mv = Timed(plan)
            so = fb.SolverOptions(residual=1e-5, max_iters=50, restart=50, max_p=12, variable_p=variable)
            x = torch.zeros(n, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            t0 = time.time()
            x, it, res = fb.gmres(mv, x, b, so)
            torch.cuda.synchronize()
            dt = time.time() - t0
            print("rep %d variable_p=%s: %d iterations %.1f ms total, matvecs %.1f ms, rest %.1f ms" %
                  (rep, variable, it, dt * 1e3, sum(mv.t) * 1e3, (dt - sum(mv.t)) * 1e3))
            print("   per-matvec ms:", " ".join("%d:%.1f" % (p, t * 1e3) for p, t in zip(mv.p, mv.t)), flush=True)


if __name__ == "__main__":
    main()
