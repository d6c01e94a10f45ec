#!/usr/bin/env python3
"""A/B of the tree passes on the bench workload: M2M and L2L by rotation (kernels_m2l_rot.hip, FMMBEM_ROT_OP = 1, 2) against the
sparse-operator kernels (kernels_far.hip), per order: stage times, whole matvec, relative difference of the results.
usage: python tools/shift_ab.py [--orders 5,6,...] [--min PAIRS] [--workload laplace|stokes_rbc]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fmm_bem_relaxed_amd as fb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="laplace")
    ap.add_argument("--orders", default="5,6,7,8,9,10,11,12")
    ap.add_argument("--recursions", type=int, default=9)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--min", type=int, default=None)
    args = ap.parse_args()
    orders = [int(x) for x in args.orders.split(",")]
    stokes = args.workload == "stokes_rbc"
    if stokes:
        v = fb.red_blood_cell(args.recursions)
        K = fb.StokesSphericalBEM(max(orders), 4, 1e-3)
        K.set_Kfine(19)
    else:
        v = np.concatenate([fb.unit_sphere(args.recursions, center=(3.0 * i, 0.0, 0.0)) for i in range(2)])
        K = fb.LaplaceSphericalBEM(max(orders), 3)
    n, dof = len(v), 3 if stokes else 1
    x = torch.rand(n * dof, dtype=torch.float64, generator=torch.Generator().manual_seed(7)).cuda()
    if args.min is not None:
        os.environ["FMMBEM_SHIFT_ROT_MIN"] = str(args.min)
    res, ys = {}, {}
    for rot in (1, 0):
        os.environ["FMMBEM_SHIFT_ROT"] = str(rot)
        plan = fb.FMM_plan(K, v, p_max=max(orders))
        y = torch.empty_like(x)
        for p in orders:
            for _ in range(2):
                plan.execute_torch(x, out=y, p=p)
            plan.set_timing(True)
            for _ in range(args.steps):
                plan.execute_torch(x, out=y, p=p)
            torch.cuda.synchronize()
            st = plan.stats()
            plan.set_timing(False)
            res.setdefault(p, {})[rot] = (st["ms_m2m"], st["ms_l2l"], st["ms_total"])
            ys.setdefault(p, {})[rot] = y.clone()
        plan.close()
    for p in orders:
        d = float(torch.linalg.vector_norm(ys[p][1] - ys[p][0]) / torch.linalg.vector_norm(ys[p][0]))
        print("p=%2d  m2m rot %.3f ms  sparse %.3f   l2l rot %.3f  sparse %.3f   matvec rot %.3f  sparse %.3f   rel.diff %.2e"
              % (p, res[p][1][0], res[p][0][0], res[p][1][1], res[p][0][1], res[p][1][2], res[p][0][2], d), flush=True)


if __name__ == "__main__":
    main()
