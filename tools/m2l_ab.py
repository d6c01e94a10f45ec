#!/usr/bin/env python3
"""A/B of the two M2L kernels on the bench workload: per order p the M2L and whole-matvec times with the rotation kernel
(kernels_m2l_rot.hip) and with the double-sum kernels (kernels_m2l.hip), and the relative difference of the results.
usage: python tools/m2l_ab.py [--workload laplace|stokes_rbc] [--orders 1,2,...]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fmm_bem_relaxed_amd as fb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="laplace")
    ap.add_argument("--orders", default="1,2,3,4,5,6,7,8,9,10,11,12")
    ap.add_argument("--recursions", type=int, default=9)
    ap.add_argument("--steps", type=int, default=10)
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
    res = {}
    ys = {}
    for rot in (1, 0):
        os.environ["FMMBEM_M2L_ROT"] = str(rot)
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
            res.setdefault(p, {})["rot" if rot else "sum"] = {"m2l_ms": st["ms_m2l"], "mh_ms": st["ms_mh"], "total_ms": st["ms_total"]}
            ys.setdefault(p, {})[rot] = y.clone()
        plan.close()
    for p in orders:
        d = float(torch.linalg.vector_norm(ys[p][1] - ys[p][0]) / torch.linalg.vector_norm(ys[p][0]))
        res[p]["rel_diff"] = d
        print("p=%2d  m2l rot %.3f ms  sum %.3f (+mh %.3f)  matvec rot %.3f  sum %.3f   rel.diff %.2e"
              % (p, res[p]["rot"]["m2l_ms"], res[p]["sum"]["m2l_ms"], res[p]["sum"]["mh_ms"], res[p]["rot"]["total_ms"],
                 res[p]["sum"]["total_ms"], d), flush=True)
    print(json.dumps({"workload": args.workload, "n_panels": n, "orders": res}))


if __name__ == "__main__":
    main()
