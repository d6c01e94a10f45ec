#!/usr/bin/env python3
"""M2L time of experiment builds of the rotation kernel (tools/rot_variant.sh) on the bench workload.
usage: python tools/rot_time.py --p 10 base10 nonop10 ...     (names under fmm-bem-relaxed_amd/variants/; '-' = the product library)"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(p, steps, workload):
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import fmm_bem_relaxed_amd as fb
    if workload == "stokes_rbc":
        v = fb.red_blood_cell(9)
        K = fb.StokesSphericalBEM(p, 4, 1e-3)
        K.set_Kfine(19)
        dof = 3
    else:
        v = np.concatenate([fb.unit_sphere(9, center=(3.0 * i, 0.0, 0.0)) for i in range(2)])
        K = fb.LaplaceSphericalBEM(p, 3)
        dof = 1
    x = torch.rand(len(v) * dof, dtype=torch.float64, generator=torch.Generator().manual_seed(7)).cuda()
    plan = fb.FMM_plan(K, v, p_max=p)
    y = torch.empty_like(x)
    for _ in range(3):
        plan.execute_torch(x, out=y, p=p)
    plan.set_timing(True)
    for _ in range(steps):
        plan.execute_torch(x, out=y, p=p)
    torch.cuda.synchronize()
    st = plan.stats()
    print("m2l %.3f ms  matvec %.3f ms  pairs %d items %d passes %d fill %.3f  |y| %.12e"
          % (st["ms_m2l"], st["ms_total"], st["m2l_pairs_owned"], st["m2l_items"], st["m2l_passes"],
             st["m2l_pairs_owned"] / (64.0 * max(st["m2l_passes"], 1)), float(torch.linalg.vector_norm(y))), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--p", type=int, default=10)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--workload", default="laplace")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("names", nargs="*")
    a = ap.parse_args()
    if a.child:
        return child(a.p, a.steps, a.workload)
    for name in a.names:
        env = dict(os.environ)
        if name != "-":
            env["FMMBEM_LIB"] = os.path.join(ROOT, "fmm-bem-relaxed_amd", "variants", "libfmmbem_hip_%s.so" % name)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--p", str(a.p), "--steps", str(a.steps),
                            "--workload", a.workload], env=env, capture_output=True, text=True)
        print("%-14s %s" % (name, (r.stdout.strip().splitlines() or [r.stderr.strip()[-300:]])[-1]), flush=True)


if __name__ == "__main__":
    main()
