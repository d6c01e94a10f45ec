#!/usr/bin/env python3
"""Hybrid near field (fmmbem_options.near_stream_fraction): near-field kernel time, stored bytes and the difference from the fully
streamed operator, for a sweep of the stored share f.  usage: near_hybrid_sweep.py laplace|stokes [recursions] [steps] [f ...]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fmm_bem_relaxed_amd as fb  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "stokes"
    rec = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    fs = [float(a) for a in sys.argv[4:]] or [1.0, 0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3]
    stokes = which == "stokes"
    dev = torch.device("cuda", 0)
    if stokes:
        v = fb.red_blood_cell(rec)
        K = fb.StokesSphericalBEM(8, 4, 1e-3)
        K.set_Kfine(19)
    else:
        v = np.concatenate([fb.unit_sphere(rec), fb.unit_sphere(rec, center=(3.0, 0.0, 0.0))])
        K = fb.LaplaceSphericalBEM(10, 3)
    n, dof = len(v), 3 if stokes else 1
    x = torch.from_numpy(np.random.default_rng(1234).random(n * dof)).to(dev)
    y_ref = None
    for f in fs:
        o = fb.FMMOptions()
        o.near_stream_fraction = f
        plan = fb.FMM_plan(K, v, o, device=0)
        y = torch.empty_like(x)
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            plan.near_device(x.data_ptr(), y.data_ptr(), s)
        torch.cuda.synchronize()
        plan.set_timing(2)
        for _ in range(steps):
            plan.near_device(x.data_ptr(), y.data_ptr(), s)
        torch.cuda.synchronize()
        st = plan.stats()
        plan.set_timing(False)
        # whole matvec
        ym = torch.empty_like(x)
        for _ in range(3):
            plan.execute_torch(x, out=ym)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            plan.execute_torch(x, out=ym)
        b.record()
        torch.cuda.synchronize()
        if y_ref is None:
            y_ref, ym_ref = y.clone(), ym.clone()
        print(json.dumps({"f": f, "near_ms": st["ms_near"], "matvec_ms": a.elapsed_time(b) / steps, "stored_GB": st["near_bytes"] / 1e9,
                          "recomputed_pairs": st["near_recomputed_pairs"], "pairs": st["near_nnz"], "side_entries": st["near_side_entries"],
                          "near_vs_first_rel_l2": float(torch.linalg.vector_norm(y - y_ref) / torch.linalg.vector_norm(y_ref)),
                          "matvec_vs_first_rel_l2": float(torch.linalg.vector_norm(ym - ym_ref) / torch.linalg.vector_norm(ym_ref)),
                          "plan_build_ms": st["build_host_ms"] + st["build_assemble_ms"]}), flush=True)
        plan.close()


if __name__ == "__main__":
    main()
