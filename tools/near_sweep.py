#!/usr/bin/env python3
"""Times the assembled near-field operator alone (fmmbem_plan_near_device: gather + near_spmv + scatter) on the bench
input -- a tuning aid, not part of the product or the bench.  Environment variables named on the command line as
NAME=VALUE are set before each timing, e.g.  python tools/near_sweep.py FMMBEM_STOKES_SYM=1"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fmm_bem_relaxed_amd as fb  # noqa: E402


def main():
    settings = [a for a in sys.argv[1:] if "=" in a] or [""]
    v = np.concatenate([fb.unit_sphere(9, center=(3.0 * i, 0, 0)) for i in range(2)])
    plan = fb.FMM_plan(fb.LaplaceSphericalBEM(10, 3), v, fb.FMMOptions(), p_max=10)
    st = plan.stats()
    nbytes = 8 * st["near_nnz"] + 16 * plan.n
    x = torch.rand(plan.n, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    ref = None
    stream = torch.cuda.current_stream().cuda_stream
    for setting in settings:
        if setting:
            k, val = setting.split("=", 1)
            os.environ[k] = val
        for _ in range(3):
            plan.near_device(x.data_ptr(), y.data_ptr(), stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            plan.near_device(x.data_ptr(), y.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        if ref is None:
            ref = y.clone()
        err = float((y - ref).abs().max() / ref.abs().max())
        print("%s: %.4f ms incl. gather/scatter  %.0f GB/s  maxdiff vs first %.2e" % (setting or "default", ms, nbytes / ms / 1e6, err),
              flush=True)


if __name__ == "__main__":
    main()
