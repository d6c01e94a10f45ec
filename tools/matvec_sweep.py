#!/usr/bin/env python3
"""Whole-matvec time on the bench input (N = 1 048 576) per order under environment settings -- a tuning aid.
  python tools/matvec_sweep.py "FMMBEM_GRAPH=0" "FMMBEM_GRAPH=1" "FMMBEM_SHIFT_ROT2=0" ... [-- p p p]
Each setting = comma-separated NAME=VALUE, applied before the plan is created (plan.hip reads them then); the result of
every setting is compared bit for bit with the first one's."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fmm_bem_relaxed_amd as fb  # noqa: E402


def main():
    args = sys.argv[1:]
    stage = None
    if args and args[0].startswith("--stage="):             # also print this stage's time (ms_p2m, ms_m2l, ...) from an instrumented pass
        stage = args.pop(0).split("=", 1)[1]
    orders = (1, 2, 3, 6, 10, 12)
    if "--" in args:
        k = args.index("--")
        orders = tuple(int(a) for a in args[k + 1:])
        args = args[:k]
    settings = args or ["FMMBEM_STOKES_SYM=1"]
    v = np.concatenate([fb.unit_sphere(9, center=(3.0 * i, 0.0, 0.0)) for i in range(2)])
    x = torch.rand(len(v), dtype=torch.float64, generator=torch.Generator().manual_seed(1)).cuda()
    y = torch.empty_like(x)
    ref = {}
    touched = set()
    for setting in settings:
        for k in touched:
            os.environ.pop(k, None)
        for kv in filter(None, setting.split(",")):
            k, val = kv.split("=", 1)
            os.environ[k] = val
            touched.add(k)
        pm = max(orders)
        K = fb.LaplaceSphericalBEM(pm, 3)
        plan = fb.FMM_plan(K, v, p_max=pm)
        line = []
        for p in orders:
            for _ in range(3):
                plan.execute_torch(x, out=y, p=p)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            h0 = time.perf_counter()
            for _ in range(20):
                plan.execute_torch(x, out=y, p=p)
            host_us = (time.perf_counter() - h0) / 20 * 1e6      # what the host spends enqueueing one matvec
            e1.record()
            torch.cuda.synchronize()
            same = ""
            if p in ref:
                same = "" if torch.equal(y, ref[p]) else " DIFFERS(%.1e)" % float((y - ref[p]).abs().max() / ref[p].abs().max())
            else:
                ref[p] = y.clone()
            extra = ""
            if stage:
                plan.set_timing(1)
                for _ in range(5):
                    plan.execute_torch(x, out=y, p=p)
                torch.cuda.synchronize()
                extra = " (%s %.3f)" % (stage, plan.stats()["ms_" + stage])
                plan.set_timing(0)
            line.append("p=%d %.3f [host %.0f us]%s%s" % (p, e0.elapsed_time(e1) / 20, host_us, extra, same))
        print("%-60s %s" % (setting, "  ".join(line)), flush=True)
        plan.close()


if __name__ == "__main__":
    main()
