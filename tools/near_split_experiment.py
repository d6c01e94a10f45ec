#!/usr/bin/env python3
"""the near field split between the STREAMED matrix (near_spmv_pipe / near_spmv_sym3) on a fraction
1 - g of the target leaves and the RECOMPUTED far regime (mf_sweep / mf_sweep3_apply) on the fraction g, the two kernels on two
streams.  Two plans of the same geometry, each restricted to its part by FMMBEM_NEAR_SUBSET (csrc/plan.hip, experiment switch).
Prints per g: stream part alone, recompute part alone, both together (wall, events), and the sum check against the full operator.

usage: near_split_experiment.py laplace|stokes [recursions] [steps]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fmm_bem_relaxed_amd as fb  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "laplace"
    rec = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
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

    def plan(subset, sparse):
        if subset:
            os.environ["FMMBEM_NEAR_SUBSET"] = subset
        else:
            os.environ.pop("FMMBEM_NEAR_SUBSET", None)
        o = fb.FMMOptions()
        o.sparse_local = sparse
        return fb.FMM_plan(K, v, o, device=0)

    def timeit(fn, streams):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        main_s = torch.cuda.current_stream()
        a.record(main_s)
        for s in streams:
            s.wait_stream(main_s)
        for _ in range(steps):
            fn()
        for s in streams:
            main_s.wait_stream(s)
        b.record(main_s)
        torch.cuda.synchronize()
        return a.elapsed_time(b) / steps

    full = plan(None, True)
    y_full = torch.empty_like(x)
    s0 = torch.cuda.current_stream()
    t_full = timeit(lambda: full.near_device(x.data_ptr(), y_full.data_ptr(), s0.cuda_stream), [])
    full.near_device(x.data_ptr(), y_full.data_ptr(), s0.cuda_stream)
    torch.cuda.synchronize()
    full.close()
    print(json.dumps({"workload": which, "n": n, "full_streamed_near_ms_incl_gather_scatter": t_full}), flush=True)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    B = 10
    for a in (1, 2, 3, 4, 5):
        pS = plan("-%d/%d" % (a, B), True)       # leaves l % 10 >= a: streamed
        pR = plan("%d/%d" % (a, B), False)       # leaves l % 10 <  a: recomputed
        yS, yR = torch.empty_like(x), torch.empty_like(x)
        tS = timeit(lambda: pS.near_device(x.data_ptr(), yS.data_ptr(), s0.cuda_stream), [])
        tR = timeit(lambda: pR.near_device(x.data_ptr(), yR.data_ptr(), s0.cuda_stream), [])

        def both():
            pR.near_device(x.data_ptr(), yR.data_ptr(), s2.cuda_stream)
            pS.near_device(x.data_ptr(), yS.data_ptr(), s1.cuda_stream)

        def both_rev():
            pS.near_device(x.data_ptr(), yS.data_ptr(), s1.cuda_stream)
            pR.near_device(x.data_ptr(), yR.data_ptr(), s2.cuda_stream)
        tB = timeit(both, [s1, s2])
        tB2 = timeit(both_rev, [s1, s2])
        torch.cuda.synchronize()
        err = float(torch.linalg.vector_norm(yS + yR - y_full) / torch.linalg.vector_norm(y_full))
        stS, stR = pS.stats(), pR.stats()
        print(json.dumps({"g": a / B, "stream_part_ms": tS, "recompute_part_ms": tR, "both_mf_first_ms": tB, "both_stream_first_ms": tB2,
                          "sum_ms": tS + tR, "max_ms": max(tS, tR), "sum_vs_full_rel_l2": err,
                          "streamed_bytes": stS["near_bytes"], "side_entries": stR["near_side_entries"]}), flush=True)
        pS.close()
        pR.close()


if __name__ == "__main__":
    main()
