#!/usr/bin/env python3
"""What ONE rank of an N-GPU run does, timed alone on one GPU: the plan of shard 0 of `world` on the bench workload, its split
execute (upward -> [exchange skipped: the receive buffer is zeros] -> downward) with stage times.  No collective is timed;
this shows how the per-rank kernel time scales and where it stops scaling.  usage: python tools/shard_time.py [--p 10]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fmm_bem_relaxed_amd as fb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--p", type=int, default=10)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--worlds", type=str, default="1,2,4,8")
    a = ap.parse_args()
    v = np.concatenate([fb.unit_sphere(9, center=(3.0 * i, 0.0, 0.0)) for i in range(2)])
    x = torch.rand(len(v), dtype=torch.float64, generator=torch.Generator().manual_seed(7)).cuda()
    s = torch.cuda.current_stream().cuda_stream
    for world in [int(t) for t in a.worlds.split(",")]:
        K = fb.LaplaceSphericalBEM(a.p, 3)
        r = min(a.rank, world - 1)
        if world == 1:
            plan = fb.FMM_plan(K, v, p_max=a.p)
            y = torch.empty_like(x)
            run = lambda: plan.execute_torch(x, out=y, p=a.p)
            nbytes = 0
        else:
            plan = fb.FMM_plan(K, v, p_max=a.p, shard=(r, world), shard_upward=2)
            plan.set_result_slices(True)
            sc, rc = plan.exchange_counts(a.p)
            send = torch.zeros(max(int(sc.sum()), 1), dtype=torch.float64, device="cuda")
            recv = torch.zeros(max(int(rc.sum()), 1), dtype=torch.float64, device="cuda")
            y = torch.empty_like(x)
            nbytes = int(rc.sum()) * 8

            def run():
                plan.upward_device(x.data_ptr(), send.data_ptr(), s, a.p)
                plan.downward_device(recv.data_ptr(), y.data_ptr(), s, a.p)
        if world > 1 and os.environ.get("FMMBEM_GRAPH", "1") != "0":
            plan.set_graphs(True)                         # what ShardedFMM does for world > 1
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()                          # host time to ISSUE a matvec (the GPU is idle at the start: nothing to wait for)
        for _ in range(a.steps):
            run()
        host_us = (time.perf_counter() - t0) / a.steps * 1e6
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            run()
        e1.record()
        torch.cuda.synchronize()
        wall = e0.elapsed_time(e1) / a.steps
        plan.set_timing(True)
        for _ in range(a.steps):
            run()
        torch.cuda.synchronize()
        st = plan.stats()
        print("world %d rank %d: %.3f ms back to back (host issue %.0f us per matvec) | near %.3f p2m %.3f m2m %.3f m2l %.3f l2l %.3f l2p %.3f gather+deliver %.3f | "
              "multipoles in %.2f MB, near nnz %.0fM, m2l pairs %d" % (
                  world, r, wall, host_us, st["ms_near"], st["ms_p2m"], st["ms_m2m"], st["ms_m2l"], st["ms_l2l"], st["ms_l2p"],
                  st["ms_gather"] + st["ms_scatter"], nbytes / 1e6, st["near_nnz"] / 1e6, st["m2l_pairs_owned"]), flush=True)
        plan.close()


if __name__ == "__main__":
    main()
