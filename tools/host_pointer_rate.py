#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point fmmbem_plan_execute (x and y in pageable host memory) next to the
device-resident rate bench.py reports, on the bench workload.  DESIGN.md section 7 quotes the result."""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fb = importlib.import_module("fmm-bem-relaxed_amd")

v = np.concatenate([fb.unit_sphere(9, center=(3.0 * i, 0.0, 0.0)) for i in range(2)])
n = len(v)
plan = fb.FMM_plan(fb.LaplaceSphericalBEM(10, 3), v, fb.FMMOptions())
x = np.random.default_rng(0).random(n)
for _ in range(3):
    y = plan.execute(x)
t0 = time.perf_counter()
for _ in range(20):
    y = plan.execute(x)
host = (time.perf_counter() - t0) / 20
xd = torch.from_numpy(x).cuda()
for _ in range(3):
    yd = plan.execute_torch(xd)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    yd = plan.execute_torch(xd)
torch.cuda.synchronize()
dev = (time.perf_counter() - t0) / 20
assert np.array_equal(yd.cpu().numpy(), y)
print("N = %d, p = 10: host pointers %.3f ms per matvec (%.1f/s); device resident %.3f ms (%.1f/s)" % (n, host * 1e3, 1 / host, dev * 1e3, 1 / dev))
