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
import fmm_bem_relaxed_amd as fb

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
# the same host buffers page-locked by the caller (fmmbem_host_register): x in place, y through a registered result buffer
import ctypes as C
from fmm_bem_relaxed_amd import _capi
L = _capi.lib()
yb = np.empty(n)
for a in (x, yb):
    _capi.check(L.fmmbem_host_register(a.ctypes.data_as(C.c_void_p), a.nbytes))
for _ in range(3):
    _capi.check(L.fmmbem_plan_execute(plan._h, 10, x.ctypes.data_as(C.c_void_p), yb.ctypes.data_as(C.c_void_p)))
t0 = time.perf_counter()
for _ in range(20):
    _capi.check(L.fmmbem_plan_execute(plan._h, 10, x.ctypes.data_as(C.c_void_p), yb.ctypes.data_as(C.c_void_p)))
reg = (time.perf_counter() - t0) / 20
assert np.array_equal(yb, y)
for a in (x, yb):
    _capi.check(L.fmmbem_host_unregister(a.ctypes.data_as(C.c_void_p)))
print("N = %d, p = 10: host pointers, pageable %.3f ms per matvec (%.1f/s); page-locked by the caller %.3f ms (%.1f/s); device resident "
      "%.3f ms (%.1f/s)" % (n, host * 1e3, 1 / host, reg * 1e3, 1 / reg, dev * 1e3, 1 / dev))
