#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_vectors.npz from the CPU ORACLE (not from the reference, which cannot be
built here).  These vectors freeze the oracle's outputs so that a later change to oracle/ that alters results is
caught (tests/test_golden.py), and they give the GPU tests a second, file-based comparison.
usage: python tests/golden/make_golden.py"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def drand48(n, seed=42):
    libc = ctypes.CDLL("libc.so.6")
    libc.drand48.restype = ctypes.c_double
    libc.srand48(seed)
    return np.array([libc.drand48() for _ in range(n)])


out = {}
for r in (4, 5):
    v = O.unit_sphere(r)
    n = len(v)
    x = drand48(n)
    for tag, bc in (("g", np.zeros(n, dtype=np.uint8)), ("dgdn", np.ones(n, dtype=np.uint8))):
        o = O.Oracle(v, bc=bc)
        out["r%d_%s_x" % (r, tag)] = x
        out["r%d_%s_direct" % (r, tag)] = o.direct(x)
        for p in (5, 10):
            out["r%d_%s_fmm_p%d" % (r, tag, p)] = o.matvec(x, p)
        rp, col, val = o.near_csr()
        rows = np.array([0, n // 2, n - 1])
        out["r%d_%s_near_rows" % (r, tag)] = rows
        for row in rows:
            out["r%d_%s_near_row%d_cols" % (r, tag, row)] = col[rp[row]:rp[row + 1]]
            out["r%d_%s_near_row%d_vals" % (r, tag, row)] = val[rp[row]:rp[row + 1]]
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.npz"), **out)
print("wrote", len(out), "arrays")
