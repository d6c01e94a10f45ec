#!/usr/bin/env python3
"""DESIGN.md section 5's accuracy table: FMM-vs-Direct of the GPU and of the oracle (the reference's algorithm on the CPU) on
the SAME rows, p = 10, theta = 0.5, ncrit = 64, charges = default_rng(1234).random(N):

    UnitSphere(8)           every row (the reference's own measure, tests/scaling.cpp:56-74: error over all bodies)
    UnitSphere(9), (10)     4 096 rows drawn with a fixed seed over the whole vector
    2 x UnitSphere(9)       the bench workload, the same sample

plus where the error lives: the share of the squared error by the tree level of the row's leaf.  One JSON line per mesh.
The oracle is the checker here (Direct sums and the CPU FMM); nothing of this is product code."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fmm_bem_relaxed_amd as fb  # noqa: E402
from oracle import oracle as O    # noqa: E402


def leaf_levels(plan, rows):
    B = plan.boxes()
    inv = np.empty(plan.n, dtype=np.int64)
    inv[plan.perm()] = np.arange(plan.n)
    leaves = np.flatnonzero(B["leaf"])
    order = leaves[np.argsort(B["bb"][leaves])]
    return B["level"][order[np.searchsorted(B["bb"][order], inv[rows], side="right") - 1]]


def main():
    meshes = [("UnitSphere(8)", 1, 8, None), ("UnitSphere(9)", 1, 9, 4096), ("UnitSphere(10)", 1, 10, 4096), ("2 x UnitSphere(9)", 2, 9, 4096)]
    if len(sys.argv) > 1:
        meshes = [m for m in meshes if m[0].replace(" ", "") in sys.argv[1:]]
    for name, spheres, r, nrows in meshes:
        v = np.concatenate([fb.unit_sphere(r, center=(3.0 * i, 0.0, 0.0)) for i in range(spheres)])
        n = len(v)
        x = np.random.default_rng(1234).random(n)
        rows = np.arange(n, dtype=np.int32) if nrows is None else np.sort(np.random.default_rng(4096).choice(n, nrows, replace=False)).astype(np.int32)
        o = O.Oracle(v)
        d = o.direct_rows(x, rows)
        rec = {"mesh": name, "n_panels": n, "rows": len(rows), "rows_drawn": "all" if nrows is None else "seed 4096 over the whole vector"}
        plan = fb.FMM_plan(fb.LaplaceSphericalBEM(12, 3), v, p_max=12)
        lv = leaf_levels(plan, rows)
        for p in (8, 10, 12):
            plan.kernel().set_p(p)
            y = plan.execute(x)
            yo = o.matvec(x, p)
            e = np.abs(y[rows] - d)
            rec["p%d" % p] = {"gpu_vs_direct": float(np.linalg.norm(e) / np.linalg.norm(d)),
                             "oracle_vs_direct": float(np.linalg.norm(yo[rows] - d) / np.linalg.norm(d)),
                             "gpu_vs_oracle_full_vector": float(np.linalg.norm(y - yo) / np.linalg.norm(yo)),
                             "row_rel_err_median": float(np.median(e / np.abs(d))), "row_rel_err_max": float(np.max(e / np.abs(d))),
                             "by_leaf_level": {int(L): {"rows": int((lv == L).sum()), "share_of_squared_error": float((e[lv == L] ** 2).sum() / (e ** 2).sum()),
                                                        "rel_l2": float(np.linalg.norm(e[lv == L]) / np.linalg.norm(d[lv == L]))} for L in np.unique(lv)}}
        plan.close()
        o.close()
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
