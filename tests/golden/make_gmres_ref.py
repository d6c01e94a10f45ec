#!/usr/bin/env python3
"""Writes tests/golden/gmres_ref_r{4,5,6}.json: the REFERENCE's relaxed GMRES (examples/BEM/GMRES.hpp + SolverOptions.hpp +
BLAS.hpp + Preconditioner.hpp, compiled unmodified by `make -C oracle ref`) driving the oracle's matvec on the first-kind
sphere problem of examples/LaplaceBEM.cpp.  Build container only (needs /root/reference); the JSON files are the fixtures
that travel.  Per case: the order set before every matvec (kernel().set_p calls), the residuals GMRES printed, iteration
count, final residual, solution checksums.  tests/test_solver.py holds solver.py to these.
usage: python tests/golden/make_gmres_ref.py"""
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CASES = {4: [(12, "1e-5"), (12, "1e-10"), (8, "1e-6")], 5: [(12, "1e-5"), (12, "1e-10")], 6: [(12, "1e-5"), (12, "1e-10")]}


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_gmres")
    for r, cases in CASES.items():
        out = {"generator": "tests/golden/make_gmres_ref.py", "solver": "reference examples/BEM/GMRES.hpp:119-252 (unmodified)",
               "matvec": "oracle (liboracle.so) orc_matvec", "problem": "examples/LaplaceBEM.cpp:163-291, UnitSphere(%d), k=3, theta=0.5, ncrit=64" % r,
               "runs": []}
        for max_p, tol in cases:
            with tempfile.TemporaryDirectory() as td:
                js = os.path.join(td, "o.json")
                txt = subprocess.run([exe, str(r), str(max_p), tol, js], check=True, capture_output=True, text=True).stdout
                run = json.load(open(js))
            its = [(int(m.group(1)), float(m.group(2)), int(m.group(3)))
                   for m in re.finditer(r"it: (\d+), res: ([0-9.e+-]+), fmm_req_p: (\d+)", txt)]
            fin = re.search(r"Final residual: ([0-9.e+-]+), after (\d+) iterations", txt)
            run["printed_residuals"] = [a[1] for a in its]        # every inner iteration but the converged one (GMRES.hpp:222-225)
            run["printed_p"] = [a[2] for a in its]
            run["final_residual"] = float(fin.group(1))
            run["iterations"] = int(fin.group(2))
            out["runs"].append(run)
            print("r=%d max_p=%d tol=%s: %d iterations, p = %s" % (r, max_p, tol, run["iterations"], run["p_set"]), file=sys.stderr)
        json.dump(out, open(os.path.join(HERE, "gmres_ref_r%d.json" % r), "w"), indent=1)


if __name__ == "__main__":
    main()
