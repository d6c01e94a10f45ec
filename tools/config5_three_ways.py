#!/usr/bin/env python3
"""SURVEY.md section 8d config 5 (relaxed GMRES, tol 1e-5, max_p 12, max_iters = restart = 50, on the config-3 input: two
disjoint UnitSphere(r), N = 1 048 576 at r = 9), solved three ways on the same GPU, times side by side:

  reference_host   the REFERENCE's GMRES.hpp / BLAS.hpp (unmodified) above the adapter header: host-side Arnoldi on
                   std::vector, every matvec through the host-pointer fmmbem_plan_execute -- today's drop-in for a
                   reference-side build (oracle/_ref/laplace_bem_sequence_ref, built in the build container by
                   `make -C oracle ref`; skipped when the binary is not there);
  test_host        the same driver source with the test-side solver of tests/cpp/relaxed_gmres.hpp (the reference's
                   loop structure; compiled here with g++): a stand-in for the above where it is missing;
  device_cpp       the same driver source with `#define GMRES fmmbem::GMRES` (-DUSE_DEVICE_SOLVER): fmmbem_gmres behind
                   the C ABI, Arnoldi vectors in HBM, x and b cross PCIe once;
  solver_py        fmm-bem-relaxed_amd/solver.py with torch tensors (device pointers throughout);
  capi_device      fmmbem_gmres_device from Python (device pointers throughout).

Each C++ program prints `solve seconds` (the GMRES call only; plans and right-hand side are built before).  One JSON line."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(tmp, name, extra):
    exe = os.path.join(tmp, name)
    lib = os.path.join(ROOT, "fmm-bem-relaxed_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-w", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "cpp"),
                           *extra, os.path.join(ROOT, "tests", "cpp", "laplace_bem_sequence.cpp"), "-o", exe, "-L" + lib,
                           "-lfmmbem_hip", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def run(exe, r, tol):
    t0 = time.time()
    out = subprocess.run([exe, str(r), "12", str(tol), "0", "2", "50"], capture_output=True, text=True, check=True).stdout.splitlines()
    ps = [int(ln.split("fmm_req_p:")[1]) for ln in out if ln.startswith("it:")]
    fin = [ln for ln in out if ln.startswith("Final residual")][0].split()
    return {"solve_s": float([ln for ln in out if ln.startswith("solve seconds")][0].split()[2]), "iterations": int(fin[4]),
            "residual": float(fin[2].rstrip(",")), "p_schedule_printed": ps, "process_s": time.time() - t0,
            "relative_error_vs_sigma_1": float([ln for ln in out if ln.startswith("relative error")][0].split()[2]),
            "solution_sum": float([ln for ln in out if ln.startswith("solution sum")][0].split()[2])}


def main():
    import tempfile
    r = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-5
    res = {"config": "2 x UnitSphere(%d), N = %d, tol %g, max_p 12, max_iters = restart = 50" % (r, 4 * 4 ** r, tol)}
    with tempfile.TemporaryDirectory() as tmp:
        ref = os.path.join(ROOT, "oracle", "_ref", "laplace_bem_sequence_ref")
        if os.path.exists(ref):
            res["reference_host"] = run(ref, r, tol)
        res["test_host"] = run(build(tmp, "host", ()), r, tol)
        res["device_cpp"] = run(build(tmp, "dev", ("-DUSE_DEVICE_SOLVER",)), r, tol)
    import numpy as np
    import torch
    import fmm_bem_relaxed_amd as fb
    v = np.concatenate([fb.unit_sphere(r, center=(3.0 * i, 0.0, 0.0)) for i in range(2)])
    n = len(v)
    K = fb.LaplaceSphericalBEM(12, 3)
    plan = fb.FMM_plan(K, v, p_max=12)
    rhs = fb.FMM_plan(fb.LaplaceSphericalBEM(12, 3), v, bc=np.ones(n, dtype=np.uint8), p_max=12)
    b = rhs.execute_torch(torch.ones(n, dtype=torch.float64, device="cuda"))
    rhs.close()
    so = fb.SolverOptions(residual=tol, max_iters=50, restart=50, max_p=12)
    for name in ("solver_py", "capi_device"):
        for rep in range(2):                              # the first solve allocates the Krylov basis
            K.set_p(12)
            x = torch.zeros(n, dtype=torch.float64, device="cuda")
            log = []
            torch.cuda.synchronize()
            t0 = time.time()
            if name == "solver_py":
                x, it, rs = fb.gmres(plan, x, b, so, log=log)
            else:
                x, it, rs, _ = fb.gmres_capi(plan, x, b, so, log=log)
            torch.cuda.synchronize()
            dt = time.time() - t0
        res[name] = {"solve_s": dt, "iterations": it, "residual": rs, "p_schedule": [p for _, p, _ in log],
                     "solution_sum": float(x.sum()), "relative_error_vs_sigma_1": float(torch.linalg.vector_norm(x - 1.0) / np.sqrt(n))}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
