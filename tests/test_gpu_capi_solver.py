"""The device-resident relaxed (F)GMRES behind the C ABI (include/fmmbem.h fmmbem_gmres_device / fmmbem_gmres;
csrc/krylov.hip) -- what a C or C++ caller of the library gets in place of the reference's host-side Arnoldi
(examples/BEM/GMRES.hpp:143-252, 276-380; GMRES_Stokes.hpp:173-320).  Held to the same pins as solver.py: the reference's
own GMRES.hpp compiled unmodified around the oracle's matvec (tests/golden/gmres_ref_r{4,5,6}.json: order before every matvec,
iteration count, printed residuals, solution), and to solver.py's solution on the same plan."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from test_solver import _check_against_reference_run, _ref_runs

pytestmark = pytest.mark.gpu


def _first_kind(fb, r, p):
    import torch
    v = fb.unit_sphere(r)
    n = len(v)
    K = fb.LaplaceSphericalBEM(p, 3)
    plan = fb.FMM_plan(K, v, p_max=p)
    rhs = fb.FMM_plan(fb.LaplaceSphericalBEM(p, 3), v, bc=np.ones(n, dtype=np.uint8), p_max=p)
    b = rhs.execute_torch(torch.ones(n, dtype=torch.float64, device="cuda"))
    rhs.close()
    return v, K, plan, b


@pytest.mark.parametrize("r", [4, 5, 6])
def test_capi_gmres_equals_reference_gmres_run(fb, r):
    import torch
    for run in _ref_runs(r):
        v, K, plan, b = _first_kind(fb, r, run["max_p"])
        so = fb.SolverOptions(residual=run["tol"], max_iters=500, max_p=run["max_p"])
        log = []
        x, it, res, secs = fb.gmres_capi(plan, torch.zeros_like(b), b, so, log=log)
        if run["tol"] >= 1e-6:
            _check_against_reference_run(run, log, x, it, res, 1e-8)
        else:                                  # deep in a 1e-10 solve a residual sits within rounding of a predict_p threshold
            ps, ref = [p for _, p, _ in log], run["p_set"]
            assert abs(it - run["iterations"]) <= 1 and res < run["tol"]
            m = min(len(ps), len(ref))
            assert ps[:6] == ref[:6] and sum(a == b_ for a, b_ in zip(ps, ref)) >= m - 3
        # and solver.py on the same plan: same schedule, solution to 1e-10
        K.set_p(run["max_p"])
        log2 = []
        x2, it2, res2 = fb.gmres(plan, torch.zeros_like(b), b, so, log=log2)
        if run["tol"] >= 1e-6:
            assert it2 == it and [p for _, p, _ in log2] == [p for _, p, _ in log]
            assert float(torch.linalg.vector_norm(x - x2) / torch.linalg.vector_norm(x2)) <= 1e-10
        plan.close()


@pytest.mark.parametrize("pc", ["diagonal", "local", "block_diagonal"])
@pytest.mark.parametrize("flexible", [False, True])
def test_capi_preconditioned_solves_match_solver_py(fb, pc, flexible):
    """GMRES(MV, x, b, opts, M) and FGMRES with Preconditioners::Diagonal, LocalInnerSolver, BlockDiagonal: the C ABI's solver
    against solver.py step for step (orders, iterations), solutions to 1e-9."""
    import torch
    v, K, plan, b = _first_kind(fb, 5, 10)
    so = fb.SolverOptions(residual=1e-6, max_iters=200, max_p=10)
    if pc == "diagonal":
        M = fb.Diagonal(plan)
    else:
        M = (fb.LocalInnerSolver if pc == "local" else fb.BlockDiagonal)(fb, fb.LaplaceSphericalBEM(10, 3), v)
    log_c, log_p = [], []
    xc, itc, resc, _ = fb.gmres_capi(plan, torch.zeros_like(b), b, so, M=M, log=log_c, flexible=flexible)
    K.set_p(10)
    if flexible:
        xp, itp, resp = fb.fgmres(plan, torch.zeros_like(b), b, so, M, log=log_p)
    else:
        xp, itp, resp = fb.gmres(plan, torch.zeros_like(b), b, so, M=M, log=log_p)
    assert itc == itp and [p for _, p, _ in log_c] == [p for _, p, _ in log_p]
    assert resc < 1e-6 and abs(resc - resp) <= 1e-6 * resp + 1e-12
    assert float(torch.linalg.vector_norm(xc - xp) / torch.linalg.vector_norm(xp)) <= 1e-9
    xs = xc.cpu().numpy()
    assert np.linalg.norm(xs - 1.0) / np.sqrt(len(xs)) < 1.5e-2
    plan.close()


def test_capi_gmres_restart_and_host_pointers(fb):
    """restart < iterations: the outer loop re-forms r = A x - b at the order of the LAST inner iteration (the kernel object
    keeps it, GMRES.hpp:169) -- same schedule as solver.py; and the host-pointer entry point gives the device one's bits."""
    import torch
    from fmm_bem_relaxed_amd import _capi
    v, K, plan, b = _first_kind(fb, 5, 12)
    so = fb.SolverOptions(residual=1e-8, max_iters=200, max_p=12, restart=4)
    log_c, log_p = [], []
    xc, itc, resc, _ = fb.gmres_capi(plan, torch.zeros_like(b), b, so, log=log_c)
    K.set_p(12)
    xp, itp, resp = fb.gmres(plan, torch.zeros_like(b), b, so, log=log_p)
    assert itc == itp and itc > 8 and [p for _, p, _ in log_c] == [p for _, p, _ in log_p]
    assert float(torch.linalg.vector_norm(xc - xp) / torch.linalg.vector_norm(xp)) <= 1e-9
    # host pointers
    from fmm_bem_relaxed_amd.solver import _c_options
    o = _c_options(so, False, False, 12)
    xh, bh = np.zeros(len(v)), b.cpu().numpy()
    lg = _capi.SolverLog()
    _capi.check(_capi.lib().fmmbem_gmres(plan._h, C.byref(o), xh.ctypes.data_as(C.c_void_p), bh.ctypes.data_as(C.c_void_p), None, C.byref(lg)))
    assert lg.iterations == itc and np.array_equal(xh, xc.cpu().numpy())
    # errors instead of exits
    o.restart = 0
    assert _capi.lib().fmmbem_gmres(plan._h, C.byref(o), xh.ctypes.data_as(C.c_void_p), bh.ctypes.data_as(C.c_void_p), None, None) == _capi.ERR_INVALID
    # b = 0: x0 returned untouched, 0 iterations (the reference divides by zero here)
    o.restart = 50
    x0 = np.full(len(v), 3.0)
    _capi.check(_capi.lib().fmmbem_gmres(plan._h, C.byref(o), x0.ctypes.data_as(C.c_void_p), np.zeros(len(v)).ctypes.data_as(C.c_void_p), None, C.byref(lg)))
    assert lg.iterations == 0 and np.all(x0 == 3.0)
    plan.close()


def test_capi_stokes_order_rule(fb):
    """GMRES_Stokes.hpp:229: p = max(p_min, predict_p - 1) on Vec<3,double> unknowns (velocity BC on a small red blood cell);
    the same schedule and solution as solver.py's stokes=True."""
    import torch
    v = fb.red_blood_cell(4)
    K = fb.StokesSphericalBEM(10, 4, 1e-3)
    K.set_Kfine(19)
    plan = fb.FMM_plan(K, v, p_max=10)
    n = len(v)
    b = torch.zeros(3 * n, dtype=torch.float64, device="cuda")
    b[0::3] = 1.0
    so = fb.SolverOptions(residual=1e-5, max_iters=100, max_p=10, p_min=5)
    log_c, log_p = [], []
    xc, itc, resc, _ = fb.gmres_capi(plan, torch.zeros_like(b), b, so, log=log_c, stokes=True)
    K.set_p(10)
    xp, itp, resp = fb.gmres(plan, torch.zeros_like(b), b, so, log=log_p, stokes=True)
    assert itc == itp and [p for _, p, _ in log_c] == [p for _, p, _ in log_p] and min(p for _, p, _ in log_c) >= 5
    assert float(torch.linalg.vector_norm(xc - xp) / torch.linalg.vector_norm(xp)) <= 1e-9
    plan.close()


_ALLOC_RETRY = r"""
import sys
import numpy as np, torch
sys.path.insert(0, %r)
import fmm_bem_relaxed_amd as fb
v = fb.unit_sphere(4); n = len(v)
plan = fb.FMM_plan(fb.LaplaceSphericalBEM(8, 3), v, p_max=8)
rhs = fb.FMM_plan(fb.LaplaceSphericalBEM(8, 3), v, bc=np.ones(n, dtype=np.uint8), p_max=8)
b = rhs.execute_torch(torch.ones(n, dtype=torch.float64, device="cuda"))
so = fb.SolverOptions(residual=1e-6, max_iters=40, max_p=8)
try:
    fb.gmres_capi(plan, torch.zeros_like(b), b, so)
    print("NOFAIL")
except fb.FmmBemError as e:
    print("FAILED", e.status if hasattr(e, "status") else "", str(e)[:200].replace("\n", " "))
x, it, res, _ = fb.gmres_capi(plan, torch.zeros_like(b), b, so)            # the retry on the SAME plan
x2, it2, res2 = fb.gmres(plan, torch.zeros_like(b), b, so)
print("RETRY", it, it2, float(torch.linalg.vector_norm(x - x2) / torch.linalg.vector_norm(x2)), float(res))
"""


@pytest.mark.parametrize("k", [2, 3, 5, 6])
def test_solver_workspace_survives_a_failed_allocation(k, tmp_path):
    """an allocation of the solver workspace that fails (injected: the k-th one, FMMBEM_KRYLOV_FAIL_GROW) must leave
    the workspace the plan keeps EMPTY, so that the next solve on the same plan allocates again -- not half-grown with its sizes
    already committed, which made the retry launch kernels on null pointers (a GPU fault instead of FMMBEM_ERR_ALLOC)."""
    import subprocess
    import sys
    script = tmp_path / "retry.py"
    script.write_text(_ALLOC_RETRY % ROOT)
    r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, FMMBEM_KRYLOV_FAIL_GROW=str(k)), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert lines[0].startswith("FAILED") and "injected" in lines[0], lines
    tag, it, it2, diff, res = lines[1].split()
    assert tag == "RETRY" and it == it2 and float(diff) <= 1e-10 and float(res) < 1e-6
