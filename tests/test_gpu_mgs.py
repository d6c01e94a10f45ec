"""fmmbem_mgs_column_device (csrc/krylov.hip): the modified Gram-Schmidt column of examples/BEM/GMRES.hpp:203-212 as one
library call, against the same loop in float64 numpy -- sizes that are odd, not multiples of four, smaller than one
workgroup; one, fourteen and fifty columns; rows of V further apart than n (even and odd strides, the odd ones leave every
other row off the 16-byte boundary); a scratch reused after a larger problem; and the solver with the call switched off."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _numpy_mgs(w, V, ncols):
    w = w.copy()
    h = np.zeros(ncols + 1)
    for k in range(ncols):
        h[k] = np.dot(w, V[k])
        w -= h[k] * V[k]
    h[ncols] = np.linalg.norm(w)
    return h, w, w / h[ncols]


def _call(lib, w, V, ncols, vnext, scratch, h):
    from fmm_bem_relaxed_amd import _capi
    _capi.check(lib.fmmbem_mgs_column_device(w.numel(), w.data_ptr(), V.data_ptr(), V.stride(0), ncols, h.data_ptr(),
                                             vnext.data_ptr(), scratch.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()


@pytest.mark.parametrize("n,ncols,pad", [(2, 1, 0), (3, 1, 0), (257, 14, 0), (1001, 1, 3), (1001, 14, 0), (4098, 50, 6),
                                         (65537, 14, 1), (300003, 50, 0), (1 << 20, 14, 0)])
def test_mgs_column_against_numpy(fb, n, ncols, pad):
    from fmm_bem_relaxed_amd import _capi
    lib = _capi.lib()
    rng = np.random.default_rng(n + ncols)
    ldv = n + pad
    Vh = np.zeros((ncols + 1, ldv))
    Vh[:ncols, :n] = np.linalg.qr(rng.standard_normal((n, min(ncols, n))))[0].T[:ncols] if n >= ncols else rng.standard_normal((ncols, n))
    wh = rng.standard_normal(n)
    dev = torch.device("cuda", 0)
    V = torch.from_numpy(Vh).to(dev)
    w = torch.from_numpy(wh).to(dev)
    h = torch.full((ncols + 1,), float("nan"), dtype=torch.float64, device=dev)
    # no initial contents assumed: fill the scratch with garbage
    scratch = torch.full((int(lib.fmmbem_mgs_scratch_doubles(ncols)),), 1e300, dtype=torch.float64, device=dev)
    _call(lib, w, V, ncols, V[ncols], scratch, h)
    hr, wr, vr = _numpy_mgs(wh, Vh[:, :n], ncols)
    scale = np.linalg.norm(wh)
    assert np.max(np.abs(h.cpu().numpy() - hr)) <= 1e-13 * scale
    assert np.max(np.abs(w.cpu().numpy() - wr)) <= 1e-13 * scale
    assert np.max(np.abs(V[ncols, :n].cpu().numpy() - vr)) <= 1e-12
    if pad:
        assert float(V[ncols, n:].abs().max()) == 0.0          # nothing written past n
    # same bits on a second run (fixed summation order), also with the scratch left over from this one
    w2 = torch.from_numpy(wh).to(dev)
    h2 = torch.empty_like(h)
    vn = torch.empty(n, dtype=torch.float64, device=dev)
    _call(lib, w2, V, ncols, vn, scratch, h2)
    assert torch.equal(h2, h) and torch.equal(w2, w)


def test_mgs_scratch_reused_after_a_larger_problem(fb):
    """The partial sums a large call leaves in the scratch must not leak into a later, smaller call (round-2 advisory)."""
    from fmm_bem_relaxed_amd import _capi
    lib = _capi.lib()
    dev = torch.device("cuda", 0)
    scratch = torch.zeros(int(lib.fmmbem_mgs_scratch_doubles(4)), dtype=torch.float64, device=dev)
    for n in (1 << 20, 1000):
        rng = np.random.default_rng(n)
        Vh = np.zeros((5, n))
        Vh[:4] = np.linalg.qr(rng.standard_normal((n, 4)))[0].T
        wh = rng.standard_normal(n)
        V, w = torch.from_numpy(Vh).to(dev), torch.from_numpy(wh).to(dev)
        h = torch.empty(5, dtype=torch.float64, device=dev)
        _call(lib, w, V, 4, V[4], scratch, h)
        hr, _, _ = _numpy_mgs(wh, Vh, 4)
        assert np.max(np.abs(h.cpu().numpy() - hr)) <= 1e-13 * np.linalg.norm(wh), n


def test_solver_with_and_without_the_fused_column(fb, monkeypatch):
    """gmres with fmmbem_mgs_column_device and with the torch calls it replaces (FMMBEM_FUSED_MGS=0): same orders, same
    iteration count, solutions equal to rounding -- on a mesh whose panel count is odd (an open patch: unaligned rows of V)."""
    v = fb.unit_sphere(5)[:2047]                            # 2 047 panels: an odd ldv
    n = len(v)
    dev = torch.device("cuda", 0)
    K = fb.LaplaceSphericalBEM(10, 3)
    plan = fb.FMM_plan(K, v, p_max=10)
    b = plan.execute_torch(torch.ones(n, dtype=torch.float64, device=dev))
    runs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("FMMBEM_FUSED_MGS", flag)
        log = []
        x, it, res = fb.gmres(plan, torch.zeros(n, dtype=torch.float64, device=dev), b, fb.SolverOptions(residual=1e-8, max_p=10), log=log)
        runs.append((x, it, [p for _, p, _ in log]))
    (x1, it1, p1), (x0, it0, p0) = runs
    assert it1 == it0 and p1 == p0
    assert float((x1 - x0).norm() / x0.norm()) < 1e-10
    assert float((x1 - 1).norm() / np.sqrt(n)) < 1e-5        # A x = A 1
