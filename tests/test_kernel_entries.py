"""fmmbem_kernel_entries: Kernel::operator()(target, source) of the two BEM kernels (kernel/LaplaceSphericalBEM.hpp:273-297,
kernel/StokesSphericalBEM.hpp:377-389) for arbitrary panel pairs, evaluated on the device, against the oracle's entries --
self, near (semi-analytic / K_fine) and far (K-point Gauss) regimes, both boundary conditions."""
import numpy as np
import pytest


def _pairs(n):
    rng = np.random.default_rng(3)
    ti = np.concatenate([np.arange(40), rng.integers(0, n, 200)])
    sj = np.concatenate([np.arange(40), np.clip(ti[40:] + rng.integers(-3, 4, 200), 0, n - 1)])      # self, neighbours ...
    ti = np.concatenate([ti, rng.integers(0, n, 200)])
    sj = np.concatenate([sj, rng.integers(0, n, 200)])                                              # ... and far pairs
    return ti.astype(np.int32), sj.astype(np.int32)


def test_symbol_and_argument_checks(fb):
    import ctypes as C
    lib = fb.lib()
    o = fb.Options()
    lib.fmmbem_options_default(C.byref(o))
    assert lib.fmmbem_kernel_entries(C.byref(o), 1, None, None, None, None) == 1          # FMMBEM_ERR_INVALID
    out = np.empty(1)
    assert lib.fmmbem_kernel_entries(C.byref(o), 0, out.ctypes.data, None, out.ctypes.data, out.ctypes.data) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("k", [3, 7])
def test_laplace_entries(fb, oracle_mod, k):
    v = oracle_mod.unit_sphere(5)
    ti, sj = _pairs(len(v))
    for bc in (0, 1):
        flags = np.full(len(v), bc, dtype=np.uint8)
        o = oracle_mod.Oracle(v, K=k, bc=flags)
        ref = o.kernel_entries(ti, sj)
        got = fb.kernel_entries(fb.LaplaceSphericalBEM(5, k), v[ti], v[sj], target_bc=flags[ti])
        assert np.max(np.abs(got - ref) / np.abs(ref)) <= 1e-12
        o.close()


@pytest.mark.gpu
def test_stokes_entries(fb, oracle_mod):
    v = oracle_mod.red_blood_cell(4)
    ti, sj = _pairs(len(v))
    o = oracle_mod.StokesOracle(v, K=4, K_fine=19, mu=1e-3)
    ref = o.kernel_entries(ti, sj)
    K = fb.StokesSphericalBEM(5, 4, 1e-3)
    K.set_Kfine(19)
    got = fb.kernel_entries(K, v[ti], v[sj])
    assert got.shape == ref.shape == (len(ti), 3, 3)
    assert np.max(np.abs(got - ref)) <= 1e-12 * np.max(np.abs(ref))
    o.close()
