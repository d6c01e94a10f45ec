"""StokesSphericalBEM, velocity boundary condition: oracle known answer
and GPU-vs-oracle parity."""
import numpy as np
import pytest

from conftest import drand48, rel_l2


@pytest.fixture(scope="module")
def stokes5(oracle_mod):
    v = oracle_mod.unit_sphere(5)
    o = oracle_mod.StokesOracle(v, K=4, K_fine=19, mu=1e-3)
    return v, o


def test_oracle_reproduces_reference_error_level(stokes5):
    """StokesBEM (velocity BC) N=2048, p=8, k=4, kfine=19: rel L2 vs Direct 1.43e-5."""
    v, o = stokes5
    x = np.tile([1.0, 0.0, 0.0], (o.n, 1))            # charges of examples/StokesBEM.cpp:257
    err = rel_l2(o.matvec(x, 8), o.direct(x))
    assert abs(err - 1.43e-5) / 1.43e-5 < 0.01, err


def test_oracle_faithful_equals_tuned(stokes5):
    v, o = stokes5
    x = drand48(3 * o.n, seed=2).reshape(o.n, 3)
    assert rel_l2(o.matvec(x, 6, faithful=True), o.matvec(x, 6)) < 1e-14


def test_oracle_sphere_drag_identity(stokes5):
    """Uniform traction 1.5 mu U / R on the unit sphere moves it with U (Stokes drag 6 pi mu R U,
    examples/StokesBEM.cpp:344-361); the single-layer operator here carries no 1/(8 pi): u * 4 pi."""
    v, o = stokes5
    f = np.tile([1.5 * o.mu, 0.0, 0.0], (o.n, 1))
    u = o.direct(f).mean(axis=0)
    assert abs(u[0] - 4 * np.pi) / (4 * np.pi) < 1e-2 and abs(u[1]) < 1e-10 and abs(u[2]) < 1e-10


@pytest.mark.gpu
def test_gpu_stokes_vs_oracle(fb, stokes5):
    v, o = stokes5
    K = fb.StokesSphericalBEM(8, 4, 1e-3)
    K.set_Kfine(19)
    pl = fb.FMM_plan(K, v, p_max=10)
    rp, col, val = o.near_csr()
    for row in (0, o.n // 2, o.n - 1):                  # 3x3 near blocks: three rows of unknowns per panel row
        for a in range(3):
            cols, vals = pl.near_row(3 * row + a)
            ref = val[rp[row]:rp[row + 1], a, :].reshape(-1)
            assert np.array_equal(cols // 3, np.repeat(col[rp[row]:rp[row + 1]], 3))
            assert np.max(np.abs(vals - ref)) / np.abs(ref).max() <= 1e-13
    for x in (np.tile([1.0, 0.0, 0.0], (o.n, 1)), drand48(3 * o.n, seed=9).reshape(o.n, 3)):
        for p in (8, 4, 10):
            K.set_p(p)
            y = pl.execute(x)
            assert y.shape == (o.n, 3)
            assert rel_l2(y, o.matvec(x, p)) <= 1e-12
    K.set_p(8)
    x = np.tile([1.0, 0.0, 0.0], (o.n, 1))
    y8, yo8 = pl.execute(x), o.matvec(x, 8)
    for which in ("M", "L"):
        E, Eo = pl.expansions(which, 8), o.expansions(8, which)
        assert np.max(np.abs(E[:, :4] - Eo[:, :4])) / np.abs(Eo).max() <= 1e-12
    err = rel_l2(y8, o.direct(x))
    assert abs(err - 1.43e-5) / 1.43e-5 < 0.01


@pytest.mark.gpu
def test_gpu_stokes_red_blood_cell(fb, oracle_mod):
    """Config-4 geometry (RedBloodCell), small: N=2048."""
    v = oracle_mod.red_blood_cell(5)
    o = oracle_mod.StokesOracle(v, K=4, K_fine=19, mu=1e-3)
    K = fb.StokesSphericalBEM(8, 4, 1e-3)
    K.set_Kfine(19)
    pl = fb.FMM_plan(K, v)
    x = drand48(3 * o.n, seed=4).reshape(o.n, 3)
    y = pl.execute(x)
    assert rel_l2(y, o.matvec(x, 8)) <= 1e-12
    assert rel_l2(y, o.direct(x)) < 1e-3


@pytest.mark.gpu
def test_gpu_stokes_rejects_traction(fb):
    v = fb.unit_sphere(3)
    with pytest.raises(fb.FmmBemError) as e:
        fb.FMM_plan(fb.StokesSphericalBEM(5, 3), v, bc=np.ones(len(v), dtype=np.uint8))
    assert e.value.status == 6
