"""The translation operators one at a time (fmmbem_ops_*, the KernelSkeleton contract: kernel/KernelSkeleton.hpp:62-212) against
the oracle's single operators (oracle/expansions.c orc_m2m / orc_m2l / orc_l2l / orc_p2m_panel / orc_l2p_panel, oracle/stokes.c
stokes_p2m_panel / stokes_l2p_panel -- kernel/LaplaceSpherical.hpp:245-411, LaplaceSphericalBEM.hpp:307-476,
StokesSphericalBEM.hpp:391-432, 512-522 restated), and end to end against Kernel::operator() the way the reference's
tests/single_level.cpp chains them.  Tolerance: 1e-12 of the largest coefficient (the device kernels factor the shifts into
rotations and an axial translation, csrc/m2l_rot.hpp; the sums round differently from the reference's double loops)."""
import numpy as np
import pytest

from conftest import ROOT  # noqa: F401

pytestmark = pytest.mark.gpu

TRANSLATIONS = [(0.75, 0.0, 0.0), (0.3, -0.41, 0.7), (0.0, 0.0, 1.5), (0.0, 0.0, -0.9), (0.0, 0.6, 0.2), (-1.1, 1.3, -0.7)]


def rand_expansion(rng, slots, p):
    """coefficients n (n + 1) / 2 + m of a real field: the m = 0 ones are real"""
    S = p * (p + 1) // 2
    E = rng.standard_normal((slots, S)) + 1j * rng.standard_normal((slots, S))
    for n in range(p):
        E[:, n * (n + 1) // 2] = E[:, n * (n + 1) // 2].real
    return np.ascontiguousarray(E)


def small_panels(rng, n, center, radius):
    """n small triangles scattered in a ball of the given radius about center"""
    c = np.asarray(center) + radius * 0.6 * rng.uniform(-1, 1, (n, 1, 3))
    return np.ascontiguousarray(c + 0.15 * radius * rng.uniform(-1, 1, (n, 3, 3)))


def close(a, b, tol=1e-12):
    a, b = np.asarray(a), np.asarray(b)
    scale = np.abs(b).max()
    assert scale > 0
    assert np.abs(a - b).max() <= tol * scale, (np.abs(a - b).max() / scale)


# generic: the sparse shift operators and the double-sum M2L (the kernels of the orders above 12) at an order the rotation kernels take
@pytest.mark.parametrize("p,generic", [(p, False) for p in (1, 2, 3, 5, 8, 10, 12, 13, 16)] + [(p, True) for p in (1, 2, 5, 10, 12)])
def test_shifts_match_the_oracle(p, generic, monkeypatch):
    import fmm_bem_relaxed_amd as fb
    from oracle import oracle
    if generic:
        monkeypatch.setenv("FMMBEM_OPS_GENERIC", "1")
    rng = np.random.default_rng(100 + p)
    K = fb.LaplaceSphericalBEM(p, 3)
    T = oracle.Tables(p)
    for tr in TRANSLATIONS:
        src = rand_expansion(rng, 2, p)
        for name, dev, ref in (("M2M", K.M2M, T.m2m), ("M2L", K.M2L, T.m2l), ("L2L", K.L2L, T.l2l)):
            before = rand_expansion(rng, 2, p)
            tgt = before.copy()
            dev(src, tgt, tr)                                        # target += Op(source)
            want = np.stack([ref(src[s], tr) for s in range(2)])
            close(tgt - before, want)


def test_shifts_take_any_number_of_slots():
    import fmm_bem_relaxed_amd as fb
    from oracle import oracle
    rng = np.random.default_rng(7)
    p = 8
    K = fb.StokesSphericalBEM(p, 3, 1e-3)
    T = oracle.Tables(p)
    for slots in (1, 4, 8, 12):                                      # 8: both groups of a Stokes multipole_type (M[2][4])
        src = rand_expansion(rng, slots, p)
        for dev, ref in ((K.M2M, T.m2m), (K.M2L, T.m2l), (K.L2L, T.l2l)):
            tgt = np.zeros_like(src)
            dev(src, tgt, (0.4, 0.9, -0.3))
            close(tgt, np.stack([ref(src[s], (0.4, 0.9, -0.3)) for s in range(slots)]))


@pytest.mark.parametrize("p", [1, 4, 10, 12, 14])
@pytest.mark.parametrize("n", [1, 7, 150])
def test_laplace_p2m_and_l2p_match_the_oracle(p, n):
    import fmm_bem_relaxed_amd as fb
    from oracle import oracle
    rng = np.random.default_rng(p * 1000 + n)
    K = fb.LaplaceSphericalBEM(p, 3)
    center = np.array([0.2, -0.1, 0.4])
    panels = small_panels(rng, n, center, 0.5)
    bc = (rng.uniform(size=n) < 0.4).astype(np.uint8)
    q = rng.standard_normal(n)
    M0 = rand_expansion(rng, 2, p)
    M = M0.copy()
    K.P2M(panels, q, center, M, bc=bc)
    close(M - M0, oracle.single_p2m(0, p, 3, panels, bc, q, center))
    # L2P: r0 added at POTENTIAL targets, r1 subtracted at NORMAL_DERIV targets
    L = rand_expansion(rng, 2, p)
    r0 = rng.standard_normal(n)
    r = r0.copy()
    K.L2P(L, center, panels, r, bc=bc)
    close(r - r0, oracle.single_l2p(0, p, 3, L, center, panels, bc))


@pytest.mark.parametrize("k", [1, 4, 7])
def test_laplace_p2m_other_quadrature_rules(k):
    import fmm_bem_relaxed_amd as fb
    from oracle import oracle
    rng = np.random.default_rng(k)
    K = fb.LaplaceSphericalBEM(6, k)
    panels = small_panels(rng, 20, (0, 0, 0), 1.0)
    bc = (rng.uniform(size=20) < 0.5).astype(np.uint8)
    q = rng.standard_normal(20)
    M = K.init_multipole()
    K.P2M(panels, q, (0.0, 0.0, 0.0), M, bc=bc)
    close(M, oracle.single_p2m(0, 6, k, panels, bc, q, (0.0, 0.0, 0.0)))


@pytest.mark.parametrize("p", [2, 8, 12, 15])
def test_stokes_p2m_and_l2p_match_the_oracle(p):
    import fmm_bem_relaxed_amd as fb
    from oracle import oracle
    rng = np.random.default_rng(p)
    mu = 0.7
    K = fb.StokesSphericalBEM(p, 3, mu)
    center = np.array([-0.3, 0.5, 0.1])
    n = 90
    panels = small_panels(rng, n, center, 0.4)
    f = rng.standard_normal((n, 3))
    M = K.init_multipole()
    assert M.shape == (4, p * (p + 1) // 2)
    K.P2M(panels, f, center, M)
    close(M, oracle.single_p2m(1, p, 3, panels, None, f, center))
    L = rand_expansion(rng, 4, p)
    r = np.zeros((n, 3))
    K.L2P(L, center, panels, r)
    close(r.reshape(-1), oracle.single_l2p(1, p, 3, L, center, panels, None, mu=mu))


def test_single_level_chain_reproduces_the_kernel():
    """tests/single_level.cpp's chain with the BEM kernel: P2M about a child centre, M2M to the parent, M2L across, L2L down to a
    child, L2P -- against K(target, source) * charge (Kernel::operator(), the Direct sum of include/Direct.hpp)"""
    import fmm_bem_relaxed_amd as fb
    rng = np.random.default_rng(3)
    for cls, dof in ((fb.LaplaceSphericalBEM, 1), (fb.StokesSphericalBEM, 3)):
        K = cls(12, 3)
        c_src_child, c_src = np.array([0.125, 0.125, 0.125]), np.array([0.25, 0.25, 0.25])
        c_tgt, c_tgt_child = np.array([2.25, 0.25, 0.25]), np.array([2.125, 0.375, 0.125])
        src = small_panels(rng, 5, c_src_child, 0.1)
        tgt = small_panels(rng, 6, c_tgt_child, 0.1)
        q = rng.standard_normal((5, dof)) if dof == 3 else rng.standard_normal(5)
        M1, M2, L1, L2 = K.init_multipole(), K.init_multipole(), K.init_local(), K.init_local()
        K.P2M(src, q, c_src_child, M1)
        K.M2M(M1, M2, c_src - c_src_child)
        K.M2L(M2, L1, c_tgt - c_src)
        K.L2L(L1, L2, c_tgt_child - c_tgt)
        r = np.zeros((6, dof)) if dof == 3 else np.zeros(6)
        K.L2P(L2, c_tgt_child, tgt, r)
        want = np.zeros_like(r)
        for i in range(6):
            e = fb.kernel_entries(K, np.repeat(tgt[i:i + 1], 5, axis=0), src)      # K(t_i, s_j), j = 0..4
            want[i] = np.einsum("jab,jb->a", e, q) if dof == 3 else e @ q
        assert np.abs(r - want).max() <= 2e-6 * np.abs(want).max()     # the expansions' truncation at p = 12, |d| / R ~ 10
        # and M2L straight from the child's multipole to the child's local expansion gives the same field
        L3 = K.init_local()
        K.M2L(M1, L3, c_tgt_child - c_src_child)
        r2 = np.zeros_like(r)
        K.L2P(L3, c_tgt_child, tgt, r2)
        assert np.abs(r2 - want).max() <= 2e-6 * np.abs(want).max()


def test_errors():
    import fmm_bem_relaxed_amd as fb
    from fmm_bem_relaxed_amd import _capi
    K = fb.StokesSphericalBEM(6, 3)
    rng = np.random.default_rng(0)
    panels = small_panels(rng, 3, (0, 0, 0), 1.0)
    with pytest.raises(_capi.FmmBemError) as e:                       # TRACTION sources: the reference's stresslet moments are not built
        K.P2M(panels, np.ones((3, 3)), (0.0, 0.0, 0.0), K.init_multipole(), bc=np.array([0, 1, 0], dtype=np.uint8))
    assert e.value.status == _capi.ERR_UNSUPPORTED
    with pytest.raises(_capi.FmmBemError):
        K.L2P(K.init_local(), (0.0, 0.0, 0.0), panels, np.zeros((3, 3)), bc=np.array([1, 1, 1], dtype=np.uint8))
    with pytest.raises(ValueError):
        K.M2M(K.init_multipole(), K.init_multipole(slots=3), (1.0, 0.0, 0.0))
    with pytest.raises(ValueError):                                   # an expansion of another order
        K.M2L(np.zeros((4, 10), dtype=np.complex128), np.zeros((4, 10), dtype=np.complex128), (1.0, 0.0, 0.0))
    K.set_p(4)                                                        # set_p changes what the operators expect
    M = K.init_multipole()
    assert M.shape == (4, 10)
    K.M2M(M, M.copy(), (1.0, 0.0, 0.0))
