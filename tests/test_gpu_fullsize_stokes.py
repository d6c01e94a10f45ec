"""BASELINE.json config 4 at full size: StokesSphericalBEM, velocity boundary condition, on
Triangulation::RedBloodCell(9) -- N = 524 288 panels, 1 572 864 unknowns, p = 8, K = 4, K_fine = 19, mu = 1e-3
(examples/StokesBEM.cpp:111-141, 216-218).  The oracle cannot run a matvec of this size in test time, so the checks are
size-independent: list statistics against the oracle's tree, linearity, bitwise repeatability, the reference's own
FMM-vs-Direct relation on sampled rows (Direct from the oracle), oracle parity of sampled near rows (the 512 KB work-item
split and the multi-chunk leaves of near_spmv_sym3 only occur at this size), shards summing bitwise."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = 8


def _kernel(fb):
    K = fb.StokesSphericalBEM(P, 4, 1e-3)
    K.set_Kfine(19)
    return K


@pytest.fixture(scope="module")
def rbc(fb):
    v = fb.red_blood_cell(9)
    plan = fb.FMM_plan(_kernel(fb), v, p_max=P)
    yield v, plan
    plan.close()


def test_counts_linearity_repeatability(fb, oracle_mod, rbc):
    v, plan = rbc
    assert plan.n == 524288 and plan.dof == 3
    o = oracle_mod.StokesOracle(v, K=4, K_fine=19, mu=1e-3)            # tree + lists; no near matrix is built
    so, s = o.stats(), plan.stats()
    assert (s["n_boxes"], s["n_leaves"], s["near_nnz_total"], s["m2l_pairs"]) == (so["boxes"], so["leaves"], so["near_nnz"], so["m2l_pairs"])
    assert s["near_bytes"] == 48 * s["near_nnz_total"]                 # symmetric 6-value blocks: 48 B per panel pair
    o.close()
    rng = np.random.default_rng(21)
    x1, x2 = rng.random((plan.n, 3)), rng.standard_normal((plan.n, 3))
    y1, y2 = plan.execute(x1), plan.execute(x2)
    y12 = plan.execute(0.75 * x1 - 2.0 * x2)
    assert np.linalg.norm(y12 - (0.75 * y1 - 2.0 * y2)) <= 1e-13 * np.linalg.norm(y12)
    assert np.array_equal(plan.execute(x1), y1)                         # bitwise repeatable
    assert np.all(np.isfinite(y1))


def test_fmm_vs_direct_sample_and_near_rows(fb, oracle_mod, rbc):
    v, plan = rbc
    o = oracle_mod.StokesOracle(v, K=4, K_fine=19, mu=1e-3)
    x = np.tile([1.0, 0.0, 0.0], (plan.n, 1))                           # examples/StokesBEM.cpp:257
    rng = np.random.default_rng(22)
    xr = rng.random((plan.n, 3))
    for xs in (x, xr):
        y = plan.execute(xs)
        for lo in (0, 262144 - 32, 524288 - 64):
            d = o.direct(xs, rows=(lo, lo + 64))
            # the reference's error level at p = 8 on its N = 2 048 sphere is 1.4e-5
            assert np.linalg.norm(y[lo:lo + 64] - d) <= 5e-5 * np.linalg.norm(d)
    perm = plan.perm()
    for row in (0, 200001, 524287):                                     # tree-order panel rows, all three components
        for a in range(3):
            cols, vals = plan.near_row(3 * row + a)
            pc = cols[::3] // 3
            ref = o.kernel_entries(np.full(len(pc), perm[row]), perm[pc])[:, a, :].reshape(-1)
            assert np.max(np.abs(vals - ref)) <= 1e-13 * np.max(np.abs(ref))
    o.close()


def test_two_shards_sum_bitwise(fb, rbc):
    v, plan = rbc
    rng = np.random.default_rng(23)
    x = rng.random((plan.n, 3))
    y = plan.execute(x)
    total = np.zeros_like(y)
    for rank in range(2):
        part = fb.FMM_plan(_kernel(fb), v, p_max=P, shard=(rank, 2))
        yp = part.execute(x)
        part.close()
        assert np.count_nonzero(yp) < 3 * plan.n
        total += yp
    assert np.array_equal(total, y)


def test_traction_targets_at_full_size_against_direct(fb, oracle_mod):
    """Config 4 read literally -- "(stresslet)": every target TRACTION, the double-layer far field of kernels_far.hip at
    N = 524 288, p = 8, against the Direct sum over the reference's traction entries on sampled rows (the answer this
    operator has: test_stokes.py); linear, and bitwise repeatable."""
    v = fb.red_blood_cell(9)
    bc = np.ones(len(v), dtype=np.uint8)
    plan = fb.FMM_plan(_kernel(fb), v, p_max=P, bc=bc)
    o = oracle_mod.StokesOracle(v, K=4, K_fine=19, mu=1e-3, bc=bc)
    rng = np.random.default_rng(31)
    x1, x2 = rng.random((plan.n, 3)), rng.standard_normal((plan.n, 3))
    y1, y2 = plan.execute(x1), plan.execute(x2)
    for lo in (0, 262144 - 32, 524288 - 64):
        d = o.direct(x1, rows=(lo, lo + 64))
        assert np.linalg.norm(y1[lo:lo + 64] - d) <= 1e-3 * np.linalg.norm(d)      # 5e-4 at p = 8 on the small meshes of test_stokes.py
    y12 = plan.execute(0.5 * x1 + 3.0 * x2)
    assert np.linalg.norm(y12 - (0.5 * y1 + 3.0 * y2)) <= 1e-13 * np.linalg.norm(y12)
    assert np.array_equal(plan.execute(x1), y1)
    o.close()
    plan.close()
