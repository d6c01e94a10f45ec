"""BASELINE.json's full-size workload (two disjoint UnitSphere(9), N = 1 048 576, p = 10)
through size-independent properties: the oracle cannot run a whole matvec of this size in test time, so the checks are
linearity, bitwise repeatability, shards summing bitwise to the whole operator, the reference's own FMM-vs-Direct
relation on a sample of rows (tests/scaling.cpp:56-74 idea, Direct from the oracle), and oracle parity of sampled
near-matrix rows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(fb):
    v = np.concatenate([fb.unit_sphere(9, center=(3.0 * i, 0.0, 0.0)) for i in range(2)])
    plan = fb.FMM_plan(fb.LaplaceSphericalBEM(10, 3), v, p_max=10)
    yield v, plan
    plan.close()


def test_counts_and_linearity_and_repeatability(fb, big):
    v, plan = big
    s = plan.stats()
    assert plan.n == 1048576 and s["near_nnz_total"] == 513008056 and s["m2l_pairs"] == 1959503
    rng = np.random.default_rng(11)
    x1, x2 = rng.random(plan.n), rng.standard_normal(plan.n)
    y1, y2 = plan.execute(x1), plan.execute(x2)
    y12 = plan.execute(0.75 * x1 - 2.0 * x2)
    assert np.linalg.norm(y12 - (0.75 * y1 - 2.0 * y2)) <= 1e-13 * np.linalg.norm(y12)
    assert np.array_equal(plan.execute(x1), y1)                                                     # bitwise repeatable
    assert np.all(np.isfinite(y1)) and np.all(y1 > 0)          # positive kernel, positive density


def test_fmm_vs_direct_sample_and_near_rows(fb, oracle_mod, big):
    v, plan = big
    o = oracle_mod.Oracle(v)                                   # tree + lists only; no near matrix is built
    so = o.stats()
    s = plan.stats()
    assert (s["n_boxes"], s["n_leaves"], s["near_nnz_total"], s["m2l_pairs"]) == (so["boxes"], so["leaves"], so["near_nnz"], so["m2l_pairs"])
    np.random.seed(0)
    x = np.random.rand(plan.n)
    y = plan.execute(x)
    for lo in (0, 524288 - 32, 1048576 - 64):                  # rows on both bodies
        d = o.direct(x, rows=(lo, lo + 64))
        assert np.linalg.norm(y[lo:lo + 64] - d) <= 1e-6 * np.linalg.norm(d)       # north-star accuracy gate at p = 10
    # near-matrix rows against the oracle's kernel evaluations (tree order rows; columns = the reference's sorted columns)
    perm = plan.perm()
    for row in (0, 333333, 1048575):
        cols, vals = plan.near_row(row)
        ref = o.kernel_entries(np.full(len(cols), perm[row]), perm[cols])
        assert np.max(np.abs(vals - ref)) <= 1e-13 * np.max(np.abs(ref))


def test_two_shards_sum_bitwise(fb, big):
    v, plan = big
    rng = np.random.default_rng(12)
    x = rng.random(plan.n)
    y = plan.execute(x)
    total = np.zeros_like(y)
    for rank in range(2):
        part = fb.FMM_plan(fb.LaplaceSphericalBEM(10, 3), v, p_max=10, shard=(rank, 2))
        yp = part.execute(x)
        part.close()
        assert np.count_nonzero(yp) < plan.n                    # zero outside the owned rows
        total += yp
    assert np.array_equal(total, y)


def test_matrix_free_near_field_at_full_size(fb, big):
    """sparse_local = 0 at N = 1 048 576: no near matrix (4.1 GB), the near-regime pairs -- a few per cent of the near pairs --
    listed and evaluated once, the far regime recomputed per matvec: the operator of the assembled plan to rounding, linear,
    repeatable bit for bit."""
    v, plan = big
    fo = fb.FMMOptions()
    fo.sparse_local = False
    mf = fb.FMM_plan(fb.LaplaceSphericalBEM(10, 3), v, fo, p_max=10)
    st = mf.stats()
    assert st["near_bytes"] == 0 and 0.02 * st["near_nnz"] < st["near_side_entries"] < 0.08 * st["near_nnz"]
    rng = np.random.default_rng(14)
    x = rng.standard_normal(plan.n)
    y = mf.execute(x)
    assert np.linalg.norm(y - plan.execute(x)) <= 1e-13 * np.linalg.norm(y)
    assert np.array_equal(mf.execute(x), y)
    mf.close()
