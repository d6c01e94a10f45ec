"""Full-VECTOR oracle parity at BASELINE.json's full sizes: every element of the GPU result against the oracle's FMM result
on the same workload -- the far field at N = 1 048 576 (8 tree levels, the long-item M2L cut, the >= 2 048-box level rule of
the tree passes, multi-chunk near leaves) compared with the CPU restatement, not only through size-independent properties.

  config 3   two disjoint UnitSphere(9), N = 1 048 576, p = 10
  config 5   the same operator at the two extreme orders of the relaxed solve, p = 12 and p = 1
  config 4   StokesSphericalBEM on RedBloodCell(9), N = 524 288 panels, p = 8 (when the host can hold the oracle's 21 GB CSR)

and the north-star accuracy gate the way the reference forms its error (tests/scaling.cpp:56-74: over the bodies; here a
seeded sample of 4 096 rows drawn over the WHOLE vector, include/Direct.hpp:99-125 for the sum): GPU-vs-Direct and
oracle-vs-Direct on the SAME rows, so that a figure above 1e-6 is attributed -- the reference algorithm's truncation error
at that order on that mesh, or the device path."""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu

GATE_ROWS, GATE_SEED = 4096, 4096


def gate_rows(n, nrows=GATE_ROWS):
    return np.sort(np.random.default_rng(GATE_SEED).choice(n, size=nrows, replace=False)).astype(np.int32)


def mem_available_gb():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                return float(ln.split()[1]) / 1048576.0
    except Exception:
        pass
    return 0.0


@pytest.fixture(scope="module")
def cfg3(fb, oracle_mod):
    v = np.concatenate([fb.unit_sphere(9, center=(3.0 * i, 0.0, 0.0)) for i in range(2)])
    K = fb.LaplaceSphericalBEM(12, 3)
    plan = fb.FMM_plan(K, v, p_max=12)
    o = oracle_mod.Oracle(v)
    o.build_near()                                             # 513 M entries, 6.2 GB of CSR on the host
    x = np.random.default_rng(1234).random(len(v))             # bench.py's charge vector
    yield v, K, plan, o, x
    o.close()
    plan.close()


@pytest.mark.parametrize("p", [10, 12, 1])
def test_config3_and_config5_orders_full_vector_vs_oracle(fb, cfg3, p):
    v, K, plan, o, x = cfg3
    K.set_p(p)
    y = plan.execute(x)
    yo = o.matvec(x, p)                                        # `tuned` mode: every stage threaded, one expansion
    assert rel_l2(y, yo) <= 1e-12, (p, rel_l2(y, yo))
    assert np.max(np.abs(y - yo)) <= 1e-11 * np.max(np.abs(yo))          # no single row off either


def leaf_level_of_rows(plan, rows):
    """Tree level of the leaf that holds each (original-order) panel of `rows`."""
    B = plan.boxes()
    level, is_leaf, bb = B["level"], B["leaf"], B["bb"]
    inv = np.empty(plan.n, dtype=np.int64)
    inv[plan.perm()] = np.arange(plan.n)
    leaves = np.flatnonzero(is_leaf)
    order = leaves[np.argsort(bb[leaves])]
    return level[order[np.searchsorted(bb[order], inv[rows], side="right") - 1]]


def test_config3_accuracy_gate_4096_seeded_rows(fb, cfg3):
    """GPU and oracle against the Direct sum on the same 4 096 rows drawn over the whole vector.  The GPU's error IS the
    oracle's -- the reference algorithm's -- error, to three digits.  Over the whole vector that error is 9e-6 at p = 10, not
    the 1.65e-7 a block of 256 neighbouring rows showed in round 3: the median row sits at 3e-7, and four fifths of the squared
    error come from the one per cent of the rows that live in COARSE leaves.  `leaf iff count <= ncrit` (Octree.hpp:641)
    makes a big box that clips a small cap of a sphere a leaf, DefaultMAC (FMMOptions.hpp:21-31) measures it by its
    half side, and its panels sit in a corner: the M2L into it converges like 0.87^p, not 0.5^p.  Same tree, same lists,
    same numbers in the reference (the oracle restates both rules), so the gate the bench prints is "reference level"."""
    v, K, plan, o, x = cfg3
    rows = gate_rows(len(v))
    d = o.direct_rows(x, rows)
    K.set_p(10)
    y = plan.execute(x)
    yo = o.matvec(x, 10)
    g, r = rel_l2(y[rows], d), rel_l2(yo[rows], d)
    assert abs(g - r) <= 1e-3 * r, (g, r)                      # the same error to three digits
    assert g < 2e-5, g                                         # the reference's level over the whole vector (measured 8.97e-6)
    e = np.abs(y[rows] - d)
    assert np.median(e / np.abs(d)) < 1e-6                     # the typical row IS below the north-star figure
    lv = leaf_level_of_rows(plan, rows)
    fine = lv >= lv.max() - 1                                  # the two deepest leaf levels: 95 % of the panels
    assert fine.mean() > 0.9
    assert np.linalg.norm(e[fine]) / np.linalg.norm(d[fine]) < 2e-6, "rows in fine leaves"
    assert (e[~fine] ** 2).sum() > 0.5 * (e ** 2).sum()        # the coarse leaves carry most of the squared error
    K.set_p(12)
    g12 = rel_l2(plan.execute(x)[rows], d)
    assert g12 < 0.6 * g, (g12, g)                             # and it falls with p


def test_single_sphere_whole_vector_meets_the_gate(fb, oracle_mod):
    """One UnitSphere(9) (N = 524 288) at p = 10 over the seeded whole-vector sample: 5.7e-7 of Direct, GPU and oracle alike
    (profiles/r04b_accuracy_table.jsonl; r = 8 over ALL rows 5.8e-7, r = 10 6.8e-7) -- below the 1e-6 north-star figure.  The
    2.5e-6 that rounds 2 and 3 recorded for these meshes was the 256 contiguous rows at n/3, which happen to sit in a coarse
    leaf; the two-sphere headline workload has such leaves on level 3 and is dominated by them (8.97e-6, see the test above)."""
    v = fb.unit_sphere(9)
    plan = fb.FMM_plan(fb.LaplaceSphericalBEM(10, 3), v, p_max=10)
    o = oracle_mod.Oracle(v)
    x = np.random.default_rng(1234).random(len(v))
    y = plan.execute(x)
    yo = o.matvec(x, 10)
    rows = gate_rows(len(v))
    d = o.direct_rows(x, rows)
    o.close()
    plan.close()
    assert rel_l2(y, yo) <= 1e-12
    g, r = rel_l2(y[rows], d), rel_l2(yo[rows], d)
    assert abs(g - r) <= 1e-3 * r, (g, r)
    assert g < 1e-6, g


def test_config4_stokes_full_vector_vs_oracle(fb, oracle_mod):
    need = 40.0                                                # the oracle's Mat3 CSR: 265 M x 76 B = 20 GB, plus the plan's host copy
    if mem_available_gb() < need:
        pytest.skip("host has %.0f GB available; the Stokes oracle at N = 524 288 needs %.0f" % (mem_available_gb(), need))
    v = fb.red_blood_cell(9)
    K = fb.StokesSphericalBEM(8, 4, 1e-3)
    K.set_Kfine(19)
    plan = fb.FMM_plan(K, v, p_max=8)
    o = oracle_mod.StokesOracle(v, K=4, K_fine=19, mu=1e-3)
    x = np.random.default_rng(1234).random(3 * len(v)).reshape(-1, 3)
    y = plan.execute(x)
    plan.close()
    # the same operator with 40 % of the near pairs recomputed instead of stored (near_stream_fraction = 0.6, the measured optimum)
    fo = fb.FMMOptions()
    fo.near_stream_fraction = 0.6
    hyb = fb.FMM_plan(K, v, fo, p_max=8)
    st = hyb.stats()
    assert 0.35 < st["near_recomputed_pairs"] / st["near_nnz"] < 0.45 and st["near_bytes"] < 0.65 * 48 * st["near_nnz"]
    yh = hyb.execute(x)
    hyb.close()
    yo = o.matvec(x, 8)
    rows = gate_rows(len(v), 512)
    d = o.direct_rows(x, rows)
    o.close()
    assert rel_l2(y, yo) <= 1e-12, rel_l2(y, yo)
    assert rel_l2(yh, yo) <= 1e-12, rel_l2(yh, yo)
    g, r = rel_l2(y[rows], d), rel_l2(yo[rows], d)
    assert abs(g - r) <= 1e-3 * r, (g, r)
    assert g < 5e-5, g                                         # the reference's level at p = 8 is 1.4e-5


@pytest.mark.gpu
@pytest.mark.parametrize("r", [7, 9])
def test_unit_density_on_two_spheres_reproduces_the_analytic_potential(fb, r):
    """A check that needs no oracle and holds at any size (profiles/r05t: run at 16.8 M panels): sigma = 1 on two unit spheres.  A
    uniformly charged sphere is a point charge at its centre for points outside it and 4 pi R on itself, so the single-layer
    potential at a collocation point is 4 pi (1 + 1 / d), d = its distance to the OTHER sphere's centre -- up to the O(h^2) of flat
    panels inscribed in the sphere and the FMM's own error.  The deviation must fall with the discretisation."""
    c2 = np.array([2.5, 0.0, 0.0])
    v = np.concatenate([fb.unit_sphere(r), fb.unit_sphere(r, center=tuple(c2))])
    n = len(v)
    plan = fb.FMM_plan(fb.LaplaceSphericalBEM(10, 3), v, fb.FMMOptions())
    y = plan.execute(np.ones(n))
    cen = v.mean(axis=1)
    other = np.where(np.arange(n) < n // 2, 1, 0)[:, None] * c2
    want = 4 * np.pi * (1.0 + 1.0 / np.linalg.norm(cen - other, axis=1))
    rel = np.abs(y - want) / want
    h2 = 4 * np.pi / (n / 2)                                            # a panel's area on the unit sphere
    assert np.median(rel) < 0.5 * h2 and rel.max() < 2e-3, (np.median(rel), rel.max(), h2)
