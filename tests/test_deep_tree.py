"""Octrees deeper than the 10 levels of the reference's 32-bit Morton keys (include/tree/Octree.hpp:82-92): many small
bodies in a large box -- the use case of Triangulation::MultipleRedBloodCell (examples/BEM/Triangulation.hpp:260-321).
The reference cannot build such a tree (its bucket shift wraps at level 10, Octree.hpp:649); the product switches to 64-bit
keys with 21 levels ONLY then, and the oracle follows with the same switch (tree.c DEEP_LEVELS, marked "not a reference
rule"), so that every tree the reference can build keeps its bits (tests/test_capi_host.py) and the deep ones are compared
list by list, then against the Direct sum."""
import numpy as np
import pytest


def cells_in_a_big_box(fb, recursions=4, cells=8, spread=2.0e4):
    """`cells` red blood cells (radius ~4) with fixed orientations, centres spread over a cube of side `spread`."""
    rng = np.random.default_rng(5)
    placement = np.concatenate([rng.random((cells, 3)) * np.pi, rng.random((cells, 3)) * spread], axis=1)
    return fb.red_blood_cells(recursions, cells, placement)


def test_deep_tree_lists_equal_oracle(fb, oracle_mod):
    v = cells_in_a_big_box(fb)
    opts = fb.FMMOptions()
    opts.set_max_per_box(32)
    pl = fb.FMM_plan(fb.LaplaceSphericalBEM(8, 3), v, opts, host_only=True)
    o = oracle_mod.Oracle(v, ncrit=32)
    s, so = pl.stats(), o.stats()
    assert s["n_levels"] > 11 and so["levels"] == s["n_levels"]             # deeper than the reference's coder resolves
    assert (s["n_boxes"], s["n_leaves"], s["near_nnz_total"], s["m2l_pairs"], s["m2m_ops"], s["l2l_ops"], s["p2p_pairs"]) == \
           (so["boxes"], so["leaves"], so["near_nnz"], so["m2l_pairs"], so["m2m_ops"], so["l2l_ops"], so["p2p_pairs"])
    assert np.array_equal(pl.perm(), o.perm())
    assert np.array_equal(pl.pairs("m2l"), o.pairs("m2l")) and np.array_equal(pl.pairs("p2p"), o.pairs("p2p"))
    b, bo = pl.boxes(), o.boxes()
    for k in ("center", "side", "level", "leaf", "bb", "be"):
        assert np.array_equal(b[k], bo[k]), k
    assert b["leaf"][b["level"] > 10].any()                                  # leaves below level 10 exist


def test_ten_levels_still_take_the_reference_coder(fb, oracle_mod):
    """The same cells closer together need 10 levels or fewer: the 32-bit path, whose first ten levels the deep coder
    reproduces (cells are extent / 2^L): the level-<=10 boxes of a deep tree are the boxes of the shallow coder."""
    v = cells_in_a_big_box(fb, spread=300.0)
    pl = fb.FMM_plan(fb.LaplaceSphericalBEM(8, 3), v, host_only=True)
    assert pl.stats()["n_levels"] <= 11
    o = oracle_mod.Oracle(v)
    assert np.array_equal(pl.perm(), o.perm()) and np.array_equal(pl.pairs("m2l"), o.pairs("m2l"))


@pytest.mark.gpu
@pytest.mark.parametrize("p", [4, 10])
def test_deep_tree_matvec_matches_oracle_and_direct(fb, oracle_mod, p):
    v = cells_in_a_big_box(fb)
    opts = fb.FMMOptions()
    opts.set_max_per_box(32)
    plan = fb.FMM_plan(fb.LaplaceSphericalBEM(p, 3), v, opts, p_max=p)
    o = oracle_mod.Oracle(v, ncrit=32, complete_l2l=True)                    # the product's default L2L list
    assert plan.stats()["n_levels"] > 11
    x = np.random.default_rng(6).random(len(v))
    y, yo, d = plan.execute(x), o.matvec(x, p), o.direct(x)
    # 2e-11, not the 1e-12 of the compact meshes: in a box 2e4 leaf sizes wide a box centre is known to 2e4 x 1e-16 of a leaf
    # (the oracle subtracts rounded centres, Octree.hpp:350-355; the product takes translations from exact integer grid
    # vectors): measured 1.6e-12
    assert np.linalg.norm(y - yo) <= 2e-11 * np.linalg.norm(yo)
    # far cells are 1e3 radii apart: their contribution converges at once; the error is the within-cell one (7.8e-4 / 6.8e-6 here;
    # with leaves of 16 coarse panels it would stall at 2e-5 for every p: MAC-accepted boxes then hold panel pairs of the kernel's
    # NEAR regime, where Direct integrates semi-analytically and P2M uses the K-point rule -- the reference's own floor)
    assert np.linalg.norm(y - d) <= {4: 2e-3, 10: 2e-5}[p] * np.linalg.norm(d)
    plan.close()


@pytest.mark.gpu
def test_deep_tree_stokes_and_shards(fb, oracle_mod):
    """Stokes on the deep tree (the paper's application), and two shards of it summing bitwise to the whole."""
    v = cells_in_a_big_box(fb, cells=4)
    K = fb.StokesSphericalBEM(6, 4, 1e-3)
    K.set_Kfine(19)
    opts = fb.FMMOptions()
    opts.set_max_per_box(32)
    plan = fb.FMM_plan(K, v, opts, p_max=6)
    assert plan.stats()["n_levels"] > 11
    o = oracle_mod.StokesOracle(v, K=4, K_fine=19, mu=1e-3, ncrit=32, complete_l2l=True)
    x = np.random.default_rng(7).random((len(v), 3))
    y = plan.execute(x)
    yo = o.matvec(x, 6)
    assert np.linalg.norm(y - yo) <= 2e-11 * np.linalg.norm(yo)          # see above: measured 1.4e-12
    total = np.zeros_like(y)
    for rank in range(2):
        part = fb.FMM_plan(K, v, opts, p_max=6, shard=(rank, 2))
        total += part.execute(x)
        part.close()
    assert np.array_equal(total, y)
    plan.close()


@pytest.mark.parametrize("case", ["two_spheres", "random", "deep"])
def test_threaded_tree_build_equals_the_serial_one(fb, case, monkeypatch):
    """From 32 768 panels up the octree is built by a few threads (one stable sort by Morton code, the boxes read off the sorted
    codes level by level, every leaf's bodies put back in their original order: csrc/host_plan.cpp build_tree); below that, and
    with FMMBEM_TREE_SERIAL=1, by the reference's breadth-first bucketing (include/tree/Octree.hpp:617-692).  Same permutation,
    same boxes, same numbering -- both coders, degenerate inputs included (coincident centroids, one crowded octant)."""
    rng = np.random.default_rng(11)
    ncrit = 64
    if case == "two_spheres":
        v = np.concatenate([fb.unit_sphere(7), fb.unit_sphere(7, center=(2.5, 0.3, 0.1))])
    elif case == "random":
        c = rng.random((40000, 1, 3)) ** 3                                # crowded towards one corner: a ragged, adaptive tree
        v = c + 1e-3 * rng.standard_normal((40000, 3, 3))
        v[1000:1040] = v[1000]                                            # forty coincident panels: one finest cell, within ncrit
    else:
        v = cells_in_a_big_box(fb, recursions=5, cells=40)               # 64-bit keys
        ncrit = 32
    assert len(v) >= 32768
    opts = fb.FMMOptions()
    opts.set_max_per_box(ncrit)
    K = fb.LaplaceSphericalBEM(4, 3)
    a = fb.FMM_plan(K, v, opts, host_only=True)
    monkeypatch.setenv("FMMBEM_TREE_SERIAL", "1")
    b = fb.FMM_plan(K, v, opts, host_only=True)
    if case == "deep":
        assert a.stats()["n_levels"] > 11
    assert np.array_equal(a.perm(), b.perm())
    ba, bb = a.boxes(), b.boxes()
    for k in ba:
        assert np.array_equal(ba[k], bb[k]), k
    assert np.array_equal(a.pairs("m2l"), b.pairs("m2l")) and np.array_equal(a.pairs("p2p"), b.pairs("p2p"))
