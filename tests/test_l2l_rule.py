"""The reference's lazy L2L rule on an adaptive tree (executor/EvalInteractionLazySparse.hpp:199-237).

resolve_LR_interactions marks a box the first time it is an M2L target and propagate_local then queues parent->child
only for children that are not marked yet.  A box that was an M2L target EARLIER in the traversal than one of its
ancestors never receives that ancestor's local expansion.  The reference's spheres never produce such an edge (the
known-answer trees have l2l_skipped == 0); a clustered soup does.  The library's default applies every edge;
FMMOptions.reference_l2l reproduces the reference's list."""
import numpy as np
import pytest

from test_random_meshes import _coverage, _soup


def _case():
    return _soup(0, 158, 1, 1.0, 1.0), 16          # panels, ncrit: found by the randomised Stokes parity test


def _opts(fb, ncrit, ref):
    o = fb.FMMOptions()
    o.set_max_per_box(ncrit)
    o.reference_l2l = ref
    return o


def test_reference_rule_omits_edges_on_this_tree(fb, oracle_mod):
    v, ncrit = _case()
    so = oracle_mod.Oracle(v, ncrit=ncrit).stats()
    assert so["l2l_skipped"] > 0
    full = fb.FMM_plan(fb.LaplaceSphericalBEM(5, 3), v, _opts(fb, ncrit, False), host_only=True).stats()
    ref = fb.FMM_plan(fb.LaplaceSphericalBEM(5, 3), v, _opts(fb, ncrit, True), host_only=True).stats()
    assert full["l2l_reference_omitted"] == ref["l2l_reference_omitted"] == so["l2l_skipped"]
    assert ref["l2l_ops"] == so["l2l_ops"] and full["l2l_ops"] == so["l2l_ops"] + so["l2l_skipped"]
    assert oracle_mod.Oracle(v, ncrit=ncrit, complete_l2l=True).stats()["l2l_ops"] == full["l2l_ops"]


def test_oracle_complete_rule_converges_reference_rule_does_not(oracle_mod):
    v, ncrit = _case()
    x = np.random.default_rng(1).standard_normal(len(v))
    ref, full = oracle_mod.Oracle(v, ncrit=ncrit), oracle_mod.Oracle(v, ncrit=ncrit, complete_l2l=True)
    yd = ref.direct(x)
    err = lambda y: np.linalg.norm(y - yd) / np.linalg.norm(yd)
    e_ref = [err(ref.matvec(x, p)) for p in (4, 8, 12)]
    e_full = [err(full.matvec(x, p)) for p in (4, 8, 12)]
    assert e_full[2] < 1e-5 and e_full[2] < 0.05 * e_full[0]           # complete list: the error falls with p
    assert e_ref[2] > 20 * e_full[2]                                     # reference list: a floor p does not lower


def test_rules_coincide_on_the_reference_sphere(fb):
    v = fb.unit_sphere(3)
    s = fb.FMM_plan(fb.LaplaceSphericalBEM(5, 3), v, fb.FMMOptions(), host_only=True).stats()
    assert s["l2l_reference_omitted"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("ref", [False, True])
def test_gpu_follows_the_chosen_rule(fb, oracle_mod, ref):
    v, ncrit = _case()
    x = np.random.default_rng(1).standard_normal(len(v))
    o = oracle_mod.Oracle(v, ncrit=ncrit, complete_l2l=not ref)
    for p in (3, 8, 12):
        y = fb.FMM_plan(fb.LaplaceSphericalBEM(p, 3), v, _opts(fb, ncrit, ref)).execute(x)
        yo = o.matvec(x, p)
        assert np.linalg.norm(y - yo) <= 1e-12 * np.linalg.norm(yo)
    yd = o.direct(x)
    e = np.linalg.norm(y - yd) / np.linalg.norm(yd)
    assert (e > 1e-4) if ref else (e < 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("ref", [False, True])
def test_gpu_stokes_follows_the_chosen_rule(fb, oracle_mod, ref):
    v, ncrit = _case()
    f = np.random.default_rng(2).standard_normal((len(v), 3))
    K = fb.StokesSphericalBEM(8, 4, 1e-3)
    u = fb.FMM_plan(K, v, _opts(fb, ncrit, ref)).execute(f)
    uo = oracle_mod.StokesOracle(v, K=4, K_fine=25, mu=1e-3, ncrit=ncrit, complete_l2l=not ref).matvec(f, 8)
    assert np.linalg.norm(u - uo) <= 1e-11 * np.linalg.norm(uo)


def test_every_source_reaches_every_target_exactly_once(fb):
    v, ncrit = _case()
    n = len(v)
    full = _coverage(fb.FMM_plan(fb.LaplaceSphericalBEM(5, 3), v, _opts(fb, ncrit, False), host_only=True), n)
    assert np.all(full == n)
    ref = _coverage(fb.FMM_plan(fb.LaplaceSphericalBEM(5, 3), v, _opts(fb, ncrit, True), host_only=True), n)
    assert np.all(ref <= n) and np.any(ref < n)        # the reference's list loses sources on this tree
    for r in (3, 5):                                   # ... and none on its own spheres
        s = fb.unit_sphere(r)
        for rule in (False, True):
            assert np.all(_coverage(fb.FMM_plan(fb.LaplaceSphericalBEM(5, 3), s, _opts(fb, 64, rule), host_only=True), len(s)) == len(s))
