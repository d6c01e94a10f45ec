"""Plans of the same panels that differ in the boundary-condition flags share one tree, one set of lists and tables
(fmmbem_plan_create_like; fmmbem_plan_create recognises the geometry of a live plan by itself): the drivers' right-hand-side plan
(examples/LaplaceBEM.cpp:218-232, StokesBEM.cpp:266-270) costs the near-matrix assembly and the P2M table, not a second tree build
and traversal.  The shared plan must give the bits of an independent one, and either plan may be destroyed first."""
import numpy as np
import pytest

from conftest import drand48

pytestmark = pytest.mark.gpu


def test_second_plan_with_flipped_flags_shares_the_geometry_and_gives_the_same_bits(fb, monkeypatch):
    v = np.concatenate([fb.unit_sphere(6), fb.unit_sphere(5, center=(2.5, 0.0, 0.3))])
    n = len(v)
    ones = np.ones(n, dtype=np.uint8)
    mixed = (np.arange(n) % 3 == 0).astype(np.uint8)
    x = drand48(n, seed=5)
    # independent plans (the recognition switched off): the reference results
    monkeypatch.setenv("FMMBEM_PLAN_SHARE", "0")
    ref = {}
    for name, bc in (("pot", None), ("nd", ones), ("mixed", mixed)):
        p = fb.FMM_plan(fb.LaplaceSphericalBEM(10, 3), v, bc=bc, p_max=10)
        assert p.stats()["geometry_shared"] == 1
        ref[name] = (p.execute(x), p.diagonal())
        p.close()
    monkeypatch.delenv("FMMBEM_PLAN_SHARE")
    K = fb.LaplaceSphericalBEM(10, 3)
    base = fb.FMM_plan(K, v, p_max=10)
    assert base.stats()["geometry_shared"] == 1
    rhs = fb.FMM_plan(fb.LaplaceSphericalBEM(10, 3), v, bc=ones, p_max=10)          # recognised: same vertices, same options
    third = base.like(mixed)                                                        # asked for
    assert base.stats()["geometry_shared"] == 3 and rhs.stats()["geometry_shared"] == 3
    assert np.array_equal(base.perm(), rhs.perm())
    for p, name in ((base, "pot"), (rhs, "nd"), (third, "mixed")):
        assert np.array_equal(p.execute(x), ref[name][0]), name
        assert np.array_equal(p.diagonal(), ref[name][1]), name
    # other options, other geometry: not shared
    other = fb.FMM_plan(fb.LaplaceSphericalBEM(8, 3), v, bc=ones, p_max=8)
    assert other.stats()["geometry_shared"] == 1
    moved = fb.FMM_plan(fb.LaplaceSphericalBEM(10, 3), v + 1e-9, bc=ones, p_max=10)
    assert moved.stats()["geometry_shared"] == 1
    other.close()
    moved.close()
    # the base goes first: the shared block lives on with the plans that still point to it
    base.close()
    assert rhs.stats()["geometry_shared"] == 2
    K2 = rhs.kernel()
    for p_ in (10, 4):
        K2.set_p(p_)
        y = rhs.execute(x)
        assert np.all(np.isfinite(y))
    K2.set_p(10)
    assert np.array_equal(rhs.execute(x), ref["nd"][0])
    fourth = fb.FMM_plan(fb.LaplaceSphericalBEM(10, 3), v, p_max=10)                 # recognised through a plan that shares it
    assert fourth.stats()["geometry_shared"] == 3
    assert np.array_equal(fourth.execute(x), ref["pot"][0])


def test_stokes_traction_plan_shares_the_velocity_plans_geometry(fb, monkeypatch):
    """StokesBEM.cpp:266-270: the right-hand side comes from a plan whose targets are TRACTION (11 expansion slots per box instead
    of 8, three gradient records per panel): a different set of flag-dependent arrays on the same tree."""
    v = fb.unit_sphere(5)
    n = len(v)
    K = fb.StokesSphericalBEM(7, 4, 1e-3)
    K.set_Kfine(19)
    x = drand48(3 * n, seed=9).reshape(n, 3)
    monkeypatch.setenv("FMMBEM_PLAN_SHARE", "0")
    ref_v = fb.FMM_plan(K, v).execute(x)
    ref_t = fb.FMM_plan(K, v, bc=np.ones(n, dtype=np.uint8)).execute(x)
    monkeypatch.delenv("FMMBEM_PLAN_SHARE")
    fo = fb.FMMOptions()
    fo.near_stream_fraction = 0.5                                                   # hybrid plans share their leaf choice and items too
    for opts in (None, fo):
        vel = fb.FMM_plan(K, v, opts)
        trac = fb.FMM_plan(K, v, opts, bc=np.ones(n, dtype=np.uint8))
        assert vel.stats()["geometry_shared"] == 2 and trac.stats()["expansion_slots"] == 11 and vel.stats()["expansion_slots"] == 8
        yv, yt = vel.execute(x), trac.execute(x)
        if opts is None:
            assert np.array_equal(yv, ref_v) and np.array_equal(yt, ref_t)
        else:
            assert np.linalg.norm(yv - ref_v) <= 1e-13 * np.linalg.norm(ref_v) and np.linalg.norm(yt - ref_t) <= 1e-13 * np.linalg.norm(ref_t)
        vel.close()
        trac.close()


def test_shards_share_per_shard(fb):
    v = fb.unit_sphere(6)
    n = len(v)
    x = drand48(n, seed=2)
    K = fb.LaplaceSphericalBEM(8, 3)
    whole = fb.FMM_plan(K, v, p_max=8).execute(x)
    ones = np.ones(n, dtype=np.uint8)
    whole_nd = fb.FMM_plan(K, v, bc=ones, p_max=8).execute(x)
    tot, tot_nd = np.zeros(n), np.zeros(n)
    for r in range(2):
        a = fb.FMM_plan(K, v, p_max=8, shard=(r, 2))
        b = fb.FMM_plan(K, v, bc=ones, p_max=8, shard=(r, 2))
        assert b.stats()["geometry_shared"] == 2
        tot += a.execute(x)
        tot_nd += b.execute(x)
    assert np.array_equal(tot, whole) and np.array_equal(tot_nd, whole_nd)
