"""Hybrid near field (fmmbem_options.near_stream_fraction < 1): the target leaves with the most rows keep no matrix; their
far-regime entries are recomputed every matvec (near_recompute3_kernel, beside the streaming kernel), their near-regime pairs are
listed and evaluated once.  Same operator as the assembled matrix (EvalP2P.hpp:47-98 + Matvec.hpp:14-33) and as the matrix-free
evaluator (EvalInteractionLazy.hpp:239-252): gated against the oracle at 1e-12 like every other path, against the fully streamed
plan at 1e-13, bitwise across shards and across repeated runs."""
import numpy as np
import pytest

from conftest import drand48, rel_l2

pytestmark = pytest.mark.gpu


def _stokes(fb, p=8, k=4, kfine=19):
    K = fb.StokesSphericalBEM(p, k, 1e-3)
    K.set_Kfine(kfine)
    return K


def _opts(fb, f):
    o = fb.FMMOptions()
    o.near_stream_fraction = f
    return o


@pytest.mark.parametrize("f", [0.7, 0.35, 0.0])
def test_hybrid_stokes_equals_oracle_and_the_streamed_plan(fb, oracle_mod, f):
    v = oracle_mod.red_blood_cell(5)
    o = oracle_mod.StokesOracle(v, K=4, K_fine=19, mu=1e-3)
    K = _stokes(fb)
    full = fb.FMM_plan(K, v)
    hyb = fb.FMM_plan(K, v, _opts(fb, f))
    sf, sh = full.stats(), hyb.stats()
    pairs = sf["near_nnz"]                                                             # panel pairs (a pair is a 3 x 3 block)
    assert sf["near_recomputed_pairs"] == 0
    assert abs(sh["near_recomputed_pairs"] / pairs - (1.0 - f)) < 0.05                 # the share asked for, to a leaf
    assert abs(sh["near_bytes"] / sf["near_bytes"] - f) < 0.05 and sh["near_nnz"] == sf["near_nnz"]
    assert 0 < sh["near_side_entries"] < 0.2 * sh["near_recomputed_pairs"]
    for seed, p in ((4, 8), (5, 3)):
        K.set_p(p)
        x = drand48(3 * o.n, seed=seed).reshape(o.n, 3)
        y = hyb.execute(x)
        assert rel_l2(y, o.matvec(x, p)) <= 1e-12
        assert rel_l2(y, full.execute(x)) <= 1e-13
        assert np.array_equal(y, hyb.execute(x))                                       # fixed summation order: same bits every run
    # introspection of a recomputed leaf: the row is evaluated on the fly by the assembly's entry functions, the self entry comes
    # from the list -- the streamed plan's numbers (which went through the symmetric 6-value storage: (a, b) and (b, a) of a block are
    # ONE stored value there, two evaluated ones here, equal to rounding)
    assert np.max(np.abs(hyb.diagonal() - full.diagonal())) <= 4e-16 * np.max(np.abs(full.diagonal()))
    for row in (0, 3 * (o.n // 2) + 1, 3 * o.n - 1):
        c1, v1 = full.near_row(row)
        c2, v2 = hyb.near_row(row)
        assert np.array_equal(c1, c2) and np.max(np.abs(v1 - v2)) <= 4e-16 * np.max(np.abs(v1))


def test_hybrid_stokes_traction_and_mixed_targets(fb, oracle_mod):
    v = np.concatenate([oracle_mod.unit_sphere(4), oracle_mod.unit_sphere(4, center=(2.4, 0.0, 0.3))])
    n = len(v)
    bc = (np.arange(n) % 3 != 0).astype(np.uint8)                                      # mostly TRACTION targets, some velocity
    x = drand48(3 * n, seed=33).reshape(n, 3)
    K = _stokes(fb, p=7)
    y = fb.FMM_plan(K, v, bc=bc).execute(x)
    for f in (0.5, 0.0):
        h = fb.FMM_plan(K, v, _opts(fb, f), bc=bc)
        assert h.stats()["near_recomputed_pairs"] > 0
        assert rel_l2(h.execute(x), y) <= 1e-13
    # shards of a hybrid plan: every shard reaches the same verdict for a leaf (the choice is made on the whole tree), so the shards'
    # rows are the single plan's rows bit for bit
    single = fb.FMM_plan(K, v, _opts(fb, 0.5), bc=bc).execute(x)
    total = np.zeros_like(single)
    for rank in range(3):
        part = fb.FMM_plan(K, v, _opts(fb, 0.5), bc=bc, shard=(rank, 3))
        total += part.execute(x)
        part.close()
    assert np.array_equal(total, single)


@pytest.mark.parametrize("k,applies", [(1, True), (3, True), (4, True), (13, False)])
def test_hybrid_rules_and_fallback(fb, oracle_mod, k, applies):
    """K = 1, 3, 4 keep the source's points in registers; longer rules (and everything else the kernel does not cover) take the
    fully streamed plan whatever the option says."""
    v = oracle_mod.unit_sphere(4)
    o = oracle_mod.StokesOracle(v, K=k, K_fine=19, mu=1e-3)
    K = _stokes(fb, p=6, k=k)
    h = fb.FMM_plan(K, v, _opts(fb, 0.4))
    assert (h.stats()["near_recomputed_pairs"] > 0) == applies
    x = drand48(3 * o.n, seed=12).reshape(o.n, 3)
    assert rel_l2(h.execute(x), o.matvec(x, 6)) <= 1e-12


def test_hybrid_is_ignored_where_it_does_not_apply(fb, oracle_mod, monkeypatch):
    v = oracle_mod.unit_sphere(4)
    K = _stokes(fb, p=6)
    o = _opts(fb, 0.3)
    o.local_evaluation, o.lazy_evaluation = True, False                                # the LOCAL evaluator (LocalPC.hpp:7-16)
    assert fb.FMM_plan(K, v, o).stats()["near_recomputed_pairs"] == 0
    o2 = _opts(fb, 0.3)
    o2.sparse_local = False                                                            # matrix-free: nothing stored anyway
    assert fb.FMM_plan(K, v, o2).stats()["near_recomputed_pairs"] == 0
    monkeypatch.setenv("FMMBEM_STOKES_SYM", "0")                                       # the 9-value rows: streamed only
    assert fb.FMM_plan(K, v, _opts(fb, 0.3)).stats()["near_recomputed_pairs"] == 0
    # Laplace with a rule of more than three points: the fully streamed plan
    KL = fb.LaplaceSphericalBEM(6, 4)
    assert fb.FMM_plan(KL, v, _opts(fb, 0.3)).stats()["near_recomputed_pairs"] == 0


def test_hybrid_under_graph_replay_and_in_the_solver(fb, oracle_mod):
    """The two kernels fork to a second stream and join inside the launch chain: captured into the hipGraph of a matvec like
    everything else; the device-resident solver runs on the hybrid operator."""
    import torch
    v = oracle_mod.unit_sphere(5)
    n = len(v)
    K = _stokes(fb, p=8)
    h = fb.FMM_plan(K, v, _opts(fb, 0.5))
    x = torch.from_numpy(drand48(3 * n, seed=7)).cuda()
    y0 = h.execute_torch(x).clone()
    h.set_graphs(True)
    for _ in range(4):                                                                 # launch by launch, capture, replay, replay
        assert torch.equal(h.execute_torch(x), y0)
    h.set_graphs(False)
    full = fb.FMM_plan(K, v)
    b = full.execute_torch(torch.ones(3 * n, dtype=torch.float64, device="cuda"))
    so = fb.SolverOptions(residual=1e-6, max_iters=60, max_p=8)
    xa, ita, _ = fb.gmres(full, torch.zeros_like(b), b, so)
    xb, itb, _ = fb.gmres(h, torch.zeros_like(b), b, so)
    assert ita == itb and float(torch.linalg.vector_norm(xa - xb) / torch.linalg.vector_norm(xa)) <= 1e-9


@pytest.mark.parametrize("f", [0.6, 0.0])
def test_hybrid_laplace(fb, oracle_mod, f):
    """One unknown per panel (near_recompute1_kernel beside near_spmv_pipe_kernel): POTENTIAL, NORMAL_DERIV and mixed targets, the
    rules K = 1 and 3, shards bitwise.  (No faster than the streamed matrix on this operator -- 8 bytes per pair are cheaper to
    stream than three reciprocal square roots are to compute, profiles/r05f -- the option trades time for footprint there.)"""
    v = np.concatenate([oracle_mod.unit_sphere(5), oracle_mod.unit_sphere(4, center=(2.4, 0.0, 0.3))])
    n = len(v)
    x = drand48(n, seed=21)
    for k, bc in ((3, None), (3, np.ones(n, dtype=np.uint8)), (3, (np.arange(n) % 3 == 0).astype(np.uint8)), (1, None)):
        o = oracle_mod.Oracle(v, bc=bc, K=k)
        K = fb.LaplaceSphericalBEM(9, k)
        full = fb.FMM_plan(K, v, bc=bc)
        hyb = fb.FMM_plan(K, v, _opts(fb, f), bc=bc)
        st = hyb.stats()
        assert abs(st["near_recomputed_pairs"] / st["near_nnz"] - (1.0 - f)) < 0.05 and abs(st["near_bytes"] / full.stats()["near_bytes"] - f) < 0.06
        y = hyb.execute(x)
        assert rel_l2(y, o.matvec(x, 9)) <= 1e-12
        assert rel_l2(y, full.execute(x)) <= 1e-13
        assert np.array_equal(y, hyb.execute(x))
        # the entry functions are the assembly's, inlined into another kernel: the compiler contracts them its own way there
        assert np.max(np.abs(hyb.diagonal() - full.diagonal())) <= 4e-16 * np.max(np.abs(full.diagonal()))
        for row in (0, n // 2, n - 1):
            a, b = hyb.near_row(row)[1], full.near_row(row)[1]
            assert np.max(np.abs(a - b)) <= 4e-16 * np.max(np.abs(b))
        o.close()
    total = np.zeros(n)
    for rank in range(3):
        total += fb.FMM_plan(fb.LaplaceSphericalBEM(9, 3), v, _opts(fb, f), shard=(rank, 3)).execute(x)
    assert np.array_equal(total, fb.FMM_plan(fb.LaplaceSphericalBEM(9, 3), v, _opts(fb, f)).execute(x))
