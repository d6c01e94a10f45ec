"""The preconditioner evaluators of make_evaluators (executor/make_executor.hpp:24-60): near-field only
(EvalLocalSparse.hpp:34-86, used by Preconditioners::LocalInnerSolver) and block diagonal (EvalDiagonalSparse.hpp:33-49,
used by Preconditioners::BlockDiagonal).  Selected the way the reference does, through FMMOptions."""
import numpy as np
import pytest


def _opts(fb, kind):
    o = fb.FMMOptions()
    o.lazy_evaluation = False
    o.sparse_local = True
    if kind == "local":
        o.local_evaluation = True                   # examples/BEM/LocalPC.hpp:7-16
    else:
        o.block_diagonal = True                     # examples/BEM/BlockDiagonalPC.hpp:53-63
    return o


@pytest.mark.parametrize("r", [4, 6])
def test_host_lists(fb, oracle_mod, r):
    v = fb.unit_sphere(r)
    full = fb.FMM_plan(fb.LaplaceSphericalBEM(5, 3), v, host_only=True)
    loc = fb.FMM_plan(fb.LaplaceSphericalBEM(5, 3), v, _opts(fb, "local"), host_only=True)
    dia = fb.FMM_plan(fb.LaplaceSphericalBEM(5, 3), v, _opts(fb, "diag"), host_only=True)
    assert np.array_equal(loc.pairs("p2p"), full.pairs("p2p"))               # same traversal, multipoles ignored
    for pl in (loc, dia):
        assert len(pl.pairs("m2l")) == len(pl.pairs("m2m")) == len(pl.pairs("l2l")) == 0
    b = dia.boxes()
    leaves = np.nonzero(b["leaf"])[0]
    assert np.array_equal(dia.pairs("p2p"), np.stack([leaves, leaves], axis=1))
    assert dia.stats()["near_nnz_total"] == int(((b["be"] - b["bb"])[leaves].astype(np.int64) ** 2).sum())
    for kind, pl in ((1, loc), (2, dia)):
        o = oracle_mod.Oracle(v, evaluator=kind)
        assert np.array_equal(pl.pairs("p2p"), o.pairs("p2p"))
        assert o.stats()["m2l_pairs"] == 0 and o.stats()["near_nnz"] == pl.stats()["near_nnz_total"]
    # lazy_evaluation wins over the other flags (make_executor.hpp:26)
    o = _opts(fb, "diag")
    o.lazy_evaluation = True
    assert len(fb.FMM_plan(fb.LaplaceSphericalBEM(5, 3), v, o, host_only=True).pairs("m2l")) == len(full.pairs("m2l"))


@pytest.mark.gpu
def test_local_and_block_diagonal_match_oracle(fb, oracle_mod):
    import torch
    v = fb.unit_sphere(6)
    rng = np.random.default_rng(5)
    x = rng.random(len(v))
    full = fb.FMM_plan(fb.LaplaceSphericalBEM(8, 3), v)
    xd = torch.from_numpy(x).cuda()
    yn = torch.empty_like(xd)
    full.near_device(xd.data_ptr(), yn.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for kind, name in ((1, "local"), (2, "diag")):
        pl = fb.FMM_plan(fb.LaplaceSphericalBEM(8, 3), v, _opts(fb, name))
        y = pl.execute(x)
        yo = oracle_mod.Oracle(v, evaluator=kind).matvec(x, 8)
        assert np.linalg.norm(y - yo) <= 1e-13 * np.linalg.norm(yo)
        if kind == 1:
            assert np.array_equal(y, yn.cpu().numpy())                      # = the FMM plan's near field, bit for bit
        else:
            # block diagonal: a unit charge only reaches the panels of its own leaf
            e = np.zeros(len(v)); e[17] = 1.0
            ye = pl.execute(e)
            b, perm = pl.boxes(), pl.perm()
            pos = int(np.nonzero(perm == 17)[0][0])
            leaf = [i for i in np.nonzero(b["leaf"])[0] if b["bb"][i] <= pos < b["be"][i]][0]
            own = perm[b["bb"][leaf]:b["be"][leaf]]
            assert np.all(ye[own] != 0) and np.count_nonzero(ye) == len(own)


@pytest.mark.gpu
def test_stokes_local_matches_oracle(fb, oracle_mod):
    v = fb.unit_sphere(4)
    rng = np.random.default_rng(6)
    f = rng.random((len(v), 3))
    K = fb.StokesSphericalBEM(6, 4, 1e-3)
    K.set_Kfine(19)
    pl = fb.FMM_plan(K, v, _opts(fb, "local"))
    u = pl.execute(f)
    uo = oracle_mod.StokesOracle(v, K=4, K_fine=19, mu=1e-3, evaluator=1).matvec(f, 6)
    assert np.linalg.norm(u - uo) <= 1e-13 * np.linalg.norm(uo)


@pytest.mark.gpu
def test_diagonal_preconditioner(fb, oracle_mod):
    """Preconditioners::Diagonal (examples/BEM/Preconditioner.hpp:19-42, LaplaceBEM.cpp:241-244, 287-296): GMRES with
    z = M(v) = v / K(s,s).  The diagonal comes out of the assembled near matrix; K(s,s) from the oracle is the check."""
    import torch
    v = fb.unit_sphere(5)
    n = len(v)
    bc = (np.arange(n) % 3 == 0).astype(np.uint8)
    plan = fb.FMM_plan(fb.LaplaceSphericalBEM(10, 3), v, bc=bc)
    dg = plan.diagonal()
    o = oracle_mod.Oracle(v, bc=bc)
    ref = o.kernel_entries(np.arange(n), np.arange(n))
    assert np.max(np.abs(dg - ref)) <= 1e-14 * np.max(np.abs(ref))
    # the same from a shard: its own rows, zeros elsewhere
    part = fb.FMM_plan(fb.LaplaceSphericalBEM(10, 3), v, bc=bc, shard=(1, 2))
    dp = part.diagonal()
    own = dp != 0
    assert 0 < own.sum() < n and np.array_equal(dp[own], dg[own])
    # GMRES with it: same solution as without (right preconditioning in the reference's GMRES, :196-241)
    pot = fb.FMM_plan(fb.LaplaceSphericalBEM(10, 3), v)
    rhs = fb.FMM_plan(fb.LaplaceSphericalBEM(10, 3), v, bc=np.ones(n, dtype=np.uint8))
    b = rhs.execute_torch(torch.ones(n, dtype=torch.float64, device="cuda"))
    so = fb.SolverOptions(residual=1e-6, max_iters=100, max_p=10)
    x1, it1, r1 = fb.gmres(pot, torch.zeros_like(b), b, so, M=fb.Diagonal(pot))
    x0, it0, r0 = fb.gmres(pot, torch.zeros_like(b), b, so)
    assert r1 < 1e-6 and abs(it1 - it0) <= 3
    assert float(torch.linalg.vector_norm(x1 - x0) / torch.linalg.vector_norm(x0)) < 1e-4
