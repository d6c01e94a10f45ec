"""Sharded upward pass (fmmbem_options.shard_upward, SURVEY.md section 8e taken one step further): every shard computes
P2M/M2M only for the boxes it owns, the multipoles are exchanged by one all-gather, the boxes spanning shards are then
translated by everybody.  On one GPU the all-gather is a concatenation of the shards' send buffers; the per-shard
results must still add up BITWISE to the single-plan result (same kernels, same operands, same order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,p", [(2, 10), (3, 10), (8, 4)])
def test_split_execute_sums_bitwise(fb, world, p):
    import torch
    v = np.concatenate([fb.unit_sphere(6), fb.unit_sphere(5, center=(3.0, 0.0, 0.0))])
    rng = np.random.default_rng(21)
    x = rng.standard_normal(len(v))
    K = fb.LaplaceSphericalBEM(p, 3)
    full = fb.FMM_plan(K, v, p_max=10).execute(x)
    xd = torch.from_numpy(x).cuda()
    plans = [fb.FMM_plan(fb.LaplaceSphericalBEM(p, 3), v, p_max=10, shard=(r, world), shard_upward=True) for r in range(world)]
    per = plans[0].exchange_doubles(p)
    assert per > 0 and all(pl.exchange_doubles(p) == per for pl in plans)
    buf = torch.full((world, per), float("nan"), dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for r, pl in enumerate(plans):
        pl.upward_device(xd.data_ptr(), buf[r].data_ptr(), s, p)
    total = torch.zeros_like(xd)
    for r, pl in enumerate(plans):
        y = torch.empty_like(xd)
        if r % 2 == 0:                                 # the near field while the all-gather would be in flight, then the rest
            pl.near_split_device(y.data_ptr(), s)
        pl.downward_device(buf.data_ptr(), y.data_ptr(), s, p)
        total += y
    torch.cuda.synchronize()
    assert np.array_equal(total.cpu().numpy(), full)
    # the whole-matvec entry point refuses such a plan instead of silently repeating the upward pass
    with pytest.raises(fb.FmmBemError) as e:
        plans[0].execute(x)
    assert e.value.status == 6
    st = plans[0].stats()
    assert st["ms_total"] >= 0


def test_mixed_bc_and_stokes_split(fb):
    import torch
    v = fb.unit_sphere(5)
    rng = np.random.default_rng(22)
    bc = (rng.random(len(v)) < 0.4).astype(np.uint8)
    x = rng.random(len(v))
    full = fb.FMM_plan(fb.LaplaceSphericalBEM(8, 3), v, bc=bc).execute(x)
    xd = torch.from_numpy(x).cuda()
    plans = [fb.FMM_plan(fb.LaplaceSphericalBEM(8, 3), v, bc=bc, shard=(r, 2), shard_upward=True) for r in range(2)]
    per = plans[0].exchange_doubles(8)
    buf = torch.zeros((2, per), dtype=torch.float64, device="cuda")
    for r, pl in enumerate(plans):
        pl.upward_device(xd.data_ptr(), buf[r].data_ptr(), 0, 8)
    torch.cuda.synchronize()
    total = torch.zeros_like(xd)
    for pl in plans:
        y = torch.empty_like(xd)
        pl.downward_device(buf.data_ptr(), y.data_ptr(), 0, 8)
        torch.cuda.synchronize()
        total += y
    assert np.array_equal(total.cpu().numpy(), full)
    # Stokes: 4 active slots
    f = rng.random((len(v), 3))
    Ks = fb.StokesSphericalBEM(6, 4, 1e-3)
    Ks.set_Kfine(19)
    fulls = fb.FMM_plan(Ks, v).execute(f)
    fd = torch.from_numpy(f).cuda()
    sp = []
    for r in range(2):
        Kr = fb.StokesSphericalBEM(6, 4, 1e-3)
        Kr.set_Kfine(19)
        sp.append(fb.FMM_plan(Kr, v, shard=(r, 2), shard_upward=True))
    per = sp[0].exchange_doubles(6)
    buf = torch.zeros((2, per), dtype=torch.float64, device="cuda")
    for r, pl in enumerate(sp):
        pl.upward_device(fd.data_ptr(), buf[r].data_ptr(), 0, 6)
    torch.cuda.synchronize()
    total = torch.zeros_like(fd)
    for pl in sp:
        y = torch.empty_like(fd)
        pl.downward_device(buf.data_ptr(), y.data_ptr(), 0, 6)
        torch.cuda.synchronize()
        total += y
    assert np.array_equal(total.cpu().numpy(), fulls)


def test_split_execute_as_graphs(fb, monkeypatch):
    """The two halves of a split execute replayed as hipGraphs (FMMBEM_GRAPH=1: one graph per half, order and exchange
    buffer): three matvecs in a row -- launch by launch, captured, replayed -- each summing bitwise to the single plan's."""
    import torch
    monkeypatch.setenv("FMMBEM_GRAPH", "1")
    v = np.concatenate([fb.unit_sphere(5), fb.unit_sphere(4, center=(3.0, 0.0, 0.0))])
    n, world, p = len(v), 3, 8
    rng = np.random.default_rng(77)
    plans = [fb.FMM_plan(fb.LaplaceSphericalBEM(p, 3), v, shard=(r, world), shard_upward=2) for r in range(world)]
    monkeypatch.delenv("FMMBEM_GRAPH")
    single = fb.FMM_plan(fb.LaplaceSphericalBEM(p, 3), v)
    counts = [pl.exchange_counts(p) for pl in plans]
    s = torch.cuda.current_stream().cuda_stream
    send = [torch.full((max(int(c[0].sum()), 1),), float("nan"), dtype=torch.float64, device="cuda") for c in counts]
    recv = [torch.full((max(int(c[1].sum()), 1),), float("nan"), dtype=torch.float64, device="cuda") for c in counts]
    for it in range(4):
        x = rng.standard_normal(n)
        xd = torch.from_numpy(x).cuda()
        for r, pl in enumerate(plans):
            pl.upward_device(xd.data_ptr(), send[r].data_ptr(), s, p)
        torch.cuda.synchronize()
        for r in range(world):
            so = np.concatenate([[0], np.cumsum(counts[r][0])])
            for q in range(world):
                ro = np.concatenate([[0], np.cumsum(counts[q][1])])
                recv[q][ro[r]:ro[r + 1]] = send[r][so[q]:so[q + 1]]
        total = torch.zeros_like(xd)
        for r, pl in enumerate(plans):
            y = torch.empty_like(xd)
            pl.downward_device(recv[r].data_ptr(), y.data_ptr(), s, p)
            total += y
        torch.cuda.synchronize()
        assert np.array_equal(total.cpu().numpy(), single.execute(x)), it


@pytest.mark.parametrize("world,p,traction", [(2, 10, False), (3, 8, False), (8, 5, False), (4, 6, True)])
def test_selective_exchange_sums_bitwise(fb, world, p, traction):
    """shard_upward = 2: every shard sends every other only the multipoles that shard's lists read, in one all-to-all with
    uneven counts (emulated here by copying the segments between the shards' buffers on one GPU).  The receive buffers are
    filled with NaN first: a box a shard reads but was not sent would poison its result.  The shards' results add up bitwise
    to the single plan's, and the traffic is a fraction of the all-gather's."""
    import torch
    v = np.concatenate([fb.unit_sphere(6), fb.unit_sphere(5, center=(3.0, 0.0, 0.0))])
    rng = np.random.default_rng(31)
    n = len(v)

    def kernel():
        if not traction:
            return fb.LaplaceSphericalBEM(p, 3)
        K = fb.StokesSphericalBEM(p, 4, 1e-3)
        K.set_Kfine(19)
        return K
    bc = (np.arange(n) % 2).astype(np.uint8) if traction else None      # Stokes: velocity and TRACTION targets, 11 slots
    x = rng.standard_normal((n, 3) if traction else n)
    full = fb.FMM_plan(kernel(), v, p_max=p, bc=bc).execute(x)
    xd = torch.from_numpy(x).cuda().reshape(-1)
    plans = [fb.FMM_plan(kernel(), v, p_max=p, bc=bc, shard=(r, world), shard_upward=2) for r in range(world)]
    counts = [pl.exchange_counts(p) for pl in plans]
    for r in range(world):
        assert counts[r][0][r] == 0 and counts[r][1][r] == 0
        for q in range(world):
            assert counts[r][0][q] == counts[q][1][r]                    # what r sends q is what q expects from r
    s = torch.cuda.current_stream().cuda_stream
    send = [torch.full((max(int(c[0].sum()), 1),), float("nan"), dtype=torch.float64, device="cuda") for c in counts]
    recv = [torch.full((max(int(c[1].sum()), 1),), float("nan"), dtype=torch.float64, device="cuda") for c in counts]
    for r, pl in enumerate(plans):
        pl.upward_device(xd.data_ptr(), send[r].data_ptr(), s, p)
    torch.cuda.synchronize()
    for r in range(world):                                              # the all-to-all
        so = np.concatenate([[0], np.cumsum(counts[r][0])])
        for q in range(world):
            ro = np.concatenate([[0], np.cumsum(counts[q][1])])
            recv[q][ro[r]:ro[r + 1]] = send[r][so[q]:so[q + 1]]
    total = torch.zeros_like(xd)
    for r, pl in enumerate(plans):
        y = torch.empty_like(xd)
        pl.downward_device(recv[r].data_ptr(), y.data_ptr(), s, p)
        total += y
    torch.cuda.synchronize()
    assert np.array_equal(total.cpu().numpy().reshape(full.shape), full)
    # against everything-to-everybody (shard_upward = 1)
    gather = fb.FMM_plan(kernel(), v, p_max=p, bc=bc, shard=(0, world), shard_upward=True).exchange_doubles(p) * (world - 1)
    assert max(int(c[1].sum()) for c in counts) < gather
