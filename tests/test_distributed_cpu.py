"""The N > 1 path under gloo on the CPU, world_size 2 (no GPU here): every rank builds the same tree
(host-only plan through the C ABI), owns a contiguous shard of the target leaves, and the ONE collective of the
matvec -- all-reduce(sum) of result vectors that are zero outside the owned rows -- reassembles the operator.
The per-shard arithmetic is supplied by the CPU oracle restricted to the owned rows (the HIP kernels need a
GPU; tests/test_gpu_parity.py::test_shards_sum_to_full_operator checks them shard by shard on one card)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fmm_bem_relaxed_amd as fb
        from oracle import oracle as O

        v = fb.unit_sphere(5)
        n = len(v)
        K = fb.LaplaceSphericalBEM(8, 3)
        orc = O.Oracle(v)
        holder = {}

        def local_execute(x):          # stand-in for plan.execute_torch: oracle rows of this shard, zeros elsewhere
            y = np.zeros(n)
            rows = holder["op"].owned_rows()
            y[rows] = orc.matvec(x.numpy(), K.P)[rows]
            return torch.from_numpy(y)

        op = fb.ShardedFMM(K, v, host_only=True, local_execute=local_execute)
        holder["op"] = op
        assert (op.rank, op.world) == (rank, world)
        st = op.plan.stats()
        x = torch.from_numpy(np.random.default_rng(5).random(n))
        y = op.execute(x)
        ref = orc.matvec(x.numpy(), K.P)
        # ownership: contiguous leaf ranges that tile the leaf list
        begins = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(begins, torch.tensor([st["owned_leaf_begin"], st["owned_leaf_end"]]))
        owned = torch.zeros(n, dtype=torch.int64)
        owned[torch.from_numpy(op.owned_rows().astype(np.int64))] = 1
        dist.all_reduce(owned)
        ok = (np.array_equal(y.numpy(), ref) and bool((owned == 1).all())
              and begins[0][0].item() == 0 and begins[0][1].item() == begins[1][0].item()
              and begins[1][1].item() == st["n_leaves"])
        # the relaxed-p knob travels with the kernel object
        K.set_p(3)
        ok = ok and np.array_equal(op.execute(x).numpy(), orc.matvec(x.numpy(), 3))
        if rank == 0:
            out.put(ok)
    finally:
        dist.destroy_process_group()


def test_sharded_matvec_world2_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert out.get(timeout=5) is True


def test_shard_balance(fb):
    """The work model gives every rank a comparable share of near entries + M2L pairs."""
    v = fb.unit_sphere(7)
    K = fb.LaplaceSphericalBEM(10, 3)
    world = 8
    nnz, pairs = [], []
    for r in range(world):
        s = fb.FMM_plan(K, v, host_only=True, shard=(r, world)).stats()
        nnz.append(s["near_nnz"])
        pairs.append(s["m2l_pairs_owned"])
    assert max(nnz) / (sum(nnz) / world) < 1.35
    assert max(pairs) / (sum(pairs) / world) < 1.6


def _split_worker(rank, world, port, out):
    """The split execute of fmm-bem-relaxed_amd/distributed.py (upward half -> all_gather_into_tensor -> downward half ->
    all_reduce) with CPU stand-ins for the two halves: every rank contributes the x values of ITS rows, the gathered
    buffer must hold every rank's piece at rank * per, and the result is again the oracle's rows of the shard."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fmm_bem_relaxed_amd as fb
        from oracle import oracle as O

        v = fb.unit_sphere(5)
        n = len(v)
        K = fb.LaplaceSphericalBEM(6, 3)
        orc = O.Oracle(v)
        holder = {}
        per = n + 1                                      # room for every row + a rank tag

        def upward(x, send):
            rows = holder["op"].owned_rows()
            send.zero_()
            send[torch.from_numpy(rows.astype(np.int64))] = x[torch.from_numpy(rows.astype(np.int64))]
            send[n] = float(rank + 1)

        def downward(recv, y):
            pieces = recv.view(world, per)
            assert [float(pieces[r, n]) for r in range(world)] == [float(r + 1) for r in range(world)]
            xfull = pieces[:, :n].sum(dim=0)             # disjoint rows: the sum reassembles x
            rows = holder["op"].owned_rows()
            y.zero_()
            y[torch.from_numpy(rows.astype(np.int64))] = torch.from_numpy(orc.matvec(xfull.numpy(), K.P)[rows])

        op = fb.ShardedFMM(K, v, host_only=True, local_split=(per, upward, downward))
        holder["op"] = op
        # the plan itself was built with the owner lists of the sharded upward pass and reports the exchange size
        assert op.split and op.plan.shard_upward and op.plan.exchange_doubles(6) > 0
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([op.plan.exchange_doubles(6)]))
        x = torch.from_numpy(np.random.default_rng(8).random(n))
        y = op.execute(x)
        ok = np.array_equal(y.numpy(), orc.matvec(x.numpy(), K.P)) and sizes[0].item() == sizes[1].item()
        # the selective exchange (all-to-all with uneven counts): host-only plans answer the count queries, and the collective
        # check bench.py's preflight runs before the first all_to_all_single holds -- Laplace and Stokes with both target kinds
        sel = fb.ShardedFMM(K, v, host_only=True)
        ok = ok and sel.split and sel.plan.exchange_mode == 2 and sel.check_exchange_symmetry(6) is True
        bcs = np.zeros(n, dtype=np.uint8)
        bcs[::3] = 1
        KS = fb.StokesSphericalBEM(6, 4, 1e-3)
        for flags, slots in ((None, 4), (np.ones(n, dtype=np.uint8), 7), (bcs, 11)):
            ss = fb.ShardedFMM(KS, v, bc=flags, host_only=True)
            sc, rc = ss.plan.exchange_counts(6)
            ok = ok and ss.check_exchange_symmetry(6) is True and int(sc.sum()) > 0 and int(sc.sum()) % (slots * 21 * 2) == 0
        if rank == 0:
            out.put(bool(ok))
    finally:
        dist.destroy_process_group()


def test_split_execute_world2_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert out.get(timeout=5) is True
