"""The N > 1 path with the REAL per-shard work: two ranks (gloo, both on GPU 0 -- this box has one card) run
ShardedFMM.execute through the HIP plan's split entry points (upward -> exchange of multipoles, selective all-to-all or
all-gather -> downward) and both
result collectives (all-gather of tree-order slices + assembly; all-reduce of zero-padded vectors), and the replicated
result must equal the single plan's bit for bit; the relaxed GMRES then runs on the sharded operator.  (The collectives go
through the host under gloo: this is a correctness rehearsal of what bench.py --gpus N does over RCCL, not a measurement.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fmm_bem_relaxed_amd as fb
        dev = torch.device("cuda", 0)
        v = np.concatenate([fb.unit_sphere(6), fb.unit_sphere(5, center=(2.5, 0.0, 0.3))])
        n = len(v)
        x = torch.from_numpy(np.random.default_rng(3).random(n)).to(dev)
        ok = {}
        single = fb.FMM_plan(fb.LaplaceSphericalBEM(10, 3), v, p_max=10) if rank == 0 else None
        ups = {}
        for split, xch in ((True, "alltoall"), (True, "allgather"), (False, "alltoall")):
            os.environ["FMMBEM_XCH"] = xch                # how the multipoles travel: only what the receiver reads / everything
            for coll in ("allgather", "allreduce"):
                K = fb.LaplaceSphericalBEM(10, 3)
                op = fb.ShardedFMM(K, v, p_max=10, device=0, shard_upward=split, y_collective=coll)
                assert op.split == split and op.y_collective == coll
                assert op.plan.exchange_mode == ((2 if xch == "alltoall" else 1) if split else 0)
                for p in (10, 3):
                    K.set_p(p)
                    y = op.execute(x)
                    if rank == 0:
                        ref = single.execute_torch(x, p=p)
                        ok[(split, xch, coll, p)] = bool(torch.equal(y, ref))
                up, down = op.exchange_bytes(10)
                assert down > 0 and (up > 0) == split
                ups[(split, xch)] = up
                op.plan.close()
        assert ups[(True, "alltoall")] < ups[(True, "allgather")]
        os.environ.pop("FMMBEM_XCH")
        # Stokes, three unknowns per panel, slices of Vec<3,double>
        vs = fb.unit_sphere(5)
        KS = fb.StokesSphericalBEM(6, 4, 1e-3)
        KS.set_Kfine(19)
        xs = torch.from_numpy(np.random.default_rng(4).random(3 * len(vs))).to(dev)
        ops = fb.ShardedFMM(KS, vs, device=0)
        ys = ops.execute(xs)
        if rank == 0:
            refs = fb.FMM_plan(KS, vs).execute_torch(xs)
            ok["stokes"] = bool(torch.equal(ys, refs))
        # the caller of the path on the sharded operator: same orders, same iteration count as on one plan
        K = fb.LaplaceSphericalBEM(12, 3)
        op = fb.ShardedFMM(K, v, p_max=12, device=0)
        rhs = fb.ShardedFMM(fb.LaplaceSphericalBEM(12, 3), v, bc=np.ones(n, dtype=np.uint8), p_max=12, device=0)
        b = rhs.execute(torch.ones(n, dtype=torch.float64, device=dev))
        log = []
        xg, it, res = fb.gmres(op, torch.zeros(n, dtype=torch.float64, device=dev), b,
                               fb.SolverOptions(residual=1e-5, max_p=12), log=log)
        if rank == 0:
            K1 = fb.LaplaceSphericalBEM(12, 3)
            p1 = fb.FMM_plan(K1, v, p_max=12)
            r1 = fb.FMM_plan(fb.LaplaceSphericalBEM(12, 3), v, bc=np.ones(n, dtype=np.uint8), p_max=12)
            b1 = r1.execute_torch(torch.ones(n, dtype=torch.float64, device=dev))
            log1 = []
            x1, it1, res1 = fb.gmres(p1, torch.zeros(n, dtype=torch.float64, device=dev), b1,
                                     fb.SolverOptions(residual=1e-5, max_p=12), log=log1)
            ok["gmres"] = (bool(torch.equal(b, b1)) and it == it1 and [q for _, q, _ in log] == [q for _, q, _ in log1]
                           and bool(torch.equal(xg, x1)))
            out.put(ok)
    except BaseException as e:                              # the parent should fail now, not after its queue timeout
        out.put({"error in rank %d: %r" % (rank, e): False})
        raise
    finally:
        dist.destroy_process_group()


def test_two_ranks_real_sharded_execute_and_solve():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    ok = out.get(timeout=280)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok.values()), ok
