"""One plan over several devices of ONE process (fmmbem_options.n_devices / FMMBEM_DEVICES; SURVEY.md section 8b "device list"): the
handle shards the target leaves over the devices, copies x to each, exchanges the multipoles the shards' lists read, brings the result
slices home -- peer copies ordered by events, no second process, no collective library.  This box has ONE GPU: the device list names it
several times, which runs the whole path (every copy, every event, both upward modes) and must give the single plan's bits.  On an
8-GPU node the same code runs over xGMI; that has not been measured."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, drand48

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,replicate", [(2, False), (4, False), (3, True), (8, False)])
def test_multi_device_plan_equals_the_single_plan_bit_for_bit(fb, world, replicate):
    v = np.concatenate([fb.unit_sphere(6), fb.unit_sphere(5, center=(2.5, 0.0, 0.3))])
    n = len(v)
    x = drand48(n, seed=3)
    K = fb.LaplaceSphericalBEM(10, 3)
    single = fb.FMM_plan(K, v, p_max=10)
    multi = fb.FMM_plan(K, v, p_max=10, devices=[0] * world, replicate_upward=replicate)
    st = multi.stats()
    assert st["n_devices"] == world and st["near_nnz"] == single.stats()["near_nnz"]
    assert st["m2l_pairs_owned"] >= single.stats()["m2l_pairs_owned"]            # ancestors shared by several shards are translated by each
    for p in (10, 3, 10):
        K.set_p(p)
        assert np.array_equal(multi.execute(x), single.execute(x)), p            # host pointers
    import torch
    xd = torch.from_numpy(x).cuda()
    K.set_p(7)
    assert torch.equal(multi.execute_torch(xd), single.execute_torch(xd))         # device pointers, the caller's stream
    assert np.array_equal(multi.diagonal(), single.diagonal())
    for row in (0, n // 2, n - 1):
        a, b = multi.near_row(row), single.near_row(row)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(multi.perm(), single.perm())


def test_multi_device_stokes_and_the_flipped_plan(fb):
    v = fb.unit_sphere(5)
    n = len(v)
    K = fb.StokesSphericalBEM(7, 4, 1e-3)
    K.set_Kfine(19)
    x = drand48(3 * n, seed=9).reshape(n, 3)
    single = fb.FMM_plan(K, v)
    multi = fb.FMM_plan(K, v, devices=[0, 0, 0])
    assert np.array_equal(multi.execute(x), single.execute(x))
    # the drivers' second plan (TRACTION targets, StokesBEM.cpp:266-270) over the same devices: every shard shares its base shard's geometry
    ones = np.ones(n, dtype=np.uint8)
    multi_t = fb.FMM_plan(K, v, bc=ones, devices=[0, 0, 0])
    assert multi_t.stats()["geometry_shared"] >= 2
    assert np.array_equal(multi_t.execute(x), fb.FMM_plan(K, v, bc=ones).execute(x))
    third = multi.like(np.zeros(n, dtype=np.uint8))
    assert np.array_equal(third.execute(x), single.execute(x))


def test_relaxed_gmres_on_a_multi_device_plan(fb):
    """fmmbem_gmres_device on the handle: the Krylov vectors live on the first device, every matvec fans out over the list."""
    import torch
    v = fb.unit_sphere(6)
    n = len(v)
    ones = np.ones(n, dtype=np.uint8)
    so = fb.SolverOptions(residual=1e-5, max_iters=50, max_p=12)
    res = []
    for devices in (None, [0, 0]):
        K = fb.LaplaceSphericalBEM(12, 3)
        plan = fb.FMM_plan(K, v, p_max=12, devices=devices)
        rhs = fb.FMM_plan(fb.LaplaceSphericalBEM(12, 3), v, bc=ones, p_max=12, devices=devices)
        b = rhs.execute_torch(torch.ones(n, dtype=torch.float64, device="cuda"))
        log = []
        xs, it, r, _ = fb.gmres_capi(plan, torch.zeros_like(b), b, so, log=log)
        res.append((xs.clone(), it, [p for _, p, _ in log]))
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2] and res[0][2][:5] == [12, 3, 2, 1, 1]
    assert torch.equal(res[0][0], res[1][0])


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "LaplaceBEM_ref")), reason="oracle/_ref/LaplaceBEM_ref not built (make -C oracle ref)")
def test_the_references_own_driver_over_a_device_list(tmp_path):
    """examples/LaplaceBEM.cpp, unmodified, with FMMBEM_DEVICES=0,0: its ONE plan in ONE process runs on the listed devices and
    reproduces the recorded schedule 12, 3, 2, 1, 1."""
    exe = os.path.join(ROOT, "oracle", "_ref", "LaplaceBEM_ref")
    outs = []
    for env in ({}, {"FMMBEM_DEVICES": "0,0"}):
        r = subprocess.run([exe, "-recursions", "6", "-p", "12", "-theta", "0.5"], capture_output=True, text=True, cwd=str(tmp_path),
                           env=dict(os.environ, **env), check=True)
        outs.append([ln for ln in r.stdout.splitlines() if "fmm_req_p" in ln or "iteration" in ln.lower() or "error" in ln.lower()])
    assert outs[0] == outs[1] and len(outs[0]) > 0
    ps = [int(ln.split("fmm_req_p:")[1].split()[0].strip(",")) for ln in outs[0] if "fmm_req_p:" in ln]
    assert ps[:5] == [12, 3, 2, 1, 1] or len(ps) == 0


def test_multi_device_with_the_other_near_field_forms(fb):
    """The device list composes with the plan's other options: a hybrid near field (every shard makes the whole tree's leaf choice, so
    the shards' rows are the single hybrid plan's rows bit for bit), the matrix-free near field, a Stokes hybrid plan with TRACTION
    targets, and the evaluators that hold no far field."""
    import torch
    v = np.concatenate([fb.unit_sphere(6), fb.unit_sphere(5, center=(2.5, 0.0, 0.3))])
    n = len(v)
    x = drand48(n, seed=13)
    K = fb.LaplaceSphericalBEM(9, 3)
    for make in ("hybrid", "matfree", "local"):
        o = fb.FMMOptions()
        if make == "hybrid":
            o.near_stream_fraction = 0.5
        elif make == "matfree":
            o.sparse_local = False
        else:
            o.local_evaluation, o.lazy_evaluation = True, False
        single = fb.FMM_plan(K, v, o, p_max=9)
        multi = fb.FMM_plan(K, v, o, p_max=9, devices=[0, 0, 0])
        assert np.array_equal(multi.execute(x), single.execute(x)), make
        if make == "hybrid":
            assert multi.stats()["near_recomputed_pairs"] == single.stats()["near_recomputed_pairs"] > 0
        single.close()
        multi.close()
    vs = fb.unit_sphere(5)
    ns = len(vs)
    KS = fb.StokesSphericalBEM(7, 4, 1e-3)
    KS.set_Kfine(19)
    xs = drand48(3 * ns, seed=14).reshape(ns, 3)
    bc = (np.arange(ns) % 3 != 0).astype(np.uint8)
    o = fb.FMMOptions()
    o.near_stream_fraction = 0.5
    a = fb.FMM_plan(KS, vs, o, bc=bc).execute(xs)
    b = fb.FMM_plan(KS, vs, o, bc=bc, devices=[0, 0]).execute(xs)
    assert np.array_equal(a, b)


def test_sharded_operator_with_a_hybrid_near_field(fb):
    """ShardedFMM's split execute (upward / exchange / downward, the near field in between) on hybrid shards, two gloo-free shards
    driven by hand: the near field of a shard runs its three kernels on their streams inside fmmbem_plan_near_split_device."""
    import torch
    v = fb.unit_sphere(6)
    n = len(v)
    K = fb.LaplaceSphericalBEM(8, 3)
    o = fb.FMMOptions()
    o.near_stream_fraction = 0.6
    x = torch.from_numpy(drand48(n, seed=4)).cuda()
    whole = fb.FMM_plan(K, v, o, p_max=8).execute_torch(x)
    shards = [fb.FMM_plan(K, v, o, p_max=8, shard=(r, 2), shard_upward=True) for r in range(2)]
    per = shards[0].exchange_doubles(8)
    send = [torch.zeros(per, dtype=torch.float64, device="cuda") for _ in range(2)]
    s = torch.cuda.current_stream().cuda_stream
    for r in range(2):
        shards[r].upward_device(x.data_ptr(), send[r].data_ptr(), s, 8)
    recv = torch.cat(send)
    total = torch.zeros_like(x)
    for r in range(2):
        y = torch.zeros_like(x)
        shards[r].near_split_device(y.data_ptr(), s)
        shards[r].downward_device(recv.data_ptr(), y.data_ptr(), s, 8)
        total += y
    assert torch.equal(total, whole)
