"""GPU parity tests: the HIP path (through the C ABI of include/fmmbem.h) against the CPU oracle on the
same seeded inputs, stage by stage and end to end.

Tolerances: near entries <= 1e-13 relative (libm differences), operator outputs
(M, L) <= 1e-12 relative, matvec GPU vs oracle <= 1e-12 relative L2; FMM vs Direct: north_star gate
1e-6 at p=10 for the G kernel.
"""
import numpy as np
import pytest

from conftest import drand48, rel_l2

pytestmark = pytest.mark.gpu

TOL_ENTRY = 1e-13
TOL_EXPANSION = 1e-12
TOL_MATVEC = 1e-12


@pytest.fixture(scope="module")
def case6(fb, oracle_mod):
    v = fb.unit_sphere(6)
    K = fb.LaplaceSphericalBEM(10, 3)
    pl = fb.FMM_plan(K, v, p_max=16)
    o = oracle_mod.Oracle(v)
    x = drand48(o.n)
    return v, K, pl, o, x


def test_extension_is_loaded(fb):
    import os
    assert os.path.exists(fb.LIB_PATH)
    fb.unit_sphere(1)                                  # first call loads the shared object
    maps = open("/proc/self/maps").read()
    assert "libfmmbem_hip.so" in maps


def test_near_matrix_entries(case6):
    _, _, pl, o, _ = case6
    rp, col, val = o.near_csr()
    rng = np.random.default_rng(1)
    worst = 0.0
    for row in np.concatenate([[0, o.n - 1], rng.integers(0, o.n, 60)]):
        cols, vals = pl.near_row(int(row))
        assert np.array_equal(cols, col[rp[row]:rp[row + 1]])
        ref = val[rp[row]:rp[row + 1]]
        worst = max(worst, float(np.max(np.abs(vals - ref) / np.abs(ref))))
    assert worst <= TOL_ENTRY, worst


def test_near_field_only(case6):
    import torch
    _, _, pl, o, x = case6
    xd = torch.from_numpy(x).cuda()
    yd = torch.zeros_like(xd)
    pl.near_device(xd.data_ptr(), yd.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rel_l2(yd.cpu().numpy(), o.near_only(x)) <= 1e-14


@pytest.mark.parametrize("p", [1, 2, 3, 4, 5, 6, 10, 11, 12, 13, 14, 15, 16])
def test_expansions_and_matvec_vs_oracle(case6, p):
    _, K, pl, o, x = case6
    K.set_p(p)
    y = pl.execute(x)
    yo = o.matvec(x, p)
    M, L = pl.expansions("M", p), pl.expansions("L", p)
    Mo, Lo = o.expansions(p, "M"), o.expansions(p, "L")
    # slot 0 (G expansion) is the active one for all-POTENTIAL panels
    scaleM = np.abs(Mo[:, 0]).max(axis=1, keepdims=True) + 1e-300
    scaleL = np.abs(Lo[:, 0]).max(axis=1, keepdims=True) + 1e-300
    assert np.max(np.abs(M[:, 0] - Mo[:, 0]) / scaleM) <= TOL_EXPANSION
    assert np.max(np.abs(L[:, 0] - Lo[:, 0]) / scaleL) <= TOL_EXPANSION
    assert rel_l2(y, yo) <= TOL_MATVEC


@pytest.mark.parametrize("p", [2, 7, 10, 12, 13, 14, 15, 16])
def test_expansions_all_slots_mixed_bc(fb, oracle_mod, p):
    """Both expansion slots live (POTENTIAL and NORMAL_DERIV panels mixed): every slot of M and L against the oracle, every
    order family -- rotation kernels (p <= 12) and the double-sum kernels (13 ... 16)."""
    v = oracle_mod.unit_sphere(5)
    rng = np.random.default_rng(17)
    bc = (rng.random(len(v)) < 0.5).astype(np.uint8)
    x = rng.standard_normal(len(v))
    o = oracle_mod.Oracle(v, bc=bc)
    K = fb.LaplaceSphericalBEM(p, 3)
    pl = fb.FMM_plan(K, v, bc=bc)
    y, yo = pl.execute(x), o.matvec(x, p)
    for which in ("M", "L"):
        got, ref = pl.expansions(which, p), o.expansions(p, which)
        assert got.shape == ref.shape and got.shape[1] == 2
        scale = np.abs(ref).max(axis=2, keepdims=True) + 1e-300
        assert np.max(np.abs(got - ref) / scale) <= TOL_EXPANSION, which
        assert np.abs(ref[:, 0]).max() > 0 and np.abs(ref[:, 1]).max() > 0
    assert rel_l2(y, yo) <= TOL_MATVEC


def test_matvec_vs_direct_north_star_gate(case6):
    _, K, pl, o, x = case6
    d = o.direct(x)
    K.set_p(10)
    err = rel_l2(pl.execute(x), d)
    assert err < 1e-6                                  # north_star: within 1e-6 relative L2 of Direct.hpp
    assert abs(err - 5.52e-7) / 5.52e-7 < 0.01         # the reference's own error level
    K.set_p(5)
    err5 = rel_l2(pl.execute(x), d)
    assert abs(err5 - 6.71e-5) / 6.71e-5 < 0.01


def test_ones_charges_and_linearity(case6):
    _, K, pl, o, x = case6
    K.set_p(8)
    one = np.ones(o.n)
    y1, yx = pl.execute(one), pl.execute(x)
    assert rel_l2(y1, o.matvec(one, 8)) <= TOL_MATVEC
    assert rel_l2(pl.execute(2.5 * x - one), 2.5 * yx - y1) <= 1e-13
    # repeatability: same input, same bits (fixed summation order, no atomics)
    assert np.array_equal(yx, pl.execute(x))


def test_relaxed_p_sequence_on_one_plan(case6):
    """The solver lowers p between executes on the same plan (GMRES.hpp:194-201)."""
    _, K, pl, o, x = case6
    for p in (12, 3, 2, 1, 1, 12):
        K.set_p(p)
        assert rel_l2(pl.execute(x), o.matvec(x, p)) <= TOL_MATVEC


@pytest.mark.parametrize("bc_kind", ["normal_deriv", "mixed"])
def test_boundary_condition_variants(fb, oracle_mod, bc_kind):
    v = fb.unit_sphere(5)
    n = len(v)
    bc = np.ones(n, dtype=np.uint8)
    if bc_kind == "mixed":
        bc[::3] = 0
    K = fb.LaplaceSphericalBEM(10, 3)
    pl = fb.FMM_plan(K, v, bc=bc)
    o = oracle_mod.Oracle(v, bc=bc)
    x = drand48(n, seed=7)
    y, yo = pl.execute(x), o.matvec(x, 10)
    assert rel_l2(y, yo) <= TOL_MATVEC
    Mo = o.expansions(10, "M")
    M = pl.expansions("M", 10)
    assert np.max(np.abs(M - Mo)) / np.abs(Mo).max() <= TOL_EXPANSION
    rp, col, val = o.near_csr()
    for row in (0, n // 2, n - 1):
        _, vals = pl.near_row(row)
        ref = val[rp[row]:rp[row + 1]]
        assert np.max(np.abs(vals - ref) / np.abs(ref)) <= TOL_ENTRY


@pytest.mark.parametrize("theta,ncrit,k", [(0.4, 64, 3), (0.6, 32, 4), (0.5, 125, 7), (0.5, 200, 1), (0.5, 64, 13), (0.5, 64, 79)])
def test_option_variants(fb, oracle_mod, theta, ncrit, k):
    v = fb.unit_sphere(5)
    opts = fb.FMMOptions()
    opts.set_mac_theta(theta)
    opts.set_max_per_box(ncrit)
    K = fb.LaplaceSphericalBEM(8, k)
    pl = fb.FMM_plan(K, v, opts)
    o = oracle_mod.Oracle(v, K=k, theta=theta, ncrit=ncrit)
    x = drand48(len(v), seed=3)
    assert rel_l2(pl.execute(x), o.matvec(x, 8)) <= TOL_MATVEC


def test_tiny_inputs(fb, oracle_mod):
    # r=1: 8 panels, a single leaf, no far field at all; r=3: 128 panels
    for r in (1, 2, 3, 4):
        v = fb.unit_sphere(r)
        pl = fb.FMM_plan(fb.LaplaceSphericalBEM(6, 3), v)
        o = oracle_mod.Oracle(v)
        x = drand48(len(v), seed=r)
        assert rel_l2(pl.execute(x), o.matvec(x, 6)) <= TOL_MATVEC


def test_two_spheres_multi_body(fb, oracle_mod):
    """Config-3 style input: two disjoint unit spheres, small version."""
    v = np.concatenate([fb.unit_sphere(5), fb.unit_sphere(5, center=(3.0, 0.0, 0.0))])
    K = fb.LaplaceSphericalBEM(10, 3)
    pl = fb.FMM_plan(K, v)
    o = oracle_mod.Oracle(v)
    assert np.array_equal(pl.pairs("m2l"), o.pairs("m2l"))
    x = drand48(len(v), seed=11)
    y = pl.execute(x)
    assert rel_l2(y, o.matvec(x, 10)) <= TOL_MATVEC
    assert rel_l2(y, o.direct(x)) < 1e-6


def test_shards_sum_to_full_operator(fb, oracle_mod):
    """Multi-GPU decomposition on one card: the per-shard results (zero outside the owned rows) add up
    to the single-plan result bit for bit (identical per-target summation order)."""
    v = fb.unit_sphere(6)
    K = fb.LaplaceSphericalBEM(10, 3)
    x = drand48(len(v))
    full = fb.FMM_plan(K, v).execute(x)
    for world in (2, 4):
        acc = np.zeros_like(full)
        for rank in range(world):
            part = fb.FMM_plan(K, v, shard=(rank, world)).execute(x)
            assert np.count_nonzero(part) <= len(v)
            acc += part
        assert np.array_equal(acc, full)


def test_device_pointer_entry_point_and_torch_stream(case6):
    import torch
    _, K, pl, o, x = case6
    K.set_p(10)
    xd = torch.from_numpy(x).cuda()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        yd = pl.execute_torch(xd)
    s.synchronize()
    assert np.array_equal(yd.cpu().numpy(), pl.execute(x))


def test_larger_case_r8_properties(fb, oracle_mod):
    """N=131072 (config 2): oracle comparison on the full vector plus a sampled Direct check."""
    v = fb.unit_sphere(8)
    K = fb.LaplaceSphericalBEM(10, 3)
    pl = fb.FMM_plan(K, v)
    s = pl.stats()
    assert (s["n_boxes"], s["n_leaves"], s["near_nnz"], s["m2l_pairs"]) == (5201, 4184, 79891400, 167480)
    o = oracle_mod.Oracle(v)
    x = drand48(o.n)
    y = pl.execute(x)
    assert rel_l2(y, o.matvec(x, 10)) <= TOL_MATVEC
    rows = (1000, 1256)
    d = o.direct(x, rows=rows)
    assert rel_l2(y[rows[0]:rows[1]], d) < 2e-6


@pytest.mark.parametrize("bc_val,quad_k", [(0, 3), (1, 3), (0, 1), (1, 1), (0, 4), (1, 13)])
def test_matrix_free_near_field(fb, oracle_mod, monkeypatch, bc_val, quad_k):
    """sparse_local = false: EvalInteractionLazy recomputes the panel integrals every matvec.
    K = 1: the only quadrature point of a panel is its centroid, so a lane past the last column that runs the far-regime
    arithmetic against tree panel 0 meets distance zero in row 0 -- its NaN must not reach the row's sum (round-2 advisory);
    K > 3: the far regime re-reads the quadrature points (mf_far_general)."""
    v = fb.unit_sphere(5)
    n = len(v)
    bc = np.full(n, bc_val, dtype=np.uint8)
    opts = fb.FMMOptions()
    opts.sparse_local = False
    K = fb.LaplaceSphericalBEM(10, quad_k)
    pl = fb.FMM_plan(K, v, opts, bc=bc)
    st = pl.stats()
    assert st["near_bytes"] == 0 and 0 < st["near_side_entries"] < 0.2 * st["near_nnz"]     # the near-regime pairs, kept; no matrix
    x = drand48(n, seed=5)
    y = pl.execute(x)
    assert np.isfinite(y).all()
    assert rel_l2(y, oracle_mod.Oracle(v, K=quad_k, bc=bc).matvec(x, 10)) <= TOL_MATVEC
    dense = fb.FMM_plan(K, v, bc=bc).execute(x)
    assert rel_l2(y, dense) <= 1e-14


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"FMMBEM_P2M_TABLE": "0"}, {"FMMBEM_L2P_GENERIC": "1"},
                                 {"FMMBEM_ROT_ITEM_PASSES": "1", "FMMBEM_ROT_LONG_MAX": "1"},
                                 {"FMMBEM_ROT_ITEM_PASSES": "5", "FMMBEM_ROT_LONG_MAX": "40"}])
def test_alternative_paths_match_oracle(fb, oracle_mod, monkeypatch, env):
    """The switches that select the fallback kernels (recurrence P2M instead of the stored moments -- what a plan whose records
    would pass 16 GB runs --, the L2P kernel that takes the order at run time -- p > 12 --) and other cuts of the M2L pair list
    into items give the same operator -- cuts and the two L2P kernels bit for bit."""
    for k, val in env.items():
        monkeypatch.setenv(k, val)
    v = oracle_mod.unit_sphere(5)
    rng = np.random.default_rng(3)
    bc = (rng.random(len(v)) < 0.5).astype(np.uint8)
    x = rng.standard_normal(len(v))
    o = oracle_mod.Oracle(v, bc=bc)
    K = fb.LaplaceSphericalBEM(10, 3)
    pl = fb.FMM_plan(K, v, bc=bc)
    res = {}
    for p in (10, 3):
        K.set_p(p)
        y, yo = pl.execute(x), o.matvec(x, p)
        assert np.linalg.norm(y - yo) <= 1e-12 * np.linalg.norm(yo)
        res[p] = y
    if "FMMBEM_L2P_GENERIC" in env or "FMMBEM_ROT_ITEM_PASSES" in env:   # not other arithmetic: the same bits
        for k in env:
            monkeypatch.delenv(k)
        K2 = fb.LaplaceSphericalBEM(10, 3)
        pl2 = fb.FMM_plan(K2, v, bc=bc)
        for p in (10, 3):
            K2.set_p(p)
            assert np.array_equal(pl2.execute(x), res[p])


@pytest.mark.gpu
@pytest.mark.parametrize("k,ncrit", [(3, 64), (3, 200), (4, 300), (1, 130), (7, 64)])
def test_assembly_kernels_give_the_same_bits(fb, monkeypatch, k, ncrit):
    """The near matrix by a thread per column (the source panel in registers down the rows of a leaf, near-regime entries queued
    and taken by full wavefronts; rules of up to four points) and by a thread per entry (FMMBEM_ASM_COLS=0; what rules of more
    points take): the same entry functions, the same matrix bit for bit -- leaves of more rows than one row block (ncrit > 64),
    mixed flags, every rule the column kernel is instantiated for."""
    monkeypatch.setenv("FMMBEM_PLAN_SHARE", "0")
    rng = np.random.default_rng(5)
    v = fb.unit_sphere(6)
    bc = (rng.random(len(v)) < 0.5).astype(np.uint8)
    x = rng.standard_normal(len(v))
    opts = fb.FMMOptions()
    opts.set_max_per_box(ncrit)
    got = []
    for cols in ("1", "0"):
        monkeypatch.setenv("FMMBEM_ASM_COLS", cols)
        pl = fb.FMM_plan(fb.LaplaceSphericalBEM(4, k), v, opts, bc=bc)
        rows = [pl.near_row(int(r)) for r in (0, 17, len(v) // 2, len(v) - 1)]
        got.append((pl.execute(x), rows))
    assert np.array_equal(got[0][0], got[1][0])
    for (ca, va), (cb, vb) in zip(got[0][1], got[1][1]):
        assert np.array_equal(ca, cb) and np.array_equal(va, vb)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["velocity_sym", "mixed_rows9", "mixed_big_leaves", "traction_k1"])
def test_stokes_assembly_kernels_give_the_same_bits(fb, monkeypatch, case):
    """The same for the Stokes blocks (`near_assemble_stokes_cols_kernel` against the pair-per-thread kernel): the symmetric
    6-value form and the 9-value rows, VELOCITY and TRACTION targets, leaves of more rows than one row block."""
    monkeypatch.setenv("FMMBEM_PLAN_SHARE", "0")
    rng = np.random.default_rng(7)
    v, k, kf, ncrit, sym = {"velocity_sym": (fb.red_blood_cell(5), 4, 19, 64, "1"), "mixed_rows9": (fb.red_blood_cell(5), 4, 19, 64, "0"),
                            "mixed_big_leaves": (fb.unit_sphere(5), 3, 25, 200, "1"), "traction_k1": (fb.unit_sphere(4), 1, 7, 64, "1")}[case]
    n = len(v)
    bc = None if case == "velocity_sym" else np.ones(n, dtype=np.uint8) if case == "traction_k1" else (rng.random(n) < 0.4).astype(np.uint8)
    monkeypatch.setenv("FMMBEM_STOKES_SYM", sym)
    x = rng.standard_normal((n, 3))
    opts = fb.FMMOptions()
    opts.set_max_per_box(ncrit)
    got = []
    for cols in ("1", "0"):
        monkeypatch.setenv("FMMBEM_ASM_COLS", cols)
        K = fb.StokesSphericalBEM(4, k, 1e-3)
        K.set_Kfine(kf)
        pl = fb.FMM_plan(K, v, opts, bc=bc)
        got.append((pl.execute(x), [pl.near_row(int(r)) for r in (0, 7, 3 * n - 1)]))
    assert np.array_equal(got[0][0], got[1][0])
    for (ca, va), (cb, vb) in zip(got[0][1], got[1][1]):
        assert np.array_equal(ca, cb) and np.array_equal(va, vb)


@pytest.mark.gpu
def test_graph_replay_is_bitwise_the_launch_chain(fb, oracle_mod):
    """fmmbem_plan_set_graphs: from its second execute at an order on, the chain between gather and delivery is a captured
    hipGraph launched on the caller's stream -- the same kernels in the same order, so the same bits as launch by launch, for
    every order of a relaxed schedule, repeated, for both kernels, and next to the near-field-only entry point."""
    import torch
    v = np.concatenate([oracle_mod.unit_sphere(5), oracle_mod.unit_sphere(4, center=(2.5, 0.3, -0.2))])
    rng = np.random.default_rng(5)
    bc = (rng.random(len(v)) < 0.3).astype(np.uint8)
    x = rng.standard_normal(len(v))
    K0, K1 = fb.LaplaceSphericalBEM(12, 3), fb.LaplaceSphericalBEM(12, 3)
    plain, graphed = fb.FMM_plan(K0, v, bc=bc), fb.FMM_plan(K1, v, bc=bc)
    graphed.set_graphs(True)
    xd = torch.from_numpy(x).cuda()
    for p in (12, 12, 12, 7, 3, 7, 3, 1, 12, 3, 7):
        a = plain.execute_torch(xd, p=p)
        b = graphed.execute_torch(xd, p=p)
        assert torch.equal(a, b), p
        yn_a, yn_b = torch.empty_like(xd), torch.empty_like(xd)
        s = torch.cuda.current_stream().cuda_stream
        plain.near_device(xd.data_ptr(), yn_a.data_ptr(), s)
        graphed.near_device(xd.data_ptr(), yn_b.data_ptr(), s)
        assert torch.equal(yn_a, yn_b)
    # on a side stream, and with other inputs than the graph was captured with
    side = torch.cuda.Stream()
    x2 = torch.from_numpy(rng.standard_normal(len(v))).cuda()
    with torch.cuda.stream(side):
        b = graphed.execute_torch(x2, p=7)
    side.synchronize()
    assert torch.equal(plain.execute_torch(x2, p=7), b)
    vs = oracle_mod.unit_sphere(4)
    KS = fb.StokesSphericalBEM(6, 4, 1e-3)
    KS.set_Kfine(19)
    f = rng.standard_normal((len(vs), 3))
    ref = fb.FMM_plan(KS, vs).execute(f)
    gs = fb.FMM_plan(KS, vs)
    gs.set_graphs(True)
    for _ in range(3):
        assert np.array_equal(gs.execute(f), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("bc_val", [0, 1])
def test_p2m_streaming_kernel(fb, oracle_mod, monkeypatch, bc_val):
    """One expansion per box and more than 32 coefficients: the streaming P2M contraction (scalar leaf records and charges,
    eight table entries in flight) -- M per box against the oracle at p = 8 ... 16 (two passes over the coefficients from
    p = 11), and bit for bit the kernel it replaces (same panel order of the additions)."""
    v = np.concatenate([oracle_mod.unit_sphere(5), oracle_mod.unit_sphere(3, center=(2.5, 0.3, -0.2))])   # leaves of 1 ... 64 panels
    bc = np.full(len(v), bc_val, dtype=np.uint8)
    x = np.random.default_rng(21).standard_normal(len(v))
    o = oracle_mod.Oracle(v, bc=bc)
    K = fb.LaplaceSphericalBEM(16, 3)
    pl = fb.FMM_plan(K, v, bc=bc)
    got = {}
    for p in (8, 10, 11, 12, 16):
        K.set_p(p)
        pl.execute(x)
        o.matvec(x, p)
        M, Mo = pl.expansions("M", p), o.expansions(p, "M")
        scale = np.abs(Mo[:, bc_val]).max(axis=1, keepdims=True) + 1e-300
        assert np.abs(Mo[:, bc_val]).max() > 0
        assert np.max(np.abs(M[:, bc_val] - Mo[:, bc_val]) / scale) <= TOL_EXPANSION
        got[p] = M
    monkeypatch.setenv("FMMBEM_P2M_STREAM", "0")
    for p in got:
        K.set_p(p)
        pl.execute(x)
        assert np.array_equal(pl.expansions("M", p), got[p]), p


@pytest.mark.gpu
@pytest.mark.parametrize("p", [4, 10])
def test_tree_passes_at_the_default_level_rule(fb, monkeypatch, p):
    """The rule the product runs with -- rotation kernels on tree levels of 2 048 boxes and more, sparse operators on the
    levels near the root -- on a mesh big enough to have both kinds of level (UnitSphere(8), 131 072 panels), both
    expansion slots live.  No oracle at this size: the two families are each held to the oracle level by level on small meshes
    (test_tree_passes_by_rotation_match_oracle, test_expansions_and_matvec_vs_oracle); here the mixed pass must give the M and L
    of the all-sparse pass box by box, and the M2L items (long ones at p = 10, short ones at p = 4) the same bits under another cut."""
    v = fb.unit_sphere(8)
    rng = np.random.default_rng(5)
    bc = (rng.random(len(v)) < 0.5).astype(np.uint8)
    x = rng.standard_normal(len(v))
    K = fb.LaplaceSphericalBEM(p, 3)
    pl = fb.FMM_plan(K, v, bc=bc)
    lev = pl.boxes()["level"]
    per_level = np.bincount(lev)
    assert per_level.max() >= 2048 and (per_level[2:] < 2048).any()          # both kernels families run
    y = pl.execute(x)
    M, L = pl.expansions("M", p), pl.expansions("L", p)
    monkeypatch.setenv("FMMBEM_SHIFT_ROT", "0")
    pl2 = fb.FMM_plan(K, v, bc=bc)
    y2 = pl2.execute(x)
    for got, ref, which in ((M, pl2.expansions("M", p), "M"), (L, pl2.expansions("L", p), "L")):
        scale = np.abs(ref).max(axis=2, keepdims=True) + 1e-300
        assert np.max(np.abs(got - ref) / scale) <= TOL_EXPANSION, which
    assert rel_l2(y, y2) <= 1e-13
    monkeypatch.delenv("FMMBEM_SHIFT_ROT")
    monkeypatch.setenv("FMMBEM_ROT_ITEM_PASSES", "3")
    monkeypatch.setenv("FMMBEM_ROT_LONG_MAX", "5")
    assert np.array_equal(fb.FMM_plan(K, v, bc=bc).execute(x), y)


@pytest.mark.gpu
@pytest.mark.parametrize("p", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_tree_passes_by_rotation_match_oracle(fb, oracle_mod, monkeypatch, p):
    """M2M and L2L through the rotation kernels (kernels_m2l_rot.hip compiled with FMMBEM_ROT_OP = 1, 2).  By default only
    levels of 2 048 boxes and more take that path, which no mesh of test size has: FMMBEM_SHIFT_ROT_MIN=0 sends every level
    there (with the one-pair-per-wavefront kernel, which small levels take by default, switched off).  Expansions and result against
    the oracle; mixed boundary conditions so that both expansion slots are live."""
    monkeypatch.setenv("FMMBEM_SHIFT_ROT_MIN", "0")
    monkeypatch.setenv("FMMBEM_SHIFT_LANES", "0")
    v = np.concatenate([oracle_mod.unit_sphere(5), oracle_mod.unit_sphere(4, center=(2.5, 0.3, -0.2))])
    rng = np.random.default_rng(11)
    bc = (rng.random(len(v)) < 0.4).astype(np.uint8)
    x = rng.standard_normal(len(v))
    o = oracle_mod.Oracle(v, bc=bc)
    K = fb.LaplaceSphericalBEM(p, 3)
    pl = fb.FMM_plan(K, v, bc=bc)
    y, yo = pl.execute(x), o.matvec(x, p)
    for which in ("M", "L"):
        got, ref = pl.expansions(which, p), o.expansions(p, which)
        scale = np.abs(ref).max(axis=2, keepdims=True) + 1e-300
        assert np.max(np.abs(got - ref) / scale) <= TOL_EXPANSION, which
    assert rel_l2(y, yo) <= TOL_MATVEC
    # the same operator as the sparse-operator kernels, to rounding
    monkeypatch.setenv("FMMBEM_SHIFT_ROT", "0")
    y2 = fb.FMM_plan(K, v, bc=bc).execute(x)
    assert rel_l2(y, y2) <= 1e-14



@pytest.mark.gpu
@pytest.mark.parametrize("p", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_shift_lanes_equal_the_one_pair_kernels_bit_for_bit(fb, oracle_mod, monkeypatch, p):
    """The tree passes with one pair per wavefront (kernels_shift.hip: lane = coefficient, the pair's coefficients through LDS)
    against the one-pair-per-lane rotation kernels (kernels_m2l_rot.hip, FMMBEM_ROT_OP = 1, 2): M and L of every box and the
    result, BIT FOR BIT -- every output is the same sequence of floating-point operations -- and against the oracle."""
    v = np.concatenate([oracle_mod.unit_sphere(5), oracle_mod.unit_sphere(4, center=(2.5, 0.3, -0.2))])
    rng = np.random.default_rng(17)
    bc = (rng.random(len(v)) < 0.4).astype(np.uint8)
    x = rng.standard_normal(len(v))
    K = fb.LaplaceSphericalBEM(p, 3)
    monkeypatch.setenv("FMMBEM_SHIFT_LANES", "1")
    pl = fb.FMM_plan(K, v, bc=bc)
    y = pl.execute(x)
    M, L = pl.expansions("M", p), pl.expansions("L", p)
    monkeypatch.setenv("FMMBEM_SHIFT_LANES", "0")
    monkeypatch.setenv("FMMBEM_SHIFT_ROT_MIN", "0")          # every level through the one-pair-per-lane rotation kernels
    pl1 = fb.FMM_plan(K, v, bc=bc)
    y1 = pl1.execute(x)
    assert np.array_equal(pl1.expansions("M", p), M)
    assert np.array_equal(pl1.expansions("L", p), L)
    assert np.array_equal(y1, y)
    o = oracle_mod.Oracle(v, bc=bc)
    assert rel_l2(y, o.matvec(x, p)) <= TOL_MATVEC
    for which, got in (("M", M), ("L", L)):
        ref = o.expansions(p, which)
        scale = np.abs(ref).max(axis=2, keepdims=True) + 1e-300
        assert np.max(np.abs(got - ref) / scale) <= TOL_EXPANSION, which


@pytest.mark.gpu
@pytest.mark.parametrize("p", [9, 10])
def test_shift_lanes_wavefront_runs_of_parents_bit_for_bit(fb, monkeypatch, p):
    """the M2M path of shift_lanes_kernel that plan.hip selects for levels of more than 1 024 parents (a contiguous run of
    parents per wavefront, the pa[4][R] chain bookkeeping) -- UnitSphere(8) with 16 panels per leaf has such levels -- against the
    one-pair-per-lane kernels: M, L and y bit for bit, and two shards summing to the whole bit for bit."""
    v = fb.unit_sphere(8)
    x = np.random.default_rng(5).standard_normal(len(v))
    K = fb.LaplaceSphericalBEM(p, 3)
    fo = fb.FMMOptions()
    fo.set_max_per_box(16)
    monkeypatch.setenv("FMMBEM_SHIFT_LANES", "1")
    monkeypatch.setenv("FMMBEM_SHIFT_LANES_MAX", "1000000")          # every level through the wavefront kernel
    pl = fb.FMM_plan(K, v, fo)
    boxes = pl.boxes()
    parents_per_level = np.bincount(boxes["level"][boxes["leaf"] == 0])
    assert parents_per_level.max() > 1024
    y = pl.execute(x)
    M, L = pl.expansions("M", p), pl.expansions("L", p)
    total = np.zeros_like(y)
    for r in range(2):
        total += fb.FMM_plan(K, v, fo, shard=(r, 2)).execute(x)
    assert np.array_equal(total, y)
    monkeypatch.setenv("FMMBEM_SHIFT_LANES", "0")
    monkeypatch.setenv("FMMBEM_SHIFT_ROT_MIN", "0")
    pl1 = fb.FMM_plan(K, v, fo)
    assert np.array_equal(pl1.execute(x), y)
    assert np.array_equal(pl1.expansions("M", p), M) and np.array_equal(pl1.expansions("L", p), L)
