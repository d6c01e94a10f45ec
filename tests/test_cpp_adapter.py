"""The header-only C++ adapter (include/fmmbem/FMM_plan.hpp) compiles with plain g++ against the C ABI and
drives a plan the way the reference's programs do (tests/scaling.cpp:41-54, examples/LaplaceBEM.cpp:203-232)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT


REF_BEM = "/root/reference/examples/BEM"          # present in the build container only


def _build(tmp_path, src="adapter_example", extra=()):
    exe = str(tmp_path / (src + ("_%08x" % (hash(tuple(extra)) & 0xffffffff) if extra else "")))
    libdir = os.path.join(ROOT, "fmm-bem-relaxed_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "tests", "cpp"), *extra,
                           os.path.join(ROOT, "tests", "cpp", src + ".cpp"), "-o", exe,
                           "-L" + libdir, "-lfmmbem_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_driver_sequence_compiles(tmp_path, gpu_available):
    """examples/LaplaceBEM.cpp:163-291's call sequence (plan, flipped-BC right-hand-side plan,
    Preconditioners::Diagonal over plan.source_begin()/source_end(), relaxed GMRES) compiles and links against the
    adapter header with the test-side solver ..."""
    exe = _build(tmp_path, "laplace_bem_sequence")
    if not gpu_available:
        r = subprocess.run([exe, "3", "8", "1e-5", "1"], capture_output=True, text=True)
        assert r.returncode == 2 and "no HIP device" in r.stdout


@pytest.mark.skipif(not os.path.isdir(REF_BEM), reason="reference tree not present (GPU box)")
def test_driver_sequence_compiles_with_the_references_own_solver_headers(tmp_path, gpu_available):
    """... and with the REFERENCE's GMRES.hpp / SolverOptions.hpp / Preconditioner.hpp / BLAS.hpp and GMRES_Stokes.hpp,
    read in place and unmodified: everything those files touch (FMM_plan typedefs, execute, kernel().set_p,
    source_begin/source_end, K(s, s), Vec<3,double> arithmetic) exists with the reference's meaning."""
    for src in ("laplace_bem_sequence", "stokes_bem_sequence"):
        exe = _build(tmp_path, src, extra=("-DUSE_REFERENCE_SOLVER", "-I" + REF_BEM))
        if not gpu_available:
            r = subprocess.run([exe, "3", "8", "1e-5", "1"], capture_output=True, text=True)
            assert r.returncode == 2 and "no HIP device" in r.stdout


def test_device_solver_call_sites_compile(tmp_path, gpu_available):
    """`#define GMRES fmmbem::GMRES` (-DUSE_DEVICE_SOLVER) turns the driver sequence's call sites into the device-resident
    solve, beside the test-side solver's names and -- in the build container -- beside the REFERENCE's GMRES.hpp, whose
    unqualified GMRES(plan, ...) stays the reference's (no ADL capture); FGMRES / InnerSolverPC / the Stokes order rule compile."""
    variants = [("laplace_bem_sequence", ("-DUSE_DEVICE_SOLVER",)), ("device_solver_sequence", ("-DX",))]
    if os.path.isdir(REF_BEM):
        variants.append(("laplace_bem_sequence", ("-DUSE_DEVICE_SOLVER", "-DUSE_REFERENCE_SOLVER", "-I" + REF_BEM)))
    for src, extra in variants:
        exe = _build(tmp_path, src, extra=extra)
        if not gpu_available:
            r = subprocess.run([exe, "3", "8", "1e-5", "1"], capture_output=True, text=True)
            assert r.returncode == 2 and "no HIP device" in r.stdout


REF_INCLUDE = "/root/reference/include"          # KernelTraits.hpp, executor/INITM.hpp, INITL.hpp: Boost-free


def test_kernel_classes_meet_the_kernel_contract(tmp_path, gpu_available):
    """kernel/KernelSkeleton.hpp:62-212: the adapter's kernel classes have init_multipole / init_local / P2M / M2M / M2L / L2L /
    L2P with the reference's argument lists (tests/cpp/traits_contract.cpp drives them the way tests/single_level.cpp does); in
    the build container the REFERENCE's own include/KernelTraits.hpp, read in place, static_asserts is_valid_fmm on them."""
    variants = [()]
    if os.path.isdir(REF_INCLUDE):
        variants.append(("-DUSE_REFERENCE_TRAITS", "-I" + REF_INCLUDE))
    for extra in variants:
        exe = _build(tmp_path, "traits_contract", extra=extra)
        if not gpu_available:
            r = subprocess.run([exe], capture_output=True, text=True)
            assert r.returncode == 2 and "no HIP device" in r.stdout


@pytest.mark.gpu
def test_single_operator_chain_through_the_adapter(tmp_path):
    """INITM, P2M, M2M, M2L, INITL, L2L, L2P of both kernel classes against K(t, s) * c summed over the sources; and the same
    program built in the container around the reference's KernelTraits.hpp / INITM.hpp / INITL.hpp (oracle/_ref, when present)"""
    exes = [_build(tmp_path, "traits_contract")]
    ref = os.path.join(ROOT, "oracle", "_ref", "traits_contract_ref")
    if os.path.exists(ref):
        exes.append(ref)
    for exe in exes:
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        vals = {ln.split()[0]: ln.split() for ln in r.stdout.splitlines()}
        assert float(vals["laplace"][3]) < 2e-6 and float(vals["stokes"][3]) < 2e-6
        assert vals["refused"] == ["refused", "1", "unsupported", "1"]
        if exe == ref:
            assert vals["traits"] == ["traits", "is_valid_fmm", "1", "1"]


def test_adapter_compiles_and_reports_missing_device(tmp_path, gpu_available):
    exe = _build(tmp_path)
    if gpu_available:
        pytest.skip("GPU present: covered by the gpu-marked test")
    r = subprocess.run([exe, "3"], capture_output=True, text=True)
    assert r.returncode == 2 and "no HIP device" in r.stdout       # fails loudly, no CPU fallback


OPT_FLAGS = ["-theta", "0.4", "-ncrit", "32", "-printtree", "-eval", "FMM", "-lazy_eval"]


def test_options_surface_compiles_and_parses(tmp_path, gpu_available):
    """get_options(argc, argv), FMMOptions::MAC() / MAC_ / print_tree (include/FMMOptions.hpp:19-106) exist with the
    reference's meaning in the stand-alone header; the flags are parsed before any device is needed."""
    exe = _build(tmp_path, "options_and_defaults")
    r = subprocess.run([exe, "3"] + OPT_FLAGS, capture_output=True, text=True)
    first = r.stdout.splitlines()[0].split()
    # theta, ncrit, printtree, lazy, evaluator == FMM, MAC accepts (0,0,0)-(2.1,0,0) with radii .5 at theta .4 (d^2 = 4.41 <= 6.25: no)
    assert first[0] == "options" and float(first[1]) == 0.4 and first[2:6] == ["32", "1", "1", "1"]
    assert first[6:] == ["0", "0"]
    d = subprocess.run([exe, "3"], capture_output=True, text=True).stdout.splitlines()[0].split()
    assert float(d[1]) == 0.5 and d[2:6] == ["64", "0", "1", "1"] and d[6:] == ["1", "0"]    # defaults; 4.41 > 4: accepted, 3.61: not
    if not gpu_available:
        assert r.returncode == 2 and "no HIP device" in r.stdout


@pytest.mark.gpu
def test_adapter_defaults_grow_with_set_p_and_accept_the_traction_rhs_plan(tmp_path, oracle_mod):
    """FMM_plan<K>(K, panels, opts) with no p_max: sized for K's order; kernel().set_p above it re-creates the plan (the
    reference's set_p resizes its tables, LaplaceSpherical.hpp:119-128) and earlier orders give the same bits afterwards;
    the right-hand-side step of StokesBEM.cpp:266-270 -- every panel TRACTION, default arguments -- is accepted."""
    exe = _build(tmp_path, "options_and_defaults")
    r = subprocess.run([exe, "5"] + OPT_FLAGS, capture_output=True, text=True, check=True)
    out = [ln.split() for ln in r.stdout.strip().splitlines()]
    v = oracle_mod.unit_sphere(5)
    o = oracle_mod.Oracle(v, theta=0.4, ncrit=32)
    one = np.ones(o.n)
    lap = [ln for ln in out if ln[0] == "laplace"]
    assert [(int(ln[2]), int(ln[3])) for ln in lap] == [(6, 6), (4, 6), (9, 9), (6, 9)]
    for ln in lap:
        y = o.matvec(one, int(ln[2]))
        assert abs(float(ln[4]) - y.sum()) / abs(y.sum()) < 1e-12
        assert abs(float(ln[5]) - y[0]) / abs(y[0]) < 1e-11 and abs(float(ln[6]) - y[-1]) / abs(y[-1]) < 1e-11
    assert lap[0][4:] == lap[3][4:]                                # p = 6 before and after the plan grew: the same digits
    tr = [ln for ln in out if ln[0] == "traction"][0]
    assert int(tr[3]) == 8 and abs(float(tr[4]) - 1.0) < 2e-2 and float(tr[5]) < 2e-2


@pytest.mark.gpu
def test_adapter_matches_oracle(tmp_path, oracle_mod):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "5"], capture_output=True, text=True, check=True)
    v = oracle_mod.unit_sphere(5)
    o = oracle_mod.Oracle(v)
    one = np.ones(o.n)
    out = [ln.split() for ln in r.stdout.strip().splitlines()]
    lines = [ln for ln in out if ln[0].isdigit()]
    assert [int(ln[1]) for ln in lines] == [12, 10, 5]
    for ln in lines:
        n, p = int(ln[0]), int(ln[1])
        y = o.matvec(one, p)
        assert n == o.n
        assert abs(float(ln[2]) - y.sum()) / abs(y.sum()) < 1e-12
        assert abs(float(ln[3]) - y[0]) / abs(y[0]) < 1e-11 and abs(float(ln[4]) - y[-1]) / abs(y[-1]) < 1e-11
    st = [ln for ln in out if ln[0] == "stokes"][0]
    so = oracle_mod.StokesOracle(v, K=3, K_fine=19, mu=1e-3)
    f = np.zeros((so.n, 3)); f[:, 0] = 1.0
    u = so.matvec(f, 8)
    for c in range(3):
        assert abs(float(st[3 + c]) - u[:, c].sum()) <= 1e-11 * np.abs(u).sum()
    tr = [ln for ln in out if ln[0] == "traction"][0]             # the double layer of the closed sphere on u = (1,0,0): 4 pi u
    assert tr[1] == "accepted" and abs(float(tr[2]) - 1.0) < 2e-2 and float(tr[3]) < 2e-2
    t14 = [ln for ln in out if ln[0] == "traction14"][0]         # orders above 12: the double-sum M2L, slot by slot
    assert t14[1] == "accepted" and abs(float(t14[2]) - 1.0) < 2e-2 and float(t14[3]) < 2e-2


@pytest.mark.gpu
def test_device_solver_sequences_match_python_solver(tmp_path, fb):
    """fmmbem::GMRES on the Stokes plan and fmmbem::FGMRES with InnerSolverPC, from C++, against solver.py on the same
    problems: schedule, iterations, solution sums."""
    import torch
    exe = _build(tmp_path, "device_solver_sequence", extra=("-DX",))
    r = subprocess.run([exe, "4", "8", "1e-5"], capture_output=True, text=True, check=True)
    out = r.stdout.splitlines()

    def block(tag):
        i = out.index(tag + " begin")
        ps = []
        for ln in out[i + 1:]:
            if ln.startswith("it:"):
                ps.append(int(ln.split("fmm_req_p:")[1]))
            elif ln.startswith("Final residual"):
                return ps, int(ln.split()[4])
    v = fb.unit_sphere(4)
    n = len(v)
    # Stokes
    K = fb.StokesSphericalBEM(8, 4, 1e-3)
    K.set_Kfine(19)
    plan = fb.FMM_plan(K, v, p_max=8)
    b = torch.zeros(3 * n, dtype=torch.float64, device="cuda")
    b[0::3] = 1.0
    so = fb.SolverOptions(residual=1e-5, max_iters=100, max_p=8, restart=100)
    log = []
    x, it, res = fb.gmres(plan, torch.zeros_like(b), b, so, log=log, stokes=True)
    ps, its = block("stokes")
    assert its == it and ps == [p for _, p, _ in log][:len(ps)]
    st = [ln for ln in out if ln.startswith("stokes sum:")][0].split()
    sums = x.view(n, 3).sum(0).cpu().numpy()
    assert np.allclose([float(t) for t in st[2:5]], sums, rtol=1e-8, atol=1e-8 * abs(sums).max())
    assert int(st[8]) == log[-1][1]                              # the kernel object ends at the last order set
    plan.close()
    # Laplace FGMRES with the two inner-solver preconditioners
    K = fb.LaplaceSphericalBEM(8, 3)
    plan = fb.FMM_plan(K, v, p_max=8)
    rhs = fb.FMM_plan(fb.LaplaceSphericalBEM(8, 3), v, bc=np.ones(n, dtype=np.uint8), p_max=8)
    b = rhs.execute_torch(torch.ones(n, dtype=torch.float64, device="cuda"))
    for kind, cls in enumerate((fb.LocalInnerSolver, fb.BlockDiagonal)):
        M = cls(fb, fb.LaplaceSphericalBEM(8, 3), v)
        K.set_p(8)
        log = []
        x, it, res = fb.fgmres(plan, torch.zeros_like(b), b, so, M, log=log)
        ps, its = block("fgmres %d" % kind)
        assert its == it and ps == [p for _, p, _ in log][:len(ps)]
        s_cpp = float([ln for ln in out if ln.startswith("fgmres %d sum:" % kind)][0].split()[3])
        assert abs(s_cpp - float(x.sum())) <= 1e-8 * abs(s_cpp)
        z_cpp = float([ln for ln in out if ln.startswith("functor %d sum:" % kind)][0].split()[3])
        assert abs(z_cpp - float(M(b).sum())) <= 1e-8 * abs(z_cpp)
    assert "shift refused %d" % 6 in out                       # FMMBEM_ERR_UNSUPPORTED


@pytest.mark.gpu
@pytest.mark.parametrize("pc", [0, 1])
@pytest.mark.parametrize("device_solver", [False, True])
def test_driver_sequence_matches_python_solver(tmp_path, fb, oracle_mod, pc, device_solver):
    """The C++ driver sequence on the GPU against the same solve through solver.py: identical order schedule,
    iteration count, solution to rounding; K(s, s) and K(t, s) of the kernel object against the oracle's entries.
    device_solver: the same source with -DUSE_DEVICE_SOLVER (GMRES -> fmmbem::GMRES, the Arnoldi process in HBM)."""
    import torch
    exe = _build(tmp_path, "laplace_bem_sequence", extra=("-DUSE_DEVICE_SOLVER",) if device_solver else ())
    r = subprocess.run([exe, "5", "12", "1e-5", str(pc)], capture_output=True, text=True, check=True)
    out = r.stdout.splitlines()
    ps_cpp = [int(ln.split("fmm_req_p:")[1]) for ln in out if ln.startswith("it:")]
    final = [ln for ln in out if ln.startswith("Final residual")][0].split()
    its_cpp = int(final[4])
    v = fb.unit_sphere(5)
    n = len(v)
    K = fb.LaplaceSphericalBEM(12, 3)
    plan = fb.FMM_plan(K, v, p_max=12)
    rhs = fb.FMM_plan(fb.LaplaceSphericalBEM(12, 3), v, bc=np.ones(n, dtype=np.uint8), p_max=12)
    b = rhs.execute_torch(torch.ones(n, dtype=torch.float64, device="cuda"))
    log = []
    M = fb.Diagonal(plan) if pc else None
    if pc:                                              # the reference applies the TREE-order reciprocals to vectors in
        perm = torch.from_numpy(plan.perm().astype(np.int64)).cuda()      # original order: so does the C++ run
        M.recip = M.recip[perm]
    x, its, res = fb.gmres(plan, torch.zeros(n, dtype=torch.float64, device="cuda"), b,
                           fb.SolverOptions(residual=1e-5, max_p=12), M=M, log=log)
    assert its == its_cpp and [p for _, p, _ in log][:len(ps_cpp)] == ps_cpp
    s_cpp = float([ln for ln in out if ln.startswith("solution sum")][0].split()[2])
    assert abs(s_cpp - float(x.sum())) <= 1e-9 * abs(s_cpp)
    o = oracle_mod.Oracle(v)
    selfs = [float(t) for t in [ln for ln in out if ln.startswith("self entry")][0].split()[2:]]
    perm = plan.perm()
    assert abs(selfs[0] - o.kernel_entries(np.array([0]), np.array([0]))[0]) <= 1e-13 * abs(selfs[0])
    assert abs(selfs[1] - o.kernel_entries(perm[:1], perm[:1])[0]) <= 1e-13 * abs(selfs[1])
    pair = float([ln for ln in out if ln.startswith("pair entry")][0].split()[2])
    assert abs(pair - o.kernel_entries(np.array([0]), np.array([1]))[0]) <= 1e-13 * abs(pair)
