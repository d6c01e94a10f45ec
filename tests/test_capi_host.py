"""CPU-only checks of the product's C-ABI library: it loads, exports every symbol include/fmmbem.h
declares, builds tree + lists on the host identically to the oracle, reports errors as codes, and has
no CPU execution path."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol(fb):
    header = open(os.path.join(ROOT, "include", "fmmbem.h")).read()
    declared = set(re.findall(r"^(?:int|void|const char \*)\s*(fmmbem_[a-z0-9_]+)\s*\(", header, flags=re.M))
    assert declared == set(fb.SYMBOLS)
    lib = ctypes.CDLL(fb.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.fmmbem_version() == 1


@pytest.mark.parametrize("r", [4, 5, 6, 8])
def test_host_lists_equal_oracle(fb, oracle_mod, r):
    v = fb.unit_sphere(r)
    assert np.array_equal(v, oracle_mod.unit_sphere(r))
    pl = fb.FMM_plan(fb.LaplaceSphericalBEM(10, 3), v, host_only=True)
    o = oracle_mod.Oracle(v)
    s, so = pl.stats(), o.stats()
    assert (s["n_boxes"], s["n_leaves"], s["n_levels"], s["near_nnz_total"], s["m2l_pairs"], s["m2m_ops"],
            s["l2l_ops"], s["p2p_pairs"]) == (so["boxes"], so["leaves"], so["levels"], so["near_nnz"],
                                              so["m2l_pairs"], so["m2m_ops"], so["l2l_ops"], so["p2p_pairs"])
    assert np.array_equal(pl.perm(), o.perm())
    assert np.array_equal(pl.pairs("m2l"), o.pairs("m2l"))          # same traversal order
    assert np.array_equal(pl.pairs("p2p"), o.pairs("p2p"))
    assert set(map(tuple, pl.pairs("m2m"))) == set(map(tuple, o.pairs("m2m")))
    assert set(map(tuple, pl.pairs("l2l"))) == set(map(tuple, o.pairs("l2l")))
    b, bo = pl.boxes(), o.boxes()
    for k in ("center", "side", "level", "leaf", "bb", "be"):
        assert np.array_equal(b[k], bo[k]), k
    # near sparsity pattern of a few rows
    rp, col, _ = (None, None, None)
    if r <= 6:
        rp, col, _ = o.near_csr()
        for row in (0, o.n // 3, o.n - 1):
            cols, _ = pl.near_row(row, values=False)
            assert np.array_equal(cols, col[rp[row]:rp[row + 1]])


def test_other_options_equal_oracle(fb, oracle_mod):
    v = fb.unit_sphere(5)
    for theta, ncrit in [(0.4, 64), (0.6, 32), (0.5, 125), (0.5, 10)]:
        opts = fb.FMMOptions()
        opts.set_mac_theta(theta)
        opts.set_max_per_box(ncrit)
        pl = fb.FMM_plan(fb.LaplaceSphericalBEM(5, 3), v, opts, host_only=True)
        o = oracle_mod.Oracle(v, theta=theta, ncrit=ncrit)
        assert np.array_equal(pl.pairs("m2l"), o.pairs("m2l"))
        assert np.array_equal(pl.pairs("p2p"), o.pairs("p2p"))
        assert pl.stats()["near_nnz_total"] == o.stats()["near_nnz"]


def test_no_cpu_execution_path(fb):
    pl = fb.FMM_plan(fb.LaplaceSphericalBEM(5, 3), fb.unit_sphere(3), host_only=True)
    with pytest.raises(fb.FmmBemError) as e:
        pl.execute(np.ones(pl.n))
    assert e.value.status == 2          # FMMBEM_ERR_NO_DEVICE


def test_error_codes(fb):
    v = fb.unit_sphere(3)
    with pytest.raises(fb.FmmBemError) as e:
        fb.FMM_plan(fb.LaplaceSphericalBEM(5, 2), v, host_only=True)         # invalid Gauss key
    assert e.value.status == 1
    with pytest.raises(ValueError):
        fb.LaplaceSphericalBEM(17, 3)
    o = fb.Options()
    fb.lib().fmmbem_options_default(ctypes.byref(o))
    o.host_only, o.evaluator = 1, 3                                           # no such evaluator
    h = ctypes.c_void_p()
    assert fb.lib().fmmbem_plan_create(ctypes.byref(o), len(v), v.ctypes.data_as(ctypes.c_void_p), None, ctypes.byref(h)) == 1
    with pytest.raises(fb.FmmBemError):
        fb.FMM_plan(fb.LaplaceSphericalBEM(5, 3), v, host_only=True, shard=(2, 2))


def test_shards_partition_the_leaves(fb):
    v = fb.unit_sphere(6)
    full = fb.FMM_plan(fb.LaplaceSphericalBEM(10, 3), v, host_only=True).stats()
    for world in (2, 3, 8):
        rows, nnz, pairs, prev_end = 0, 0, 0, 0
        for rank in range(world):
            s = fb.FMM_plan(fb.LaplaceSphericalBEM(10, 3), v, host_only=True, shard=(rank, world)).stats()
            assert s["owned_leaf_begin"] == prev_end
            prev_end = s["owned_leaf_end"]
            rows += s["owned_row_end"] - s["owned_row_begin"]
            nnz += s["near_nnz"]
            pairs += s["m2l_pairs_owned"]
        assert prev_end == full["n_leaves"] and rows == full["n_panels"] and nnz == full["near_nnz_total"]
        assert pairs >= full["m2l_pairs"]            # shared top-of-tree targets are recomputed per shard
        assert pairs < 1.2 * full["m2l_pairs"]


def test_rotation_m2l_work_list_items_are_runs_of_whole_targets():
    """The rotation M2L kernel's work list (HostPlan::build_rot_items): every owned M2L pair exactly once, a target's pairs
    contiguous and in traversal order, every item a run of WHOLE targets (no target is shared by two wavefronts), and few
    idle lanes in the items' last passes."""
    import fmm_bem_relaxed_amd as fb
    v = fb.unit_sphere(5)
    K = fb.LaplaceSphericalBEM(5, 3)
    for shard in ((0, 1), (1, 3)):
        plan = fb.FMM_plan(K, v, host_only=True, shard=shard)
        work, items = plan.pairs("m2l_work"), plan.pairs("m2l_items")
        st = plan.stats()
        assert len(work) == st["m2l_pairs_owned"] and len(items) == st["m2l_items"]
        assert items[0, 0] == 0 and items[-1, 1] == len(work) and np.array_equal(items[1:, 0], items[:-1, 1])
        # per target: the kernel's list restricted to it == the traversal's list restricted to it, same order
        lr = plan.pairs("m2l")
        ref = {int(t): [] for t in np.unique(work[:, 1])}
        for s, t in lr:
            if int(t) in ref:
                ref[int(t)].append(int(s))
        got, last = {}, None
        for s, t in work:
            t = int(t)
            if last != t:
                assert t not in got, "a target's pairs must be contiguous"
                got[t] = []
                last = t
            got[t].append(int(s))
        assert got == ref
        passes = 0
        for b, e in items:
            assert e > b
            if b > 0:
                assert work[b - 1, 1] != work[b, 1], "an item must begin with a new target"
            passes += (e - b + 63) // 64
        assert passes == st["m2l_passes"]
        assert st["m2l_pairs_owned"] / (64.0 * passes) > 0.8
        # the long cut (orders at one wavefront per SIMD) is another cut of the same list at target boundaries
        long_items = plan.pairs("m2l_items_long")
        assert long_items[0, 0] == 0 and long_items[-1, 1] == len(work) and np.array_equal(long_items[1:, 0], long_items[:-1, 1])
        for b, e in long_items:
            assert e > b and (b == 0 or work[b - 1, 1] != work[b, 1])
        assert len(long_items) <= len(items)
        plan.close()


def test_long_m2l_items_fill_the_chip_in_even_rounds():
    """HostPlan::build_rot_items, the cut for the orders that run one wavefront per SIMD: the items are long (up to 16 passes) and
    come in whole rounds of 1 024 -- ONE round where items of at most 16 passes cover the list (a shard, or an operator of this
    size: every item's first pass runs without operands fetched ahead, so fewer and longer items), two even rounds above that, never
    a handful beyond a round (measured: thirty items too many cost 0.92 ms against 0.54) -- and the lanes are full."""
    import fmm_bem_relaxed_amd as fb
    v = np.concatenate([fb.unit_sphere(8), fb.unit_sphere(8, center=(3.0, 0.0, 0.0))])
    plan = fb.FMM_plan(fb.LaplaceSphericalBEM(10, 3), v, host_only=True)
    work, short, long_items = plan.pairs("m2l_work"), plan.pairs("m2l_items"), plan.pairs("m2l_items_long")
    n_pass = lambda it: int(((it[:, 1] - it[:, 0] + 63) // 64).sum())
    assert len(work) > 64 * 1024 * 4                            # enough pairs for the rule to matter
    rounds = 1 if (len(work) + 63) // 64 <= 16 * 1024 else 2
    assert rounds == 1 and len(long_items) <= 1024 * rounds     # whole rounds over 1 024 SIMDs, not one item more
    per_item = (long_items[:, 1] - long_items[:, 0] + 63) // 64
    want = -(-(len(work) // 64) // (1024 * rounds))             # passes per item the rule aims at
    assert np.median(per_item) in (want, want + 1) and per_item.max() <= 17
    assert np.mean(per_item >= want) > 0.95                     # the few short ones sit in front of targets longer than an item
    assert len(work) / (64.0 * n_pass(long_items)) > 0.95 > len(work) / (64.0 * n_pass(short))     # fuller lanes than the short cut
    assert np.median((short[:, 1] - short[:, 0] + 63) // 64) <= 2
    plan.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_selective_exchange_lists_cover_what_a_shard_reads(world):
    """shard_upward = 2 (HostPlan::xsel_*): what shard r sends shard q is what q expects from r, nothing goes to oneself, and
    every multipole a shard's M2L work list reads is either its own, translated by every shard (a box spanning shards), or on
    its receive list -- counted here through fmmbem_plan_exchange_counts against the shard's own work list."""
    import fmm_bem_relaxed_amd as fb
    v = np.concatenate([fb.unit_sphere(5), fb.unit_sphere(4, center=(2.5, 0.0, 0.0))])
    K = fb.LaplaceSphericalBEM(6, 3)
    plans = [fb.FMM_plan(K, v, host_only=True, shard=(r, world), shard_upward=2) for r in range(world)]
    per = 2 * (6 * 7 // 2)                                    # doubles per box: one live expansion (all POTENTIAL), order 6
    counts = [pl.exchange_counts(6) for pl in plans]
    cut = plans[0].shard_rows(world)
    B = plans[0].boxes()
    owner = np.full(len(B["bb"]), -1)
    for r in range(world):
        owner[(B["bb"] >= cut[r]) & (B["be"] <= cut[r + 1])] = r
    for r in range(world):
        assert counts[r][0][r] == 0 and counts[r][1][r] == 0
        for q in range(world):
            assert counts[r][0][q] == counts[q][1][r] and counts[r][0][q] % per == 0
        src = np.unique(plans[r].pairs("m2l_work")[:, 0])
        for q in range(world):
            if q != r:                                        # at least the M2L sources private to q (plus children of spanning parents)
                assert counts[r][1][q] // per >= np.count_nonzero(owner[src] == q)
    # less than everything to everybody
    gather = fb.FMM_plan(K, v, host_only=True, shard=(0, world), shard_upward=True).exchange_doubles(6) * (world - 1)
    assert max(int(c[1].sum()) for c in counts) < gather
    for pl in plans:
        pl.close()


def test_plan_builder_threads_survive_concurrent_builds_and_fork():
    """The plan builder keeps a few worker threads for the life of the process (csrc/host_plan.cpp host_parallel).  Plans built
    from several threads at once (the shards of a device list are), in a forked child (no workers there: plain threads), and a
    child that exits normally (static state that names the parent's threads must not be torn down) all give the serial result."""
    code = r"""
import os, sys, threading, numpy as np
sys.path.insert(0, %r)
import fmm_bem_relaxed_amd as fb
v = fb.unit_sphere(7)
K = fb.LaplaceSphericalBEM(6, 3)
base = fb.FMM_plan(K, v, host_only=True).perm()
out = [None] * 4
def work(i): out[i] = fb.FMM_plan(K, v, host_only=True).perm()
th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
[t.start() for t in th]; [t.join() for t in th]
assert all(np.array_equal(o, base) for o in out)
for how in ("_exit", "exit"):
    sys.stdout.flush()
    pid = os.fork()
    if pid == 0:
        ok = np.array_equal(fb.FMM_plan(K, v, host_only=True).perm(), base)
        (os._exit if how == "_exit" else sys.exit)(0 if ok else 1)
    _, st = os.waitpid(pid, 0)
    assert os.WIFEXITED(st) and os.WEXITSTATUS(st) == 0, (how, st)
print("ok")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
