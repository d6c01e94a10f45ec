#!/usr/bin/env python3
"""bench.py -- FMM matvecs/s of the LaplaceBEM operator on MI355X, with the P2P roofline and a CPU baseline.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched through
torch.distributed.run (one rank per GPU, RCCL) -- by the driver, or, when WORLD_SIZE is not set, by bench.py itself
(launch_ranks: N child processes, started before this process touches the GPU).  WORLD_SIZE != --gpus is an error line
and exit code 2, never a silently different measurement.  Rank 0 prints ONE JSON line.

  step      one FMM matvec  y = A x  (near-field block SpMV + P2M/M2M/M2L/L2L/L2P at p=10), x and y
            resident in HBM, result replicated on every rank (one all-reduce per matvec when N > 1)
  workload  BASELINE.json metric "LaplaceBEM sphere N=1e6 p=10": two disjoint unit spheres, each
            Triangulation::UnitSphere(recursions=9), N = 1 048 576 panels,
            k=3, theta=0.5, ncrit=64, all panels POTENTIAL (first-kind operator, int G)
  value     matvecs/s of the whole job (K matvecs / max-over-ranks wall time)
  roofline  the HBM-bound P2P kernel (near_spmv): algorithmic bytes per launch / mean launch duration
            measured with HIP events on the launch stream during the timed steps
  cpu_baseline  the oracle ("port" of the reference's OpenMP path, faithful structure, the reference's compiler flags)
            timed on this box's host cores in a child process (tools/cpu_baseline.py): a quarter-size sample of the
            same geometry first, then the workload itself when the sample says it fits --cpu-budget seconds
            (`extrapolated` says which of the two `value` is)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6      # MI355X FP64 vector peak (AMD datasheet); the guide lists no FP64 figure


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--recursions", type=int, default=9, help="UnitSphere recursions per sphere")
    ap.add_argument("--spheres", type=int, default=2)
    ap.add_argument("--p", type=int, default=10)
    ap.add_argument("--theta", type=float, default=0.5)
    ap.add_argument("--ncrit", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-accuracy", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=150.0,
                    help="seconds of host time the cpu_baseline leg may spend on the full workload (else: quarter-size sample, "
                         "extrapolated); the full N = 1M Laplace workload takes ~45 s on a 128-core host, Stokes config 4 ~100 s")
    ap.add_argument("--collectives", choices=["default", "plain"], default="default",
                    help="N > 1: default = upward pass sharded by owner + one all-to-all of the multipoles each shard reads + "
                         "all-gather of the result slices; plain = upward pass repeated by every rank + ONE all-reduce of y "
                         "(BASELINE.json's wording; the fallback of the preflight)")
    ap.add_argument("--preflight-timeout", type=float, default=240.0,
                    help="N > 1: seconds the preflight (first contact with RCCL) may take before rank 0 prints an error line and the job exits")
    ap.add_argument("--matrix-free", action="store_true",
                    help="sparse_local = false: the near field recomputed every matvec (EvalInteractionLazy, SURVEY a8) "
                         "instead of the assembled matrix; the roofline object then reports FP64 flop/s, not HBM GB/s")
    ap.add_argument("--one-process", action="store_true",
                    help="--gpus N in ONE process: one plan over the device list 0..N-1 (fmmbem_options.n_devices; peer copies over xGMI "
                         "instead of torch.distributed collectives).  Not the form the SCALE driver launches; an alternative to compare with it")
    ap.add_argument("--near-stream-fraction", type=float, default=1.0,
                    help="share of the near-field pairs kept as a matrix (fmmbem_options.near_stream_fraction); < 1: the rest is "
                         "recomputed every matvec beside the streamed part (Stokes workloads; the roofline object then states "
                         "streamed bytes AND recomputed pairs)")
    ap.add_argument("--workload", choices=["laplace", "stokes_rbc", "stokes_rbc_traction"], default="laplace",
                    help="laplace: the BASELINE metric workload (default); stokes_rbc: SURVEY 8(d) config 4 "
                         "(StokesSphericalBEM velocity BC on RedBloodCell(r), p=8, k=4, K_fine=19, mu=1e-3)")
    return ap.parse_args()


def cpu_baseline(args, stokes, y_out=None):
    """The oracle in faithful mode on this box's host cores, in a child process (tools/cpu_baseline.py): the reference's
    flags, OMP_PROC_BIND=close, threads = the CPUs this process may run on; a quarter-size sample first, the full
    workload as well when that fits --cpu-budget seconds."""
    import subprocess
    threads = len(os.sched_getaffinity(0))
    try:                                                     # one thread per PHYSICAL core this process may use
        lines = subprocess.check_output(["lscpu", "-p=CPU,CORE,SOCKET"], text=True).splitlines()
        cpus = os.sched_getaffinity(0)
        cores = {tuple(ln.split(",")[1:]) for ln in lines if ln and not ln.startswith("#") and int(ln.split(",")[0]) in cpus}
        threads = len(cores) or threads
    except Exception:
        pass
    cmd = [sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py")]
    if stokes:
        cmd += ["stokes", str(args.recursions), str(args.p)]
    else:
        cmd += ["laplace", str(args.spheres), str(args.recursions), str(args.p)]
    cmd += [str(args.theta), str(args.ncrit), str(threads), str(args.cpu_budget)]
    if y_out:
        cmd += [y_out, str(X_SEED)]
    out = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout
    return json.loads(out.strip().splitlines()[-1])


STAGE = ["start"]                 # where this rank is: printed by the error line and by the watchdog


def stage(rank, what):
    """Breadcrumb on stderr (every rank) so that a hang or a crash on first contact with RCCL says where it happened."""
    STAGE[0] = what
    print("[bench rank %d] %s" % (rank, what), file=sys.stderr, flush=True)


def die(rank, world, msg, code=1):
    """One self-explaining JSON line instead of a bare traceback, then a hard exit (a rank stuck in a collective would
    otherwise keep the job alive until the launcher's timeout)."""
    print(json.dumps({"error": msg, "stage": STAGE[0], "rank": rank, "n_gpus": world, "metric": "FMM matvecs/s", "value": None}), flush=True)
    sys.stderr.flush()
    os._exit(code)


GATE_ROWS, GATE_SEED, X_SEED = 4096, 4096, 1234


def gate_rows(np, n, nrows=GATE_ROWS):
    """The target rows of the accuracy gate: `nrows` panels drawn with a fixed seed over the WHOLE vector (BASELINE.md 3.5;
    the reference forms its error over all bodies, tests/scaling.cpp:56-74 -- a seeded sample of them is what O(N^2) allows)."""
    return np.sort(np.random.default_rng(GATE_SEED).choice(n, size=min(nrows, n), replace=False)).astype(np.int32)


def direct_check(args, np, v, x, y, stokes, bc, nrows=GATE_ROWS, y_oracle=None):
    """north_star gate: relative L2 of the result against the O(N^2) Direct sum (include/Direct.hpp:99-125; the oracle as
    checker) on the seeded row sample.  Returns (gpu_vs_direct, oracle_vs_direct or None, rows used)."""
    from oracle import oracle as O
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:                                            # torchrun exports OMP_NUM_THREADS=1: the Direct sum runs on rank 0 only
        O.set_num_threads(max(1, len(os.sched_getaffinity(0)) // 2))
    n = len(v)
    rows = gate_rows(np, n, nrows // 4 if stokes else nrows)          # a Stokes row costs ~10x a Laplace row on the host
    if stokes:
        o = O.StokesOracle(v, K=4, K_fine=19, mu=1e-3, theta=args.theta, ncrit=args.ncrit, bc=bc)
        d = o.direct_rows(x.cpu().numpy().reshape(n, 3), rows)
        ys = y.cpu().numpy().reshape(n, 3)[rows]
        yo = None if y_oracle is None else y_oracle.reshape(n, 3)[rows]
    else:
        o = O.Oracle(v, K=3, theta=args.theta, ncrit=args.ncrit)
        d = o.direct_rows(x.cpu().numpy(), rows)
        ys = y.cpu().numpy()[rows]
        yo = None if y_oracle is None else y_oracle[rows]
    o.close()
    nd = np.linalg.norm(d)
    e = np.abs(ys - d).reshape(len(rows), -1)
    e2 = np.sort((e ** 2).sum(axis=1))[::-1]
    dist = {"row_rel_err_median": float(np.median(np.linalg.norm(e, axis=1) / np.linalg.norm(d.reshape(len(rows), -1), axis=1))),
            "row_rel_err_max": float(np.max(np.linalg.norm(e, axis=1) / np.linalg.norm(d.reshape(len(rows), -1), axis=1))),
            "share_of_squared_error_in_worst_1pct_rows": float(e2[:max(1, len(rows) // 100)].sum() / e2.sum())}
    return float(np.linalg.norm(ys - d) / nd), (None if yo is None else float(np.linalg.norm(yo - d) / nd)), len(rows), dist


def preflight(args, fb, make_op, x, v, stokes, bc, rank, world, dev):
    """N > 1, BEFORE the timed region and in the same processes: the first collectives this code ever issues on a backend
    should produce a verdict, not a hang inside the measurement.  In order of first execution:
      1. all_reduce of one double                       (communicator bring-up)
      2. the PLAIN path: upward pass repeated on every rank, ONE all_reduce(sum) of zero-padded result vectors
      3. rank 0 checks that result against the Direct sum; the verdict is all-reduced
      4. the DEFAULT path (unless --collectives plain): all_gather of the exchange count vectors and their symmetry
         (what r sends q == what q expects from r), then a matvec through all_to_all_single (uneven splits) +
         all_gather_into_tensor of the result slices; torch.equal against the plain result, verdict all-reduced
    A default path that disagrees (or raises) makes the run fall back to the plain path and says so; a plain path that
    disagrees with Direct is an error line and a non-zero exit.  Returns (the operator to time, the preflight record)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    info = {"backend": dist.get_backend(), "requested": args.collectives}
    stage(rank, "preflight 1/4: all_reduce of one double")
    one = torch.ones(1, dtype=torch.float64, device=dev)
    dist.all_reduce(one)
    if int(one.item()) != world:
        die(rank, world, "all_reduce of ones returned %r on %d ranks" % (one.item(), world))
    stage(rank, "preflight 2/4: plain path (replicated upward pass + all_reduce of y)")
    op_plain = make_op("plain")
    y_plain = op_plain.execute(x).clone()
    torch.cuda.synchronize()
    stage(rank, "preflight 3/4: Direct-sum check of the plain result on rank 0")
    flag = torch.zeros(2, dtype=torch.float64, device=dev)
    if rank == 0 and not args.no_accuracy:
        flag[0] = direct_check(args, np, v, x, y_plain, stokes, bc, nrows=512)[0]
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    info["direct_rel_l2"] = None if args.no_accuracy else float(flag[0].item())
    gate = 1e-3                      # not the north-star gate (reported as is): a wrong collective gives O(1), p = 8 Stokes 3e-5
    if not args.no_accuracy and not (flag[0].item() < gate):
        die(rank, world, "plain path (all_reduce) disagrees with the Direct sum: rel L2 %.3e" % flag[0].item())
    if args.collectives == "plain":
        info.update(paths_equal=None, used="plain")
        return op_plain, info
    stage(rank, "preflight 4/4: default path (count symmetry, all_to_all_single, all_gather_into_tensor)")
    verdict = torch.ones(1, dtype=torch.float64, device=dev)
    why = None
    op = None
    try:
        op = make_op("default")
        info["exchange_symmetric"] = op.check_exchange_symmetry() if op.split else None
        if info["exchange_symmetric"] is False:
            raise RuntimeError("exchange counts are not symmetric across ranks")
        y_def = op.execute(x)
        torch.cuda.synchronize()
        if not torch.equal(y_def, y_plain):
            verdict[0] = 0
            why = "default path result differs from the plain path on rank %d (max abs diff %.3e)" % (rank, float((y_def - y_plain).abs().max()))
    except Exception as e:                     # a collective that raised leaves the communicator in an unknown state:
        die(rank, world, "default collectives path raised: %r (rerun with --collectives plain)" % (e,))
    dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
    info["paths_equal"] = bool(verdict.item() == 1)
    if info["paths_equal"]:
        op_plain.plan.close()
        info["used"] = "default"
        return op, info
    if why:
        print("[bench rank %d] %s" % (rank, why), file=sys.stderr, flush=True)
    op.plan.close()
    info.update(used="plain", fallback="default path disagreed with the plain path (see stderr); timed the plain path")
    return op_plain, info


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start the N ranks as CHILD processes
    through torch.distributed.run, one per GPU, and hand back the launcher's exit code.  Called before this process has
    imported torch or touched the GPU (a process that has initialised HIP must not exec or fork GPU work), and as a child
    process, never an exec.  Rank 0's JSON line goes to the stdout the children inherit.  tools/launch_scale.sh is the same
    recipe for a shell."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without a launcher: starting %d ranks: %s" % (args.gpus, args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    one_process = args.one_process and args.gpus > 1
    if one_process:
        if env_world not in (None, "1"):
            raise SystemExit("--one-process runs without a launcher")
        env_world = "1"
        os.environ["WORLD_SIZE"] = "1"
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args))
    if env_world is not None and int(env_world) != args.gpus and not one_process:
        # a launcher that started W ranks for --gpus N would otherwise time W GPUs and be read as N (or the reverse)
        STAGE[0] = "launch"
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"error": "--gpus %d but the launcher set WORLD_SIZE=%s: start one rank per GPU (python -m "
                                       "torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..., tools/launch_scale.sh) "
                                       "or run `python bench.py --gpus %d` bare and let it start its own ranks"
                                       % (args.gpus, env_world, args.gpus, args.gpus, args.gpus),
                              "stage": "launch", "rank": 0, "n_gpus": int(env_world), "metric": "FMM matvecs/s", "value": None}), flush=True)
        sys.exit(2)
    import datetime
    import threading
    import numpy as np
    import torch
    import torch.distributed as dist
    import fmm_bem_relaxed_amd as fb

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        if world > 1:
            if rank != 0:
                time.sleep(3.0)                              # rank 0 speaks first: a launcher that sees any rank fail kills the others
                os._exit(4)
            die(rank, world, "bench.py needs a GPU: the product has no CPU execution path", 4)
        raise SystemExit("bench.py needs a GPU: the product has no CPU execution path")
    # one rank per GPU over RCCL.  FMMBEM_BENCH_BACKEND=gloo is a rehearsal aid for a one-GPU box: the ranks then share
    # device 0 and the collectives go through the host (numbers from such a run mean nothing).
    backend = os.environ.get("FMMBEM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    watchdog = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        stage(rank, "init_process_group(%s)" % backend)
        # a stuck collective fails after two minutes instead of the launcher's limit
        tmo = datetime.timedelta(seconds=120)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
        watchdog = threading.Timer(args.preflight_timeout, lambda: die(
            rank, world, "preflight did not finish in %.0f s" % args.preflight_timeout, 3))
        watchdog.daemon = True
        watchdog.start()

    stokes = args.workload in ("stokes_rbc", "stokes_rbc_traction")
    traction = args.workload == "stokes_rbc_traction"       # config 4 read literally: the double layer (stresslet), every target TRACTION
    bc = None
    if stokes:
        # ---- config 4: one red blood cell (examples/BEM/Triangulation.hpp:184-255), identity rotation, no shift ----
        v = fb.red_blood_cell(args.recursions)
        if args.p == 10:
            args.p = 8
        K = fb.StokesSphericalBEM(args.p, 4, 1e-3)
        K.set_Kfine(19)
    else:
        # ---- workload: `spheres` disjoint unit spheres, centres 3 apart on the x axis ----
        parts = [fb.unit_sphere(args.recursions, center=(3.0 * i, 0.0, 0.0)) for i in range(args.spheres)]
        v = np.concatenate(parts) if len(parts) > 1 else parts[0]
        K = fb.LaplaceSphericalBEM(args.p, 3)
    n = len(v)
    dof = 3 if stokes else 1
    opts = fb.FMMOptions()
    opts.set_mac_theta(args.theta)
    opts.set_max_per_box(args.ncrit)
    opts.sparse_local = not args.matrix_free
    opts.near_stream_fraction = args.near_stream_fraction
    if traction:
        bc = np.ones(len(v), dtype=np.uint8)

    build = {}

    class OneProcess:
        """One plan over the device list (MultiDevice in csrc/plan.hip) behind the attributes the rest of this file reads."""
        split, y_collective = False, "peer copies of tree-order slices"

        def __init__(self):
            ndev = torch.cuda.device_count()
            self.devices = [i % ndev for i in range(args.gpus)]        # a box with fewer GPUs names them again (a rehearsal, not a measurement)
            self.plan = fb.FMM_plan(K, v, opts, bc=bc, devices=self.devices)

        def execute(self, xx, out=None):
            return self.plan.execute_torch(xx, out=out)

    def make_op(which):
        t_b = time.time()
        if one_process:
            o = OneProcess()
            build[id(o)] = time.time() - t_b
            return o
        if which == "plain":                                 # upward pass repeated, ONE all_reduce(sum) of y per matvec
            o = fb.ShardedFMM(K, v, opts, bc=bc, device=local_rank, shard_upward=False, y_collective="allreduce")
        else:                                                # distributed.py's defaults (the FMMBEM_* switches apply)
            o = fb.ShardedFMM(K, v, opts, bc=bc, device=local_rank)
        build[id(o)] = time.time() - t_b
        return o

    # the charges: numpy's seeded stream, so that the cpu_baseline child regenerates the SAME x without torch
    x = torch.from_numpy(np.random.default_rng(X_SEED).random(n * dof)).to(dev)
    y = torch.empty_like(x)
    pre = None
    if world > 1:
        try:
            op, pre = preflight(args, fb, make_op, x, v, stokes, bc, rank, world, dev)
        except Exception as e:
            die(rank, world, "preflight raised: %r" % (e,))
        watchdog.cancel()
        torch.cuda.empty_cache()
        stage(rank, "preflight done: timing the %s path" % pre["used"])
    else:
        op = make_op("default")
    build_s = build[id(op)]
    plan = op.plan

    def step():
        op.execute(x, out=y)                                 # local matvec + the collective(s) of fmm-bem-relaxed_amd/distributed.py

    for _ in range(args.warmup):
        step()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # HIP events around the near-field kernel only (the roofline kernel), no added syncs: events around every stage put a
    # packet between every two kernels and cost 85 us per matvec (profiles/r03b_p10_timeline.md); the per-stage breakdown
    # (`stage_ms`) comes from a second, fully instrumented pass OUTSIDE the timed region
    plan.set_timing(2)
    stage(rank, "timed region: %d steps" % args.steps)
    try:
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
    except Exception as e:
        if world == 1:
            raise
        die(rank, world, "timed region raised: %r" % (e,))
    st_near = plan.stats()
    plan.set_timing(1)
    for _ in range(min(args.steps, 10)):
        step()
    fence()
    st = plan.stats()                                        # every stage bracketed: the breakdown, not the throughput
    plan.set_timing(False)
    per_rank, replicas_equal = None, None
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # OUTSIDE the timed region: a few more steps with the collectives bracketed by device syncs, so that a SCALE line
        # explains itself -- per rank: kernel stage times of its shard, seconds spent in collectives, bytes received
        op.profile, op.collective_s, c0 = True, 0.0, op.calls
        for _ in range(5):
            step()
        fence()
        op.profile = False
        up_b, y_b = op.exchange_bytes()
        mine = [st["ms_total"], st["ms_near"], st["ms_p2m"] + st["ms_m2m"], st["ms_m2l"], st["ms_l2l"] + st["ms_l2p"],
                op.collective_s / (op.calls - c0) * 1e3, float(up_b), float(y_b), float(st["near_nnz"]), float(st["m2l_pairs_owned"])]
        allr = [torch.zeros(len(mine), dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(allr, torch.tensor(mine, dtype=torch.float64, device=dev))
        keys = ("kernels_ms", "near_ms", "upward_ms", "m2l_ms", "downward_ms", "collectives_ms_synced", "multipole_bytes_in",
                "result_bytes_in", "near_nnz", "m2l_pairs")
        per_rank = [dict(zip(keys, t.tolist())) for t in allr]
        # every rank must hold the same replicated result
        sig = torch.stack([y.sum(), y.abs().max(), y[::4097].sum()])
        lo, hi = sig.clone(), sig.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_equal = bool(torch.equal(lo, hi))
    ms_per_step = elapsed / args.steps * 1e3

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the P2P kernel on this rank's shard (SURVEY.md section 8d "Algorithmic bytes, P2P") ----
    rows = st["owned_row_end"] - st["owned_row_begin"]
    # Stokes: the 3x3 block of a panel pair is symmetric and is streamed as 6 values, not the 9 of SURVEY 8d
    hybrid = st["near_recomputed_pairs"] > 0
    # a hybrid plan streams only the blocks it stores (near_bytes); the recomputed pairs cost arithmetic, stated beside the bytes
    p2p_bytes = (st["near_bytes"] if hybrid else st["near_nnz"] * 8 * (6 if stokes else 1)) + n * 8 * dof + rows * 8 * dof
    near_ms = st_near["ms_near"]                              # measured live over the timed region, on the launch stream
    p2p_gbs = p2p_bytes / (near_ms * 1e-3) / 1e9 if near_ms > 0 else 0.0
    traffic = None
    try:                                                     # PMC traffic comes from a separate rocprofv3 --pmc pass
        prof = json.load(open(os.path.join(ROOT, "profiles", "pmc_near_spmv_stokes.json" if stokes else "pmc_near_spmv.json")))
        import hashlib
        sha = hashlib.sha256(open(os.path.join(ROOT, "fmm-bem-relaxed_amd", "csrc", "kernels_near.hip"), "rb").read()).hexdigest()
        # a figure measured on another build of the near-field kernels is stale: report none rather than that
        # ... and so is one measured on another configuration of the near field (the committed passes are the streamed plans of
        # the laplace and stokes_rbc workloads: tools/round_profile.sh, tools/stokes_profile.sh)
        if (prof.get("n_panels") == n and prof.get("n_gpus") == world and prof.get("kernels_near_sha256") == sha and not hybrid
                and not traction and not args.matrix_free):
            traffic = prof.get("hbm_bytes_per_launch")
    except Exception:
        pass
    P = args.p
    n_exp = (7 if traction else 4) if stokes else 1                              # live expansions per box
    # reference operation count: the double sum of LaplaceSpherical::M2L, 4 FMAs = 8 flop per complex multiply-add ...
    m2l_ref_flops = n_exp * st["m2l_pairs_owned"] * 8.0 * (P * (P + 1) // 2) * P * P
    # ... and what the kernel executes: rotation / axial translation / rotation (csrc/m2l_rot.hpp), per pair and expansion
    rot_fma = 4 * sum(n * n + n + 1 for n in range(1, P)) + sum((P - k) ** 2 * (2 if k else 1) for k in range(P))
    rot_mul = 4 * 2 * sum(P - m for m in range(1, P)) + 2 * (P * P - 1)        # z rotations (2 mul + 2 FMA per coefficient), powers of rho
    rot_fma += 4 * 2 * sum(P - m for m in range(1, P))
    rot_on = st["m2l_kernel"] == 1                           # which kernel the plan launched (fmmbem_stats), not a guess from the environment
    # the double-sum kernels (p > 12, FMMBEM_M2L_ROT=0) execute 280 of the reference's 400 real FMAs per output and source at
    # p = 10 (real radial table x complex phase, kernels_m2l.hip header): 0.70 of the reference's count
    m2l_flops = n_exp * st["m2l_pairs_owned"] * (2.0 * rot_fma + rot_mul) if rot_on else 0.70 * m2l_ref_flops
    m2l_tflops = m2l_flops / (st["ms_m2l"] * 1e-3) / 1e12 if st["ms_m2l"] > 0 else 0.0

    if args.matrix_free:
        # a matrix-free P2P has no meaningful HBM figure; report FP64 flop/s with its count,
        # (3 rsqrt + 20 flop) * K per far panel pair (Laplace; the few near-regime pairs cost more and are not counted)
        kq = 4 if stokes else 3
        mf_flops = st["near_nnz"] * kq * 23.0 * (4 if stokes else 1)
        # What bounds it is FP64 instruction issue, and of that the reciprocal square roots cost most per instruction: measured on
        # this part (tools/microbench/rsq64.hip, profiles/r04m_rsq64_issue_rate.txt) v_rsq_f64 issues 9.09e12 lane-operations/s
        # against 33.4e12 for v_fma_f64.  Ceiling = per quadrature point one v_rsq_f64 and ~16 other FP64 instructions (the
        # difference vector, r^2, the Newton step, the accumulation; kernels_near.hip mf_far) at those rates; `frac` = ceiling time
        # / launch time.  The FMA-peak view of SURVEY 8(d) is kept beside it, labelled: it is not the bound.
        RSQ_RATE, FMA_RATE, OTHER = 9.09e12, 33.4e12, 16
        points = st["near_nnz"] * kq
        ceil_ms = points * (1.0 / RSQ_RATE + OTHER / FMA_RATE) * 1e3
        mf = {"kernel": "mf_sweep (P2P recomputed)", "bound": "fp64 instruction issue (v_rsq_f64 at 0.27 of the FMA rate, measured)",
              "achieved": points / (near_ms * 1e-3) / 1e12, "peak": points / (ceil_ms * 1e-3) / 1e12, "unit": "T quadrature points/s",
              "frac": ceil_ms / near_ms, "traffic": None, "panel_pairs": st["near_nnz"], "launch_ms": near_ms, "ceiling_ms": ceil_ms,
              "fma_peak_view": {"algorithmic_flops_per_launch": mf_flops, "tflops": mf_flops / (near_ms * 1e-3) / 1e12,
                                "frac_of_fma_peak_not_the_bound": mf_flops / (near_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS}}
    out = {
        "metric": (("FMM matvecs/s (StokesBEM red blood cell, %s) + achieved HBM GB/s on P2P" % ("TRACTION targets: double layer" if traction else "velocity BC")) if stokes else
                   "FMM matvecs/s (LaplaceBEM sphere N=1e6 p=10) + achieved HBM GB/s on P2P"),
        "value": args.steps / elapsed, "unit": "matvecs/s", "n_gpus": args.gpus if one_process else world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("StokesSphericalBEM " + ("TRACTION targets (stresslet)" if traction else "velocity BC") + ", RedBloodCell(r=%d), N=%d panels (%d unknowns), p=%d, k=4, "
                                "K_fine=19, mu=1e-3, theta=%g, ncrit=%d; %d GPU(s)"
                                % (args.recursions, n, 3 * n, P, args.theta, args.ncrit, world)) if stokes else
                               ("LaplaceSphericalBEM, %d disjoint UnitSphere(r=%d), N=%d panels, p=%d, k=3, theta=%g, "
                                "ncrit=%d, all POTENTIAL; target leaves sharded over %d GPU(s), %s"
                                % (args.spheres, args.recursions, n, P, args.theta, args.ncrit, args.gpus if one_process else world,
                                   ("ONE process, one plan over the device list: x, the multipoles each shard reads and the result slices by peer copies"
                                    if one_process else
                                    ((("1 all-to-all of the multipoles each shard reads + " if op.plan.exchange_mode == 2
                                       else "1 all-gather of the multipoles + ") if op.split else "") +
                                     ("1 all-gather of the result slices per matvec" if op.y_collective == "allgather" else
                                      "1 all-reduce of y per matvec"))))),
                   "n_panels": n, "p": P, "theta": args.theta, "ncrit": args.ncrit, "near_nnz": st["near_nnz_total"], "m2l_pairs": st["m2l_pairs"],
                   "boxes": st["n_boxes"], "leaves": st["n_leaves"],
                   # which parent->child L2L edges ran: the plan's default is the COMPLETE list; the reference's lazy list omits
                   # `l2l_reference_omitted` of them on this tree (0: the two rules are one list here, as on every mesh the reference's
                   # generators produce -- FMMOptions.reference_l2l selects the reference's list where they differ)
                   "l2l_rule": "complete", "l2l_reference_omitted": st["l2l_reference_omitted"],
                   "near_stream_fraction": args.near_stream_fraction},
        "roofline": mf if args.matrix_free else
                    {"kernel": "near_spmv (P2P)", "bound": "hbm", "achieved": p2p_gbs, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": p2p_gbs / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": p2p_bytes, "launch_ms": near_ms,
                     "timed_launches": st_near["timed_executes"],
                     **({"near_stream_fraction": args.near_stream_fraction, "streamed_bytes": st["near_bytes"],
                         "recomputed_pairs": st["near_recomputed_pairs"], "recomputed_quadrature_points": st["near_recomputed_pairs"] * (4 if stokes else 3),
                         "listed_near_regime_pairs": st["near_side_entries"],
                         "note": "hybrid near field: launch_ms spans the streaming kernel and the recompute kernel running side by side; "
                                 "`achieved` counts the streamed bytes only, so it falls short of the HBM rate of the streaming kernel itself "
                                 "when the recompute kernel is the longer of the two"} if hybrid else {})},
        "roofline_m2l": {"kernel": {1: "m2l_rot", 2: "m2l (double sum)", 3: "m2l_small (double sum, lanes = sources)"}.get(st["m2l_kernel"], "?"), "bound": "fp64 vector FMA", "achieved": m2l_tflops,
                         "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": m2l_tflops / FP64_PEAK_TFLOPS,
                         "executed_flops_per_launch": m2l_flops, "launch_ms": st["ms_m2l"],
                         "reference_flops_per_launch": m2l_ref_flops,
                         "reference_equivalent_tflops": m2l_ref_flops / (st["ms_m2l"] * 1e-3) / 1e12 if st["ms_m2l"] > 0 else 0.0},
        "stage_ms": {k[3:]: st[k] for k in ("ms_total", "ms_gather", "ms_near", "ms_scatter", "ms_p2m", "ms_m2m",
                                            "ms_mh", "ms_m2l", "ms_l2l", "ms_l2p")},
        "stage_ms_note": "a second pass of %d matvecs with HIP events around every stage, outside the timed region (the events "
                         "themselves lengthen such a matvec; `ms_per_step` is the un-instrumented figure)" % min(args.steps, 10),
        "per_rank": per_rank, "replicas_equal": replicas_equal, "preflight": pre,
        "one_process": ({"devices": op.devices, "transport": "hipMemcpyPeerAsync ordered by events (no collective library)"} if one_process else None),
        "collectives": None if world == 1 else {"upward": ("all-to-all of the multipoles each shard reads" if op.plan.exchange_mode == 2 else "all-gather of multipoles") if op.split else "none (upward pass repeated)",
                                                "result": op.y_collective},
        "plan_build_s": build_s, "near_assemble_s": st["build_assemble_ms"] * 1e-3,
        "host_lists_s": st["build_host_ms"] * 1e-3,
    }

    # The oracle is only ever touched in this CPU leg of the bench (rank 0): as the checker of the result just
    # computed (Direct sum on a row sample) and as the timed CPU baseline -- never inside the timed region.
    cpu_leg = world == 1 and not args.no_cpu_baseline
    y_oracle = None
    if cpu_leg and traction:
        # the oracle restates the reference, whose far field for this operator is wrong: there is no CPU FMM to time
        out["cpu_baseline"] = None
    elif cpu_leg:
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            y_path = os.path.join(td, "y_oracle.npy")
            out["cpu_baseline"] = cpu_baseline(args, stokes, y_path)
            if os.path.exists(y_path):                       # the child ran the workload itself: its result vector, same x
                y_oracle = np.load(y_path)
        if out["cpu_baseline"]:
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    # full-vector parity: the GPU result against the oracle's FMM result on the bench workload itself, every element
    if y_oracle is not None:
        yg = y.cpu().numpy()
        out["parity_vs_oracle_full"] = float(np.linalg.norm(yg - y_oracle) / np.linalg.norm(y_oracle))
        out["parity_vs_oracle_full_max_abs_over_max"] = float(np.abs(yg - y_oracle).max() / np.abs(y_oracle).max())
    else:
        out["parity_vs_oracle_full"] = None
    if not args.no_accuracy:                                  # rank 0, any N (the result is replicated); independent of the baseline switch
        g_d, o_d, nr, rowdist = direct_check(args, np, v, x, y, stokes, bc, y_oracle=y_oracle)
        gate = 1e-6
        out["rel_l2_vs_direct_sample"] = g_d
        out["accuracy_gate"] = {"rows": nr, "rows_seed": GATE_SEED, "drawn_over": "the whole vector (numpy default_rng choice without replacement)",
                                "gpu_vs_direct": g_d, "oracle_vs_direct": o_d, "gate": gate, "pass": bool(g_d < gate), **rowdist,
                                "note": ("below the north-star gate" if g_d < gate else
                                         "reference level: %.3e -- above 1e-6; the oracle's FMM (the reference's algorithm on the CPU) "
                                         "sits at %s on the same rows: the truncation error of p = %d, theta = %g on this tree (rows in "
                                         "coarse leaves that clip a cap of the surface carry most of it, DESIGN.md section 5), not a "
                                         "defect of the device path" % (g_d, "%.3e" % o_d if o_d is not None else "n/a (no CPU leg)", P, args.theta))}
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
