#!/usr/bin/env python3
"""bench.py's cpu_baseline leg, as a process of its own: the oracle ("port" of the reference's OpenMP path) in FAITHFUL
mode -- serial near-field SpMV, serial M2M/L2L, OpenMP P2M/M2L/L2P, both expansions, the reference's parallel structure
(EvalInteractionLazySparse.hpp:120-168) -- built with the reference's flags (Makefile:16,26; oracle/Makefile refflags) and
timed as tests/scaling.cpp:44-54 does: 1 warm-up + 3 executes.  A child process so that OMP_NUM_THREADS /
OMP_PROC_BIND=close take effect before any OpenMP runtime starts, and nothing of the GPU process is involved.

  python tools/cpu_baseline.py laplace <spheres> <recursions> <p> <theta> <ncrit> <threads> <budget_s> [y_out.npy x_seed]
  python tools/cpu_baseline.py stokes  <recursions> <p> <theta> <ncrit> <threads> <budget_s> [y_out.npy x_seed]

With the two optional arguments the charges are numpy's default_rng(x_seed).random(n * dof) -- the vector bench.py runs the
GPU on -- and, when the full workload is executed, the result vector of its last execute is saved to y_out.npy: bench.py
compares the GPU's result with it element by element ("parity_vs_oracle_full"), instead of throwing the vector away.

Bounded: a sample of the workload (two-sphere r-1, a quarter of the panels; Stokes: r-2, a sixteenth) is timed first; the
full workload is timed too only when the sample says it fits the budget.  Prints one JSON object."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def physical_cores():
    try:
        out = subprocess.check_output(["lscpu", "-p=CORE,SOCKET"], text=True)
        cores = {ln for ln in out.splitlines() if ln and not ln.startswith("#")}
        return len(cores)
    except Exception:
        return None


def mem_cap_bytes():
    """What the oracle's near matrix may take: 60 % of MemAvailable (16 GB if /proc/meminfo cannot be read)."""
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                return 0.6 * float(ln.split()[1]) * 1024.0
    except Exception:
        pass
    return 16e9


def main():
    kind = sys.argv[1]
    args = sys.argv[2:]
    y_out, x_seed = None, 0
    if len(args) >= 2 and args[-2].endswith(".npy"):
        y_out, x_seed = args[-2], int(args[-1])
        args = args[:-2]
    threads, budget = int(args[-2]), float(args[-1])
    os.environ["OMP_NUM_THREADS"] = str(threads)
    os.environ["OMP_PROC_BIND"] = "close"
    os.environ["FMM_ORACLE_FLAVOR"] = "refflags"
    sys.path.insert(0, ROOT)
    import numpy as np
    from oracle import oracle as O

    last = [None]

    def timed(o, x, p):
        t0 = time.time()
        if kind == "laplace":
            o.build_near()
        elif O.lib().orc_stokes_build_near(o._h):
            raise MemoryError("near matrix")
        build_s = time.time() - t0
        t0 = time.time()
        y = o.matvec(x, p, faithful=True)                    # warm-up
        warm = time.time() - t0
        last[0] = y
        if warm * 3 > 4 * budget:                            # a runaway host: one execute is the sample
            return warm, build_s, 1
        ts = []
        for _ in range(3):
            t0 = time.time()
            last[0] = o.matvec(x, p, faithful=True)
            ts.append(time.time() - t0)
        return sum(ts) / 3, build_s, 3

    if kind == "laplace":
        spheres, r, p, theta, ncrit = int(args[0]), int(args[1]), int(args[2]), float(args[3]), int(args[4])

        def make(rr):
            v = np.concatenate([O.unit_sphere(rr, center=(3.0 * i, 0.0, 0.0)) for i in range(spheres)])
            return O.Oracle(v, K=3, theta=theta, ncrit=ncrit), np.random.default_rng(x_seed).random(len(v))
        sample_r, what = r - 1, "%d disjoint UnitSphere(r=%%d)" % spheres
    else:
        r, p, theta, ncrit = int(args[0]), int(args[1]), float(args[2]), int(args[3])

        def make(rr):
            v = O.red_blood_cell(rr)
            return O.StokesOracle(v, K=4, K_fine=19, mu=1e-3, theta=theta, ncrit=ncrit), np.random.default_rng(x_seed).random(len(v) * 3).reshape(len(v), 3)
        sample_r, what = r - 2, "RedBloodCell(r=%d)"

    o, x = make(sample_r)
    n_s = o.n
    t_s, build_s, reps = timed(o, x, p)
    near_bytes_s = o.stats()["near_nnz"] * (12 if kind == "laplace" else 76)      # values + column indices of the oracle's CSR
    o.close()
    n_full = n_s * 4 ** (r - sample_r)
    scale = n_full / n_s
    res = {"kind": "port", "note": "port, faster than the reference's own M2L (BASELINE.md section 4: the survey's reference build took "
                                   "1.4-2.0 s in M2L at N = 131 072 where this port takes 0.73 s): a conservative baseline; the calibration "
                                   "against the real reference that BASELINE.md 3.2 planned was never possible (Boost absent)",
           "unit": "matvecs/s", "cores": threads, "physical_cores": physical_cores(),
           "omp_proc_bind": "close", "flags": "-O3 -fopenmp -funroll-loops (the reference's, Makefile:16,26)",
           "mode": "faithful (serial SpMV/M2M/L2L, OpenMP P2M/M2L/L2P, both expansions)",
           "sample_n": n_s, "sample_s_per_matvec": t_s}
    est_full = t_s * scale
    cap = mem_cap_bytes()
    if 4 * est_full + build_s * scale <= budget and near_bytes_s * scale <= cap:
        o, x = make(r)
        t_f, build_f, reps_f = timed(o, x, p)
        o.close()
        if y_out:
            np.save(y_out, np.asarray(last[0]).reshape(-1))
        res.update(value=1.0 / t_f, extrapolated=False, full_s_per_matvec=t_f,
                   sample=(what + " N=%d p=%d, the bench workload itself: %.3f s/matvec (mean of %d after 1 warm-up; near-matrix "
                           "build %.1f s not counted); the quarter-size sample predicted %.3f s") % (r, n_full, p, t_f, reps_f, build_f, est_full))
    else:
        res.update(value=1.0 / est_full, extrapolated=True,
                   sample=(what + " N=%d p=%d: %.3f s/matvec (mean of %d after 1 warm-up; near-matrix build %.1f s not counted), "
                           "scaled x%.0f by O(N) to N=%d (the full workload: est. %.0f s for 4 executes + %.0f s build against a budget of %.0f s, "
                           "%.1f GB of near matrix against a cap of %.0f)")
                   % (sample_r, n_s, p, t_s, reps, build_s, scale, n_full, 4 * est_full, build_s * scale, budget, near_bytes_s * scale / 1e9, cap / 1e9))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
