#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into the small files committed under profiles/.

  python tools/prof_summary.py kernel-stats <dir> <out.md>        (from --kernel-trace --stats)
  python tools/prof_summary.py pmc <fetch_dir> <write_dir> <out.json> <n_panels> <n_gpus>
  python tools/prof_summary.py timeline <dir> <out.md> [first kernel [matvecs to skip at the end]]
                                                   (from --kernel-trace: one matvec, dispatch by dispatch)
PMC units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming
read, so the read side of the streaming near_spmv kernel is doubled.
"""
import collections
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def near_source_sha():
    """Identity of the kernel the traffic figure belongs to: sha256 of csrc/kernels_near.hip (bench.py drops
    `roofline.traffic` when the current source differs)."""
    return hashlib.sha256(open(os.path.join(ROOT, "fmm-bem-relaxed_amd", "csrc", "kernels_near.hip"), "rb").read()).hexdigest()


def short(name):
    name = name.replace("fmmbem::(anonymous namespace)::", "").replace("void ", "").replace("fmmbem::", "")
    return name.split("(")[0]


def kernel_stats(d, out):
    f = glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    with open(out, "w") as o:
        o.write("| kernel | calls | avg us | min us | max us | % of GPU time |\n|---|---|---|---|---|---|\n")
        for r in rows:
            o.write("| %s | %s | %.1f | %.1f | %.1f | %s |\n" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                              float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    print(open(out).read())


def pmc_mean(d, counter):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def pmc(fetch_dir, write_dir, out, n_panels, n_gpus):
    fe, wr = pmc_mean(fetch_dir, "FETCH_SIZE"), pmc_mean(write_dir, "WRITE_SIZE")
    res = {"n_panels": int(n_panels), "n_gpus": int(n_gpus), "kernels": {}, "kernels_near_sha256": near_source_sha()}
    try:
        res["commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        res["commit"] = None                          # the GPU box has no .git: filled in when the file is copied to profiles/
    for k in sorted(set(fe) | set(wr)):
        f_kib, nf = fe.get(k, (0.0, 0))
        w_kib, nw = wr.get(k, (0.0, 0))
        res["kernels"][k] = {"FETCH_SIZE_KiB_raw": f_kib, "WRITE_SIZE_KiB": w_kib, "dispatches": max(nf, nw)}
    ns = next((v for k, v in res["kernels"].items() if k.startswith("near_spmv")), None)
    if ns:
        # gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide streaming read -> x2 (guide, HBM section)
        res["hbm_bytes_per_launch"] = (2.0 * ns["FETCH_SIZE_KiB_raw"] + ns["WRITE_SIZE_KiB"]) * 1024.0
        res["note"] = "near_spmv kernel: (2*FETCH_SIZE + WRITE_SIZE) KiB, separate --pmc passes"
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


def timeline(d, out, first="gather_x", last_n=8, skip_last=0):
    """One matvec as the GPU saw it (from --kernel-trace): per dispatch its start relative to the matvec's first kernel, its
    duration and the idle gap in front of it, averaged over the last `last_n` matvecs of the trace (a matvec = from one
    `first` kernel to the next; FMMBEM graphs and fused kernels change the dispatch list, not this definition)."""
    f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(f))]
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2].startswith(first)]
    if len(starts) < last_n + 2:
        raise SystemExit("fewer than %d matvecs in the trace" % (last_n + 2))
    skip_last = int(skip_last)                   # bench.py: the last 10 matvecs of a run are the fully instrumented pass
    runs = [rows[starts[k]:starts[k + 1]] for k in range(len(starts) - 1 - last_n - skip_last, len(starts) - 1 - skip_last)]
    n = min(len(r) for r in runs)
    runs = [r for r in runs if len(r) == n]
    with open(out, "w") as o:
        o.write("| # | kernel | start us | duration us | gap before us |\n|---|---|---|---|---|\n")
        tot_k = tot_g = 0.0
        for i in range(n):
            st = sum(r[i][0] - r[0][0] for r in runs) / len(runs) / 1e3
            du = sum(r[i][1] - r[i][0] for r in runs) / len(runs) / 1e3
            gap = sum((r[i][0] - r[i - 1][1]) if i else 0 for r in runs) / len(runs) / 1e3
            tot_k += du
            tot_g += gap
            o.write("| %d | %s | %.1f | %.1f | %.1f |\n" % (i, runs[0][i][2], st, du, gap))
        span = sum(r[-1][1] - r[0][0] for r in runs) / len(runs) / 1e3
        o.write("\n%d dispatches per matvec; kernels %.1f us + gaps %.1f us = span %.1f us (mean of %d matvecs)\n" % (n, tot_k, tot_g, span, len(runs)))
    print(open(out).read())


if __name__ == "__main__":
    if sys.argv[1] == "kernel-stats":
        kernel_stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "timeline":
        timeline(sys.argv[2], sys.argv[3], *(sys.argv[4:5] or ["gather_x"]), 8, *(sys.argv[5:6] or [0]))
    else:
        pmc(*sys.argv[2:7])
