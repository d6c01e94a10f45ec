#!/usr/bin/env python3
"""Per-kernel means of a rocprofv3 --pmc pass:  python tools/pmc_kernel.py <dir> [kernel-substring]"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("fmmbem::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if sub in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
