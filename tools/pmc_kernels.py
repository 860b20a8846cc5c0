#!/usr/bin/env python3
"""Per-kernel sums of the counters of a rocprofv3 --pmc run (all dispatches of a kernel name added up).
usage: pmc_kernels.py <rocprof_out_dir>"""
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.Counter())
n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k] += 1
for k in sorted(tot, key=lambda k: -sum(tot[k].values())):
    print(f"{k:40s} " + ", ".join(f"{c}={v:.4g}" for c, v in sorted(tot[k].items())))
