"""Summarise rocprofv3 CSV output of `bench.py` runs into the text files committed under profiles/.
The replay launches of the timed region are the dispatches with the largest grid of each kernel name
(the host pipeline's small launches during the build phase are ignored).
usage: profile_summary.py <rocprof_out_dir> [<rocprof_out_dir> ...]"""
import collections, csv, glob, sys

def rows_of(d, pattern):
    out = []
    for f in glob.glob(d + "/**/" + pattern, recursive=True):
        out += list(csv.DictReader(open(f)))
    return out

for d in sys.argv[1:]:
    print(f"== {d}")
    tr = rows_of(d, "*kernel_trace.csv")
    if tr:
        by = collections.defaultdict(list)
        for r in tr:
            if "mrp_" not in r["Kernel_Name"]:
                continue
            g = int(r["Grid_Size_X"])
            by[(r["Kernel_Name"].split("(")[0], g, r["Workgroup_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        best = {}
        for (name, g, wg), v in by.items():
            if len(v) <= 64:  # replay launches repeat steps+warmup times; build-phase grids are mostly unique/small
                best.setdefault(name, []).append((g, wg, v))
        print("kernel trace, replay launches (grid, workgroup, calls, avg ms, min ms, max ms):")
        for name, lst in sorted(best.items()):
            lst.sort(key=lambda x: -x[0])
            for g, wg, v in lst[:3]:
                if g < 50000:
                    continue
                print(f"  {name:32s} grid {g:>10d} wg {wg:>4s} calls {len(v):3d} avg {sum(v)/len(v):8.3f} min {min(v):8.3f} max {max(v):8.3f}")
    pm = rows_of(d, "*counter_collection.csv")
    if pm:
        by = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in pm:
            if "mrp_" not in r["Kernel_Name"]:
                continue
            by[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]), r["Workgroup_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("PMC, replay launches (mean per dispatch):")
        for (name, g, wg), dd in sorted(by.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
            n = len(next(iter(dd.values())))
            if g < 50000 or n > 64:
                continue
            vals = ", ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(dd.items()))
            print(f"  {name:32s} grid {g:>10d} wg {wg:>4s} n={n:2d}: {vals}")
