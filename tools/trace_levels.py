#!/usr/bin/env python3
"""Per-dispatch listing of a rocprofv3 --kernel-trace run of tools/pipeline_probe.py: every mrp_ kernel of the LAST
mrp_phase_reads_many call in time order (name, grid, workgroup, duration, gap to the previous kernel's end), so the
levels of the resident pipeline can be read off.  usage: trace_levels.py <rocprof_out_dir> [n_last_dispatches]"""
import csv, glob, sys

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "mrp_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call: cut at the last mrp_cross_kernel that follows a traceback
last_tb = max((i for i, r in enumerate(rows[:-1]) if "traceback" in r["Kernel_Name"]), default=-1)
tbs = [i for i, r in enumerate(rows) if "traceback" in r["Kernel_Name"]]
start = tbs[-2] + 1 if len(tbs) >= 2 else 0
sel = rows[start:]
t0 = int(sel[0]["Start_Timestamp"])
prev_end = t0
tot = {}
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    print(f"{(s - t0) / 1e6:9.3f} ms  {name:34s} grid {int(r['Grid_Size_X']):>9d} wg {r['Workgroup_Size_X']:>4s}  {(e - s) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:8.1f} us")
    prev_end = max(prev_end, e)
    tot[name] = tot.get(name, 0.0) + (e - s) / 1e6
print("totals (ms):", {k: round(v, 3) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])})
print(f"span {(prev_end - t0) / 1e6:.3f} ms, busy {sum(tot.values()):.3f} ms")
