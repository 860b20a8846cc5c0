#!/usr/bin/env python3
"""Per stream (= hardware queue) of the LAST mrp_phase_reads_many call of a rocprofv3 --kernel-trace run of tools/pipeline_probe.py:
span, time with a kernel of the stream running, idle time between its kernels, and the summed durations by kernel family -- a
batch's critical path, as opposed to the device's occupancy (trace_busy.py).  usage: trace_batches.py <rocprof_out_dir> <n_groups>"""
import collections, csv, glob, sys

d, groups = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "mrp_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tbs = [i for i, r in enumerate(rows) if "traceback" in r["Kernel_Name"]]
start = tbs[-groups - 1] + 1 if len(tbs) > groups else 0
sel = rows[start:]
t0 = min(int(r["Start_Timestamp"]) for r in sel)
key = "Queue_Id" if "Queue_Id" in sel[0] else "Stream_Id"
by = collections.defaultdict(list)
for r in sel:
    by[r[key]].append(r)
print(f"columns: {list(sel[0].keys())}")
print(f"{len(sel)} kernels on {len(by)} queues; times in ms from the call's first kernel")
for q, rs in sorted(by.items(), key=lambda kv: int(kv[1][0]["Start_Timestamp"])):
    rs.sort(key=lambda r: int(r["Start_Timestamp"]))
    s0, e1 = int(rs[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rs)
    busy, cur_s, cur_e = 0, s0, int(rs[0]["End_Timestamp"])
    for r in rs[1:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > cur_e:
            busy += cur_e - cur_s; cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    fam = collections.Counter()
    for r in rs:
        n = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("mrp_", "").replace("_kernel", "")
        fam[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    top = ", ".join(f"{k} {v:.1f}" for k, v in fam.most_common(7))
    print(f"queue {q:>4s}: {len(rs):4d} kernels  first {(s0 - t0) / 1e6:6.1f}  last end {(e1 - t0) / 1e6:6.1f}  busy {busy / 1e6:6.1f}  idle {(e1 - s0 - busy) / 1e6:6.1f} | {top}")
