#!/usr/bin/env python3
"""Where a batch's stream sits idle: per hardware queue of the LAST mrp_phase_reads_many call of a rocprofv3 --kernel-trace run of
tools/pipeline_probe.py, the gaps between one kernel's end and the next kernel's start, summed by (kernel before -> kernel after) over
all queues, and the longest ones listed.  usage: trace_gaps.py <rocprof_out_dir> <n_groups> [min_gap_ms]"""
import collections, csv, glob, sys

d, groups = sys.argv[1], int(sys.argv[2])
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
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


def fam(r):
    return r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("mrp_", "").replace("_kernel", "")


by = collections.defaultdict(list)
for r in sel:
    by[r[key]].append(r)
pair_ms, pair_n, longest = collections.Counter(), collections.Counter(), []
tot_gap = 0.0
for q, rs in by.items():
    if len(rs) < 20:
        continue  # the copy / structure stream of a batch
    rs.sort(key=lambda r: int(r["Start_Timestamp"]))
    cur_e, last = int(rs[0]["End_Timestamp"]), rs[0]
    for r in rs[1:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > cur_e:
            g = (s - cur_e) / 1e6
            tot_gap += g
            pair_ms[(fam(last), fam(r))] += g
            pair_n[(fam(last), fam(r))] += 1
            if g >= min_gap:
                longest.append((g, (cur_e - t0) / 1e6, q, fam(last), fam(r), r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", "?")))
        if e > cur_e:
            cur_e, last = e, r
print(f"{len(sel)} kernels; idle time between kernels on the batches' main queues: {tot_gap:.1f} ms summed over {sum(1 for v in by.values() if len(v) >= 20)} queues")
print("by transition (kernel before -> kernel after): summed ms, count, mean ms")
for k, v in pair_ms.most_common(25):
    print(f"  {k[0]:>18s} -> {k[1]:<18s} {v:8.1f} ms  {pair_n[k]:4d}  {v / pair_n[k]:6.2f}")
print(f"longest gaps (>= {min_gap} ms): length, at, queue, before -> after (grid of the kernel after)")
for g in sorted(longest, reverse=True)[:60]:
    print(f"  {g[0]:7.2f} ms at {g[1]:7.1f}  q{g[2]:>3s}  {g[3]} -> {g[4]} ({g[5]})")
if len(sys.argv) > 4:  # every kernel that starts before <until_ms>, in start order: start, end, queue, family, grid
    until = float(sys.argv[4])
    print(f"kernels starting before {until} ms:")
    for r in sel:
        s = (int(r["Start_Timestamp"]) - t0) / 1e6
        if s > until:
            break
        print(f"  {s:8.2f} .. {(int(r['End_Timestamp']) - t0) / 1e6:8.2f}  q{r[key]:>3s}  {fam(r):18s} grid {r.get('Grid_Size', r.get('Grid_Size_X', '?'))}")
if len(sys.argv) > 5:  # every kernel of the main queue that finishes <rank>-th (0: first to finish), as trace_levels.py lists a call
    rank = int(sys.argv[5])
    mains = sorted(((max(int(r["End_Timestamp"]) for r in rs), q) for q, rs in by.items() if len(rs) >= 20))
    q = mains[min(rank, len(mains) - 1)][1]
    rs = sorted(by[q], key=lambda r: int(r["Start_Timestamp"]))
    print(f"queue {q} (finishes {rank}-th of {len(mains)}):")
    prev_end, tot = int(rs[0]["Start_Timestamp"]), {}
    for r in rs:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"  {(s - t0) / 1e6:9.3f} ms  {fam(r):18s} grid {int(r.get('Grid_Size', r.get('Grid_Size_X', 0))):>9d}  {(e - s) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:8.1f} us")
        prev_end = max(prev_end, e)
        tot[fam(r)] = tot.get(fam(r), 0.0) + (e - s) / 1e6
    print("  totals (ms):", {k: round(v, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}, f"busy {sum(tot.values()):.1f}")
