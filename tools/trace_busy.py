#!/usr/bin/env python3
"""Device occupancy of the LAST mrp_phase_reads_many call of a rocprofv3 --kernel-trace run of tools/pipeline_probe.py when the
call's batches run concurrently: union of the kernels' busy intervals against the call's span, and summed durations per kernel.
usage: trace_busy.py <rocprof_out_dir> <n_groups>"""
import csv, glob, sys

d, groups = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "mrp_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tbs = [i for i, r in enumerate(rows) if "traceback" in r["Kernel_Name"]]
# the last call holds `groups` trace back kernels; it starts after the trace back before them
start = tbs[-groups - 1] + 1 if len(tbs) > groups else 0
sel = rows[start:]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
t0, t1 = iv[0][0], max(e for _, e in iv)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
# idle gaps of the device inside the call (longer than 0.3 ms), relative to the first kernel
gaps, ce = [], iv[0][1]
for s_, e_ in iv[1:]:
    if s_ > ce and s_ - ce > 300000:
        gaps.append(((ce - t0) / 1e6, (s_ - ce) / 1e6))
    ce = max(ce, e_)
print("idle gaps > 0.3 ms (at ms, length ms):", [(round(a, 1), round(b, 2)) for a, b in gaps])
tot = {}
for r in sel:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    tot[n] = tot.get(n, 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print(f"kernels {len(sel)}, span {(t1 - t0) / 1e6:.2f} ms, device busy (union) {busy / 1e6:.2f} ms = {100.0 * busy / (t1 - t0):.1f} %, summed durations {sum(tot.values()):.2f} ms")
print("summed durations (ms):", {k: round(v, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])})
# timeline: per 10 ms of the call, how many kernels run on average, what share of the device's thread slots their grids could fill
# (grid threads / (256 CUs x 2 048), at most 1 per kernel: an upper bound, LDS and registers limit it further) and who they are
CAP = 256 * 2048.0
BIN = 10e6
nb = int((t1 - t0) / BIN) + 1
run = [0.0] * nb; fill = [0.0] * nb; who = [dict() for _ in range(nb)]
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = min(1.0, float(r["Grid_Size_X"]) * float(r.get("Grid_Size_Y", 1) or 1) / CAP)
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mrp_", "").replace("_kernel", "")
    b = int((s - t0) / BIN)
    while b < nb and t0 + b * BIN < e:
        ov = min(e, t0 + (b + 1) * BIN) - max(s, t0 + b * BIN)
        if ov > 0:
            run[b] += ov / BIN; fill[b] += g * ov / BIN; who[b][n] = who[b].get(n, 0.0) + g * ov / BIN
        b += 1
print("timeline (10 ms bins): kernels running | thread slots their grids could fill (1 = the whole device) | largest shares")
for b in range(nb):
    top = sorted(who[b].items(), key=lambda kv: -kv[1])[:4]
    print(f"  {10 * b:4d} ms  {run[b]:5.1f}  {fill[b]:5.2f}  " + ", ".join(f"{k} {v:.2f}" for k, v in top))
