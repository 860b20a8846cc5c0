#!/bin/bash
# section clocks of the prune chain wave (alt_lib/libmargin_rphmm_clk2.so = build with -DPRUNE_EXP_CLOCK2), first hmm of every level
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/clk2; mkdir -p $O
name=${1:-run}; shift
( cd $R; for kv in "$@"; do export "$kv"; done
  MRP_LIB_OVERRIDE=$R/alt_lib/libmargin_rphmm_clk2.so MRP_TIMING=1 MRP_PHASE_GROUPS=1 timeout -k 10 300 python3 tools/pipeline_probe.py --chunks 96 --repeat 2 --check-host 0 > $O/$name.log 2> $O/$name.err ) || { tail -5 $O/$name.err; exit 1; }
grep -B1 "prune clocks" $O/$name.err | tail -20 | grep -A1 "level: \(96\|192\|1285\) hmms" > $O/${name}_sections.txt
python3 - <<PY
import re
rows = open("$O/${name}_sections.txt").read().split("\n")
lv = None
for r in rows:
    m = re.search(r"level: (\d+) hmms (\d+) cols (\d+) cells", r)
    if m: lv = m.groups(); continue
    m = re.search(r"prune clocks.*?: (.*)", r)
    if m and lv:
        v = [int(x) for x in m.group(1).split()]
        ncol = sum(v[8:12])
        if not ncol: continue
        names = ["0 kept merge", "1 keep-all", "2 hist", "3 cutoff", "4 select", "5 fence/ties", "6 tail", "7 barrier"]
        print(f"level {lv[0]} hmms: first hmm {ncol} columns (keep-all {v[8]}, sorted {v[9]}, hist {v[10]}+{v[11]}): " +
              ", ".join(f"{n} {v[i] / ncol:.0f}" for i, n in enumerate(names)) + f" | total {sum(v[:8]) / ncol:.0f} cycles per column")
PY
