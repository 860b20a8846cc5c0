#!/bin/bash
# work / barrier-wait clocks per role of the prune kernel (alt_lib/libmargin_rphmm_clk1.so = build with -DPRUNE_EXP_CLOCK), first hmm of every level
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/clk1; mkdir -p $O
name=${1:-run}; shift
( cd $R; for kv in "$@"; do export "$kv"; done
  MRP_LIB_OVERRIDE=$R/alt_lib/libmargin_rphmm_clk1.so MRP_TIMING=1 MRP_PHASE_GROUPS=1 timeout -k 10 300 python3 tools/pipeline_probe.py --chunks 96 --repeat 2 --check-host 0 > $O/$name.log 2> $O/$name.err ) || { tail -5 $O/$name.err; exit 1; }
grep -B1 "prune clocks" $O/$name.err | tail -20 | grep -A1 "level: \(96\|192\|1285\) hmms" > $O/${name}_roles.txt
python3 - <<PY
import re
lv = None
for r in open("$O/${name}_roles.txt").read().split("\n"):
    m = re.search(r"level: (\d+) hmms (\d+) cols (\d+) cells", r)
    if m: lv = m.groups(); continue
    m = re.search(r"prune clocks.*?: (.*)", r)
    if m and lv:
        v = [int(x) for x in m.group(1).split()]
        names = ["chain", "lists1", "lists2", "tables", "bins g0", "bins g1"]
        print(f"level {lv[0]} hmms, first hmm, kilocycles work/wait: " + ", ".join(f"{n} {v[2 * i] / 1e3:.0f}/{v[2 * i + 1] / 1e3:.0f}" for i, n in enumerate(names)))
PY
