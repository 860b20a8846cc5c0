#!/bin/bash
# Kernel trace of ONE 96-chunk batch, level by level (run on the GPU box): tools/level_trace.sh NAME [ENV=VALUE ...]
# writes gpurun_out/lt/levels_NAME.txt and prints the per-kernel totals.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lt; mkdir -p $O
name=$1; shift
( for kv in "$@"; do export "$kv"; done; export MRP_PHASE_GROUPS=1
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/t_$name -o t -- python3 $R/tools/pipeline_probe.py --chunks 96 --repeat 3 --check-host 0 > $O/probe_$name.log 2>&1 ) || { echo "$name failed"; grep -v "^[EWI]2026" $O/probe_$name.log | tail -5; exit 1; }
python3 $R/tools/trace_levels.py $O/t_$name > $O/levels_$name.txt
rm -rf $O/t_$name
tail -2 $O/levels_$name.txt | cut -c1-900
