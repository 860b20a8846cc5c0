#!/bin/bash
# device occupancy of one resident call with concurrent batches (run on the GPU box): tools/busy_trace.sh NAME CHUNKS [ENV=VALUE ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/busy; mkdir -p $O
name=$1; chunks=$2; shift; shift
( for kv in "$@"; do export "$kv"; done
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/t_$name -o t -- python3 $R/tools/pipeline_probe.py --chunks $chunks --repeat 3 --check-host 0 > $O/probe_$name.log 2>&1 ) || { echo "$name failed"; tail -5 $O/probe_$name.log; exit 1; }
python3 $R/tools/trace_busy.py $O/t_$name 8 > $O/busy_$name.txt
rm -rf $O/t_$name
grep "^run" $O/probe_$name.log; cut -c1-1200 $O/busy_$name.txt
