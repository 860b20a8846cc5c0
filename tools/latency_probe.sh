#!/bin/bash
# Development: where one small call's time goes: 1 chunk of the headline kind and 640 chunks of ~130 sites (configs[2]).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lat; mkdir -p $O
( cd $R && MRP_TIMING=1 timeout -k 10 200 python3 tools/pipeline_probe.py --chunks 1 --repeat 6 --check-host 0 > $O/one.log 2> $O/one.err ) || exit 1
( cd $R && MRP_TIMING=1 timeout -k 10 200 python3 tools/pipeline_probe.py --chunks 640 --sites 130 --repeat 6 --check-host 0 > $O/c2.log 2> $O/c2.err ) || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/t_one -o t -- python3 $R/tools/pipeline_probe.py --chunks 1 --repeat 4 --check-host 0 > $O/one_trace.log 2>&1 || exit 1
python3 $R/tools/trace_levels.py $O/t_one > $O/levels_one.txt
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/t_c2 -o t -- python3 $R/tools/pipeline_probe.py --chunks 640 --sites 130 --repeat 4 --check-host 0 > $O/c2_trace.log 2>&1 || exit 1
python3 $R/tools/trace_busy.py $O/t_c2 8 > $O/busy_c2.txt
rm -rf $O/t_one $O/t_c2
grep "^run" $O/one.log $O/c2.log | cut -c1-120; tail -3 $O/levels_one.txt | cut -c1-600; tail -3 $O/busy_c2.txt | cut -c1-900
