#!/bin/bash
# kernel traces of a resident call and of a queue call of the same 576 chunks (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qt; rm -rf $O; mkdir -p $O
for leg in resident queue; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t_$leg -o t -- python3 $R/tools/queue_timeline.py --only $leg > $O/$leg.log 2>&1 || { grep -v "^[EWI]2026" $O/$leg.log | tail -5; exit 1; }
  python3 $R/tools/trace_busy.py $O/t_$leg 8 > $O/busy_$leg.txt
  rm -rf $O/t_$leg
  echo "== $leg"; cut -c1-700 $O/busy_$leg.txt
done
