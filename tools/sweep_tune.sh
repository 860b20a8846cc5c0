#!/bin/bash
# developer tool: try block sizes for the recursion kernel size classes
for cfg in "$@"; do
  set -- $cfg
  MRP_T_WIDE=$1 MRP_T_MID=$2 MRP_T_NARROW=$3 python bench.py --chunks 64 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d['roofline']['whole_step']; print('$cfg', '%.3e'%d['value'], 'ms %.2f planes %.2f emis %.2f sweep %.2f'%(d['ms_per_step'], w['planes_ms'], w['emission_ms'], w['sweep_ms']))"
done
