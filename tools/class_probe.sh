#!/bin/bash
# developer experiment: time each size class of the recursion kernel alone (results are NOT valid when skipping)
for skip in 6 5 3 0; do
  MRP_SKIP=$skip MRP_T_WIDE=${1:-1024} MRP_T_MID=${2:-512} MRP_T_NARROW=${3:-64} python bench.py --chunks 64 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d['roofline']['whole_step']; print('skip=$skip', 'sweep %.2f ms'%(w['sweep_ms']), d['config']['hmms_per_gpu'], d['config']['cells_per_gpu'])"
done
