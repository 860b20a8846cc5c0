#!/bin/bash
# Development: the work queue's policies side by side on one device: the library's choice against explicit batch sizes / lanes.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qp; mkdir -p $O; cd $R
for n in 2304 4608; do
  echo "== $n chunks, library default"
  timeout -k 10 300 python3 tools/queue_long.py --chunks $n --runs 3 --skip-resident 1 2>&1 | grep "queue:" | cut -c1-120 || exit 1
  for cfg in "1152 1" "576 2" "288 4"; do
    set -- $cfg
    echo "== $n chunks, batches of $1, $2 lane(s)"
    MRP_QUEUE_LANES=$2 timeout -k 10 300 python3 tools/queue_long.py --chunks $n --runs 3 --batch $1 --skip-resident 1 2>&1 | grep "queue:" | cut -c1-120 || exit 1
  done
done
