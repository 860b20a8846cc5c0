#!/bin/bash
# Development: the work queue's policies side by side on one device (short-queue threshold, lanes), 1 152 and 2 304 chunks.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qp; mkdir -p $O; cd $R
for n in 1152 2304; do
  for short in 640 1280 2560; do
    echo "== $n chunks, one call per device up to $short chunks"
    MRP_QUEUE_SHORT_CHUNKS=$short timeout -k 10 300 python3 tools/queue_long.py --chunks $n --runs 4 --skip-resident 1 2>&1 | grep "queue:" | cut -c1-120 || exit 1
  done
done
