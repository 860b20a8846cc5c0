#!/bin/bash
# instruction-cache counters of every kernel of one 96-chunk batch (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ic; rm -rf $O; mkdir -p $O
export MRP_PHASE_GROUPS=1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $O/t1 -o t -- python3 $R/tools/pipeline_probe.py --chunks 96 --repeat 1 --check-host 0 > $O/p1.log 2>&1 || { tail -5 $O/p1.log; exit 1; }
python3 $R/tools/pmc_kernels.py $O/t1 > $O/pmc_icache.txt
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/t2 -o t -- python3 $R/tools/pipeline_probe.py --chunks 96 --repeat 1 --check-host 0 > $O/p2.log 2>&1 || { tail -5 $O/p2.log; exit 1; }
python3 $R/tools/pmc_kernels.py $O/t2 > $O/pmc_ifetch.txt
rm -rf $O/t1 $O/t2
cut -c1-260 $O/pmc_icache.txt | head -12; cut -c1-260 $O/pmc_ifetch.txt | head -12
