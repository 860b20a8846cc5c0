#!/bin/bash
# instruction-fetch counters of one 96-chunk batch (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ic; rm -rf $O; mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_INSTS_SALU\|SQ_INST_CYCLES[A-Z_]*\|SQ_BUSY_CYCLES" $O/avail.txt | sort -u > $O/names.txt
cat $O/names.txt | tr '\n' ' '; echo
export MRP_PHASE_GROUPS=1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES --output-format csv -d $O/t1 -o t -- python3 $R/tools/pipeline_probe.py --chunks 96 --repeat 1 --check-host 0 > $O/p1.log 2>&1 || { tail -5 $O/p1.log; exit 1; }
python3 $R/tools/r02_summary.py pmc $O/t1 > $O/pmc_icache.txt
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/t2 -o t -- python3 $R/tools/pipeline_probe.py --chunks 96 --repeat 1 --check-host 0 > $O/p2.log 2>&1 || { tail -5 $O/p2.log; exit 1; }
python3 $R/tools/r02_summary.py pmc $O/t2 > $O/pmc_ifetch.txt
rm -rf $O/t1 $O/t2
grep -c "" $O/pmc_icache.txt $O/pmc_ifetch.txt
