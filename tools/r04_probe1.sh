#!/bin/bash
# Round 4, first look (GPU box): section clocks of the prune chain wave (clk2 build), counters of every kernel of ONE 96-chunk
# batch in situ.  Output under gpurun_out/r04a/.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. section clocks of the chain wave, first (longest) hmm of every level
( cd $R && MRP_LIB_OVERRIDE=$R/alt_lib/libmargin_rphmm_clk2.so MRP_TIMING=1 MRP_PHASE_GROUPS=1 timeout -k 10 300 python3 tools/pipeline_probe.py --chunks 96 --repeat 2 --check-host 0 > $O/clk2.log 2> $O/clk2.err ) || { tail -5 $O/clk2.err; exit 1; }
grep -B1 "prune clocks" $O/clk2.err | tail -40 > $O/clk2_sections.txt
# 2. counters of one batch, in situ (SQ pass, then the two TCC passes)
export MRP_PHASE_GROUPS=1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/t_sq -o t -- python3 $R/tools/pipeline_probe.py --chunks 96 --repeat 2 --check-host 0 > $O/pmc_sq.log 2>&1 || tail -5 $O/pmc_sq.log
python3 $R/tools/pmc_all.py $O/t_sq 20000 > $O/pmc_sq_insitu.txt
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES --output-format csv -d $O/t_lds -o t -- python3 $R/tools/pipeline_probe.py --chunks 96 --repeat 2 --check-host 0 > $O/pmc_lds.log 2>&1 || tail -5 $O/pmc_lds.log
python3 $R/tools/pmc_all.py $O/t_lds 20000 > $O/pmc_lds_insitu.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/t_$c -o t -- python3 $R/tools/pipeline_probe.py --chunks 96 --repeat 2 --check-host 0 > $O/pmc_$c.log 2>&1 || tail -5 $O/pmc_$c.log
  python3 $R/tools/pmc_all.py $O/t_$c 20000 > $O/pmc_${c}_insitu.txt
done
rm -rf $O/t_sq $O/t_lds $O/t_FETCH_SIZE $O/t_WRITE_SIZE
ls -la $O
