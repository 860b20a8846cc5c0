#!/bin/bash
# stall counters of mrp_cross_emit_kernel, summed over the launches of one 96-chunk batch (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pxe; rm -rf $O; mkdir -p $O
export MRP_PHASE_GROUPS=1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" "SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/t$i -o t -- python3 $R/tools/pipeline_probe.py --chunks 96 --repeat 1 --check-host 0 > $O/p$i.log 2>&1 || { grep -v "^[EWI]2026" $O/p$i.log | tail -5; echo "set $i failed: $set"; rm -rf $O/t$i; continue; }
  python3 $R/tools/pmc_kernels.py $O/t$i | grep "mrp_cross_emit_kernel" >> $O/pmc_xe.txt
  rm -rf $O/t$i
done
cat $O/pmc_xe.txt | cut -c1-400
