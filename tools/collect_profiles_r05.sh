#!/bin/bash
# Round-5 measurements on the GPU box (run through gpurun from the repo root, possibly in parts: tools/collect_profiles_r05.sh [part ...],
# parts: bench trace traffic insitu levels host queue clocks).  Everything lands under gpurun_out/r05p/; the text / JSON files are
# then copied into profiles/r05/.  rocprofv3 needs cd /tmp and TMPDIR=/tmp on this pool; the program goes directly after `--`.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
parts="${*:-bench trace traffic insitu levels host queue clocks}"
has() { case " $parts " in *" $1 "*) return 0;; esac; return 1; }
BENCH="python3 $R/bench.py --threads 8 --no-cpu-baseline --align-chunks 0 --queue-runs 0 --sum-chunks 0 --shape-runs 0 --genome-chunks 0"
PROBE="python3 $R/tools/pipeline_probe.py --chunks 96 --repeat 2 --check-host 0"
if has bench; then   # 1. the default bench line
  ( cd $R && timeout -k 10 1000 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) || exit 1
fi
if has trace; then   # 2. kernel trace + stats of the bench command (end-to-end steps and the replay leg)
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_stats -o t -- $BENCH --steps 5 --warmup 1 > $O/bench_under_trace.json 2> $O/bench_under_trace.err || exit 1
  python3 $R/tools/r05_summary.py stats $O/t_stats > $O/bench_kernel_stats.txt
  rm -rf $O/t_stats
fi
if has traffic; then # 3. HBM traffic of the replay leg ALONE (--steps 0: no end-to-end step in the trace), separate passes per counter
  RS=20; NL=$((RS + 3)) # (bench.py launches the replay batch three times before it starts counting)
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/t_$c -o t -- $BENCH --chunks 96 --steps 0 --warmup 0 --roofline-steps $RS > $O/bench_$c.json 2> $O/bench_$c.err || exit 1
  done
  python3 $R/tools/r04_summary.py traffic $O/t_FETCH_SIZE $O/t_WRITE_SIZE 96 $NL $O/traffic.json profiles/r05 $R/margin_amd/csrc/mrp_kernels.hip > /dev/null
  python3 $R/tools/pmc_all.py $O/t_FETCH_SIZE 100000 > $O/pmc_FETCH_SIZE_replay.txt
  python3 $R/tools/pmc_all.py $O/t_WRITE_SIZE 100000 > $O/pmc_WRITE_SIZE_replay.txt
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/t_sqr -o t -- $BENCH --chunks 96 --steps 0 --warmup 0 --roofline-steps 4 > $O/bench_sq_replay.json 2> $O/bench_sq_replay.err || exit 1
  python3 $R/tools/pmc_all.py $O/t_sqr 100000 > $O/pmc_sq_replay.txt
  rm -rf $O/t_FETCH_SIZE $O/t_WRITE_SIZE $O/t_sqr
fi
if has insitu; then  # 4. counters of EVERY kernel of one 96-chunk batch in situ (prune variants, sweeps as the pipeline launches them)
  export MRP_PHASE_GROUPS=1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/t_sq -o t -- $PROBE > $O/pmc_sq.log 2>&1 || exit 1
  python3 $R/tools/pmc_all.py $O/t_sq 20000 > $O/pmc_sq_insitu.txt
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES --output-format csv -d $O/t_lds -o t -- $PROBE > $O/pmc_lds.log 2>&1 || exit 1
  python3 $R/tools/pmc_all.py $O/t_lds 20000 > $O/pmc_lds_insitu.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/t_i$c -o t -- $PROBE > $O/pmc_i$c.log 2>&1 || exit 1
    python3 $R/tools/pmc_all.py $O/t_i$c 20000 > $O/pmc_${c}_insitu.txt
  done
  # the bytes that really cross the HBM interface per 96-chunk call, every kernel: what bench.py quotes as roofline.traffic (scaled to the step)
  python3 $R/tools/r05_summary.py path_traffic $O/t_iFETCH_SIZE $O/t_iWRITE_SIZE 96 $O/path_traffic.json profiles/r05 $R/margin_amd/csrc/mrp_kernels.hip $R/margin_amd/csrc/mrp_engine_kernels.hip > /dev/null
  rm -rf $O/t_sq $O/t_lds $O/t_iFETCH_SIZE $O/t_iWRITE_SIZE
  unset MRP_PHASE_GROUPS
fi
if has levels; then  # 5. the levels of the resident pipeline: one batch (per-dispatch listing), eight concurrent batches (device occupancy)
  MRP_PHASE_GROUPS=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t_lv -o t -- python3 $R/tools/pipeline_probe.py --chunks 96 --repeat 3 --check-host 0 > $O/probe_96_g1.log 2>&1 || exit 1
  python3 $R/tools/trace_levels.py $O/t_lv > $O/pipeline_levels_96chunks_1batch.txt
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t_g8 -o t -- python3 $R/tools/pipeline_probe.py --chunks 576 --repeat 3 --check-host 0 > $O/probe_576_g8.log 2>&1 || exit 1
  python3 $R/tools/trace_busy.py $O/t_g8 8 > $O/pipeline_busy_576chunks_8batches.txt
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t_g8b -o t -- python3 $R/tools/pipeline_probe.py --chunks 1152 --repeat 3 --check-host 0 > $O/probe_1152_g8.log 2>&1 || exit 1
  python3 $R/tools/trace_busy.py $O/t_g8b 8 > $O/pipeline_busy_1152chunks_8batches.txt
  rm -rf $O/t_lv $O/t_g8 $O/t_g8b
fi
if has host; then    # 6. host threads: the same 1 152-chunk call with 16, 12, 8, 6 and 4 threads in the library's pool (wall, process CPU time),
                     #    and a CPU sampling profile of the call's host side (tools/sampler: no perf on these boxes)
  for t in 16 12 8 6 4; do
    ( cd $R && MRP_HOST_THREADS=$t timeout -k 10 300 python3 tools/pipeline_probe.py --chunks 1152 --repeat 5 --check-host 0 > $O/probe_1152_t$t.log 2>&1 ) || exit 1
  done
  ( cd $R && timeout -k 10 300 python3 tools/pipeline_probe.py --chunks 576 --repeat 5 --check-host 0 > $O/probe_576_t16.log 2>&1 ) || exit 1
  ( cd $O && grep -H "^run" probe_1152_t16.log probe_1152_t12.log probe_1152_t8.log probe_1152_t6.log probe_1152_t4.log probe_576_t16.log > probe_runs.txt )
  ( cd $R && gcc -O2 -g -shared -fPIC -o tools/sampler/libcpusampler.so tools/sampler/cpusampler.c -lpthread &&
    timeout -k 10 400 python3 tools/pipeline_probe.py --chunks 1152 --repeat 8 --check-host 0 --sample $O/host.samples > $O/host_sample.log 2>&1 &&
    python3 tools/sampler/resolve.py $O/host.samples 40 | cut -c1-160 > $O/host_profile.txt; rm -f $O/host.samples ) || exit 1
fi
if has queue; then   # 7. the work queue from host memory: one batch beside the resident call, and a long queue
  ( cd $R && timeout -k 10 600 python3 tools/queue_probe.py --runs 4 2>&1 | grep " ms" > $O/queue_probe.txt ) || exit 1
  ( cd $R && timeout -k 10 600 python3 tools/queue_long.py --chunks 2304 2>&1 | grep "queue:\|resident:" > $O/queue_long.txt ) || exit 1
  ( cd $R && timeout -k 10 600 python3 tools/queue_long.py --chunks 4608 --skip-resident 1 2>&1 | grep "queue:" >> $O/queue_long.txt ) || exit 1
  ( cd $R && timeout -k 10 600 python3 tools/adaptor_probe.py --chunks 8 --threads 8 > $O/adaptor_probe.txt 2>&1 ) || exit 1
fi
# (alt_lib/ is listed in .gpurunignore since the end of round 4: to run this part again, build the two libraries -- tools/clk1_probe.sh,
#  tools/clk2_probe.sh say how -- and take alt_lib/ out of .gpurunignore for that call)
if has clocks && [ -f $R/alt_lib/libmargin_rphmm_clk1.so ]; then  # 8. in-kernel clocks of the prune kernel (development builds with -DPRUNE_EXP_CLOCK / -DPRUNE_EXP_CLOCK2)
  ( cd $R && bash tools/clk1_probe.sh r04 > $O/prune_role_clocks.txt 2>&1 && bash tools/clk2_probe.sh r04 > $O/prune_chain_sections.txt 2>&1 ) || exit 1
fi
ls -la $O
