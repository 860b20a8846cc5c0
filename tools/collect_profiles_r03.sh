#!/bin/bash
# Round-3 measurements on the GPU box (run through gpurun from the repo root): everything lands under gpurun_out/r03p/,
# the text / JSON files are then copied into profiles/r03/ by hand.  rocprofv3 needs cd /tmp and TMPDIR=/tmp on this pool.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --threads 2 --no-cpu-baseline --align-chunks 0 --queue-runs 0 --sum-chunks 0 --shape-runs 0"
# 1. the default bench line
( cd $R && timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) || exit 1
# 2. kernel trace + stats of the bench command (end-to-end steps and the replay leg)
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_stats -o t -- $BENCH --steps 5 --warmup 1 > $O/bench_under_trace.json 2> $O/bench_under_trace.err || exit 1
python3 $R/tools/r02_summary.py stats $O/t_stats > $O/bench_kernel_stats.txt
# 3. HBM traffic of the replay leg (separate passes), then counters of the one-pass cross product + emission kernel
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/t_$c -o t -- $BENCH --steps 1 --warmup 0 --roofline-steps 4 > $O/bench_$c.json 2> $O/bench_$c.err || exit 1
  python3 $R/tools/r02_summary.py pmc $O/t_$c > $O/pmc_$c.txt
done
python3 $R/tools/r02_summary.py traffic $O/t_FETCH_SIZE $O/t_WRITE_SIZE 96 $O/traffic.json profiles/r03 $R/margin_amd/csrc/mrp_kernels.hip > /dev/null
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/t_sq -o t -- $BENCH --steps 1 --warmup 0 --roofline-steps 4 > $O/bench_sq.json 2> $O/bench_sq.err || exit 1
python3 $R/tools/r02_summary.py pmc $O/t_sq > $O/pmc_sq.txt
# 4. the levels of the resident pipeline: one batch (per-dispatch listing), eight concurrent batches (device occupancy)
MRP_PHASE_GROUPS=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t_lv -o t -- python3 $R/tools/pipeline_probe.py --chunks 96 --repeat 3 --check-host 0 > $O/probe_96_g1.log 2>&1 || exit 1
python3 $R/tools/trace_levels.py $O/t_lv > $O/pipeline_levels_96chunks_1batch.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t_g4 -o t -- python3 $R/tools/pipeline_probe.py --chunks 288 --repeat 3 --check-host 0 > $O/probe_288_g8.log 2>&1 || exit 1
python3 $R/tools/trace_busy.py $O/t_g4 8 > $O/pipeline_busy_288chunks_8batches.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t_g8 -o t -- python3 $R/tools/pipeline_probe.py --chunks 576 --repeat 3 --check-host 0 > $O/probe_576_g8.log 2>&1 || exit 1
python3 $R/tools/trace_busy.py $O/t_g8 8 > $O/pipeline_busy_576chunks_8batches.txt
# 5. host threads: the same 288-chunk call with 16 and with 4 threads in the library's pool (wall, process CPU time)
for t in 16 4; do
  ( cd $R && MRP_HOST_THREADS=$t timeout -k 10 300 python3 tools/pipeline_probe.py --chunks 288 --repeat 6 --check-host 0 > $O/probe_288_t$t.log 2>&1 ) || exit 1
done
for t in 16 8 4; do
  ( cd $R && MRP_HOST_THREADS=$t timeout -k 10 300 python3 tools/pipeline_probe.py --chunks 576 --repeat 5 --check-host 0 > $O/probe_576_t$t.log 2>&1 ) || exit 1
done
( cd $R && timeout -k 10 300 python3 tools/pipeline_probe.py --chunks 1152 --repeat 4 --check-host 0 > $O/probe_1152_t16.log 2>&1 ) || exit 1
( cd $O && grep -H "^run" probe_96_g1.log probe_288_g8.log probe_576_g8.log probe_288_t16.log probe_288_t4.log probe_576_t16.log probe_576_t8.log probe_576_t4.log probe_1152_t16.log > probe_runs.txt )
# 5b. the work queue from host memory: one batch (576 chunks) beside the resident call, and a queue of twelve batches (2 304 chunks)
( cd $R && timeout -k 10 600 python3 tools/queue_probe.py --runs 4 2>&1 | grep " ms" > $O/queue_probe.txt ) || exit 1
( cd $R && timeout -k 10 600 python3 tools/queue_long.py --chunks 2304 2>&1 | grep "queue:\|resident:" > $O/queue_long.txt ) || exit 1
# 6. what linking the adaptor alone gives (the seam per hmm / per merge call, beside the oracle and the whole-chunk path)
( cd $R && timeout -k 10 600 python3 tools/adaptor_probe.py --chunks 8 --threads 8 > $O/adaptor_probe.txt 2>&1 ) || exit 1
rm -rf $O/t_stats $O/t_FETCH_SIZE $O/t_WRITE_SIZE $O/t_sq $O/t_lv $O/t_g4 $O/t_g8
ls -la $O
