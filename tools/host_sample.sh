#!/bin/bash
# Development: CPU sampling profile of the host side of one 1 152-chunk call (tools/sampler), with the stock library and with a
# build whose host functions are not inlined (alt_lib/libmargin_rphmm_prof.so, optional).  Output: gpurun_out/hs/.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/hs; mkdir -p $O
cd $R
gcc -O2 -g -shared -fPIC -o tools/sampler/libcpusampler.so tools/sampler/cpusampler.c -lpthread || exit 1
N=${1:-1152}
MRP_TIMING=1 timeout -k 10 400 python3 tools/pipeline_probe.py --chunks $N --repeat 6 --check-host 0 --sample $O/stock.samples > $O/stock.log 2> $O/stock.err || { tail -5 $O/stock.err; exit 1; }
python3 tools/sampler/resolve.py $O/stock.samples 60 > $O/stock_profile.txt
if [ -f alt_lib/libmargin_rphmm_prof.so ]; then
  MRP_LIB_OVERRIDE=$R/alt_lib/libmargin_rphmm_prof.so timeout -k 10 400 python3 tools/pipeline_probe.py --chunks $N --repeat 6 --check-host 0 --sample $O/prof.samples > $O/prof.log 2> $O/prof.err || { tail -5 $O/prof.err; exit 1; }
  python3 tools/sampler/resolve.py $O/prof.samples 70 > $O/prof_profile.txt
fi
grep "^run" $O/stock.log $O/prof.log
rm -f $O/*.samples
