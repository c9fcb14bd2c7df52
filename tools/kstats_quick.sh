#!/bin/bash
# on the GPU box: rocprofv3 kernel stats of 3 steps (1 warm-up) of bench.py for one engine; prints the rows matching a pattern.  usage: kstats_quick.sh <dtype> <grep pattern>
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_q
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_q -- python3 $R/bench.py --dtype ${1:-tf32h} --steps 3 --warmup 1 --steps-only > $R/gpurun_out/prof_q.log 2>&1
f=$(find $R/gpurun_out/prof_q -name '*kernel_stats.csv' | head -1)
cp $f $R/gpurun_out/kstats_quick_${1:-tf32h}.csv
grep -E "${2:-skinny}" $f | cut -c1-170
tail -1 $R/gpurun_out/prof_q.log | cut -c1-200
rm -rf $R/gpurun_out/prof_q
