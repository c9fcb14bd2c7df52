#!/bin/bash
# on the GPU box: rocprofv3 --kernel-trace --stats of the benched step (3 steps after 1 warm-up) -> gpurun_out/<tag>_kernel_stats.csv.  usage: kstats_step.sh <tag> [dtype]
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-step}; DT=${2:-tf32h}; O=$R/gpurun_out
rm -rf $O/ks_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$TAG -- python3 $R/bench.py --dtype $DT --steps 3 --warmup 1 --steps-only > $O/ks_$TAG.log 2>&1
find $O/ks_$TAG -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $O/${TAG}_kernel_stats.csv
rm -rf $O/ks_$TAG
tail -1 $O/ks_$TAG.log | cut -c1-300
