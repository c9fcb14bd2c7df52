#!/bin/bash
# on the GPU box: the round's rocprofv3 evidence for bench.py's roofline object.  usage: prof_r04.sh [dtype]   (default: bench.py's default, tf32h)
#   (1) --kernel-trace --stats of the bench command (3 steps after 1 warm-up, steps only): per-kernel average durations;
#   (2) --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, counters only) over one step: HBM-side bytes of the persistent GEMM.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
DT=${1:-tf32h}
rm -rf $R/gpurun_out/prof_r04
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04 -- python3 $R/bench.py --dtype $DT --steps 3 --warmup 1 --steps-only > $R/gpurun_out/prof_r04.log 2>&1
find $R/gpurun_out/prof_r04 -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $R/gpurun_out/r04_bench_p32_kernel_stats_$DT.csv
rm -rf $R/gpurun_out/prof_r04
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_gemm_$c
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_gemm_$c -- python3 $R/bench.py --dtype $DT --steps 1 --warmup 1 --steps-only > $R/gpurun_out/pmc_gemm_$c.log 2>&1
done
python3 $R/tools/pmc_gemm_traffic.py $R/gpurun_out/pmc_gemm_FETCH_SIZE $R/gpurun_out/pmc_gemm_WRITE_SIZE $R/gpurun_out/r04_pmc_gemm_traffic_$DT.json
rm -rf $R/gpurun_out/pmc_gemm_FETCH_SIZE $R/gpurun_out/pmc_gemm_WRITE_SIZE
head -12 $R/gpurun_out/r04_bench_p32_kernel_stats_$DT.csv | cut -c1-160; tail -2 $R/gpurun_out/prof_r04.log | cut -c1-400
