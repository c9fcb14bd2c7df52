#!/bin/bash
# on the GPU box: only the two --pmc passes of tools/prof_r04.sh (per-instantiation FETCH_SIZE / WRITE_SIZE of the persistent GEMM over one step)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
DT=${1:-tf32h}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_gemm_$c
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_gemm_$c -- python3 $R/bench.py --dtype $DT --steps 1 --warmup 1 --steps-only > $R/gpurun_out/pmc_gemm_$c.log 2>&1
done
python3 $R/tools/pmc_gemm_traffic.py $R/gpurun_out/pmc_gemm_FETCH_SIZE $R/gpurun_out/pmc_gemm_WRITE_SIZE $R/gpurun_out/r04_pmc_gemm_traffic_$DT.json
rm -rf $R/gpurun_out/pmc_gemm_FETCH_SIZE $R/gpurun_out/pmc_gemm_WRITE_SIZE
