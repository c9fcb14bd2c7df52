#!/bin/bash
# on the GPU box: the two rocprofv3 --pmc passes behind profiles/r04_pmc_cost_volume_traffic_rows.json (the kept-row forward the tf32h trainer runs)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_cv_rows_$c
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_cv_rows_$c -- python3 $R/tools/bench_kernels.py pmc_cv_rows > $R/gpurun_out/pmc_cv_rows_$c.log 2>&1
done
kept=$(grep "kept rows" $R/gpurun_out/pmc_cv_rows_FETCH_SIZE.log | awk '{print $3, $4}')
python3 $R/tools/pmc_cv_traffic.py $R/gpurun_out/pmc_cv_rows_FETCH_SIZE $R/gpurun_out/pmc_cv_rows_WRITE_SIZE $R/gpurun_out/r04_pmc_cost_volume_traffic_rows.json rows $kept
for d in $R/gpurun_out/pmc_cv_rows_*_SIZE; do f=$(ls $d/*/*counter_collection.csv | head -1); grep -E "Kernel_Name|cv_" $f > $d.csv; rm -rf $d; done
