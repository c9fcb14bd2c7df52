#!/bin/bash
# on the GPU box: the four rocprofv3 --pmc passes behind profiles/r03_pmc_cost_volume_traffic_{kp,full}.json
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in kp full; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc_cv_${tag}_$c
    rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_cv_${tag}_$c -- python3 $R/tools/bench_kernels.py pmc_cv_$tag > $R/gpurun_out/pmc_cv_${tag}_$c.log 2>&1
  done
done
kept=$(grep "kept rows" $R/gpurun_out/pmc_cv_kp_FETCH_SIZE.log | awk '{print $3, $4}')
python3 $R/tools/pmc_cv_traffic.py $R/gpurun_out/pmc_cv_kp_FETCH_SIZE $R/gpurun_out/pmc_cv_kp_WRITE_SIZE $R/gpurun_out/r03_pmc_cost_volume_traffic_kp.json kp $kept
python3 $R/tools/pmc_cv_traffic.py $R/gpurun_out/pmc_cv_full_FETCH_SIZE $R/gpurun_out/pmc_cv_full_WRITE_SIZE $R/gpurun_out/r03_pmc_cost_volume_traffic_full.json full 43808 43808
# keep the merged scratch small: the counter CSVs of the cv kernels only
for d in $R/gpurun_out/pmc_cv_*_SIZE; do f=$(ls $d/*/*counter_collection.csv | head -1); grep -E "Kernel_Name|cv_" $f > $d.csv; rm -rf $d; done
