#!/bin/bash
# on the GPU box: per-kernel times of the cost-volume forward + backward with sparse keypoint masks, dense backward against the kept-row form
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  export GD_CV_BWD_ROWS=$v
  rm -rf $R/gpurun_out/prof_cvb$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cvb$v -- python3 $R/tools/cv_bwd_rows_ab.py h > $R/gpurun_out/prof_cvb$v.log 2>&1
  find $R/gpurun_out/prof_cvb$v -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $R/gpurun_out/r03_cv_bwd_rows${v}_kernel_stats.csv
  rm -rf $R/gpurun_out/prof_cvb$v
  echo "== GD_CV_BWD_ROWS=$v"; head -16 $R/gpurun_out/r03_cv_bwd_rows${v}_kernel_stats.csv | cut -c1-150; tail -1 $R/gpurun_out/prof_cvb$v.log
done
