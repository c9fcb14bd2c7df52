#!/bin/bash
# on the GPU box: anatomy of cv_bwd_rows_kernel — rocprofv3 kernel time with parts of the kernel switched off (GD_CV_DBG bits: 1 no teacher loads,
# 2 no G stores, 4 no MFMA main loop, 8 no exp / KL arithmetic); results are wrong with any bit set, only the kernel's duration is read
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export GD_CV_BWD_ROWS=1
for v in 0 1 2 4 8 3 7 15; do
  export GD_CV_DBG=$v
  rm -rf $R/gpurun_out/prof_anat
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_anat -- python3 $R/tools/cv_bwd_rows_ab.py h > /dev/null 2>&1
  f=$(find $R/gpurun_out/prof_anat -name '*kernel_stats.csv' | head -1)
  echo "GD_CV_DBG=$v $(grep cv_bwd_rows_kernel $f | cut -d, -f2-4)"
  rm -rf $R/gpurun_out/prof_anat
done
