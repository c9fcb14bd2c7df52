#!/bin/bash
# on the GPU box: the round's rocprofv3 evidence, taken on ONE build.  usage: HEAD=<git rev> bash tools/prof_r06.sh [dtype]   (default: bench.py's default, tf32h)
#   (1) --kernel-trace --stats of the bench command (3 steps after 1 warm-up, steps only): per-kernel average durations + tools/kernel_gaps.py on the trace;
#   (2) --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, counters only) over one step: HBM-side bytes of the persistent GEMM;
#   (3) the cost-volume forward — dense sweep (every row kept) and the kept-row form the trainer runs: FETCH_SIZE / WRITE_SIZE passes.
# Every JSON it writes carries the commit (HEAD, passed in: the box has no .git) and the sha256 of the library the passes ran on.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
DT=${1:-tf32h}
O=$R/gpurun_out
SHA=$(sha256sum $R/3d-vlm-gd_amd/lib/libgd_hip.so | cut -c1-16)
stamp() { python3 - "$1" <<PY
import json, sys
p = sys.argv[1]
d = json.load(open(p))
d["commit"] = "${HEAD:-unknown}"
d["libgd_hip_sha256_16"] = "$SHA"
json.dump(d, open(p, "w"), indent=1)
PY
}
rm -rf $O/prof_r06
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r06 -- python3 $R/bench.py --dtype $DT --steps 3 --warmup 1 --steps-only > $O/prof_r06.log 2>&1
find $O/prof_r06 -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $O/r06_bench_p32_kernel_stats_$DT.csv
sed -i "1i # commit ${HEAD:-unknown}, libgd_hip.so sha256 $SHA: rocprofv3 --kernel-trace --stats -- python3 bench.py --dtype $DT --steps 3 --warmup 1 --steps-only (4 steps in the trace)" $O/r06_bench_p32_kernel_stats_$DT.csv
python3 $R/tools/kernel_gaps.py $O/prof_r06 > $O/r06_kernel_gaps_$DT.txt 2>&1
sed -i "1i # commit ${HEAD:-unknown}, libgd_hip.so sha256 $SHA" $O/r06_kernel_gaps_$DT.txt
rm -rf $O/prof_r06
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_gemm_$c
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_gemm_$c -- python3 $R/bench.py --dtype $DT --steps 1 --warmup 1 --steps-only > $O/pmc_gemm_$c.log 2>&1
done
python3 $R/tools/pmc_gemm_traffic.py $O/pmc_gemm_FETCH_SIZE $O/pmc_gemm_WRITE_SIZE $O/r06_pmc_gemm_traffic_$DT.json && stamp $O/r06_pmc_gemm_traffic_$DT.json
rm -rf $O/pmc_gemm_FETCH_SIZE $O/pmc_gemm_WRITE_SIZE
for tag in full rows; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_cv_${tag}_$c
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_cv_${tag}_$c -- python3 $R/tools/bench_kernels.py pmc_cv_$tag > $O/pmc_cv_${tag}_$c.log 2>&1
  done
done
kept=$(grep "kept rows" $O/pmc_cv_rows_FETCH_SIZE.log | awk '{print $3, $4}')
python3 $R/tools/pmc_cv_traffic.py $O/pmc_cv_full_FETCH_SIZE $O/pmc_cv_full_WRITE_SIZE $O/r06_pmc_cost_volume_traffic_full.json full 43808 43808 && stamp $O/r06_pmc_cost_volume_traffic_full.json
python3 $R/tools/pmc_cv_traffic.py $O/pmc_cv_rows_FETCH_SIZE $O/pmc_cv_rows_WRITE_SIZE $O/r06_pmc_cost_volume_traffic_rows.json rows $kept && stamp $O/r06_pmc_cost_volume_traffic_rows.json
rm -rf $O/pmc_cv_*_SIZE
head -12 $O/r06_bench_p32_kernel_stats_$DT.csv | cut -c1-160; tail -2 $O/prof_r06.log | cut -c1-300; cat $O/r06_kernel_gaps_$DT.txt | head -4
