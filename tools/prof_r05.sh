#!/bin/bash
# on the GPU box: the round's rocprofv3 evidence.  usage: prof_r05.sh [dtype]   (default: bench.py's default, tf32h)
#   (1) --kernel-trace --stats of the bench command (3 steps after 1 warm-up, steps only): per-kernel average durations;
#   (2) --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, counters only) over one step: HBM-side bytes of the persistent GEMM;
#   (3) the cost-volume forward — dense sweep (every row kept) and the kept-row form the trainer runs: FETCH_SIZE / WRITE_SIZE passes, and the SQ
#       wave-cycle / MFMA-busy / VALU-MFMA co-execution passes of tools/pmc_attn_r05.sh on the same two launches.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
DT=${1:-tf32h}
O=$R/gpurun_out
rm -rf $O/prof_r05
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r05 -- python3 $R/bench.py --dtype $DT --steps 3 --warmup 1 --steps-only > $O/prof_r05.log 2>&1
find $O/prof_r05 -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $O/r05_bench_p32_kernel_stats_$DT.csv
rm -rf $O/prof_r05
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_gemm_$c
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_gemm_$c -- python3 $R/bench.py --dtype $DT --steps 1 --warmup 1 --steps-only > $O/pmc_gemm_$c.log 2>&1
done
python3 $R/tools/pmc_gemm_traffic.py $O/pmc_gemm_FETCH_SIZE $O/pmc_gemm_WRITE_SIZE $O/r05_pmc_gemm_traffic_$DT.json
rm -rf $O/pmc_gemm_FETCH_SIZE $O/pmc_gemm_WRITE_SIZE
# ---- cost-volume forward
for tag in full rows; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_cv_${tag}_$c
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_cv_${tag}_$c -- python3 $R/tools/bench_kernels.py pmc_cv_$tag > $O/pmc_cv_${tag}_$c.log 2>&1
  done
done
kept=$(grep "kept rows" $O/pmc_cv_rows_FETCH_SIZE.log | awk '{print $3, $4}')
python3 $R/tools/pmc_cv_traffic.py $O/pmc_cv_full_FETCH_SIZE $O/pmc_cv_full_WRITE_SIZE $O/r05_pmc_cost_volume_traffic_full.json full 43808 43808
python3 $R/tools/pmc_cv_traffic.py $O/pmc_cv_rows_FETCH_SIZE $O/pmc_cv_rows_WRITE_SIZE $O/r05_pmc_cost_volume_traffic_rows.json rows $kept
for d in $O/pmc_cv_*_SIZE; do f=$(ls $d/*/*counter_collection.csv | head -1); grep -E "Kernel_Name|cv_" $f > $d.csv; rm -rf $d; done
S=$O/pmc_cv_sq_r05
rm -rf $S; mkdir -p $S
rocprofv3 -L > $S/counters.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $S/counters.txt | sort -u > $S/sq_counters.txt
run() { n=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $S/$n -- python3 $R/tools/bench_kernels.py pmc_cv_full pmc_cv_rows > $S/$n.log 2>&1 || echo "pass $n failed" >> $S/failed.txt; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
for c in SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_VALU; do
  if grep -qx "$c" $S/sq_counters.txt; then run c_$c $c; else echo "$c: not listed by rocprofv3 -L" >> $S/failed.txt; fi
done
python3 $R/tools/pmc_attn_summary.py $S $O/r05_pmc_cost_volume_stall.json cv_fwd "rocprofv3 --pmc passes (counters only + --kernel-trace) over tools/bench_kernels.py pmc_cv_full pmc_cv_rows: 32 pairs, hw = 1369, C = 768, bf16 / fp16 features; cv_fwd_persist_kernel = the dense sweep (every row kept), cv_fwd_rows_kernel = the kept-row form (keypoint-patch masks)"
rm -rf $S/a $S/b $S/c_*
head -12 $O/r05_bench_p32_kernel_stats_$DT.csv | cut -c1-160; tail -2 $O/prof_r05.log | cut -c1-400
