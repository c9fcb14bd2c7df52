#!/bin/bash
# on the GPU box: MFMA-pipe occupancy and the clock the chip holds, per kernel of one tf32h step.
#   rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE with --kernel-trace (durations of the same dispatches), counters only
#   (no other trace domain), summarised by tools/pmc_mfma_summary.py.  usage: pmc_mfma_r04.sh [dtype]
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
DT=${1:-tf32h}
rm -rf $R/gpurun_out/pmc_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma -- python3 $R/bench.py --dtype $DT --steps 1 --warmup 1 --steps-only > $R/gpurun_out/pmc_mfma.log 2>&1
ls $R/gpurun_out/pmc_mfma/*/ | head
python3 $R/tools/pmc_mfma_summary.py $R/gpurun_out/pmc_mfma $R/gpurun_out/r04_pmc_mfma_busy_$DT.json
rm -rf $R/gpurun_out/pmc_mfma
