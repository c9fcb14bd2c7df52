#!/bin/bash
# on the GPU box: where do the attention kernels' wave cycles go?    Three counter-only rocprofv3 passes over the same
# micro-run (tools/bench_kernels.py pmc_attn: forward + both backward kernels at the benched shape 64 images x 12 heads x 1370 tokens, fp16 operands =
# the tf32h engine's attention), 8 SQ counters per pass (the block's slot limit), plus one pass per optional counter whose name differs between ROCm
# versions (a pass with an unknown counter fails as a whole).  Summarised by tools/pmc_attn_summary.py into gpurun_out/r05_pmc_attention_stall.json.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export GD_PMC_F16=1
O=$R/gpurun_out/pmc_attn_r05
rm -rf $O; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $O/counters.txt | sort -u > $O/sq_counters.txt
run() { # name, counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -- python3 $R/tools/bench_kernels.py pmc_attn > $O/$n.log 2>&1 || echo "pass $n failed" >> $O/failed.txt
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
for c in SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MFMA SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_MFMA_F16 SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_IFETCH SQ_WAIT_IFETCH; do
  if grep -qx "$c" $O/sq_counters.txt; then run c_$c $c; else echo "$c: not listed by rocprofv3 -L" >> $O/failed.txt; fi
done
python3 $R/tools/pmc_attn_summary.py $O $R/gpurun_out/r05_pmc_attention_stall.json
cat $O/failed.txt 2>/dev/null
