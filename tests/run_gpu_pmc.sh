#!/bin/bash
# PMC passes over the per-layer conv A/B script (one tile mode): which pipe bounds the kernel.
# usage: tests/run_gpu_pmc.sh <tag> <mode>
cd "$(dirname "$0")/.."
R=$(pwd); TAG=${1:-pmc}; MODE=${2:-4}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
CMD="python $R/tests/gpu_convbench.py --modes $MODE --rounds 1 --iters 1"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1
  tail -2 $OUT/p$i.log
done
cd $R; find $OUT -type f -size +8M -delete; ls $OUT
