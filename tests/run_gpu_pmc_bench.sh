#!/bin/bash
# MFMA utilisation / clock / L2 hit rate per kernel of the benchmark step, from rocprofv3 PMC passes
# (counters only with --kernel-trace; separate passes per counter set).
# usage: tests/run_gpu_pmc_bench.sh <tag>
cd "$(dirname "$0")/.."
R=$(pwd); TAG=${1:-pmc_bench}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --eager --skip-cpu-baseline --skip-api --queries 8192 --gallery 81920"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1
  tail -1 $OUT/p$i.log
done
cd $R; find $OUT -type f -size +8M -delete; ls $OUT
