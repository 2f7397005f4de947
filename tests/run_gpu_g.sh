#!/bin/bash
cd "$(dirname "$0")/.."
R=$(pwd)
OUT=$R/gpurun_out/${1:-r02g}
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest_gpu.log | tail -20
grep -E "split-K on vs off" $OUT/pytest_gpu.log | tail -12
timeout 600 python tests/gpu_latency.py $OUT/latency.md 2>&1 | grep -v amdgpu.ids | tee $OUT/latency.log
for b in 1 2 4 8; do
  timeout 300 python tests/gpu_timing.py --batch $b --precision bf16 --iters 20 2>&1 | grep -E "whole|conv1[0-2]|conv0[789]" | sed "s/^/b=$b /" | tee -a $OUT/timing_small.log
done
