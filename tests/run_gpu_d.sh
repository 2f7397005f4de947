#!/bin/bash
cd "$(dirname "$0")/.."
R=$(pwd)
OUT=$R/gpurun_out/${1:-r02d}
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_matching.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -k "argsort or prepared or beyond_1024 or first_hit" > $OUT/pytest_new.log 2>&1
echo "pytest new exit $?" | tee -a $OUT/pytest_new.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest_new.log | tail -20
for p in bf16 bf16x3 fp32; do
  timeout 300 python tests/gpu_timing.py --batch 1 --precision $p --iters 20 2>&1 | grep -v amdgpu.ids | tee $OUT/timing_b1_$p.log
done
timeout 300 python tests/gpu_timing.py --batch 1 --precision bf16 --iters 20 --height 479 --width 637 2>&1 | grep -v amdgpu.ids | tee $OUT/timing_b1_odd_bf16.log
for a in 0 1 2 3; do
  timeout 300 python tests/gpu_timing.py --batch 32 --precision bf16x3 --layers 2 --tile 2 --ablate $a 2>&1 | grep "conv01" | sed "s/^/ablate=$a /" | tee -a $OUT/ablate_conv01_x3.log
done
timeout 300 python tests/gpu_shardbench.py 1,2,4,8 bf16 2>&1 | grep -v amdgpu.ids | tee $OUT/shardbench_bf16.log
timeout 300 python tests/gpu_shardbench.py 1,2,4,8 bf16x3 2>&1 | grep -v amdgpu.ids | tee $OUT/shardbench_bf16x3.log
