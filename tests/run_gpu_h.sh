#!/bin/bash
cd "$(dirname "$0")/.."
R=$(pwd)
OUT=$R/gpurun_out/${1:-r02h}
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_gpu_x3.py -m gpu -q --tb=short --timeout 300 -p no:cacheprovider -k "stem" > $OUT/pytest_stem.log 2>&1
echo "pytest stem exit $?" | tee -a $OUT/pytest_stem.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest_stem.log | tail -10
timeout 300 python tests/gpu_stem3_prof.py 2>&1 | grep -v amdgpu.ids | tee $OUT/stem3_prof.log

timeout 600 python tests/gpu_timing.py --batch 32 --precision bf16x3 2>&1 | grep -v amdgpu.ids | tee $OUT/timing_bf16x3.log
