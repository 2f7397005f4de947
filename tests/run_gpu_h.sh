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
for pr in 0 3; do timeout 300 python tests/gpu_stem3_prof.py $pr 2>&1 | grep -v amdgpu.ids | grep -E "ticks/tile" | sed "s/^/prio=$pr /" | tee -a $OUT/stem3_prof.log; done
