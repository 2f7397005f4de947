#!/bin/bash
cd "$(dirname "$0")/.."
R=$(pwd)
OUT=$R/gpurun_out/${1:-r02f}
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests/test_gpu_matching.py tests/test_gpu_conv.py tests/test_gpu_extensions.py tests/test_gpu_evaluator.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest_gpu.log | tail -20
timeout 300 python tests/gpu_shardbench.py 1,2,4,8 bf16 2>&1 | grep -v amdgpu.ids | tee $OUT/shardbench.log
timeout 300 python tests/gpu_shardbench.py 1,2,4,8 bf16x3 2>&1 | grep -v amdgpu.ids | tee -a $OUT/shardbench.log
timeout 600 python bench.py --skip-cpu-baseline --skip-api --steps 10 2> $OUT/bench_err.log | tee $OUT/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); m=d['matching']; print('x3', d['value'], 'match', m['value'], m['ms_per_step'], 'fast', m['fast_mode']['value'], m['fast_mode']['ms_per_step'])"
# what bounds the chunk-major K order: instruction overhead or the access pattern?
for ko in 0 1; do
  timeout 300 python tests/gpu_convbench.py --korder $ko --ablate 3 --rounds 3 2>&1 | grep -v amdgpu.ids | tail -12 | sed "s/^/korder=$ko /" | tee -a $OUT/korder_ablate.log
done
