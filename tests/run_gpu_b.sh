#!/bin/bash
cd "$(dirname "$0")/.."
R=$(pwd)
OUT=$R/gpurun_out/${1:-r02b}
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -s > $OUT/pytest_gpu.log 2>&1
echo "pytest all exit $?" | tee -a $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | tail -40
grep -E "configs\[1\]|worst single|recalls|PCA\.|test.py sequence|bf16x3 vs fp64|desc \(bf16x3\)" $OUT/pytest_gpu.log | tail -40
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -2 $OUT/smoke.log
timeout 600 python tests/gpu_timing.py --batch 32 --precision bf16x3 2>&1 | grep -v amdgpu.ids | tee $OUT/timing_bf16x3.log
timeout 600 python tests/gpu_timing.py --batch 32 --precision bf16x3 --tile 2 2>&1 | grep -v amdgpu.ids | grep -E "conv01|whole" | tee $OUT/timing_bf16x3_t2.log
timeout 600 python bench.py --steps 10 --warmup 3 --precision bf16x3 --skip-cpu-baseline 2> $OUT/bench_x3_err.log | tee $OUT/bench_x3.json
tail -3 $OUT/bench_x3_err.log
