#!/bin/bash
# first GPU contact of the bf16x3 mode: its parity tests, the whole suite, per-layer timing, a bench line
cd "$(dirname "$0")/.."
R=$(pwd)
OUT=$R/gpurun_out/${1:-x3a}
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_x3.py -m gpu -q --tb=short --timeout 300 -p no:cacheprovider -s > $OUT/pytest_x3.log 2>&1
echo "pytest x3 exit $?" | tee -a $OUT/pytest_x3.log
grep -E "passed|failed|rel_l2|max abs|agreement|vs fp64" $OUT/pytest_x3.log | tail -60
timeout 900 python -m pytest tests -m gpu -q --tb=short --timeout 600 -p no:cacheprovider --deselect tests/test_gpu_x3.py > $OUT/pytest_gpu.log 2>&1
echo "pytest all exit $?" | tee -a $OUT/pytest_gpu.log
tail -n 8 $OUT/pytest_gpu.log
timeout 600 python tests/gpu_timing.py --batch 32 --precision bf16x3 2>&1 | grep -v amdgpu.ids | tee $OUT/timing_bf16x3.log
timeout 600 python tests/gpu_timing.py --batch 32 --precision bf16 2>&1 | grep -v amdgpu.ids | tee $OUT/timing_bf16.log
timeout 600 python bench.py --steps 10 --warmup 3 --precision bf16x3 --skip-cpu-baseline 2> $OUT/bench_x3_err.log | tee $OUT/bench_x3.json
tail -5 $OUT/bench_x3_err.log
