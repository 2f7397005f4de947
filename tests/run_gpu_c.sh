#!/bin/bash
cd "$(dirname "$0")/.."
R=$(pwd)
OUT=$R/gpurun_out/${1:-r02c}
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
echo "pytest all exit $?" | tee -a $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest_gpu.log | tail -30
for t in 1 2 3 5; do
  timeout 300 python tests/gpu_timing.py --batch 32 --precision bf16x3 --layers 2 --tile $t 2>&1 | grep "tile=" | tee -a $OUT/timing_conv01_variants.log
done
timeout 900 python bench.py 2> $OUT/bench_err.log | tee $OUT/bench.json
tail -3 $OUT/bench_err.log
timeout 300 python bench.py --sustain 12 --precision bf16x3 2>> $OUT/bench_err.log > $OUT/sustain_bf16x3.json; tail -c 600 $OUT/sustain_bf16x3.json; echo
timeout 300 python bench.py --sustain 12 --precision bf16 2>> $OUT/bench_err.log > $OUT/sustain_bf16.json; tail -c 300 $OUT/sustain_bf16.json; echo
