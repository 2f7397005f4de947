#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
for f in conv descriptor matching; do
  timeout 900 python -m pytest tests/test_gpu_$f.py -m gpu -q --tb=short --timeout 300 -p no:cacheprovider -rP \
      > gpurun_out/pytest_$f.log 2>&1
  echo "test_gpu_$f exit $?"; grep -h "passed\|failed\|^FAILED" gpurun_out/pytest_$f.log | tail -8
done
for t in 0 1 2 3; do
timeout 600 python tests/gpu_timing.py --batch 32 --precision bf16 --tile $t 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/timing32.log
done
timeout 900 python bench.py --steps 10 --warmup 3 --skip-cpu-baseline 2> gpurun_out/bench_err.log | tee gpurun_out/bench.json
