#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q --tb=short --timeout 300 -p no:cacheprovider -rP > gpurun_out/pytest_conv.log 2>&1
echo "conv exit $?"; grep -h "passed\|failed\|^FAILED\|fraction" gpurun_out/pytest_conv.log | tail -20
timeout 600 python -m pytest tests/test_gpu_descriptor.py -m gpu -q --tb=short --timeout 300 -p no:cacheprovider -rP > gpurun_out/pytest_descriptor.log 2>&1
echo "descriptor exit $?"; grep -h "passed\|failed\|^FAILED\|bf16 rel" gpurun_out/pytest_descriptor.log | tail
timeout 600 python tests/gpu_timing.py --batch 32 --precision bf16 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/timing_r4.log
timeout 900 python bench.py --steps 10 --warmup 3 --skip-cpu-baseline --skip-matching 2> gpurun_out/bench_err.log | tee gpurun_out/bench.json
