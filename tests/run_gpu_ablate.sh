#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python -m pytest tests/test_gpu_conv.py -m gpu -q --tb=short --timeout 300 -p no:cacheprovider 2>&1 | tail -3
for ab in 0 4; do
timeout 600 python tests/gpu_timing.py --batch 32 --precision bf16 --ablate $ab 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ablate.log
done
