#!/bin/bash
# round 5, call E: the fused NetVLAD layer (parity + timing)
cd "$(dirname "$0")/.."
R=$(pwd); OUT=$R/gpurun_out/r5f; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_netvlad_pca.py tests/test_gpu_descriptor.py tests/test_gpu_api.py -q --tb=short --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest.log
grep -v amdgpu.ids $OUT/pytest.log | tail -n 25
timeout 300 python tests/gpu_head_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/head_bench.log
timeout 300 python tests/gpu_latency.py $OUT/latency.md 2>&1 | grep -v amdgpu.ids | tail -12 | tee $OUT/latency.log
