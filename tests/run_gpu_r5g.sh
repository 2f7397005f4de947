#!/bin/bash
# round 5, call G: two-phase f16r across shards (two real ranks on the one GPU), shard projection, fused NetVLAD tests
cd "$(dirname "$0")/.."
R=$(pwd); OUT=$R/gpurun_out/r5g; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_f16r.py tests/test_gpu_netvlad_pca.py -q --tb=short --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest.log
grep -v amdgpu.ids $OUT/pytest.log | tail -n 25 | cut -c1-300
for p in f16r f16mx bf16x3 bf16; do
  timeout 300 python tests/gpu_shardbench.py 1,2,4,8 $p 2>&1 | grep -v amdgpu.ids | grep -v "^precision" | tee -a $OUT/shardbench.log
done
