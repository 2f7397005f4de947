#!/bin/bash
# round 5, call D: are the two workgroups of a CU in phase (halo4 stamps)?  halo4 on every layer; f16r with the
# two-launch selection; shard projection with both exchanges pipelined
cd "$(dirname "$0")/.."
R=$(pwd); OUT=$R/gpurun_out/r5d; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python tests/gpu_halo4_phase.py 2>&1 | grep -v amdgpu.ids | tee $OUT/halo4_phase.log
timeout 600 python tests/gpu_precbench.py --rounds 3 2>&1 | grep -v amdgpu.ids | tee $OUT/precbench.log
timeout 300 python tests/gpu_matchbench.py --only prepared:bf16,f16r --iters 5 2>&1 | grep -v amdgpu.ids | tee $OUT/matchbench.log
for p in f16r f16mx bf16; do
  timeout 300 python tests/gpu_shardbench.py 1,2,4,8 $p 2>&1 | grep -v amdgpu.ids | tee -a $OUT/shardbench.log
done
timeout 900 python -m pytest tests/test_gpu_f16r.py "tests/test_gpu_fullsize.py" -q --tb=short --timeout 900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest.log
grep -v amdgpu.ids $OUT/pytest.log | tail -5
