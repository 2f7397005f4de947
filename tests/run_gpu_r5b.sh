#!/bin/bash
# round 5, call B: the 4-wave halo kernel of conv2_x (parity + timing against the ring), f16r on two lanes
cd "$(dirname "$0")/.."
R=$(pwd); OUT=$R/gpurun_out/r5b; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_mx.py tests/test_gpu_f16r.py tests/test_gpu_range.py tests/test_gpu_splitk.py -q --tb=short --timeout 600 -p no:cacheprovider -x > $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest.log
grep -v amdgpu.ids $OUT/pytest.log | tail -n 30
timeout 600 python tests/gpu_precbench.py --rounds 5 2>&1 | grep -v amdgpu.ids | tee $OUT/precbench.log
timeout 300 python tests/gpu_matchbench.py --only prepared:bf16,f16r --iters 5 2>&1 | grep -v amdgpu.ids | tee $OUT/matchbench.log
timeout 600 python bench.py --steps 20 --warmup 3 --skip-api --skip-fast-mode --skip-cpu-baseline 2> $OUT/bench_err.log | tee $OUT/bench.json | cut -c1-300
tail -3 $OUT/bench_err.log
