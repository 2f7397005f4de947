#!/bin/bash
# round 5, call C: de-phased halo4 (variants), fused f16r selection, the new parity tests
cd "$(dirname "$0")/.."
R=$(pwd); OUT=$R/gpurun_out/r5c; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python tests/gpu_precbench.py --rounds 5 2>&1 | grep -v amdgpu.ids | tee $OUT/precbench.log
timeout 300 python tests/gpu_matchbench.py --only prepared:bf16,f16r --iters 5 2>&1 | grep -v amdgpu.ids | tee $OUT/matchbench.log
timeout 1500 python -m pytest tests/test_gpu_f16r.py tests/test_gpu_splitk.py tests/test_gpu_fullsize.py tests/test_gpu_evaluator.py tests/test_gpu_range.py "tests/test_gpu_mx.py::test_conv_mx_halo4_repeatable_under_load" "tests/test_gpu_mx.py::test_conv3x3_mx" -q --tb=short --timeout 900 -p no:cacheprovider -s > $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest.log
grep -v amdgpu.ids $OUT/pytest.log | grep -i "calibrated\|recalls\|configs\[\|fast mode\|passed\|failed\|error\|per-layer\|gathered" | tail -n 40
