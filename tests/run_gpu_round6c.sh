#!/bin/bash
# round 6, stem: parity tests of everything the f16mx stem feeds + its role breakdown + per-layer timing + bench line.
cd "$(dirname "$0")/.."
R=$(pwd)
TAG=${1:-r06_c}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests/test_gpu_mx.py tests/test_gpu_u8.py tests/test_gpu_range.py tests/test_gpu_api.py tests/test_gpu_descriptor.py -q --tb=short -x -p no:cacheprovider > $OUT/pytest_stem.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_stem.log
tail -n 6 $OUT/pytest_stem.log
timeout 300 python tests/gpu_stem_mx_bench.py 0 2>&1 | grep -v amdgpu.ids | tee $OUT/stem_roles.txt
timeout 600 python tests/gpu_precbench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/precbench.log
timeout 600 python bench.py --skip-cpu-baseline --skip-api --skip-fast-mode 2> $OUT/bench_err.log | tee $OUT/bench.json | cut -c1-600
