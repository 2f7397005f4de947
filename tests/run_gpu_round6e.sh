#!/bin/bash
# round 6: the continuous-issue halo4 K loop — parity (f16mx layer shapes, variants, repeatability under load), per-layer
# timing, per-workgroup stamps.
cd "$(dirname "$0")/.."
R=$(pwd)
TAG=${1:-r06_e}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests/test_gpu_mx.py tests/test_gpu_splitk.py tests/test_gpu_range.py -q --tb=short -x -p no:cacheprovider > $OUT/pytest_halo4.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_halo4.log
tail -n 4 $OUT/pytest_halo4.log
timeout 600 python tests/gpu_precbench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/precbench.log
timeout 200 python tests/gpu_halo4_phase.py 2>&1 | grep -v amdgpu.ids | tee $OUT/halo4_phase.log
timeout 300 python tests/gpu_conv_sweep.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee $OUT/conv_sweep.log
timeout 600 python bench.py --skip-cpu-baseline --skip-api --skip-fast-mode --skip-matching 2> $OUT/bench_err.log | tee $OUT/bench.json | cut -c1-300
