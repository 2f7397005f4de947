#!/bin/bash
# Round 4, first GPU call: the full GPU suite (range guard, graph store), the one-barrier schedule A/B
# (timing, bit identity, race screen) and a bench line.   usage: tests/run_gpu_r4a.sh <tag>
cd "$(dirname "$0")/.."
R=$(pwd)
TAG=${1:-r04_a}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python tests/gpu_bar1_ab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/bar1_ab.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
tail -n 30 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log
timeout 900 python bench.py --skip-cpu-baseline 2> $OUT/bench_err.log | tee $OUT/bench.json | cut -c1-600
tail -n 5 $OUT/bench_err.log
