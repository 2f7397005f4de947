#!/bin/bash
# Round 4, second GPU call: the full GPU suite after the stem fix + split-K, smoke, bench line, latency table,
# shard projection.   usage: tests/run_gpu_r4b.sh <tag>
cd "$(dirname "$0")/.."
R=$(pwd)
TAG=${1:-r04_b}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests -m gpu -q --tb=short --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | tail -n 40
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log
timeout 900 python bench.py --skip-cpu-baseline 2> $OUT/bench_err.log | tee $OUT/bench.json | cut -c1-700
tail -n 3 $OUT/bench_err.log
timeout 600 python tests/gpu_latency.py $OUT/latency.md 2>&1 | grep -v amdgpu.ids | tee $OUT/latency.log
