#!/bin/bash
# round 6: the full GPU parity suite + smoke + matching timings (k = 10 / 120).   usage: tests/run_gpu_round6b.sh <tag>
cd "$(dirname "$0")/.."
R=$(pwd)
TAG=${1:-r06_b}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 2400 python -m pytest tests -m gpu -q --tb=short --timeout 900 -p no:cacheprovider --durations=15 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
tail -n 30 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log
for k in 10 120; do
  python tests/gpu_matchbench.py --q 315 --g 75984 --k $k --iters 20 --only prepared:f16mx,f16r 2>&1 | grep -v amdgpu.ids | sed "s/^/[315x75984 k=$k] /" >> $OUT/matchbench.txt
done
cat $OUT/matchbench.txt
