#!/bin/bash
# round 6, matching: Tokyo-shape parity tests, f16r tests, k = 10 / 120 timing + rocprofv3 kernel stats at the Tokyo
# shape and at the benchmark shape.      usage: tests/run_gpu_round6a.sh <tag>
cd "$(dirname "$0")/.."
R=$(pwd)
TAG=${1:-r06_a}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
python -m pytest tests/test_gpu_tokyo.py tests/test_gpu_f16r.py -x -q -s -p no:cacheprovider > $OUT/tokyo_tests.log 2>&1; echo "rc=$?" >> $OUT/tokyo_tests.log
tail -3 $OUT/tokyo_tests.log
python -m pytest tests/test_gpu_matching.py tests/test_gpu_fullsize.py tests/test_gpu_two_ranks.py -x -q -p no:cacheprovider > $OUT/match_tests.log 2>&1; echo "rc=$?" >> $OUT/match_tests.log
tail -3 $OUT/match_tests.log
for k in 10 120; do
  python tests/gpu_matchbench.py --q 315 --g 75984 --k $k --iters 20 --only prepared:f16mx,f16r 2>&1 | grep -v amdgpu.ids | sed "s/^/[315x75984 k=$k] /" >> $OUT/matchbench.txt
  python tests/gpu_matchbench.py --q 8192 --g 81920 --k $k --iters 5 --only prepared:f16mx,f16r 2>&1 | grep -v amdgpu.ids | sed "s/^/[8192x81920 k=$k] /" >> $OUT/matchbench.txt
done
cat $OUT/matchbench.txt
cd /tmp && export TMPDIR=/tmp
for cfg in "315 75984 10" "315 75984 120" "8192 81920 120"; do
  set -- $cfg
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$1_k$3 -o m -- python $R/tests/gpu_matchbench.py --q $1 --g $2 --k $3 --iters 10 --only prepared:f16r > $OUT/prof_$1_k$3.log 2>&1
done
cd $R
find $OUT -type f -size +8M -print -delete
find $OUT -name "*kernel_stats.csv" | head
