#!/bin/bash
# small-problem crossover between f16mx and bf16x3 + kernel stats of a 224x224 forward in both
tag=${1:-r04_i}; out=gpurun_out/$tag; mkdir -p $out
timeout 300 python tests/gpu_small_sizes.py $out/small_sizes.md > $out/small_sizes.txt 2>&1
export TMPDIR=/tmp
for prec in f16mx bf16x3; do
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$prec -o p -- python $GRAFT_REPO_ROOT/tests/gpu_small_sizes.py --loop $prec 224 224 1 > /tmp/prof_$prec.log 2>&1)
  f=$(find /tmp/prof_$prec -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && head -40 "$f" | cut -c1-260 > $out/stats_224_$prec.csv
done
tail -15 $out/small_sizes.txt
