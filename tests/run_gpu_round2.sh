#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
for f in netvlad_pca matching; do
  timeout 900 python -m pytest tests/test_gpu_$f.py -m gpu -q --tb=short --timeout 300 -p no:cacheprovider -rP \
      > gpurun_out/pytest_$f.log 2>&1
  echo "test_gpu_$f exit $?"; grep -h "passed\|failed" gpurun_out/pytest_$f.log | tail -2
done
timeout 600 python -m pytest tests/test_gpu_descriptor.py -m gpu -q --tb=short --timeout 300 -p no:cacheprovider -rP > gpurun_out/pytest_descriptor.log 2>&1
grep -h "rel_l2\|cosine" gpurun_out/pytest_descriptor.log | head -40
timeout 600 python tests/gpu_timing.py --batch 32 --precision bf16 2>&1 | tee gpurun_out/timing32.log
timeout 900 python bench.py --steps 10 --warmup 3 2> gpurun_out/bench_err.log | tee gpurun_out/bench.json
tail -5 gpurun_out/bench_err.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --skip-matching --skip-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_stats.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_stats -type f | head -20
f=$(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f"
