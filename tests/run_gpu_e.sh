#!/bin/bash
cd "$(dirname "$0")/.."
R=$(pwd)
OUT=$R/gpurun_out/${1:-r02e}
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest_gpu.log | tail -20
for p in bf16x3 bf16; do
  timeout 600 python tests/gpu_timing.py --batch 32 --precision $p 2>&1 | grep -v amdgpu.ids | tee $OUT/timing_$p.log
done
timeout 600 python bench.py --skip-cpu-baseline --skip-api --skip-matching 2> $OUT/bench_err.log | tee $OUT/bench.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
SKIP="--skip-matching --skip-cpu-baseline --skip-api --skip-fast-mode"
for p in bf16x3 bf16; do
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch_$p -o bench -- python $R/bench.py --precision $p --steps 5 --warmup 2 $SKIP > $OUT/prof_fetch_$p.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write_$p -o bench -- python $R/bench.py --precision $p --steps 5 --warmup 2 $SKIP > $OUT/prof_write_$p.log 2>&1
done
cd $R
find $OUT -type f -size +8M -print -delete
