#!/bin/bash
# round 5, call A: the f16r matching path — parity tests, timing against bf16 / f16mx, kernel stats
cd "$(dirname "$0")/.."
R=$(pwd); OUT=$R/gpurun_out/r5a; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_f16r.py tests/test_gpu_fullsize.py -q --tb=short --timeout 600 -p no:cacheprovider -s > $OUT/pytest_f16r.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_f16r.log
grep -v amdgpu.ids $OUT/pytest_f16r.log | tail -n 40
timeout 300 python tests/gpu_matchbench.py --only prepared --iters 5 2>&1 | grep -v amdgpu.ids | tee $OUT/matchbench.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_f16r -o m -- python $R/tests/gpu_matchbench.py --only prepared:f16r --iters 5 > $OUT/prof_f16r.log 2>&1
cd $R
python - <<'P' 2>&1 | tee $OUT/prof_f16r_top.txt
import csv, glob
for f in glob.glob("gpurun_out/r5a/prof_f16r/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(f"{r['Name'][:100]:100s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f} %")
P
timeout 600 python bench.py --steps 10 --warmup 3 --skip-api --skip-fast-mode --skip-cpu-baseline 2> $OUT/bench_err.log | tee $OUT/bench.json | cut -c1-300
tail -5 $OUT/bench_err.log
find $OUT -type f -size +8M -delete
