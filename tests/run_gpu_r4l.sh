#!/bin/bash
# the 8-threads-per-line split-K reduction: parity tests, single-image numbers, its kernel stats
tag=${1:-r04_l}; R=$(pwd); out=$R/gpurun_out/$tag; mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_gpu_splitk.py tests/test_gpu_range.py tests/test_gpu_mx.py -q --tb=short -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest exit $?" | tee -a $out/pytest.log
tail -3 $out/pytest.log
timeout 300 python tests/gpu_small_sizes.py $out/small_sizes.md 2>&1 | grep -v amdgpu.ids | tee $out/small_sizes.log
timeout 300 python bench.py --steps 20 --warmup 3 --skip-cpu-baseline --skip-api --skip-matching 2> $out/bench_err.log | tee $out/bench.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_single_f16mx_480 -o one -- python $R/tests/gpu_small_sizes.py --loop f16mx 480 640 1 > $out/prof_single.log 2>&1
cd $R
find $out -type f -size +8M -print -delete
