#!/bin/bash
# Round 4, fourth GPU call: the fused uint8 stems (new tests + the uint8 tests they touch), the API leg of the bench.
cd "$(dirname "$0")/.."
R=$(pwd)
TAG=${1:-r04_d}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_u8.py tests/test_gpu_x3.py tests/test_gpu_range.py tests/test_gpu_descriptor.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -s > $OUT/pytest_u8.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_u8.log
grep -E "^(FAILED|ERROR)|passed|failed|uint8 stem vs" $OUT/pytest_u8.log | tail -n 40
timeout 900 python bench.py --skip-cpu-baseline --skip-matching 2> $OUT/bench_err.log | tee $OUT/bench.json | cut -c1-300
tail -n 3 $OUT/bench_err.log
python - <<'P'
import json
d=json.load(open("gpurun_out/r04_d/bench.json"))
print("headline", d["value"], "frac", d["roofline"]["frac"], "api", {k: v["value"] for k, v in d["api"].items()})
P
timeout 300 python tests/gpu_u8_ab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/u8_ab.log
