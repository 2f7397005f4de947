#!/bin/bash
# Run every GPU test file in its own process (a device fault in one file must not hide the others),
# then the per-stage timing.  Logs go to gpurun_out/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
for f in core conv netvlad_pca matching descriptor evaluator; do
  echo "=== test_gpu_$f" | tee -a gpurun_out/suite.log
  timeout 900 python -m pytest tests/test_gpu_$f.py -m gpu -q --tb=short --timeout 300 -p no:cacheprovider \
      > gpurun_out/pytest_$f.log 2>&1
  echo "exit $?" | tee -a gpurun_out/suite.log
  tail -n 3 gpurun_out/pytest_$f.log | tee -a gpurun_out/suite.log
done
for args in "--batch 8 --precision bf16" "--batch 8 --precision bf16 --regstage 1" "--batch 4 --precision fp32"; do
  timeout 600 python tests/gpu_timing.py $args 2>&1 | tee -a gpurun_out/timing.log
done
grep -h "FAILED\|ERROR\|passed\|failed" gpurun_out/pytest_*.log | tail -60
