#!/bin/bash
# One GPU call per checkpoint: full GPU parity suite, smoke, bench line, per-layer timing,
# rocprofv3 kernel stats and HBM PMC passes.  Everything lands in gpurun_out/<tag>/.
# usage: tests/run_gpu_round.sh <tag> [quick]
cd "$(dirname "$0")/.."
R=$(pwd)
TAG=${1:-r01}
QUICK=${2:-}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt
if [ -z "$QUICK" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --tb=short --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
  tail -n 5 $OUT/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log
fi
timeout 900 python bench.py --steps 20 --warmup 3 2> $OUT/bench_err.log | tee $OUT/bench.json
# the launch line the driver uses for N > 1, with one rank: RCCL init / barrier / all-reduce path
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 \
  bench.py --gpus 1 --steps 10 --warmup 2 --skip-cpu-baseline 2> $OUT/bench_torchrun_err.log | tee $OUT/bench_torchrun.json
timeout 600 python tests/gpu_timing.py --batch 32 --precision bf16 2>&1 | grep -v amdgpu.ids | tee $OUT/timing_bf16.log
if [ -z "$QUICK" ]; then
  timeout 600 python tests/gpu_timing.py --batch 8 --precision fp32 2>&1 | grep -v amdgpu.ids | tee $OUT/timing_fp32.log
  timeout 300 python tests/gpu_pcie_rate.py 2>&1 | grep -v amdgpu.ids | tee $OUT/pcie.log
  timeout 300 python tests/gpu_shardbench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/shardbench.log
fi
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --skip-matching --skip-cpu-baseline"
# kernel stats over a run long enough that steady-state launches dominate the averages
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- python $R/bench.py --steps 40 --warmup 5 --skip-matching --skip-cpu-baseline > $OUT/prof_stats.log 2>&1
if [ -z "$QUICK" ]; then
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o bench -- python $R/bench.py $ARGS > $OUT/prof_fetch.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o bench -- python $R/bench.py $ARGS > $OUT/prof_write.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_match -o bench -- python $R/bench.py --steps 2 --warmup 1 --skip-cpu-baseline > $OUT/prof_match.log 2>&1
fi
cd $R
# keep the merged output small: drop per-dispatch traces larger than 8 MiB
find $OUT -type f -size +8M -print -delete
find $OUT -type f | head -50
