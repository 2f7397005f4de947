#!/bin/bash
# Round profile: bench line, rocprofv3 kernel stats, HBM PMC passes, PCIe-inclusive rate.
cd "$(dirname "$0")/.."
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python bench.py --steps 20 --warmup 3 2> gpurun_out/bench_err.log | tee gpurun_out/bench.json
timeout 300 python tests/gpu_pcie_rate.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pcie.log
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --skip-matching --skip-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o bench -- python $R/bench.py $ARGS > $R/gpurun_out/prof_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o bench -- python $R/bench.py $ARGS > $R/gpurun_out/prof_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -o bench -- python $R/bench.py $ARGS > $R/gpurun_out/prof_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_match -o bench -- python $R/bench.py --steps 2 --warmup 1 --skip-cpu-baseline > $R/gpurun_out/prof_match.log 2>&1
cd $R; find gpurun_out -name "*.db" -exec ls -la {} \;
tail -2 gpurun_out/prof_fetch.log
