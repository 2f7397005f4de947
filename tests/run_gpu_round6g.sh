#!/bin/bash
# round 6: sharded matching with query-sliced post-processing — two real ranks on the GPU, shard projection (xGMI time charged)
cd "$(dirname "$0")/.."
R=$(pwd)
TAG=${1:-r06_g}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_fullsize.py -q --tb=short -x -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/pytest_sharded.log
for p in f16r f16mx bf16x3 bf16; do
  timeout 300 python tests/gpu_shardbench.py 1,2,4,8 $p 2 2 1 2>&1 | grep -v amdgpu.ids | tee -a $OUT/shardbench_sliced.log
  timeout 300 python tests/gpu_shardbench.py 1,8 $p 2 2 0 2>&1 | grep -v amdgpu.ids | tee -a $OUT/shardbench_replicated.log
done
OIBL_BENCH_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --skip-cpu-baseline --skip-api --skip-fast-mode 2> $OUT/bench_2ranks_err.log | tee $OUT/bench_2ranks_shared.json | cut -c1-300
