#!/bin/bash
# One GPU call per evidence set (round 3): full GPU parity suite, smoke, the bench line (f16mx headline), the
# self-launched 2-rank flow check, per-layer timing in the three matrix-core modes, single-image latency,
# shard scaling, a sustained run with clock / power, rocprofv3 kernel stats (f16mx, bf16) and the HBM PMC
# passes for f16mx.  Everything lands in gpurun_out/<tag>/; tools/prof_summary.py condenses it into profiles/.
# usage: tests/run_gpu_round3.sh <tag> [quick]
cd "$(dirname "$0")/.."
R=$(pwd)
TAG=${1:-r03}
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
timeout 900 python bench.py 2> $OUT/bench_err.log | tee $OUT/bench.json | cut -c1-400
# the launch line the driver uses for N > 1, with one rank: RCCL init / barrier / all-reduce path
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 \
  bench.py --gpus 1 --steps 10 --warmup 3 --skip-cpu-baseline --skip-api 2> $OUT/bench_torchrun_err.log | tee $OUT/bench_torchrun.json | cut -c1-300
# `python bench.py --gpus 2` launches its two ranks itself; on this one-GPU box they share the GPU over gloo
# (flow check of barrier / max-reduce / sharded matching with two real ranks; the numbers mean nothing)
OIBL_BENCH_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --skip-cpu-baseline --skip-api \
  2> $OUT/bench_2ranks_shared_err.log | tee $OUT/bench_2ranks_shared.json | cut -c1-300
timeout 600 python tests/gpu_precbench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/precbench.log
if [ -z "$QUICK" ]; then
  timeout 600 python tests/gpu_latency.py $OUT/latency.md 2>&1 | grep -v amdgpu.ids | tee $OUT/latency.log
  for p in f16mx bf16x3 bf16; do
    timeout 300 python tests/gpu_shardbench.py 1,2,4,8 $p 2>&1 | grep -v amdgpu.ids | tee -a $OUT/shardbench.log
  done
  timeout 200 python tests/gpu_mfma_peak.py 6 $OUT/mfma_peak.md 2>&1 | grep -v amdgpu.ids | tee $OUT/mfma_peak.log
  timeout 100 build/mx_probe 2 > $OUT/mx_probe.log 2>&1
  timeout 300 python bench.py --sustain 12 --precision f16mx 2>> $OUT/bench_err.log > $OUT/sustain_f16mx.json
  timeout 300 python bench.py --sustain 12 --precision bf16 2>> $OUT/bench_err.log > $OUT/sustain_bf16.json
  timeout 120 python tests/gpu_mx_stamps.py 2>&1 | grep -v amdgpu.ids | tee $OUT/mx_stamps.log
  timeout 120 python tests/gpu_stem_mx_bench.py 0 2>&1 | grep -v amdgpu.ids | tee $OUT/stem_mx.log
  timeout 120 python tests/gpu_pca_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/pca_small.log
fi
cd /tmp && export TMPDIR=/tmp
# --no-pipeline: one lane, so that the per-kernel durations are those of the roofline's span leg
SKIP="--no-pipeline --skip-matching --skip-cpu-baseline --skip-api --skip-fast-mode"
for p in f16mx bf16; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_$p -o bench -- python $R/bench.py --precision $p --steps 40 --warmup 5 $SKIP > $OUT/prof_stats_$p.log 2>&1
done
if [ -z "$QUICK" ]; then
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch_f16mx -o bench -- python $R/bench.py --precision f16mx --steps 5 --warmup 2 $SKIP > $OUT/prof_fetch_f16mx.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write_f16mx -o bench -- python $R/bench.py --precision f16mx --steps 5 --warmup 2 $SKIP > $OUT/prof_write_f16mx.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_match -o bench -- python $R/bench.py --steps 2 --warmup 1 --skip-cpu-baseline --skip-api > $OUT/prof_match.log 2>&1
fi
cd $R
# keep the merged output small: drop per-dispatch traces larger than 8 MiB
find $OUT -type f -size +8M -print -delete
find $OUT -type f | wc -l
