#!/bin/bash
# One GPU call per checkpoint (round 2): full GPU parity suite, smoke, the bench line, the one-rank
# torchrun line, per-layer timing in both matrix-core modes, single-image latency, shard scaling,
# sustained runs with clock / power, rocprofv3 kernel stats and HBM PMC passes for bf16x3 and bf16.
# Everything lands in gpurun_out/<tag>/;  tools/prof_summary.py condenses it into profiles/.
# usage: tests/run_gpu_round2.sh <tag> [quick]
cd "$(dirname "$0")/.."
R=$(pwd)
TAG=${1:-r02}
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
# flow check of the N = 2 launch line on this one-GPU box: two ranks share the GPU over gloo (the
# numbers mean nothing; what is checked is that the line comes out: barrier / max-reduce / sharded
# matching with two real ranks)
OIBL_BENCH_SHARED_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29543 \
  bench.py --gpus 2 --steps 5 --warmup 2 --skip-cpu-baseline --skip-api 2> $OUT/bench_2ranks_shared_err.log | tee $OUT/bench_2ranks_shared.json | cut -c1-300
for p in bf16x3 bf16; do
  timeout 600 python tests/gpu_timing.py --batch 32 --precision $p 2>&1 | grep -v amdgpu.ids | tee $OUT/timing_$p.log
done
timeout 600 python tests/gpu_timing.py --batch 32 --precision bf16x3 --raster 1 2>&1 | grep -v amdgpu.ids | tee $OUT/timing_bf16x3_raster1.log
timeout 600 python tests/gpu_timing.py --batch 32 --precision bf16 --raster 1 2>&1 | grep -v amdgpu.ids | tee $OUT/timing_bf16_raster1.log
if [ -z "$QUICK" ]; then
  timeout 600 python tests/gpu_timing.py --batch 8 --precision fp32 2>&1 | grep -v amdgpu.ids | tee $OUT/timing_fp32.log
  timeout 600 python tests/gpu_latency.py $OUT/latency.md 2>&1 | grep -v amdgpu.ids | tee $OUT/latency.log
  timeout 300 python tests/gpu_pcie_rate.py 2>&1 | grep -v amdgpu.ids | tee $OUT/pcie.log
  timeout 300 python tests/gpu_shardbench.py 1,2,4,8 bf16 2>&1 | grep -v amdgpu.ids | tee $OUT/shardbench.log
  timeout 300 python tests/gpu_shardbench.py 1,2,4,8 bf16x3 2>&1 | grep -v amdgpu.ids | tee -a $OUT/shardbench.log
  # the matrix pipe with nothing else to do (random register operands): the power-capped ceiling
  timeout 200 python tests/gpu_mfma_peak.py 6 $OUT/mfma_peak.md 2>&1 | grep -v amdgpu.ids | tee $OUT/mfma_peak.log
  timeout 300 python bench.py --sustain 12 --precision bf16x3 2>> $OUT/bench_err.log > $OUT/sustain_bf16x3.json
  timeout 300 python bench.py --sustain 12 --precision bf16 2>> $OUT/bench_err.log > $OUT/sustain_bf16.json
fi
cd /tmp && export TMPDIR=/tmp
# --no-pipeline: one lane, so that the per-kernel durations are those of the roofline's span leg
# (with two lanes in flight the launches of consecutive steps overlap and every duration stretches)
SKIP="--no-pipeline --skip-matching --skip-cpu-baseline --skip-api --skip-fast-mode"
for p in bf16x3 bf16; do
  # kernel stats over a run long enough that steady-state launches dominate the averages
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_$p -o bench -- python $R/bench.py --precision $p --steps 40 --warmup 5 $SKIP > $OUT/prof_stats_$p.log 2>&1
  if [ -z "$QUICK" ]; then
    timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch_$p -o bench -- python $R/bench.py --precision $p --steps 5 --warmup 2 $SKIP > $OUT/prof_fetch_$p.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write_$p -o bench -- python $R/bench.py --precision $p --steps 5 --warmup 2 $SKIP > $OUT/prof_write_$p.log 2>&1
  fi
done
if [ -z "$QUICK" ]; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_match -o bench -- python $R/bench.py --steps 2 --warmup 1 --skip-cpu-baseline --skip-api > $OUT/prof_match.log 2>&1
fi
cd $R
if [ -z "$QUICK" ]; then
  # MFMA-busy / clock / L2 hit / LDS bank conflicts per kernel, both modes + matching (eager launches)
  bash tests/run_gpu_pmc_bench.sh ${TAG}/pmc > $OUT/pmc.log 2>&1
fi
# keep the merged output small: drop per-dispatch traces larger than 8 MiB
find $OUT -type f -size +8M -print -delete
find $OUT -type f | wc -l
