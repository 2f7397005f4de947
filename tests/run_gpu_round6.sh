#!/bin/bash
# One GPU call per evidence set (round 6): full GPU parity suite, smoke, the bench line (f16mx headline, matching in
# f16r), the 1-rank torchrun line and the self-launched 2-rank flow check, per-layer timing (halo4 against the ring on
# conv2_x), single-image latency, shard projection with both exchanges pipelined, a sustained run with clock / power,
# rocprofv3 kernel stats (f16mx, bf16, matching), the HBM PMC passes for f16mx, bf16 AND the matching step, and the
# MFMA-busy / L2-hit / LDS-conflict PMC passes.  Everything lands in gpurun_out/<tag>/; tools/prof_summary.py +
# tools/pmc_summary.py condense it.        usage: tests/run_gpu_round6.sh <tag> [quick]
cd "$(dirname "$0")/.."
R=$(pwd)
TAG=${1:-r06}
QUICK=${2:-}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt
if [ -z "$QUICK" ]; then
  timeout 1800 python -m pytest tests -m gpu -q --tb=short --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
  tail -n 5 $OUT/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log
fi
timeout 900 python bench.py 2> $OUT/bench_err.log | tee $OUT/bench.json | cut -c1-400
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 \
  bench.py --gpus 1 --steps 10 --warmup 3 --skip-cpu-baseline --skip-api 2> $OUT/bench_torchrun_err.log | tee $OUT/bench_torchrun.json | cut -c1-300
OIBL_BENCH_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --skip-cpu-baseline --skip-api \
  2> $OUT/bench_2ranks_shared_err.log | tee $OUT/bench_2ranks_shared.json | cut -c1-300
timeout 600 python tests/gpu_precbench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/precbench.log
timeout 300 python tests/gpu_matchbench.py --only prepared:bf16,f16mx,f16r --iters 5 2>&1 | grep -v amdgpu.ids | tee $OUT/matchbench.log
for k in 10 120; do   # BASELINE configs[4]'s shape: Tokyo 24/7, 315 x 75 984, Recall@N (k = 10) and with spatial NMS (k = 120)
  timeout 300 python tests/gpu_matchbench.py --q 315 --g 75984 --k $k --iters 20 --only prepared:f16mx,f16r 2>&1 | grep -v amdgpu.ids | sed "s/^/[315x75984 k=$k] /" | tee -a $OUT/matchbench.log
done
if [ -z "$QUICK" ]; then
  timeout 600 python tests/gpu_latency.py $OUT/latency.md 2>&1 | grep -v amdgpu.ids | tee $OUT/latency.log
  timeout 300 python tests/gpu_small_sizes.py $OUT/small_sizes.md 2>&1 | grep -v amdgpu.ids | tee $OUT/small_sizes.log
  for p in f16r f16mx bf16x3 bf16; do   # query-sliced post-processing, xGMI time charged (tests/gpu_shardbench.py)
    timeout 300 python tests/gpu_shardbench.py 1,2,4,8 $p 2>&1 | grep -v amdgpu.ids | tee -a $OUT/shardbench.log
  done
  timeout 200 python tests/gpu_mfma_peak.py 6 $OUT/mfma_peak.md 2>&1 | grep -v amdgpu.ids | tee $OUT/mfma_peak.log
  timeout 300 python bench.py --sustain 12 --precision f16mx 2>> $OUT/bench_err.log > $OUT/sustain_f16mx.json
  timeout 300 python bench.py --sustain 12 --precision bf16 2>> $OUT/bench_err.log > $OUT/sustain_bf16.json
  timeout 200 python tests/gpu_halo4_phase.py 2>&1 | grep -v amdgpu.ids | tee $OUT/halo4_phase.log
  timeout 200 python tests/gpu_stem_mx_bench.py 0 2>&1 | grep -v amdgpu.ids | tee $OUT/stem_roles.log
  timeout 200 python tests/gpu_head_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/head_bench.log
  timeout 200 python tests/gpu_pca_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/pca_bench.log
  timeout 200 python tests/gpu_scale_probe2.py 2>&1 | grep -v amdgpu.ids | tee $OUT/scale_probe.log
fi
cd /tmp && export TMPDIR=/tmp
# --no-pipeline: one lane, so that the per-kernel durations are those of the roofline's span leg
SKIP="--no-pipeline --skip-matching --skip-cpu-baseline --skip-api --skip-fast-mode"
for p in f16mx bf16; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_$p -o bench -- python $R/bench.py --precision $p --steps 40 --warmup 5 $SKIP > $OUT/prof_stats_$p.log 2>&1
done
MATCH="python $R/tests/gpu_matchbench.py --only prepared:f16r,bf16 --iters 3"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_match -o bench -- $MATCH > $OUT/prof_match.log 2>&1
if [ -z "$QUICK" ]; then
  for p in f16mx bf16; do
    timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch_$p -o bench -- python $R/bench.py --precision $p --steps 5 --warmup 2 $SKIP > $OUT/prof_fetch_$p.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write_$p -o bench -- python $R/bench.py --precision $p --steps 5 --warmup 2 $SKIP > $OUT/prof_write_$p.log 2>&1
  done
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch_match -o bench -- $MATCH > $OUT/prof_fetch_match.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write_match -o bench -- $MATCH > $OUT/prof_write_match.log 2>&1
  for cfg in "f16mx 480 640" "f16mx 224 224"; do
    set -- $cfg
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_single_$1_$2 -o one -- python $R/tests/gpu_small_sizes.py --loop $1 $2 $3 1 > $OUT/prof_single_$1_$2.log 2>&1
  done
  # MFMA busy / clock / L2 hit / LDS bank conflicts per kernel (tools/pmc_summary.py): counters only, separate passes
  CMD="python $R/bench.py --precision f16mx --steps 3 --warmup 1 --eager --skip-cpu-baseline --skip-api --skip-fast-mode"
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1
    tail -1 $OUT/p$i.log
  done
fi
cd $R
# keep the merged output small: drop per-dispatch traces larger than 8 MiB
find $OUT -type f -size +8M -print -delete
find $OUT -type f | wc -l
