#!/usr/bin/env python
"""Benchmark of the descriptor + matching hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: BASELINE.json configs[1], i.e. a batch of 32
synthetic 480x640 images (already resident in HBM, fp32 NCHW as the reference's loader hands them
over) through VGG16-conv5 -> NetVLAD -> PCA-4096 in bf16, producing 32 descriptors, through the
reference's own API (`hubconf.vgg16_netvlad()` -> `model(x)`).  Every rank runs its own batch
(weak scaling, no data-path collective); `value` is images/s over all ranks.

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline      the matrix-core convolution kernels (12 launches per step, 99.8 % of the FLOPs: the
                fused conv1_1+conv1_2+pool stem and the 11 implicit-GEMM launches conv2_1..conv5_3):
                algorithmic FLOPs of the 12 launches / their measured span, bracketed with HIP
                events recorded inside the C ABI on the launching stream, against the dense bf16
                MFMA peak (2.5 PFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md).  In fp32 mode
                conv1_1 runs on the vector ALU outside the span and is not counted.
  cpu_baseline  the CPU oracle (a port of the reference's path onto plain torch-CPU ops) timed on
                this box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
  matching      secondary metric of BASELINE.json: query x gallery squared-L2 pairs/s on a
                synthetic 8192 x 81920 x 4096-d problem, gallery sharded over the ranks, per-shard
                top-k + all_gather merge (strong scaling: the gallery size is fixed).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

BF16_MFMA_PEAK_TFLOPS = 2500.0    # dense; MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
F32_MFMA_PEAK_TFLOPS = 157.3
HEIGHT, WIDTH = 480, 640
# VGG16 conv stack (cin, cout, spatial divisor); layer 0 (conv1_1) runs on the vector ALU
_LAYERS = [(3, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 2), (128, 256, 4), (256, 256, 4),
           (256, 256, 4), (256, 512, 8), (512, 512, 8), (512, 512, 8), (512, 512, 16),
           (512, 512, 16), (512, 512, 16)]


def igemm_flops_per_image(h=HEIGHT, w=WIDTH) -> float:
    return float(sum(2 * (h // d) * (w // d) * cout * 9 * cin for cin, cout, d in _LAYERS[1:]))


def total_flops_per_image(h=HEIGHT, w=WIDTH) -> float:
    p = (h // 16) * (w // 16)
    return (igemm_flops_per_image(h, w) + 2 * h * w * 64 * 27      # conv1_1
            + 2 * 2 * p * 64 * 512 + 2 * 4096 * 32768)            # NetVLAD (2 GEMMs) + PCA


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16x3", "fp32"])
    ap.add_argument("--eager", action="store_true", help="time eager launches instead of a hipGraph replay")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run NetVLAD+PCA of a step on the same stream instead of overlapping it with the next backbone")
    ap.add_argument("--skip-matching", action="store_true")
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--queries", type=int, default=8192)
    ap.add_argument("--gallery", type=int, default=81920)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the hot path has no CPU implementation")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    # under torch.distributed.run (RANK set) the RCCL group is always created — also for one rank,
    # so that a 1-GPU box exercises the same init / barrier / all-reduce path as an 8-GPU node
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if world != args.gpus and rank == 0:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    import hubconf
    from openibl_amd import ops, sharded, synth

    def barrier():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- model + inputs (synthetic, seeded) -------------------------------------------------
    model = hubconf.vgg16_netvlad(pretrained=False)
    model.load_state_dict(synth.embednetpca_state(0))
    model = model.to(dev).eval().set_precision(args.precision)
    base = synth.images(4, HEIGHT, WIDTH, seed=100 + rank)
    x = base.repeat((args.batch + 3) // 4, 1, 1, 1)[: args.batch].contiguous().to(dev)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    for a, b in ev:           # create the underlying hipEvents
        a.record()
        b.record()
    # Timed steps replay the forward as two hipGraphs (backbone: the 12 matrix-core launches; head:
    # NetVLAD + PCA) — `model.graphed(x)`, the same kernels on the same data as `model(x)`, but two
    # graph launches per step instead of ~30 kernel launches, so that the number does not depend on
    # how quickly a shared, possibly busy host core issues launches.  The head of step i runs on a
    # second stream while the backbone of step i+1 starts (--no-pipeline: one stream); every step's
    # head has completed when the closing barrier returns.  The span events are recorded on the
    # launching stream around the backbone graph.  --eager times `model(x)` launch by launch.
    launch_mode = "eager"
    fwd = None
    with torch.no_grad():
        model(x)                        # packs the weights, sizes the workspaces (not a timed path)
        if not args.eager:
            try:
                fwd = model.graphed(x, pipeline=not args.no_pipeline)
                ref = model(x)
                got = [fwd(), fwd()]    # both pipeline slots, in flight together
                fwd.wait()
                torch.cuda.synchronize(dev)
                assert all(torch.equal(g_, ref) for g_ in got), "graph replay differs from the eager forward"
                launch_mode = ("hipGraph x2 per step" if args.no_pipeline else
                               "hipGraph x2 per step, head of step i overlapped with backbone of step i+1")
            except Exception as e:      # capture unsupported on this stack: time the eager launches
                print(f"[bench] hipGraph capture failed ({e!r}); timing eager launches", file=sys.stderr)
                fwd = None

        def step(events):
            if fwd is not None:
                return fwd(events=events)
            model.base_model.profile_events = events
            return model(x)

        # W untimed warm-up steps of exactly the timed path, issued right before the timed region:
        # after ~20 ms of idling (graph instantiation, the host-side checks above) the chip needs
        # ~15 ms of work to return to its sustained clocks (tests/gpu_graph_overhead.py: +2.3 ms on
        # the first 20 steps after an idle gap, gone after a few untimed steps)
        barrier()                       # the first RCCL barrier builds its channels: not before t0
        for _ in range(max(args.warmup, 0)):
            step(None)
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            out = step(ev[k])
        barrier()
        t1 = time.perf_counter()
    model.base_model.profile_events = None
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    assert tuple(out.shape) == (args.batch, 4096) and bool(torch.isfinite(out).all())
    with torch.no_grad():     # the timed steps produced the descriptors the eager forward produces
        assert torch.equal(out, model(x)), "timed forward differs from model(x)"

    images = args.batch * args.steps * world
    value = images / elapsed
    span_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)          # 12 matrix-core launches
    fl_conv11 = 2.0 * HEIGHT * WIDTH * 64 * 27 if args.precision == "bf16" else 0.0   # inside the stem
    fl_igemm = (igemm_flops_per_image() + fl_conv11) * args.batch
    achieved = fl_igemm / (span_ms * 1e-3) / 1e12
    peak = F32_MFMA_PEAK_TFLOPS if args.precision == "fp32" else BF16_MFMA_PEAK_TFLOPS
    # HBM bytes per launch come from the committed PMC passes of this same command (they cannot be
    # collected inside the timed run): tools/prof_summary.py writes the digest next to the tables
    traffic, traffic_src = None, None
    tj = ROOT / "profiles" / "hbm_traffic_latest.json"
    if args.precision == "bf16" and tj.exists():
        try:
            tdata = json.loads(tj.read_text())
            traffic, traffic_src = round(float(tdata["bytes_per_launch"])), tdata["source"]
        except Exception:
            traffic = None
    roofline = {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
        "kernel": ("oibl::vgg_stem_kernel (conv1_1+conv1_2+pool) + oibl::conv3x3_ring_kernel "
                   "(conv2_1..conv5_3), 12 launches/step" if args.precision == "bf16" else
                   "oibl::conv3x3_igemm_kernel (conv1_2..conv5_3, 12 launches/step)"),
        "launches_per_step": 12, "avg_launch_ms": round(span_ms / 12, 5),
        "flops_per_launch_avg": fl_igemm / 12,
        "end_to_end_tflops": round(total_flops_per_image() * value / 1e12, 2),
        "end_to_end_frac": round(total_flops_per_image() * value / 1e12 / (peak * world), 4),
    }

    # ---- secondary metric: query x gallery matching, gallery sharded ---------------------------
    matching = None
    if not args.skip_matching:
        Q, G = args.queries, args.gallery
        start, per, n_valid = sharded.slice_bounds(G, rank, world)
        gq = torch.Generator(device=dev).manual_seed(7)
        q = torch.nn.functional.normalize(torch.randn((Q, 4096), generator=gq, device=dev), dim=1)
        gg = torch.Generator(device=dev).manual_seed(11 + rank)
        g = torch.nn.functional.normalize(torch.randn((n_valid, 4096), generator=gg, device=dev), dim=1)
        msteps = 10
        for _ in range(3):
            sharded.sharded_topk(q, g, 10, start, args.precision)
        barrier()
        t0 = time.perf_counter()
        for _ in range(msteps):
            vals, idx = sharded.sharded_topk(q, g, 10, start, args.precision)
        barrier()
        t1 = time.perf_counter()
        mt = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(mt, op=dist.ReduceOp.MAX)
        pairs = float(Q) * G * msteps / float(mt.item())
        matching = {"metric": "query_gallery_pairs_per_sec", "value": pairs, "unit": "pairs/s",
                    "ms_per_step": float(mt.item()) / msteps * 1e3, "scaling": "strong",
                    "tflops": round(pairs * 8192 / 1e12, 2),
                    "config": {"workload": f"{Q} queries x {G} gallery x 4096-d squared-L2 + top-10, "
                                           f"gallery sharded {world}-way, top-k all_gather merge",
                               "precision": args.precision}}
        del q, g

    # ---- CPU baseline: the oracle on this box's host cores (rank 0, single-GPU run only) -----------
    cpu = None
    if rank == 0 and world == 1 and not args.skip_cpu_baseline:
        from oracle import descriptor as od
        sd = synth.embednetpca_state(0)
        xb = synth.images(4, HEIGHT, WIDTH, seed=100).repeat(2, 1, 1, 1)      # batch 8
        with torch.no_grad():
            od.embednetpca(xb[:2], sd)                                       # warm-up
            reps, t0 = 0, time.perf_counter()
            while reps < 3 or (time.perf_counter() - t0 < 10.0 and reps < 6):
                od.embednetpca(xb, sd)
                reps += 1
            dt = time.perf_counter() - t0
        cpu = {"value": round(8 * reps / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(),
               "kind": "port",
               "sample": f"{reps} x batch 8 of 480x640 through oracle.descriptor.embednetpca "
                         f"(torch {torch.__version__} CPU fp32, {os.cpu_count()} logical cores)"}

    if rank == 0:
        line = {
            "metric": "descriptors_per_sec", "value": round(value, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "batch=32 VGG16-conv5 + NetVLAD(64x512) + PCA-4096 descriptor "
                                   "extraction, synthetic 480x640 inputs resident in HBM "
                                   "(BASELINE.json configs[1])",
                       "global_batch": args.batch * world, "image": f"3x{HEIGHT}x{WIDTH}",
                       "parallelism": f"dp{world}", "launch": launch_mode,
                       "weights": "seeded random init (openibl_amd.synth, seed 0)"},
            "roofline": roofline, "cpu_baseline": cpu, "matching": matching,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
