#!/usr/bin/env python
"""Benchmark of the descriptor + matching hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: BASELINE.json configs[1], i.e. a batch of 32
synthetic 480x640 images (already resident in HBM, fp32 NCHW as the reference's loader hands them
over) through VGG16-conv5 -> NetVLAD -> PCA-4096, producing 32 descriptors, through the reference's
own API (`hubconf.vgg16_netvlad()` -> `model(x)`).  Every rank runs its own batch (weak scaling, no
data-path collective); `value` is images/s over all ranks.

The headline line is measured in **f16mx**: every operand travels as hi = fp16(v) plus block-scaled
MX-fp6 (e2m3) images of hi and of lo = v - hi; a product is hi.hi on v_mfma_f32_32x32x16_f16 plus BOTH
cross terms on ONE v_mfma_scale_f32_32x32x64_f8f6f4 (K-concatenated), fp32 accumulate — 1.5 bf16-MFMA
times per product instead of the 3 of bf16x3 — and the descriptors stay within north_star's 1e-4 of
the reference CPU path (tests/test_gpu_mx.py).  Only conv1_1 (K = 27, 0.56 % of the FLOPs; inside the
fused stem) stays in split bf16 in this mode.  Plain bf16 — descriptors at ~3e-3 — is measured in the same run
and reported under `fast_mode`, labelled as what it is; `--precision bf16x3` gives round 2's headline.

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline      the matrix-core convolution kernels (12 launches per step: the fused stem — conv1_1 +
                conv1_2 + pool in one launch — and the 11 launches conv2_1..conv5_3, in every matrix-core
                mode alike): ALGORITHMIC FLOPs of those launches / their measured span, bracketed with
                HIP events recorded on the launching stream, against the dense bf16 MFMA peak
                (2.5 PFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md).  The headline steps run on two
                lanes (two streams in flight, openibl_amd/extract.py) whose launches overlap; the span
                is therefore taken in a second timed region of the same K steps on ONE lane
                (`measured_with`).  In bf16x3 the kernels issue three MFMAs per algorithmic product:
                `issued_frac` = 3 x `frac` is the share of the matrix pipe's peak actually used,
                `frac` stays the algorithmic figure.
  fast_mode     the same step in plain bf16 with its own roofline.
  api           images/s THROUGH `ibl.evaluators.extract_features` on an in-memory loader of pinned
                host batches (PCIe copy, per-batch launch work, gather and the fname dict included; a
                3-batch warm-up call comes first, whose captured graphs the timed call replays).
  cpu_baseline  the CPU oracle (a port of the reference's path onto plain torch-CPU ops) timed on
                this box's host cores on a bounded sample of the same workload (rank 0, N=1 only);
                `matching.cpu_baseline` is the same for pairwise_distance + evaluate_all.
  matching      secondary metric of BASELINE.json: query x gallery squared-L2 pairs/s on a
                synthetic 8192 x 81920 x 4096-d problem, gallery sharded over the ranks and resident
                as prepared operands, per-shard top-k + all_gather merge (strong scaling).

`--sustain S` replays the timed path for >= S seconds instead and prints per-second throughput with
the GPU clock / power read from rocm-smi (profiles/r02_*_sustain.md).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import threading
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

BF16_MFMA_PEAK_TFLOPS = 2500.0    # dense; MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
F32_MFMA_PEAK_TFLOPS = 157.3
HEIGHT, WIDTH = 480, 640
# VGG16 conv stack (cin, cout, spatial divisor)
_LAYERS = [(3, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 2), (128, 256, 4), (256, 256, 4),
           (256, 256, 4), (256, 512, 8), (512, 512, 8), (512, 512, 8), (512, 512, 16),
           (512, 512, 16), (512, 512, 16)]


def igemm_flops_per_image(h=HEIGHT, w=WIDTH) -> float:
    return float(sum(2 * (h // d) * (w // d) * cout * 9 * cin for cin, cout, d in _LAYERS[1:]))


def conv11_flops_per_image(h=HEIGHT, w=WIDTH) -> float:
    return 2.0 * h * w * 64 * 27


def total_flops_per_image(h=HEIGHT, w=WIDTH) -> float:
    p = (h // 16) * (w // 16)
    return (igemm_flops_per_image(h, w) + conv11_flops_per_image(h, w)
            + 2 * 2 * p * 64 * 512 + 2 * 4096 * 32768)            # NetVLAD (2 GEMMs) + PCA


class Ctx:
    pass


def self_launch_argv(n_gpus, argv, port=None):
    """The command line `python bench.py --gpus N` replaces itself with when nobody launched it as a
    rank: one process per GPU on this node over RCCL, rendezvous on 127.0.0.1 and a free port."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *argv]


def time_extraction(c, model, x, precision, steps, warmup, eager=False, pipeline=True):
    """K timed steps of the descriptor path in `precision`: barrier + synchronize on both sides, max
    over ranks.  Returns the result dict (value = images/s over all ranks)."""
    dev, dist = c.dev, c.dist
    model.set_precision(precision)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:           # create the underlying hipEvents
        a.record()
        b.record()
    # Timed steps replay the forward as two hipGraphs (backbone: the matrix-core launches; head:
    # NetVLAD + PCA) — `model.graphed(x)`, the same kernels on the same data as `model(x)`, but two
    # graph launches per step instead of ~30 kernel launches, so that the number does not depend on
    # how quickly a shared, possibly busy host core issues launches.  Step i runs on lane i % 2 (two
    # streams = two hardware queues, each with its own buffers: openibl_amd/extract.py), so the head
    # and the partial last rounds of one step's launches are back-filled by the other lane's
    # (--no-pipeline: one stream); every step has completed when the closing barrier returns.
    # With two lanes the launch durations of consecutive steps overlap, so the matrix-core span of
    # the roofline is measured in a second timed region of K steps on ONE lane (events recorded on
    # the launching stream around the backbone graph).  --eager times `model(x)` launch by launch.
    launch_mode, fwd, fwd1 = "eager", None, None
    bm = model.base_model
    runs0, fb0 = dict(bm.precision_runs), bm.range_fallbacks
    with torch.no_grad():
        ref = model(x).clone()          # packs the weights, sizes the workspaces (not a timed path)
        if not eager:
            try:
                fwd = model.graphed(x, pipeline=pipeline)
                got = [fwd(), fwd()]    # both pipeline slots, in flight together
                fwd.wait()
                torch.cuda.synchronize(dev)
                assert all(torch.equal(g_, ref) for g_ in got), "graph replay differs from the eager forward"
                launch_mode = ("hipGraph x2 per step, step i on lane i % 2 (two streams in flight)"
                               if pipeline else "hipGraph x2 per step")
                if pipeline:
                    fwd1 = model.graphed(x, pipeline=False)
                    assert torch.equal(fwd1(), ref), "one-lane graph replay differs from the eager forward"
            except Exception as e:      # capture unsupported on this stack: time the eager launches
                print(f"[bench] hipGraph capture failed ({e!r}); timing eager launches", file=sys.stderr)
                fwd = None

        def step(events, f=None):
            f = f or fwd
            if f is not None:
                return f(events=events)
            model.base_model.profile_events = events
            return model(x)

        # W untimed warm-up steps of exactly the timed path, issued right before the timed region:
        # after ~20 ms of idling (graph instantiation, the host-side checks above) the chip needs
        # ~15 ms of work to return to its sustained clocks (tests/gpu_graph_overhead.py)
        c.barrier()                     # the first RCCL barrier builds its channels: not before t0
        for _ in range(max(warmup, 0)):
            step(None)
        c.barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            out = step(None if fwd1 is not None else ev[k])
        if fwd is not None:
            fwd.wait()                  # f16mx: the range flags of the last two batches are checked INSIDE the region
        c.barrier()
        t1 = time.perf_counter()
        model.base_model.profile_events = None
        elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
        if c.use_dist:
            dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        elapsed = float(elapsed.item())
        assert tuple(out.shape) == (x.shape[0], 4096) and bool(torch.isfinite(out).all())
        assert torch.equal(out, ref), "timed forward differs from model(x)"
        # f16mx range guard: batches of the TIMED region that were re-run in bf16x3 (a replayed forward counts them
        # itself, the eager path on the backbone module) — a fallback-heavy run must not report the f16mx rate
        timed_fallbacks = (fwd.range_fallbacks if fwd is not None else bm.range_fallbacks - fb0)
        one_lane = None
        if fwd1 is not None:            # the span leg: the same K steps on one lane, with events
            for _ in range(max(warmup, 0)):
                step(None, fwd1)
            c.barrier()
            t0 = time.perf_counter()
            for k in range(steps):
                out = step(ev[k], fwd1)
            fwd1.wait()
            c.barrier()
            one_lane = x.shape[0] * steps / (time.perf_counter() - t0)
            assert torch.equal(out, ref), "one-lane timed forward differs from model(x)"
    batch = int(x.shape[0])
    value = batch * steps * c.world / elapsed
    span_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    # launches inside the span and their algorithmic FLOPs (2 flop per MAC of the convolution; the
    # hi/lo split of bf16x3 is an implementation detail of the arithmetic, not more algorithm)
    extra = 0
    if precision in ("bf16", "bf16x3", "f16mx"):
        # conv1_1 + conv1_2 + pool are ONE launch (the fused stem) and the span starts with it; a layer whose
        # last round is contracted split-K (f16mx: conv5_x at batch 32) is three launches: full rounds,
        # split remainder, reduction
        if precision == "f16mx":
            from openibl_amd import lib as _l, ops as _o
            hh, ww = HEIGHT // 2, WIDTH // 2
            for (cin, cout, _relu, pool) in _o.VGG16_CFG[2:]:
                if _l.load().oibl_conv3x3_workspace_bytes(batch, hh, ww, cin, cout, pool, _o.F16MX) > 0:
                    extra += 2
                if pool:
                    hh, ww = hh // 2, ww // 2
        launches, fl = 12 + extra, (igemm_flops_per_image() + conv11_flops_per_image()) * batch
        kernel = {"bf16": "oibl::vgg_stem_kernel (conv1_1+conv1_2+pool) + oibl::conv3x3_ring_kernel "
                          "(conv2_1..conv5_3), 12 launches/step",
                  "bf16x3": "oibl::vgg_stem_x3_kernel (conv1_1+conv1_2+pool) + oibl::conv3x3_ring_kernel<..., RING_X3> "
                            "(conv2_1..conv5_3), 12 launches/step",
                  "f16mx": "oibl::vgg_stem_x3_kernel<MX> (conv1_1 in bf16x3 + conv1_2 in f16mx + pool) + "
                           "oibl::conv3x3_ring_kernel<..., RING_MX> (conv2_x, conv4_x, conv5_x) + "
                           "oibl::conv3x3_halo_kernel (conv3_x): 12 layers in %d launches/step (conv5_x: full round "
                           "+ split-K remainder + oibl::conv_mx_splitk_reduce8_kernel)" % (12 + extra)}[precision]
    elif fwd is not None:
        # fp32: the replayed backbone graph holds conv1_1 too: 13 launches inside the span
        launches, fl = 13, (igemm_flops_per_image() + conv11_flops_per_image()) * batch
        kernel = "oibl::conv1_1_kernel + oibl::conv3x3_igemm_kernel (conv1_2..conv5_3), 13 launches/step"
    else:
        launches, fl = 12, igemm_flops_per_image() * batch
        kernel = ("oibl::conv3x3_igemm_kernel (conv1_2..conv5_3), 12 launches/step; "
                  "conv1_1 runs before the span")
    achieved = fl / (span_ms * 1e-3) / 1e12
    peak = F32_MFMA_PEAK_TFLOPS if precision == "fp32" else BF16_MFMA_PEAK_TFLOPS
    traffic, traffic_src = None, None
    tj = ROOT / "profiles" / "hbm_traffic_latest.json"
    if tj.exists():
        try:
            tdata = json.loads(tj.read_text())
            ent = tdata.get(precision) if isinstance(tdata.get(precision), dict) else \
                (tdata if precision == "bf16" and "bytes_per_launch" in tdata else None)
            if ent:
                # per launch like `achieved`: the digest's bytes per forward over this run's launch count
                per = float(ent["bytes_per_forward"]) / launches if "bytes_per_forward" in ent else float(ent["bytes_per_launch"])
                traffic, traffic_src = round(per), ent["source"]
        except Exception:
            traffic = None
    roof = {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
        "kernel": kernel, "launches_per_step": launches, "avg_launch_ms": round(span_ms / launches, 5),
        "flops_per_launch_avg": fl / launches,
        "end_to_end_tflops": round(total_flops_per_image() * value / 1e12, 2),
        "end_to_end_frac": round(total_flops_per_image() * value / 1e12 / (peak * c.world), 4),
    }
    pj = ROOT / "profiles" / "mfma_peak_latest.json"
    if pj.exists():
        # what the matrix pipe sustains on this kind of chip with NOTHING else to do (random register
        # operands, tests/gpu_mfma_peak.py): the power cap, not the 2.5 PFLOP/s of the data sheet, is
        # the ceiling a long MFMA-bound kernel can reach
        try:
            ceil_tf = float(json.loads(pj.read_text())["tflops"])
            issued = achieved * {"bf16x3": 3.0, "f16mx": 1.5}.get(precision, 1.0)
            roof["power_capped_mfma_tflops"] = ceil_tf
            roof["issued_frac_of_power_capped"] = round(issued / ceil_tf, 4)
            roof["power_capped_source"] = "profiles/mfma_peak_latest.json (tests/gpu_mfma_peak.py, measured)"
        except Exception:
            pass
    if one_lane is not None:
        roof["measured_with"] = ("one lane: a second timed region of the same K steps on one stream, so that "
                                 "launch durations do not overlap (this rank: %.1f images/s)" % one_lane)
    if precision == "bf16x3":
        roof["mfma_per_product"] = 3
        roof["issued_frac"] = round(3 * achieved / peak, 4)
    if precision == "f16mx":
        # per 32x32x32 block: 2 x v_mfma_f32_32x32x16_f16 + 1 x v_mfma_scale_f32_32x32x64_f8f6f4 = 96 cycles
        # of the matrix pipe against 64 for bf16; conv1_1 (0.56 % of the FLOPs, K = 27 padded to 32) runs at 3x in
        # bf16x3 inside the stem — ONCE per tile since round 6 (one workgroup serves both halves of conv1_2's output
        # channels; rounds 3-5 ran it in both workgroups of a tile: a factor 2 on this term)
        c11 = conv11_flops_per_image() / (igemm_flops_per_image() + conv11_flops_per_image())
        roof["matrix_pipe_time_per_product_vs_bf16"] = round((1.0 - c11) * 1.5 + c11 * 3.0 * 32.0 / 27.0, 3)
        roof["issued_frac"] = round(roof["matrix_pipe_time_per_product_vs_bf16"] * achieved / peak, 4)
    runs = {k_: v_ - runs0.get(k_, 0) for k_, v_ in bm.precision_runs.items() if v_ - runs0.get(k_, 0)}
    return {"value": round(value, 2), "ms_per_step": round(elapsed / steps * 1e3, 4),
            "launch": launch_mode, "roofline": roof, "dtype": precision,
            "range_fallbacks": int(timed_fallbacks + (fwd1.range_fallbacks if fwd1 is not None else 0)),
            "range_fallbacks_timed_region": int(timed_fallbacks), "timed_batches": int(steps),
            # backbone passes by the arithmetic they RAN in, eager + captured (a replay repeats its capture's)
            "precision_runs": runs}


class _MemLoader:
    """In-memory stand-in for DataLoader(pin_memory=True): yields (images, names) batches that sit in
    pinned host memory; a handful of distinct batches is cycled."""

    def __init__(self, batches, n_batches):
        self.batches, self.n = batches, n_batches
        self.sampler = range(n_batches * int(batches[0].shape[0]))

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            yield (self.batches[i % len(self.batches)], None)


def time_api(c, model, precision, batch, n_batches, u8):
    """images/s through ibl.evaluators.extract_features (PCIe included) on rank 0's device."""
    from ibl.evaluators import extract_features, extract_cnn_feature
    from openibl_amd import synth
    model.set_precision(precision)
    base = synth.images(batch, HEIGHT, WIDTH, seed=900)
    if u8:
        from ibl.utils.data import MEAN, STD
        mean = torch.tensor(MEAN).view(1, 3, 1, 1)
        std = torch.tensor(STD).view(1, 3, 1, 1)
        base = ((base * std + mean) * 255.0).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    pinned = [base.roll(s, 0).contiguous().pin_memory() for s in range(3)]
    names = [(f"im{i:06d}.jpg", i, 0.0, 0.0) for i in range(n_batches * batch)]
    # one untimed pass of the timed length: captures the graphs, and pays what a process pays once on this path
    # (the first pass from freshly pinned fp32 batches measured 5-8 % below the following ones)
    extract_features(model, _MemLoader(pinned, n_batches), names, print_freq=10 ** 9, gpu=c.dev.index)
    torch.cuda.synchronize(c.dev)
    t0 = time.perf_counter()
    feats = extract_features(model, _MemLoader(pinned, n_batches), names, print_freq=10 ** 9, gpu=c.dev.index)
    dt = time.perf_counter() - t0
    first = torch.stack([feats[n[0]] for n in names[:batch]])
    with torch.no_grad():
        eager = extract_cnn_feature(model, pinned[0], gpu=c.dev.index).cpu()
    assert torch.equal(first, eager), "extract_features differs from the eager extract_cnn_feature"
    from openibl_amd import extract as _ex
    store = _ex._GRAPH_STORES.get(_ex.unwrap_model(model))
    fallbacks = sum(f.range_fallbacks for f in store[1].values()) if store else 0
    return {"value": round(n_batches * batch / dt, 1), "unit": "images/s", "images": n_batches * batch,
            "seconds": round(dt, 4), "precision": precision, "range_fallbacks": int(fallbacks),
            "input": "uint8 NHWC, pinned host batches" if u8 else "fp32 NCHW (normalised), pinned host batches",
            "through": "ibl.evaluators.extract_features (H2D copy, forward, extra L2 normalise, D2H gather, "
                       "fname dict); descriptors torch.equal to extract_cnn_feature"}


def time_matching(c, precision, Q, G, msteps=10):
    """`precision` is the model's; the fused distance + top-k runs in ops.topk_precision(precision): an f16mx
    model's fp32 descriptors are matched by the fp16 filter pass + exact rescoring ("f16r": fp32-exact lists)."""
    from openibl_amd import ops, sharded
    dev, dist = c.dev, c.dist
    mp = {ops.BF16: "bf16", ops.F32: "fp32", ops.BF16X3: "bf16x3", ops.F16MX: "f16mx",
          ops.F16R: "f16r"}[ops.topk_precision(precision)]
    start, per, n_valid = sharded.slice_bounds(G, c.rank, c.world)
    gq = torch.Generator(device=dev).manual_seed(7)
    q = torch.nn.functional.normalize(torch.randn((Q, 4096), generator=gq, device=dev), dim=1)
    gg = torch.Generator(device=dev).manual_seed(11 + c.rank)
    g = torch.nn.functional.normalize(torch.randn((n_valid, 4096), generator=gg, device=dev), dim=1)
    # the gallery shard is resident: its norms / operand rows are prepared once, like an index that
    # serves many query batches.  The QUERIES enter every step the way they leave extraction: rank r holds
    # the Q / W rows it extracted, prepares THOSE (norms + operand rows) and the prepared parts are
    # all-gathered (north_star's all-gather; sharded.gather_prepared_queries) — inside the timed step.  The
    # local top-k then runs in query blocks whose exchange + merge overlap the next block's matrix work.
    gp = ops.PreparedRows(g, mp)
    qs, qper, _ = sharded.slice_bounds(Q, c.rank, c.world)
    q_local = q[(qs + torch.arange(qper, device=dev)) % Q].contiguous() if c.world > 1 else q   # (wrapped slice)
    blocks = 2 if c.world > 1 else 1

    def step():
        if c.world > 1:     # both exchanges in sub-blocks under the matrix work (sharded.sharded_topk_pipelined)
            return sharded.sharded_topk_pipelined(q_local, Q, gp, 10, start, mp, blocks=blocks)
        qp = sharded.gather_prepared_queries(q_local, Q, mp)
        return sharded.sharded_topk(qp, gp, 10, start, mp, blocks=blocks)
    for _ in range(3):
        step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(msteps)]
    c.barrier()
    t0 = time.perf_counter()
    for k_ in range(msteps):
        ev[k_][0].record()
        vals, idx = step()
        ev[k_][1].record()
    c.barrier()
    t1 = time.perf_counter()
    mt = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if c.use_dist:
        dist.all_reduce(mt, op=dist.ReduceOp.MAX)
    pairs = float(Q) * G * msteps / float(mt.item())
    span_ms = sum(a.elapsed_time(b) for a, b in ev) / msteps
    # correctness of the TIMED lists (rank 0's shard when sharded: world 1 only): a 256-row block against the
    # top-k of the materialised matrix.  bf16 / f16mx / bf16x3: the fused lists ARE the top-k of their own matrix
    # (torch.equal); f16r has no matrix of its own — its lists are those of the fp32 matrix except where fp64
    # calls the two candidates a near-tie (the fp32 MFMA matrix carries ~1e-6 of rounding, the f16r values do not)
    check = None
    if c.world == 1:
        rows = slice(Q // 2, Q // 2 + 256)
        dm = ops.pairwise_sqdist(q[rows].contiguous(), g, "fp32" if mp == "f16r" else mp)
        wv, wi = ops.row_topk(dm, 10)
        if mp != "f16r":
            assert torch.equal(idx[rows], wi) and torch.equal(vals[rows], wv), "timed matching lists != top-k of the matrix"
            check = "timed lists torch.equal to row_topk of the materialised matrix on rows %d..%d" % (rows.start, rows.stop)
        else:
            assert float((vals[rows] - wv).abs().max()) <= 2e-6, "timed f16r values differ from the fp32 matrix"
            differ = (idx[rows] != wi).nonzero()
            worst = 0.0
            for r_, c_ in differ.tolist():
                qq = q[rows.start + r_].double()
                a = float(((qq - g[int(idx[rows][r_, c_])].double()) ** 2).sum())
                b = float(((qq - g[int(wi[r_, c_])].double()) ** 2).sum())
                worst = max(worst, abs(a - b))
            assert worst < 4e-6, "timed f16r lists differ from the fp32 matrix beyond fp32 near-ties"
            check = ("timed lists against row_topk of the fp32 matrix on rows %d..%d: values within 2e-6, %d of 2560 "
                     "indices differ, all fp64 near-ties within %.1e" % (rows.start, rows.stop, len(differ), worst))
        del dm
    flops = 8192.0 * Q * G
    achieved = flops / (span_ms * 1e-3) / 1e12
    traffic, traffic_src = None, None
    tj = ROOT / "profiles" / "hbm_traffic_latest.json"
    if tj.exists():
        try:
            ent = json.loads(tj.read_text()).get("matching_" + mp)
            if ent:
                traffic, traffic_src = round(float(ent["bytes_per_launch"])), ent["source"]
        except Exception:
            traffic = None
    kern = {"f16r": "oibl::pairwise_f16r_kernel<true> (fp16 filter pass: 98 % of the step) + sample pass, row_topk x2, "
                    "oibl::f16r_rescore_kernel",
            "bf16": "oibl::pairwise_ring_kernel<true, RING_BF16> + sample pass, row_topk x2",
            "f16mx": "oibl::pairwise_ring_kernel<true, RING_MX_EARLY> + sample pass, row_topk x2",
            "bf16x3": "oibl::pairwise_ring_kernel<true, RING_X3> + sample pass, row_topk x2"}.get(mp, mp)
    roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "kernel": kern, "span_ms": round(span_ms, 4),
            "algorithmic_flops_per_step": flops,
            "measured_with": "HIP events on the launching stream around every timed step (rank 0; the whole step: "
                             "query preparation, sample pass, filter pass, selection, rescoring)"}
    del q, g, gp
    return {"metric": "query_gallery_pairs_per_sec", "value": pairs, "unit": "pairs/s",
            "ms_per_step": float(mt.item()) / msteps * 1e3, "scaling": "strong",
            "tflops": round(pairs * 8192 / 1e12, 2), "roofline": roof, "correctness": check,
            "config": {"workload": f"{Q} queries x {G} gallery x 4096-d squared-L2 + top-10, gallery sharded "
                                   f"{c.world}-way and resident as prepared operands; per step: the Q/{c.world} queries "
                                   f"of every rank travel in {blocks} sub-block(s) (all_gather under the previous "
                                   f"sub-block's matrix work), local top-k per sub-block, top-k all_gather + merge under "
                                   f"the next one",
                       "precision": mp, "model_precision": precision,
                       "arithmetic": {"f16r": "fp16 filter pass (v_mfma_f32_32x32x16_f16, per-row power-of-two scales) "
                                              "with a rigorous per-pair error bound + fp64-accumulated rescoring of the "
                                              "k + few survivors per query from the resident fp32 rows: fp32-exact lists",
                                      "bf16": "bf16 operands (fast mode: lists at bf16 accuracy)"}.get(mp, mp)}}


def cpu_baselines():
    """The oracle on this box's host cores: extraction (batch 8) and matching (1000 x 10000)."""
    from openibl_amd import synth
    from oracle import descriptor as od
    from oracle import matching as om
    sd = synth.embednetpca_state(0)
    xb = synth.images(4, HEIGHT, WIDTH, seed=100).repeat(2, 1, 1, 1)      # batch 8
    with torch.no_grad():
        od.embednetpca(xb[:2], sd)                                       # warm-up
        reps, t0 = 0, time.perf_counter()
        while reps < 3 or (time.perf_counter() - t0 < 10.0 and reps < 6):
            od.embednetpca(xb, sd)
            reps += 1
        dt = time.perf_counter() - t0
    # BASELINE.md §3 lists N = 1, 4..32: one image (latency-shaped) and ONE pass of the bench's own batch of 32 beside
    # the batch-8 figure the value is quoted on (bounded: the whole CPU leg stays around 30 s on the GPU box's host)
    by_batch = {"8": round(8 * reps / dt, 3)}
    with torch.no_grad():
        t0 = time.perf_counter()
        for _ in range(3):
            od.embednetpca(xb[:1], sd)
        by_batch["1"] = round(3 / (time.perf_counter() - t0), 3)
        x32 = xb.repeat(4, 1, 1, 1)
        t0 = time.perf_counter()
        od.embednetpca(x32, sd)
        by_batch["32"] = round(32 / (time.perf_counter() - t0), 3)
    ext = {"value": round(8 * reps / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(),
           "kind": "port", "images_per_s_by_batch": by_batch,
           "sample": f"port, batch 8: {reps} x 8 images of 480x640 through oracle.descriptor.embednetpca — the CPU "
                     f"restatement of hubconf.vgg16_netvlad().forward (BASELINE.md §3), pinned to outputs of the "
                     f"reference itself (tests/golden), not the reference's own modules (/root/reference is not on "
                     f"this box); also 3 x 1 image and 1 x 32 images (images_per_s_by_batch); torch "
                     f"{torch.__version__} CPU fp32, {os.cpu_count()} logical cores.  A baseline, never a speed-up claim"}
    q, g, gt, pids = synth.retrieval_problem(1000, 10000, seed=4)
    om.pairwise_distance(q[:50], g[:500])
    t0 = time.perf_counter()
    d = om.pairwise_distance(q, g)
    t_pair = time.perf_counter() - t0
    t0 = time.perf_counter()
    om.evaluate_all(d.numpy(), gt, pids)
    t_eval = time.perf_counter() - t0
    mat = {"value": round(1000 * 10000 / (t_pair + t_eval), 1), "unit": "pairs/s",
           "cores": torch.get_num_threads(), "kind": "port",
           "pairwise_distance_s": round(t_pair, 4), "evaluate_all_s": round(t_eval, 4),
           "sample": "1000 queries x 10000 gallery x 4096-d through oracle.matching.pairwise_distance "
                     "(addmm on the host) + evaluate_all (full argsort + per-query loop), "
                     "ibl/evaluators.py:105-167"}
    return ext, mat


class _SmiSampler(threading.Thread):
    """rocm-smi clock / power once per second while the sustained run is in flight."""

    def __init__(self, dev_index):
        super().__init__(daemon=True)
        self.dev_index, self.rows, self.stop_flag = dev_index, [], False

    def run(self):
        while not self.stop_flag:
            t = time.perf_counter()
            try:
                r = subprocess.run(["rocm-smi", "-d", str(self.dev_index), "--showclocks", "--showpower",
                                    "--showtemp", "--json"], capture_output=True, text=True, timeout=5)
                self.rows.append((t, json.loads(r.stdout)))
            except Exception as e:
                self.rows.append((t, {"error": repr(e)}))
            time.sleep(max(0.0, 1.0 - (time.perf_counter() - t)))


def sustain(c, model, x, precision, seconds):
    model.set_precision(precision)
    with torch.no_grad():
        model(x)
        fwd = model.graphed(x, pipeline=True)
    smi = _SmiSampler(c.dev.index)
    smi.start()
    per_sec, t_start = [], time.perf_counter()
    n_chunk = 20
    while time.perf_counter() - t_start < seconds:
        torch.cuda.synchronize(c.dev)
        t0 = time.perf_counter()
        for _ in range(n_chunk):
            fwd()
        fwd.wait()
        torch.cuda.synchronize(c.dev)
        dt = time.perf_counter() - t0
        per_sec.append((round(t0 - t_start, 3), round(n_chunk * x.shape[0] / dt, 1)))
        n_chunk = max(5, int(1.0 / (dt / n_chunk)))        # ~1 s per chunk
    smi.stop_flag = True
    smi.join(timeout=3)
    rates = [r for _, r in per_sec]
    smi_rows = []
    for t, d in smi.rows:
        card = next(iter(d.values())) if isinstance(d, dict) and d and "error" not in d else {}
        smi_rows.append({"t": round(t - t_start, 2), **{k: v for k, v in card.items()
                                                         if any(s in k.lower() for s in ("sclk", "mclk", "power", "temperature (sensor junction)"))}})
    return {"metric": "descriptors_per_sec_sustained", "precision": precision, "seconds": round(time.perf_counter() - t_start, 2),
            "images_per_s_per_chunk": per_sec, "min": min(rates), "max": max(rates),
            "mean": round(sum(rates) / len(rates), 1), "rocm_smi": smi_rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--precision", default="f16mx", choices=["bf16", "f16mx", "bf16x3", "fp32"],
                    help="arithmetic of the headline line (default: f16mx, the fastest mode inside 1e-4)")
    ap.add_argument("--eager", action="store_true", help="time eager launches instead of a hipGraph replay")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run NetVLAD+PCA of a step on the same stream instead of overlapping it with the next backbone")
    ap.add_argument("--skip-matching", action="store_true")
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--skip-fast-mode", action="store_true")
    ap.add_argument("--skip-api", action="store_true")
    ap.add_argument("--api-batches", type=int, default=48)
    ap.add_argument("--queries", type=int, default=8192)
    ap.add_argument("--gallery", type=int, default=81920)
    ap.add_argument("--sustain", type=float, default=0.0, help="sustained-run mode: replay for >= S seconds")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` launches its N ranks itself — one process per GPU under
        # torch.distributed.run, as the reference does (scripts/test_dist.sh:27-32)
        os.execv(sys.executable, self_launch_argv(args.gpus, sys.argv[1:]))

    c = Ctx()
    c.rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    c.world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the hot path has no CPU implementation")
    # Flow check of the N > 1 path on a ONE-GPU box (tests/run_gpu_round2.sh): all ranks share GPU 0
    # and talk over gloo (RCCL refuses two ranks on one device).  The numbers of such a run mean
    # nothing — the ranks time-share the chip — and the line says so in config.note.
    shared_gpu = os.environ.get("OIBL_BENCH_SHARED_GPU", "") == "1"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    c.dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    c.dist = dist
    # under torch.distributed.run (RANK set) the RCCL group is always created — also for one rank,
    # so that a 1-GPU box exercises the same init / barrier / all-reduce path as an 8-GPU node
    c.use_dist = c.world > 1 or "RANK" in os.environ
    if c.use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if shared_gpu:
            dist.init_process_group("gloo", rank=c.rank, world_size=c.world)
        else:
            dist.init_process_group("nccl", rank=c.rank, world_size=c.world, device_id=c.dev)
    if c.world != args.gpus:
        # never emit a line whose rank count differs from what was asked for
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={c.world}")
    ranks_seen = 1
    if c.use_dist:
        one = torch.ones(1, dtype=torch.float32, device=c.dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())

    def barrier():
        torch.cuda.synchronize(c.dev)
        if c.use_dist:
            dist.barrier()
        torch.cuda.synchronize(c.dev)
    c.barrier = barrier

    import hubconf
    from openibl_amd import synth

    # ---- model + inputs (synthetic, seeded; 32 distinct images per rank) -------------------------
    model = hubconf.vgg16_netvlad(pretrained=False)
    model.load_state_dict(synth.embednetpca_state(0))
    model = model.to(c.dev).eval()
    x = synth.images(args.batch, HEIGHT, WIDTH, seed=100 + c.rank).contiguous().to(c.dev)

    if args.sustain > 0:
        res = sustain(c, model, x, args.precision, args.sustain)
        if c.rank == 0:
            print(json.dumps(res), flush=True)
        if c.use_dist:
            dist.destroy_process_group()
        return

    head = time_extraction(c, model, x, args.precision, args.steps, args.warmup, eager=args.eager,
                           pipeline=not args.no_pipeline)
    fast = None
    if not args.skip_fast_mode and args.precision != "bf16":
        fast = time_extraction(c, model, x, "bf16", args.steps, args.warmup, eager=args.eager,
                               pipeline=not args.no_pipeline)
        fast["note"] = ("plain bf16 operands: descriptors at ~3e-3 relative of the reference (NOT the 1e-4 "
                        "mode), a third of the matrix work")
        fast["unit"] = "images/s"

    api = None
    if not args.skip_api and c.rank == 0:
        api = {}
        with torch.no_grad():
            api[args.precision] = time_api(c, model, args.precision, args.batch, args.api_batches, u8=False)
            if args.precision != "bf16":
                api[args.precision + "_uint8_input"] = time_api(c, model, args.precision, args.batch,
                                                                args.api_batches, u8=True)
                api["bf16"] = time_api(c, model, "bf16", args.batch, args.api_batches, u8=False)
                api["bf16_uint8_input"] = time_api(c, model, "bf16", args.batch, args.api_batches, u8=True)
    barrier()

    matching = None
    if not args.skip_matching:
        matching = time_matching(c, args.precision, args.queries, args.gallery)
        if args.precision != "bf16":
            matching["fast_mode"] = time_matching(c, "bf16", args.queries, args.gallery)

    cpu = None
    if c.rank == 0 and c.world == 1 and not args.skip_cpu_baseline:
        cpu, cpu_match = cpu_baselines()
        if matching is not None:
            matching["cpu_baseline"] = cpu_match

    if c.rank == 0:
        line = {
            "metric": "descriptors_per_sec", "value": head["value"], "unit": "images/s",
            "n_gpus": c.world, "rccl_ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "batch=32 VGG16-conv5 + NetVLAD(64x512) + PCA-4096 descriptor "
                                   "extraction, 32 distinct synthetic 480x640 inputs per rank resident in "
                                   "HBM (BASELINE.json configs[1])",
                       "global_batch": args.batch * c.world, "image": f"3x{HEIGHT}x{WIDTH}",
                       "parallelism": f"dp{c.world}", "launch": head["launch"],
                       "weights": "seeded random init (openibl_amd.synth, seed 0)",
                       "arithmetic": {"f16mx": "fp16 main term + both cross terms on one MX-fp6 instruction (hi = fp16, "
                                               "block-scaled e2m3 images of hi and lo), fp32 accumulate; descriptors "
                                               "within 1e-4 of the reference CPU path; conv1_1 (K = 27) in split bf16",
                                      "bf16x3": "split bf16: (hi, lo) bf16 operand pairs, 3 MFMAs per product, fp32 "
                                                "accumulate; descriptors within 1e-4 of the reference CPU path",
                                      "bf16": "bf16 operands, fp32 accumulate; descriptors at ~3e-3",
                                      "fp32": "exact fp32 MFMA"}[args.precision]},
            "roofline": head["roofline"], "fast_mode": fast, "api": api, "cpu_baseline": cpu,
            "matching": matching,
            # f16mx range guard (activations beyond fp16 re-run a batch in bf16x3): what the TIMED batches did
            "range_fallbacks": head["range_fallbacks"], "range_fallbacks_timed_region": head["range_fallbacks_timed_region"],
            "timed_batches": head["timed_batches"], "precision_runs": head["precision_runs"],
        }
        if shared_gpu:
            line["config"]["note"] = (f"FLOW CHECK ONLY: {c.world} ranks time-share ONE GPU over gloo "
                                      "(OIBL_BENCH_SHARED_GPU=1); not a performance number")
        print(json.dumps(line), flush=True)
    if c.use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
