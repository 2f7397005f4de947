"""A/B of the ring schedule's barrier structure (diagnostic, not a pytest): two barriers per phase (rounds 1-3)
against ONE (ring_core.h, BAR1) on the backbone layers at the benchmark batch, the halo kernel, the distance
kernels; the two must be bit-identical, and the one-barrier kernels repeat their bits under memory load.
    python tests/gpu_bar1_ab.py [--reps 60]"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import lib, ops  # noqa: E402

LAYERS = [  # (cin, cout, H, W, relu, pool)
    (64, 128, 240, 320, 1, 0), (128, 128, 240, 320, 1, 1), (128, 256, 120, 160, 1, 0), (256, 256, 120, 160, 1, 0),
    (256, 256, 120, 160, 1, 1), (256, 512, 60, 80, 1, 0), (512, 512, 60, 80, 1, 0),
    (512, 512, 60, 80, 1, 1), (512, 512, 30, 40, 1, 0), (512, 512, 30, 40, 0, 0)]


def timed(fn, iters, rounds=5):
    ts = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters)
    return sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=60)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    h = lib.debug_hooks()
    g = torch.Generator(device=dev).manual_seed(5)
    N = a.batch
    precs = ("f16mx", "bf16", "bf16x3")
    tot = {(p, b): 0.0 for p in precs for b in (0, 1)}
    split = {"f16mx": ops.mx_split, "bf16x3": ops.x3_split, "bf16": lambda t: t.to(torch.bfloat16)}
    big = torch.randn((8192, 8192), device=dev)
    junk = torch.empty((1 << 28,), dtype=torch.float32, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    bad_total = 0
    for cin, cout, H, W, relu, pool in LAYERS:
        xf = torch.relu(torch.randn((N, H, W, cin), generator=g, device=dev)) * 3.0
        w = torch.randn((cout, cin, 3, 3), generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5
        b = torch.randn((cout,), generator=g, device=dev) * 0.1
        line = f"{cin:4d}->{cout:4d} {H:3d}x{W:3d}{' pool' if pool else '     '}"
        for p in precs:
            x, wp = split[p](xf), ops.pack_conv3x3(w, p)
            run = lambda: ops.conv3x3_nhwc(x, wp, b, bool(relu), bool(pool), p)   # noqa: E731
            outs, t = {}, {}
            for bar1 in (0, 1):
                h.oibl_debug_set_ring_bar1(bar1)
                outs[bar1] = run()
                torch.cuda.synchronize()
                t[bar1] = timed(run, 4)
                tot[(p, bar1)] += t[bar1]
            same = torch.equal(outs[0], outs[1])
            # race screen of the one-barrier kernel: same bits launch after launch, other streams loading memory
            bad = 0
            if p == "f16mx" or (cin, H) in ((512, 60), (64, 240)):
                for i in range(a.reps):
                    if i % 2:
                        with torch.cuda.stream(s1):
                            big @ big
                        with torch.cuda.stream(s2):
                            junk.add_(1.0)
                    if not torch.equal(run(), outs[1]):
                        bad += 1
                torch.cuda.synchronize()
            bad_total += bad + (0 if same else 1)
            line += f" | {p}: {t[0]:6.3f} -> {t[1]:6.3f} ms ({t[0] / t[1]:4.2f}x){'' if same else ' DIFFERENT BITS'}" \
                    f"{'' if not bad else f' {bad} UNSTABLE'}"
        print(line, flush=True)
    print("ring + halo layers" + "".join(f" | {p}: {tot[(p, 0)]:6.3f} -> {tot[(p, 1)]:6.3f} ms" for p in precs), flush=True)

    Q, G, D, K = 8192, 81920, 4096, 10
    q = torch.nn.functional.normalize(torch.randn((Q, D), generator=g, device=dev), dim=1)
    gal = torch.nn.functional.normalize(torch.randn((G, D), generator=g, device=dev), dim=1)
    for p in precs:
        qp, gp = ops.PreparedRows(q, p), ops.PreparedRows(gal, p)
        res, t = {}, {}
        for bar1 in (0, 1):
            h.oibl_debug_set_match_bar1(bar1)
            res[bar1] = ops.sqdist_topk_prepared(qp, gp, K)
            torch.cuda.synchronize()
            t[bar1] = timed(lambda: ops.sqdist_topk_prepared(qp, gp, K, defer_check=True), 3)
        same = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        bad = 0
        for i in range(a.reps // 4):
            if i % 2:
                with torch.cuda.stream(s2):
                    junk.add_(1.0)
            r = ops.sqdist_topk_prepared(qp, gp, K)
            bad += 0 if (torch.equal(r[0], res[1][0]) and torch.equal(r[1], res[1][1])) else 1
        torch.cuda.synchronize()
        bad_total += bad + (0 if same else 1)
        print(f"matching {Q} x {G} x {D} + top-{K} {p}: {t[0]:6.3f} -> {t[1]:6.3f} ms ({Q * G / t[1] / 1e6:7.1f} Gpairs/s)"
              f"{'' if same else ' DIFFERENT RESULTS'}{'' if not bad else f' {bad} UNSTABLE'}", flush=True)
        qs, gs = q[:2048].contiguous(), gal[:20000].contiguous()
        mats = {}
        for bar1 in (0, 1):
            h.oibl_debug_set_match_bar1(bar1)
            mats[bar1] = ops.pairwise_sqdist(qs, gs, p)
        if not torch.equal(mats[0], mats[1]):
            bad_total += 1
            print(f"pairwise matrix {p}: DIFFERENT BITS", flush=True)
    print("BAR1 problems:", bad_total, flush=True)


if __name__ == "__main__":
    main()
