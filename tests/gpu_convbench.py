"""Per-layer A/B of the implicit-GEMM convolution kernels on the GPU (diagnostic, not a pytest).

For every VGG16 layer shape the ring-schedule kernel is eligible for, at the benchmark batch:
bit-exact comparison against the generic kernel (same K order -> identical bf16 tensors) and
interleaved timing of the tile modes given on the command line.

    python tests/gpu_convbench.py [--batch 32] [--modes 3,4] [--rounds 5]
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops  # noqa: E402

LAYERS = [  # (cin, cout, H, W, relu, pool)
    (64, 128, 240, 320, 1, 0), (128, 128, 240, 320, 1, 1), (128, 256, 120, 160, 1, 0), (256, 256, 120, 160, 1, 0),
    (256, 256, 120, 160, 1, 1), (256, 512, 60, 80, 1, 0), (512, 512, 60, 80, 1, 0),
    (512, 512, 60, 80, 1, 1), (512, 512, 30, 40, 1, 0), (512, 512, 30, 40, 0, 0)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--modes", default="3,4")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--ablate", default="", help="timing experiment: comma list of ring ablate modes "
                    "(1 = pixel loads, 2 = weight loads, 3 = both served from one line; results wrong)")
    ap.add_argument("--korder", type=int, default=-1, help="K order hook (ops.set_conv_korder; -1 = per-layer default)")
    a = ap.parse_args()
    ops.set_conv_korder(a.korder)
    modes = [int(m) for m in a.modes.split(",")]
    from openibl_amd import lib as _l
    abl = [int(v) for v in a.ablate.split(",")] if a.ablate else []
    if abl:                       # pseudo-modes 100 + ablate code, all on the auto (ring) kernel
        modes = [0] + [100 + v for v in abl]
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(5)
    tot = {m: 0.0 for m in modes}
    totfl = 0.0
    for cin, cout, H, W, relu, pool in LAYERS:
        N = a.batch
        x = (torch.randn((N, H, W, cin), generator=g, device=dev)).to(torch.bfloat16)
        w = torch.randn((cout, cin, 3, 3), generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5
        b = torch.randn((cout,), generator=g, device=dev) * 0.1
        wp = ops.pack_conv3x3(w, "bf16")
        outs, times = {}, {m: [] for m in modes}
        def select(m):
            ops.set_conv_tile(0 if m >= 100 else m)
            _l.debug_hooks().oibl_debug_set_ring_ablate(m - 100 if m >= 100 else 0)
        for m in modes:
            select(m)
            outs[m] = ops.conv3x3_nhwc(x, wp, b, bool(relu), bool(pool), "bf16")
        torch.cuda.synchronize()
        for _ in range(a.rounds):
            for m in modes:
                select(m)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(a.iters):
                    ops.conv3x3_nhwc(x, wp, b, bool(relu), bool(pool), "bf16")
                e.record()
                torch.cuda.synchronize()
                times[m].append(s.elapsed_time(e) / a.iters)
        select(0)
        fl = 2.0 * N * H * W * cout * 9 * cin
        totfl += fl
        ref = outs[modes[0]]
        line = f"{cin:4d}->{cout:4d} {H:3d}x{W:3d}{' pool' if pool else '     '}"
        for m in modes:
            t = sorted(times[m])[len(times[m]) // 2]
            tot[m] += t
            eq = torch.equal(outs[m], ref)
            nbad = 0 if eq else int((outs[m].float() != ref.float()).sum())
            if not eq:   # different summation order (Cin = 64 resident kernel): report the distance
                d = (outs[m].float() - ref.float()).norm() / ref.float().norm()
                nbad = f"{nbad}, rel {float(d):.1e}"
            line += f" | mode{m}: {t:7.3f} ms {fl / t / 1e9:7.1f} TF {'==' if eq else f'DIFF({nbad})'}"
        print(line, flush=True)
    print("total" + "".join(f" | mode{m}: {tot[m]:7.3f} ms {totfl / tot[m] / 1e9:7.1f} TF" for m in modes))


if __name__ == "__main__":
    main()
