"""Per-layer timing of the ring convolutions in the three matrix-core arithmetics (diagnostic, not a
pytest):  bf16 | bf16x3 | f16mx  on the VGG16 layer shapes behind the stem at the benchmark batch, plus
the whole backbone per precision.      python tests/gpu_precbench.py [--batch 32] [--rounds 5]"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops, synth  # noqa: E402

LAYERS = [  # (cin, cout, H, W, relu, pool)
    (64, 128, 240, 320, 1, 0), (128, 128, 240, 320, 1, 1), (128, 256, 120, 160, 1, 0), (256, 256, 120, 160, 1, 0),
    (256, 256, 120, 160, 1, 1), (256, 512, 60, 80, 1, 0), (512, 512, 60, 80, 1, 0),
    (512, 512, 60, 80, 1, 1), (512, 512, 30, 40, 1, 0), (512, 512, 30, 40, 0, 0)]
PRECS = ("bf16", "bf16x3", "f16mx", "f16mx-ring", "f16mx-halo", "f16mx-late", "f16mx-s13")


def sel(p):
    from openibl_amd import lib
    # 0: default dispatch, 1: ring kernels only, 3: halo kernel wherever it applies, 2: ring kernels with the
    # LDS-DMA issue inside the COMPUTE segments
    # 13: EVERY layer on the 4-wave halo kernel of conv2_x (conv_halo4.h; experiment)
    lib.debug_hooks().oibl_debug_set_mx_variant(3 if p.endswith("halo") else 1 if p.endswith("ring") else
                                                2 if p.endswith("late") else int(p[-2:]) if p[-3] == "s" else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=4)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(5)
    tot = {p: 0.0 for p in PRECS}
    totfl = 0.0
    N = a.batch
    for cin, cout, H, W, relu, pool in LAYERS:
        xf = torch.relu(torch.randn((N, H, W, cin), generator=g, device=dev)) * 3.0
        w = torch.randn((cout, cin, 3, 3), generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5
        b = torch.randn((cout,), generator=g, device=dev) * 0.1
        xs = {"bf16": xf.to(torch.bfloat16), "bf16x3": ops.x3_split(xf), "f16mx": ops.mx_split(xf)}
        xs["f16mx-halo"] = xs["f16mx-ring"] = xs["f16mx-late"] = xs["f16mx-s13"] = xs["f16mx"]
        wp = {p: ops.pack_conv3x3(w, p.split("-")[0]) for p in PRECS}
        times = {p: [] for p in PRECS}
        outs = {}
        for p in PRECS:
            sel(p)
            outs[p] = ops.conv3x3_nhwc(xs[p], wp[p], b, bool(relu), bool(pool), p.split("-")[0])
        torch.cuda.synchronize()
        d = (ops.mx_join(outs["f16mx-ring"]) - ops.mx_join(outs["f16mx-halo"])).norm() / ops.mx_join(outs["f16mx-halo"]).norm()
        assert float(d) < 3e-5, f"halo and ring kernels disagree: {float(d):.2e}"
        for _ in range(a.rounds):
            for p in PRECS:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                sel(p)
                for _ in range(a.iters):
                    ops.conv3x3_nhwc(xs[p], wp[p], b, bool(relu), bool(pool), p.split("-")[0])
                e.record()
                torch.cuda.synchronize()
                times[p].append(s.elapsed_time(e) / a.iters)
        fl = 2.0 * N * H * W * cout * 9 * cin
        totfl += fl
        line = f"{cin:4d}->{cout:4d} {H:3d}x{W:3d}{' pool' if pool else '     '}"
        for p in PRECS:
            t = sorted(times[p])[len(times[p]) // 2]
            tot[p] += t
            line += f" | {p}: {t:7.3f} ms {fl / t / 1e9:7.1f} TF"
        print(line, flush=True)
    print("ring layers" + "".join(f" | {p}: {tot[p]:7.3f} ms {totfl / tot[p] / 1e9:7.1f} TF" for p in PRECS), flush=True)

    # whole backbone (stem included)
    import hubconf
    sd = synth.embednetpca_state(0)
    m = hubconf.vgg16_netvlad(pretrained=False)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    x = synth.images(N, 480, 640, seed=1).to(dev)
    for p in PRECS:
        sel(p)
        m.set_precision(p.split("-")[0])
        for _ in range(2):
            m(x)
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.rounds):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                m(x)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 3)
        t = sorted(ts)[len(ts) // 2]
        print(f"whole forward (eager, one stream) {p}: {t:.3f} ms per batch of {N} = {N / t * 1e3:.0f} images/s", flush=True)


if __name__ == "__main__":
    main()
