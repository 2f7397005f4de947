"""Phase groups for the first round of a ring-kernel layer (csrc/conv_ring.h, RingParams::stagger; debug library):
all workgroups of a layer run for the same time, so the chip's CUs load and store in phase; delaying the first
round's workgroups by (blockIdx & 3) * n sleeps of 8128 cycles spreads the store bursts (diagnostic, not a pytest).
    python tests/gpu_stagger_ab.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import lib, ops  # noqa: E402

LAYERS = [(64, 128, 240, 320, 1, 0), (128, 128, 240, 320, 1, 1), (256, 512, 60, 80, 1, 0), (512, 512, 60, 80, 1, 0),
          (512, 512, 60, 80, 1, 1), (512, 512, 30, 40, 1, 0)]
STAGGERS = (0, 1, 2, 3, 5, 0)
if "early" in sys.argv[1:]:      # the two 512 x 128 layers only, alternating settings (boxes drift by several per cent)
    LAYERS, STAGGERS = LAYERS[:2], (0, 2, 0, 2, 0, 2, 0, 3, 0, 3, 0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
h = lib.debug_hooks()


def timed(fn, iters=4, rounds=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / iters)
    return sorted(ts)[len(ts) // 2]


for prec in (("f16mx",) if "early" in sys.argv[1:] else ("f16mx", "bf16")):
    for cin, cout, H, W, relu, pool in LAYERS:
        xf = torch.relu(torch.randn((32, H, W, cin), generator=g, device=dev)) * 3.0
        w = torch.randn((cout, cin, 3, 3), generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5
        b = torch.randn((cout,), generator=g, device=dev) * 0.1
        x = ops.mx_split(xf) if prec == "f16mx" else xf.to(torch.bfloat16)
        wp = ops.pack_conv3x3(w, prec)
        run = lambda: ops.conv3x3_nhwc(x, wp, b, bool(relu), bool(pool), prec)   # noqa: E731
        ref, cells = None, []
        for s in STAGGERS:
            h.oibl_debug_set_ring_stagger(s)
            out = run()
            ref = out if ref is None else ref
            cells.append(f"{s}: {timed(run):.3f}{'' if torch.equal(out, ref) else ' BITS DIFFER'}")
        h.oibl_debug_set_ring_stagger(0)
        print(f"{prec} {cin:4d}->{cout:4d} {H:3d}x{W:3d}{' pool' if pool else '     '} ms by stagger | " + " | ".join(cells), flush=True)
lib.use_product_library()
