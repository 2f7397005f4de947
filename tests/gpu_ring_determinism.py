"""Diagnostic soak: do the ring convolution / distance kernels give the same bits launch after launch while
other streams load the memory system?  (The halo kernel's first wait counts did not: conv_halo.h.  The ring
kernels also issue LDS-DMA instructions whose lanes are all out of range — taps outside the image — and count
them in their waits, 5 phases deep.)      python tests/gpu_ring_determinism.py [reps]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops  # noqa: E402
dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator(device=dev).manual_seed(5)
big = torch.randn((8192, 8192), device=dev)
junk = torch.empty((1 << 28,), dtype=torch.float32, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
total_bad = 0
for prec in ("f16mx", "bf16x3", "bf16"):
    split = {"f16mx": ops.mx_split, "bf16x3": ops.x3_split, "bf16": lambda t: t.to(torch.bfloat16)}[prec]
    for (N, H, W, cin, cout, pool) in [(32, 120, 160, 256, 256, 1), (32, 60, 80, 512, 512, 0), (32, 30, 40, 512, 512, 0),
                                       (16, 240, 320, 64, 128, 0), (16, 240, 320, 128, 128, 1)]:
        xf = torch.relu(torch.randn((N, H, W, cin), generator=g, device=dev)) * 3.0
        w = torch.randn((cout, cin, 3, 3), generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5
        b = torch.randn((cout,), generator=g, device=dev) * 0.1
        x, wp = split(xf), ops.pack_conv3x3(w, prec)
        ref = ops.conv3x3_nhwc(x, wp, b, True, bool(pool), prec)
        bad = 0
        for i in range(reps):
            if i % 2:
                with torch.cuda.stream(s1):
                    big @ big
                with torch.cuda.stream(s2):
                    junk.add_(1.0)          # streams 1 GB through HBM
            y = ops.conv3x3_nhwc(x, wp, b, True, bool(pool), prec)
            if not torch.equal(y, ref):
                bad += 1
        torch.cuda.synchronize()
        total_bad += bad
        print(f"conv {prec:7s} {cin}->{cout} {H}x{W} pool={pool}: {bad}/{reps} launches differ", flush=True)
    q = torch.nn.functional.normalize(torch.randn((2048, 4096), generator=g, device=dev), dim=1)
    gal = torch.nn.functional.normalize(torch.randn((20000, 4096), generator=g, device=dev), dim=1)
    ref = ops.pairwise_sqdist(q, gal, prec)
    bad = 0
    for i in range(reps // 4):
        if i % 2:
            with torch.cuda.stream(s2):
                junk.add_(1.0)
        if not torch.equal(ops.pairwise_sqdist(q, gal, prec), ref):
            bad += 1
    torch.cuda.synchronize()
    total_bad += bad
    print(f"pairwise {prec:7s} 2048 x 20000 x 4096: {bad}/{reps // 4} launches differ", flush=True)
print("TOTAL differing launches:", total_bad)
