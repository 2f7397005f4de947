"""Diagnostic: is the f16mx halo kernel deterministic?  Repeated launches on the same inputs, alone and with a
second stream keeping the chip busy, compared bit for bit; and against the ring kernels (K order differs ->
close, not equal)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops, lib  # noqa: E402
dev = torch.device("cuda", 0)
L = lib.debug_hooks()
g = torch.Generator(device=dev).manual_seed(5)
import os
VARIANT = int(os.environ.get("VARIANT", "3"))   # 3 = halo kernel, 19 = its unsafe-wait variant, 0 = ring
REPS = int(os.environ.get("REPS", "60"))
print("variant", VARIANT)
for (N, H, W, cin, cout, pool) in [(32, 120, 160, 256, 256, 1), (32, 60, 80, 512, 512, 0), (32, 60, 80, 512, 512, 1)]:
    xf = torch.relu(torch.randn((N, H, W, cin), generator=g, device=dev)) * 3.0
    w = torch.randn((cout, cin, 3, 3), generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn((cout,), generator=g, device=dev) * 0.1
    x = ops.mx_split(xf)
    wp = ops.pack_conv3x3(w, "f16mx")
    L.oibl_debug_set_mx_variant(1)
    ring = ops.conv3x3_nhwc(x, wp, b, True, bool(pool), "f16mx")
    L.oibl_debug_set_mx_variant(VARIANT)
    ref = ops.conv3x3_nhwc(x, wp, b, True, bool(pool), "f16mx")
    torch.cuda.synchronize()
    bad = 0
    for i in range(REPS):
        y = ops.conv3x3_nhwc(x, wp, b, True, bool(pool), "f16mx")
        if not torch.equal(y, ref):
            bad += 1
            d = (y != ref)
            idx = d.nonzero()
            if bad <= 3:
                print(f"   run {i}: {int(d.sum())} differing words, first at {idx[0].tolist()} last at {idx[-1].tolist()}")
    side = torch.cuda.Stream()
    bad2 = 0
    big = torch.randn((8192, 8192), device=dev)
    for i in range(REPS // 2):
        with torch.cuda.stream(side):
            for _ in range(3):
                big @ big
        y = ops.conv3x3_nhwc(x, wp, b, True, bool(pool), "f16mx")
        torch.cuda.synchronize()
        if not torch.equal(y, ref):
            bad2 += 1
    rel = float((ops.mx_join(ref) - ops.mx_join(ring)).norm() / ops.mx_join(ring).norm())
    print(f"{cin}->{cout} {H}x{W} pool={pool}: {bad}/{REPS} repeats differ alone, {bad2}/{REPS // 2} with a busy second stream; halo vs ring rel {rel:.2e}", flush=True)
