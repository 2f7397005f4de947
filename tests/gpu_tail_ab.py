"""f16mx fragment tail (8 e2m3 bytes + the scale byte of an operand row): ds_read_b64 + ds_read_b32 — the product
library; 2-way and 4-way bank conflicts, 12 LDS cycles — against ONE ds_read_b128 — the debug library of this
build (openibl_amd/build.py, DBG_EXPERIMENT_FLAGS); conflict-free, 4 cycles, but the 6-register operand of the
scaled MFMA is then re-assembled by copies (diagnostic, not a pytest).     python tests/gpu_tail_ab.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import lib, ops  # noqa: E402

LAYERS = [(64, 128, 240, 320, 1, 0), (128, 128, 240, 320, 1, 1), (128, 256, 120, 160, 1, 0), (256, 256, 120, 160, 1, 0),
          (256, 256, 120, 160, 1, 1), (256, 512, 60, 80, 1, 0), (512, 512, 60, 80, 1, 0), (512, 512, 60, 80, 1, 1),
          (512, 512, 30, 40, 1, 0)]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)


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


tot = [0.0, 0.0]
for cin, cout, H, W, relu, pool in LAYERS:
    xf = torch.relu(torch.randn((32, H, W, cin), generator=g, device=dev)) * 3.0
    w = torch.randn((cout, cin, 3, 3), generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn((cout,), generator=g, device=dev) * 0.1
    x, wp = ops.mx_split(xf), ops.pack_conv3x3(w, "f16mx")
    run = lambda: ops.conv3x3_nhwc(x, wp, b, bool(relu), bool(pool), "f16mx")   # noqa: E731
    t, out = [], []
    for which in (0, 1, 0, 1):
        if which:
            lib.debug_hooks()
        else:
            lib.use_product_library()
        out.append(run())
        t.append(timed(run))
    lib.use_product_library()
    t0, t1 = min(t[0], t[2]), min(t[1], t[3])
    tot[0] += t0
    tot[1] += t1
    print(f"{cin:4d}->{cout:4d} {H:3d}x{W:3d}{' pool' if pool else '     '}: b64 + b32 tail {t0:6.3f} ms | b128 tail {t1:6.3f} ms "
          f"({t0 / t1:4.2f}x) | same bits: {torch.equal(out[0], out[1])}", flush=True)
print(f"layers behind the stem (conv5 once): {tot[0]:.3f} -> {tot[1]:.3f} ms")
