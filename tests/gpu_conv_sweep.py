"""Random-shape sweep of the f16mx convolution kernels against fp64 (diagnostic): the 4-wave halo kernel
(Cout = 128), the 8-wave halo kernel (Cout = 256) and the ring kernels (Cout = 512), pooled and unpooled, batch 1-5,
maps from 2 x 2 up to ~130 x 170 with odd sides (ragged patches, ragged pool rows / columns, single-tile launches).
    python tests/gpu_conv_sweep.py [cases=48] [first seed=0]"""
import sys
import time
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 48
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
LAYERS = [(64, 128), (128, 128), (128, 256), (256, 256), (256, 512), (512, 512)]
bad = 0
t0 = time.time()
for seed in range(first, first + cases):
    g = torch.Generator().manual_seed(20_000 + seed)
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))      # noqa: E731
    cin, cout = LAYERS[seed % len(LAYERS)]
    pool = bool(r(0, 1))
    big = r(0, 3) == 0
    N, H, W = r(1, 5), (r(60, 130) if big else r(2, 40)), (r(60, 170) if big else r(2, 50))
    x = torch.relu(torch.randn((N, cin, H, W), generator=g)) * 4.0
    w = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.3
    want = F.conv2d(x.double().to(dev), w.double().to(dev), b.double().to(dev), padding=1).relu()
    if pool:
        want = F.max_pool2d(want, 2, 2)
    flag = ops.new_range_flag(dev)
    xd = ops.mx_split(ops.nchw_f32_to_nhwc(x.to(dev), "fp32"))
    got = ops.conv3x3_nhwc(xd, ops.pack_conv3x3(w.to(dev), "f16mx"), b.to(dev), True, pool, "f16mx", range_flag=flag)
    y = ops.mx_join(got).permute(0, 3, 1, 2).double()
    ok = tuple(y.shape) == tuple(want.shape)
    err = float((y - want).norm() / want.norm().clamp_min(1e-30)) if ok and want.numel() else 0.0
    worst = float((y - want).abs().max() / want.abs().max().clamp_min(1e-30)) if ok and want.numel() else 0.0
    ok = ok and err < 4e-5 and worst < 2e-4 and int(flag.item()) == 0 and bool(torch.isfinite(y).all())
    bad += int(not ok)
    print(f"seed {seed:3d} {cin:3d}->{cout:3d} pool={int(pool)} N={N} {H:3d}x{W:3d}: rel-L2 {err:.2e}, worst element "
          f"{worst:.1e} of the peak {'ok' if ok else 'FAILED'}", flush=True)
print(f"{cases} cases, {bad} failed, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
