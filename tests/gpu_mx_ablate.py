"""Timing experiment (diagnostic): conv4_2 (512 -> 512 at 60 x 80, batch 32) in f16mx with one ingredient of
the ring loop removed (results wrong): which one sets the K-tile period?"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops, lib  # noqa: E402
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
N, H, W, cin, cout = 32, 60, 80, 512, 512
xf = torch.relu(torch.randn((N, H, W, cin), generator=g, device=dev)) * 3.0
w = torch.randn((cout, cin, 3, 3), generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5
b = torch.zeros((cout,), device=dev)
x = ops.mx_split(xf)
wp = ops.pack_conv3x3(w, "f16mx")
L = lib.debug_hooks()
names = {0: "late (default)", 1: "early", 4: "early, no MFMA", 5: "early, no LDS-DMA", 6: "early, no fragment reads", 7: "early, no barriers"}
for abl in (0, 3):
    L.oibl_debug_set_ring_ablate(abl)
    for v in (1, 0, 4, 5, 6, 7):
        L.oibl_debug_set_mx_variant(v)
        for _ in range(3):
            ops.conv3x3_nhwc(x, wp, b, True, False, "f16mx")
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(8):
                ops.conv3x3_nhwc(x, wp, b, True, False, "f16mx")
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 8)
        print(f"loads {'-> one line' if abl else 'real'} | {names[v]:28s}: {sorted(ts)[2]:.3f} ms", flush=True)
L.oibl_debug_set_mx_variant(0)
L.oibl_debug_set_ring_ablate(0)
