"""Phase breakdown (shader clocks, block 0 / wave 0) of the ring convolution kernel (diagnostic).
    python tests/gpu_ring_prof.py [bf16|bf16x3]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops, lib

dev = torch.device("cuda", 0)
L = lib.debug_hooks()
buf = torch.zeros(8, dtype=torch.int64, device=dev)
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
kdiv = 32 if prec == "bf16x3" else 64
names = ["prologue (index math, loaders)", "main loop", "epilogue: regs -> LDS", "epilogue: LDS -> global"]
for (cin, cout, H, W, pool) in [(64, 128, 240, 320, 0), (128, 128, 240, 320, 1), (128, 256, 120, 160, 0),
                                (256, 256, 120, 160, 1), (512, 512, 60, 80, 0), (512, 512, 30, 40, 0)]:
    N = 32
    x = torch.randn((N, H, W, cin), device=dev)
    x = ops.x3_split(x) if prec == "bf16x3" else x.to(torch.bfloat16)
    w = ops.pack_conv3x3(torch.randn((cout, cin, 3, 3), device=dev) * 0.05, prec)
    b = torch.zeros(cout, device=dev)
    ops.conv3x3_nhwc(x, w, b, True, bool(pool), prec)
    torch.cuda.synchronize()
    L.oibl_debug_set_prof_buffer(buf.data_ptr())
    buf.zero_()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.conv3x3_nhwc(x, w, b, True, bool(pool), prec)
    e.record()
    torch.cuda.synchronize()
    L.oibl_debug_set_prof_buffer(None)
    t = buf.cpu().tolist()[:4]
    tot = sum(t)
    print(f"{prec} {cin}->{cout} {H}x{W} pool={pool}: {s.elapsed_time(e):.3f} ms; K-tiles {9 * cin // kdiv}; one tile = {tot} ticks")
    for n, v in zip(names, t):
        print(f"   {n:34s} {v:8d} ticks  {100.0 * v / tot:5.1f} %")
