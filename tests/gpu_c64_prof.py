"""Phase breakdown (shader clocks) of the Cin=64 resident kernel, block 0 / wave 0 (diagnostic)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops, lib

dev = torch.device("cuda", 0)
L = lib.debug_hooks()
buf = torch.zeros(8, dtype=torch.int64, device=dev)
names = ["issue_halo", "mfma loop", "wait vm/lgkm", "barrier A", "epilogue", "barrier B"]
for (N, H, W, cout, pool) in [(32, 480, 640, 64, True), (32, 240, 320, 128, False)]:
    x = torch.randn((N, H, W, 64), device=dev).to(torch.bfloat16)
    w = ops.pack_conv3x3(torch.randn((cout, 64, 3, 3), device=dev) * 0.05, "bf16")
    b = torch.zeros(cout, device=dev)
    ops.conv3x3_nhwc(x, w, b, True, pool, "bf16")
    torch.cuda.synchronize()
    L.oibl_debug_set_prof_buffer(buf.data_ptr())
    buf.zero_()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.conv3x3_nhwc(x, w, b, True, pool, "bf16")
    e.record()
    torch.cuda.synchronize()
    L.oibl_debug_set_prof_buffer(None)
    t = buf.cpu().tolist()[:6]
    tiles = N * ((H + 7) // 8) * ((W + 31) // 32)
    per_block = tiles / (256 // (cout // 64))
    tot = sum(t)
    print(f"N={N} {H}x{W} cout={cout} pool={pool}: {s.elapsed_time(e):.3f} ms, tiles/block={per_block:.1f}, "
          f"clock ticks total {tot} ({tot / per_block:.0f} per tile)")
    for n, v in zip(names, t):
        print(f"   {n:14s} {v / per_block:9.0f} ticks/tile  {100.0 * v / tot:5.1f} %")
