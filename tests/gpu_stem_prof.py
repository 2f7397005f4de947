"""Role breakdown (shader clocks) of the fused VGG stem kernel, block 0 (diagnostic)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops, lib

dev = torch.device("cuda", 0)
L = lib.debug_hooks()
buf = torch.zeros(8, dtype=torch.int64, device=dev)
N, H, W = 32, 480, 640
x = torch.randn((N, 3, H, W), device=dev) * 50
w1 = torch.randn((64, 3, 3, 3), device=dev) * 0.2
b1 = torch.zeros(64, device=dev)
w2 = ops.pack_conv3x3(torch.randn((64, 64, 3, 3), device=dev) * 0.05, "bf16")
b2 = torch.zeros(64, device=dev)
for _ in range(2):
    ops.vgg16_stem(x, w1, b1, w2, b2)
torch.cuda.synchronize()
L.oibl_debug_set_prof_buffer(buf.data_ptr())
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
ops.vgg16_stem(x, w1, b1, w2, b2)
e.record()
torch.cuda.synchronize()
L.oibl_debug_set_prof_buffer(None)
t = buf.cpu().tolist()
tiles = N * ((H + 7) // 8) * ((W + 31) // 32) / 256
print(f"stem {N}x{H}x{W}: {s.elapsed_time(e):.3f} ms, {tiles:.0f} tiles per workgroup")
for n, v in zip(["consumer: mfma loop", "consumer: barrier wait", "consumer: epilogue", "-",
                 "producer: produce", "producer: barrier wait"], t):
    if n != "-":
        print(f"   {n:24s} {v / tiles:9.0f} ticks/tile")
