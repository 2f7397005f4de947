"""Shader-clock stamps of phases P0 / P1 of the last steady-state K-tile of the f16mx ring kernel, block 0,
wave 0 (group 0) and wave 4 (group 1)  (diagnostic; RING_MX_PROF: the two-barrier schedule with the LDS-DMA issue
inside COMPUTE).     python tests/gpu_mx_stamps.py [conv2_1]     (conv2_1: the 512 x 128 tile, 64 -> 128 at 240x320)"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops, lib  # noqa: E402
dev = torch.device("cuda", 0)
L = lib.debug_hooks()
buf = torch.zeros(8 + 28, dtype=torch.int64, device=dev)
N, H, W, cin, cout = (32, 240, 320, 64, 128) if "conv2_1" in sys.argv[1:] else (32, 60, 80, 512, 512)
KT = 9 * cin // 32
g = torch.Generator(device=dev).manual_seed(5)
x = ops.mx_split(torch.relu(torch.randn((N, H, W, cin), generator=g, device=dev)) * 3.0)
w = ops.pack_conv3x3(torch.randn((cout, cin, 3, 3), generator=g, device=dev) * 0.02, "f16mx")
b = torch.zeros(cout, device=dev)
L.oibl_debug_set_mx_variant(8)
for _ in range(3):
    ops.conv3x3_nhwc(x, w, b, True, False, "f16mx")
torch.cuda.synchronize()
L.oibl_debug_set_prof_buffer(buf.data_ptr())
ops.conv3x3_nhwc(x, w, b, True, False, "f16mx")
torch.cuda.synchronize()
L.oibl_debug_set_prof_buffer(None)
L.oibl_debug_set_mx_variant(0)
t = buf.cpu().tolist()
print("kernel sections (wave 0): prologue, loop, regs->LDS, rest of the epilogue:", t[:4], " main loop per K-tile:", t[1] / KT)
print("epilogue, pass 0: pack", t[4], "copy-out (stores issued)", t[5], "| pass 1: pack", t[6], "copy-out", t[7])
names = ["phase start", "reads issued", "LDS-DMA issued", "vmcnt wait passed", "barrier passed", "MFMAs issued", "closing barrier passed"]
for gidx in (0, 1):
    st = t[8 + 14 * gidx: 8 + 14 * gidx + 14]
    print(f"group {gidx} (wave {4 * gidx}):")
    for ph in (0, 1):
        s = st[7 * ph: 7 * ph + 7]
        print(f"  P{ph}: " + "  ".join(f"{n} +{s[i] - s[i - 1]}" for i, n in enumerate(names) if i > 0) + f"   | total {s[6] - s[0]}")
    print(f"  P0 start -> P1 start: {st[7] - st[0]}")
g0, g1 = t[8:22], t[22:36]
print("group 1 P0 start minus group 0 P0 start:", g1[0] - g0[0])
