"""Role breakdown (shader clocks, block (0, 0)) of the fused bf16x3 stem per issue-priority setting.
argv: hook values = producer priority | consumer priority << 2   (default: a sweep)"""
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
w2 = ops.pack_conv3x3(torch.randn((64, 64, 3, 3), device=dev) * 0.05, "bf16x3")
b2 = torch.zeros(64, device=dev)
tiles = N * ((H + 7) // 8) * ((W + 31) // 32) / 128
for _ in range(10):
    ops.vgg16_stem_x3(x, w1, b1, w2, b2)
for mode in [int(a) for a in sys.argv[1:]] or [0, 3, 5, 10, 12, 15]:
    L.oibl_debug_set_stem3_prio(mode)
    for _ in range(3):
        ops.vgg16_stem_x3(x, w1, b1, w2, b2)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.vgg16_stem_x3(x, w1, b1, w2, b2)
    e.record()
    torch.cuda.synchronize()
    L.oibl_debug_set_prof_buffer(buf.data_ptr())
    ops.vgg16_stem_x3(x, w1, b1, w2, b2)
    torch.cuda.synchronize()
    L.oibl_debug_set_prof_buffer(None)
    t = buf.cpu().tolist()
    print(f"prod prio {mode & 3} cons prio {(mode >> 2) & 3}: {s.elapsed_time(e) / 10:.3f} ms | "
          f"consumer loops {t[0] / tiles:8.0f} wait {t[1] / tiles:7.0f} | "
          f"producer work {t[4] / tiles:8.0f} wait {t[5] / tiles:7.0f} ticks/tile")
L.oibl_debug_set_stem3_prio(0)
