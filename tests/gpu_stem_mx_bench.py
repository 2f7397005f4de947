"""Fused stems, batch 32 at 480x640: bf16x3 against f16mx — time per launch and the role breakdown of block (0, 0)
(shader clocks per tile: consumer passes / barrier waits / epilogue, producer work / waits).
    python tests/gpu_stem_mx_bench.py"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops, lib  # noqa: E402

dev = torch.device("cuda", 0)
L = lib.debug_hooks()
buf = torch.zeros(8, dtype=torch.int64, device=dev)
N, H, W = 32, 480, 640
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn((N, 3, H, W), generator=g, device=dev)
w1 = torch.randn((64, 3, 3, 3), generator=g, device=dev) * 0.27
b1 = torch.randn((64,), generator=g, device=dev) * 0.1
w2 = torch.randn((64, 64, 3, 3), generator=g, device=dev) * 0.06
b2 = torch.randn((64,), generator=g, device=dev) * 0.1
tiles = N * ((H + 7) // 8) * ((W + 31) // 32) / 128    # per workgroup, two workgroups per tile (bf16x3; f16mx with -DOIBL_STEM_SPLIT)
tiles_mx = N * ((H + 7) // 8) * ((W + 31) // 32) / 256  # f16mx, one workgroup per tile (round 6): 256 tile streams
modes = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [0]
for name, fn, prec, mode in [("bf16x3", ops.vgg16_stem_x3, "bf16x3", 0)] + [("f16mx", ops.vgg16_stem_mx, "f16mx", m) for m in modes]:
    L.oibl_debug_set_stem3_prio(mode)   # producer priority | consumer priority << 2
    name = f"{name} prio {mode & 3}/{(mode >> 2) & 3}"
    wp = ops.pack_conv3x3(w2, prec)
    for _ in range(3):
        fn(x, w1, b1, wp, b2)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn(x, w1, b1, wp, b2)
    b.record()
    torch.cuda.synchronize()
    buf.zero_()
    L.oibl_debug_set_prof_buffer(buf.data_ptr())
    fn(x, w1, b1, wp, b2)
    torch.cuda.synchronize()
    L.oibl_debug_set_prof_buffer(None)
    t = buf.cpu().tolist()
    tiles = tiles_mx if ("f16mx" in name and "--split" not in sys.argv) else N * ((H + 7) // 8) * ((W + 31) // 32) / 128
    print(f"{name} stem: {a.elapsed_time(b) / 10:.3f} ms | per tile: consumer passes {t[0] / tiles:7.0f} waits {t[1] / tiles:6.0f} "
          f"epilogue {t[2] / tiles:6.0f} | producer work {t[4] / tiles:7.0f} waits {t[5] / tiles:6.0f}"
          + (f" || beside pass 0: consumers wait {t[3] / tiles:6.0f}, producers work {t[6] / tiles:6.0f} wait {t[7] / tiles:6.0f}"
             if "f16mx" in name else ""))
L.oibl_debug_set_stem3_prio(0)
