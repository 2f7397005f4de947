"""Is the f16mx arithmetic invariant under a power-of-two scaling of the activations?  (diagnostic)
    python tests/gpu_scale_probe.py"""
import sys
from pathlib import Path
import torch
import torch.nn.functional as F
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
for cin, cout in [(256, 256), (512, 512), (128, 128)]:
    x = torch.randn((2, cin, 24, 40), generator=g).relu() * 5.0
    w = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.5
    want = F.conv2d(x.double(), w.double(), b.double(), padding=1).relu()
    xd = ops.nchw_f32_to_nhwc(x.to(dev), "fp32")
    pw = ops.pack_conv3x3(w.to(dev), "f16mx")
    outs = {}
    for sh in (0, 1, 5, 8):
        s = 2.0 ** -sh
        xs = ops.mx_split((xd * s).contiguous())
        back = ops.mx_join(xs) / s
        if sh == 0:
            back0, xs0 = back, xs
        y = ops.mx_join(ops.conv3x3_nhwc(xs, pw, (b * s).to(dev), True, False, "f16mx")) / s
        outs[sh] = y
        err = float((y.permute(0, 3, 1, 2).cpu().double() - want).norm() / want.norm())
        print(f"{cin}->{cout} shift {sh}: packed input equal to shift 0 (x 2^sh): {bool(torch.equal(back, back0))}; "
              f"hi bytes equal up to exponent: n/a; layer rel-L2 vs fp64 {err:.2e}; "
              f"output bit-equal to shift 0: {float((y == outs[0]).float().mean()):.4f}")
