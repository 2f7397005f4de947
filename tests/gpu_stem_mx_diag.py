import sys, torch
import torch.nn.functional as F
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops
dev = torch.device('cuda', 0)
def case(N, H, W, cin, cout, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((N, cin, H, W), generator=g)
    w = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.1
    return x, w, b
for (N, H, W) in [(1, 8, 32), (1, 64, 96), (1, 70, 90), (2, 17, 65), (4, 128, 640)]:
    x, w1, b1 = case(N, H, W, 3, 64, seed=5 * H + W)
    x = x * 60.0
    _, w2, b2 = case(1, 4, 4, 64, 64, seed=H + 9 * W)
    wp2 = ops.pack_conv3x3(w2.to(dev), "f16mx")
    y = ops.vgg16_stem_mx(x.to(dev), w1.to(dev), b1.to(dev), wp2, b2.to(dev))
    h1 = F.relu(F.conv2d(x.double(), w1.double(), b1.double(), padding=1))
    want = F.max_pool2d(F.relu(F.conv2d(h1, w2.double(), b2.double(), padding=1)), 2, 2)
    got = ops.nhwc_to_nchw_f32(ops.mx_join(y, 0)).cpu().double()
    err = (got - want).abs() / want.abs().amax()
    print(N, H, W, "rel", float((got - want).norm() / want.norm()))
    ey = err.amax((0, 1, 3)); ex = err.amax((0, 1, 2)); en = err.amax((1, 2, 3))
    print("  bad rows y:", [i for i, v in enumerate(ey) if v > 1e-3][:30], " bad cols x:", [i for i, v in enumerate(ex) if v > 1e-3][:40], " per image:", [round(float(v), 4) for v in en])
