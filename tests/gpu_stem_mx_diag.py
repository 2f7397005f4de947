import sys, torch
import torch.nn.functional as F
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from openibl_amd import ops
dev = torch.device('cuda', 0)
def case(N, H, W, cin, cout, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((N, cin, H, W), generator=g)
    w = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.1
    return x, w, b
for (N, H, W) in [(1, 8, 32), (1, 64, 96), (5, 40, 136)]:
    x, w1, b1 = case(N, H, W, 3, 64, seed=5 * H + W)
    x = x * 60.0
    _, w2, b2 = case(1, 4, 4, 64, 64, seed=H + 9 * W)
    wp2 = ops.pack_conv3x3(w2.to(dev), "f16mx")
    y = ops.vgg16_stem_mx(x.to(dev), w1.to(dev), b1.to(dev), wp2, b2.to(dev))
    h1 = F.relu(F.conv2d(x.double(), w1.double(), b1.double(), padding=1))
    want = F.max_pool2d(F.relu(F.conv2d(h1, w2.double(), b2.double(), padding=1)), 2, 2)
    got = ops.nhwc_to_nchw_f32(ops.mx_join(y, 0)).cpu().double()
    err = (got - want).abs()
    print(N, H, W, "rel", float((got - want).norm() / want.norm()), "max", float(err.max()))
    # error per channel, per y, per x
    e = err / want.abs().amax()
    print(" per channel (x1e5):", [round(float(v) * 1e5, 1) for v in e.amax((0, 2, 3))])
    print(" per y:", [round(float(v) * 1e5, 1) for v in e.amax((0, 1, 3))])
    print(" per x:", [round(float(v) * 1e5, 1) for v in e.amax((0, 1, 2))])
    hi = ops.mx_join(y, 1)
    again = ops.mx_split(hi)
    print(" hi eq", bool(torch.equal(ops.mx_join(again, 1), hi)), "hi6 mismatches", int((ops.mx_join(again, 2) != ops.mx_join(y, 2)).sum()), "of", hi.numel())
    d = (ops.mx_join(again, 2) != ops.mx_join(y, 2))
    if d.any():
        idx = d.nonzero()[:10].tolist(); print(" first mismatches (n,y,x,c):", idx)
    gmax = hi.abs().reshape(-1, 32).amax(-1, keepdim=True)
    lo6 = ops.mx_join(y, 3).reshape(-1, 32)
    print(" lo6 bound violations", int((lo6.abs() > gmax * 2.0 ** -11 * 1.07 + 1e-30).sum()))
