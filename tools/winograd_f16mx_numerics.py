#!/usr/bin/env python
"""Go / no-go numerics for VERDICT r04 item 9 (CPU only, no product code): Winograd F(2x2, 3x3) with the f16mx
product — hi = fp16(v) main term + both cross terms on block-scaled e2m3 — applied to the TRANSFORMED tiles
(V = B^T d B) and filters (U = G g G^T).  2.25x fewer products per output than the direct convolution; the
question is what the transforms do to the error of an arithmetic that sits 3x inside the 1e-4 bar when used
directly (descriptor 1.7e-5, feature map 3.9e-5 after 13 layers: tools/f16mx_numerics.py).

   python tools/winograd_f16mx_numerics.py"""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent))
from f16mx_numerics import mxq, rel  # noqa: E402  (runs that study's table on import: ~1 min)

torch.manual_seed(0)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1.]])
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1.]])
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1.]])


def mx_matmul(a, b):
    """a [.., M, K] @ b [.., K, N] in f16mx: fp16 hi.hi + q6(hi).q6(lo) + q6(lo).q6(hi), blocks of 32 along K."""
    ah, bh = a.half().float(), b.half().float()
    al, bl = a - ah, b - bh
    aq, bq = mxq(ah, "e2m3", -1), mxq(bh, "e2m3", -2)
    alq = mxq(al, "e2m3", -1, scale_from=ah, scale_shift=-12)
    blq = mxq(bl, "e2m3", -2, scale_from=bh, scale_shift=-12)
    return ah @ bh + (aq @ blq + alq @ bq)


def direct_mx(x, w):
    n, c, h, wd = x.shape
    cols = F.unfold(x, 3, padding=1).transpose(1, 2)                  # [n, h*w, c*9]  (c-major, tap-minor)
    cols = cols.reshape(n, h * wd, c, 9).transpose(2, 3).reshape(n, h * wd, 9 * c)   # K = (tap, channel): blocks of 32 channels
    wm = w.reshape(w.shape[0], c, 9).transpose(1, 2).reshape(w.shape[0], 9 * c).t()  # [9c, cout]
    return mx_matmul(cols, wm).transpose(1, 2).reshape(n, w.shape[0], h, wd)


def winograd_mx(x, w, matmul):
    n, c, h, wd = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)             # [n, c, th, tw, 4, 4]
    V = BT @ tiles @ BT.T                                  # input transform in fp32 (adds / subtracts only)
    U = (G.double() @ w.double() @ G.double().T).float()   # filter transform, once per checkpoint
    th, tw = V.shape[2], V.shape[3]
    Vp = V.permute(4, 5, 0, 2, 3, 1).reshape(4, 4, n * th * tw, c)
    Up = U.permute(2, 3, 1, 0)                             # [4, 4, cin, cout]
    M = matmul(Vp, Up)
    M = M.reshape(4, 4, n, th, tw, -1).permute(2, 5, 3, 4, 0, 1)
    Y = AT @ M @ AT.T
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(n, -1, th * 2, tw * 2)


print("\n==== one layer, VGG-like data (post-ReLU inputs x3, He filters), rel-L2 against fp64")
for cin, cout, h, wd in ((128, 128, 48, 48), (256, 256, 24, 24), (512, 512, 12, 16)):
    x = F.relu(torch.randn(2, cin, h, wd)) * 3.0
    w = torch.randn(cout, cin, 3, 3) * (2.0 / (9 * cin)) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    print(f"{cin}->{cout} {h}x{wd}: f16mx direct {rel(direct_mx(x, w), ref):.2e} | f16mx F(2x2,3x3) "
          f"{rel(winograd_mx(x, w, mx_matmul), ref):.2e} | fp32 F(2x2,3x3) {rel(winograd_mx(x, w, torch.matmul), ref):.2e}",
          flush=True)

print("\n==== conv3_1..conv5_3 as a stack (10 layers, pools as in VGG16), fp64 reference, inputs from an exact front")
cfg = [(128, 256), (256, 256), (256, 256), "M", (256, 512), (512, 512), (512, 512), "M", (512, 512), (512, 512), (512, 512)]
x = F.relu(torch.randn(1, 128, 48, 64)) * 3.0
ref, a, b = x.double(), x, x
li = 0
for v in cfg:
    if v == "M":
        ref, a, b = F.max_pool2d(ref, 2), F.max_pool2d(a, 2), F.max_pool2d(b, 2)
        continue
    w = torch.randn(v[1], v[0], 3, 3) * (2.0 / (9 * v[1])) ** 0.5
    li += 1
    last = li == 9
    ref = F.conv2d(ref, w.double(), padding=1)
    a, b = direct_mx(a, w), winograd_mx(b, w, mx_matmul)
    if not last:
        ref, a, b = F.relu(ref), F.relu(a), F.relu(b)
    print(f"layer {li} ({v[0]}->{v[1]}): f16mx direct {rel(a, ref):.2e} | f16mx F(2x2,3x3) {rel(b, ref):.2e}", flush=True)
