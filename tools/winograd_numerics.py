#!/usr/bin/env python
"""Numerics study for DESIGN.md §9 item 1 (CPU only, no product code): how far from an fp64
convolution is a 3x3 layer evaluated
  (a) directly in bf16x3  (operands split hi = bf16(v), lo = bf16(v - hi); hi.hi + hi.lo + lo.hi, fp32 sums)
  (b) as Winograd F(2x2, 3x3) with the SAME split applied to the transformed tiles / filters
  (c) as Winograd F(4x4, 3x3) likewise
on VGG-like data (He-initialised filters, post-ReLU inputs)?   python tools/winograd_numerics.py"""
import numpy as np
import torch
import torch.nn.functional as F

torch.manual_seed(0)


def split(v):
    hi = v.to(torch.bfloat16).float()
    lo = (v - hi).to(torch.bfloat16).float()
    return hi, lo


def x3_matmul(a, b):
    """a [.., M, K] @ b [.., K, N] with the three-term split product, fp32 accumulation."""
    ah, al = split(a)
    bh, bl = split(b)
    return al @ bh + ah @ bl + ah @ bh


def direct_x3(x, w):
    n, c, h, wd = x.shape
    cols = F.unfold(x, 3, padding=1)                       # [n, c*9, h*w]
    out = x3_matmul(w.reshape(w.shape[0], -1), cols)       # [n, cout, h*w]
    return out.reshape(n, w.shape[0], h, wd)


def winograd(x, w, m):
    """F(m x m, 3 x 3), transforms in fp32, the per-position GEMMs in bf16x3."""
    if m == 2:
        BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1.]])
        G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1.]])
        AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1.]])
    else:
        BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                           [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1.]])
        G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                          [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1.]])
        AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1.]])
    t = m + 2
    n, c, h, wd = x.shape
    assert h % m == 0 and wd % m == 0
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, t, m).unfold(3, t, m)             # [n, c, th, tw, t, t]
    V = BT @ tiles @ BT.T                                  # input transform (fp32)
    U = (G.double() @ w.double() @ G.double().T).float()   # filter transform, rounded once to fp32
    th, tw = V.shape[2], V.shape[3]
    Vp = V.permute(4, 5, 0, 2, 3, 1).reshape(t, t, n * th * tw, c)         # [t, t, tiles, cin]
    Up = U.permute(2, 3, 1, 0)                                             # [t, t, cin, cout]
    M = x3_matmul(Vp, Up)                                                  # [t, t, tiles, cout]
    M = M.reshape(t, t, n, th, tw, -1).permute(2, 5, 3, 4, 0, 1)           # [n, cout, th, tw, t, t]
    Y = AT @ M @ AT.T                                                      # [n, cout, th, tw, m, m]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(n, -1, th * m, tw * m)


def rel(a, ref):
    return float(((a.double() - ref) ** 2).sum().sqrt() / (ref ** 2).sum().sqrt())


for cin, cout, h, wd in ((64, 128, 48, 48), (256, 256, 24, 24), (512, 512, 12, 12)):
    x = F.relu(torch.randn(2, cin, h, wd)) * 3.0
    w = torch.randn(cout, cin, 3, 3) * (2.0 / (9 * cin)) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    fp32 = F.conv2d(x, w, padding=1)
    print(f"{cin}->{cout} {h}x{wd}: fp32 direct {rel(fp32, ref):.2e} | bf16x3 direct {rel(direct_x3(x, w), ref):.2e} | "
          f"bf16x3 F(2x2,3x3) {rel(winograd(x, w, 2), ref):.2e} | bf16x3 F(4x4,3x3) {rel(winograd(x, w, 4), ref):.2e} | "
          f"bf16 direct {rel(F.conv2d(x.bfloat16().float(), w.bfloat16().float(), padding=1), ref):.2e}", flush=True)

# error growth through a stack of layers (ReLU between them), everything relative to an fp64 stack
x = F.relu(torch.randn(1, 128, 24, 24)) * 3.0
ws = [torch.randn(128, 128, 3, 3) * (2.0 / (9 * 128)) ** 0.5 for _ in range(8)]
ref, a, b, c = x.double(), x, x, x
for i, w in enumerate(ws):
    ref = F.relu(F.conv2d(ref, w.double(), padding=1))
    a = F.relu(direct_x3(a, w))
    b = F.relu(winograd(b, w, 2))
    c = F.relu(winograd(c, w, 4))
    print(f"after layer {i + 1}: bf16x3 direct {rel(a, ref):.2e} | F(2x2,3x3) {rel(b, ref):.2e} | F(4x4,3x3) {rel(c, ref):.2e}")
