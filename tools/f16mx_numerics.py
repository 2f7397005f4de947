#!/usr/bin/env python
"""Numerics study (CPU only, no product code): can the two cross terms of a split product ride the
MX pipe?  A 3x3 layer is evaluated as
    a.b ~= hi(a).hi(b)  [fp16 or bf16 MFMA]  +  q(hi(a)).q(lo(b)) + q(lo(a)).q(hi(b))  [MX fp6 / fp8, K-concatenated]
with hi = fp16(v) (or bf16), lo = v - hi, q = block-scaled (32 along K, e8m0 scale) e2m3 / e4m3.
Error is measured against an fp64 stack through 13 VGG-like layers.   python tools/f16mx_numerics.py"""
import torch
import torch.nn.functional as F

torch.manual_seed(0)


def _round_to_grid(x, fmt):
    """x already divided by the block scale; round-to-nearest-even onto the element grid, saturating."""
    ax = x.abs()
    if fmt == "e2m3":      # 1-2-3, bias 1: subnormal step 1/8 below 1, max 7.5
        e = torch.floor(torch.log2(ax.clamp(min=1.0))).clamp(max=2)
        step = torch.pow(2.0, e - 3)
        q = torch.round(ax / step) * step
        q = q.clamp(max=7.5)
    elif fmt == "e3m2":    # 1-3-2, bias 3: min normal 0.25, max 28
        e = torch.floor(torch.log2(ax.clamp(min=0.25))).clamp(max=4)
        step = torch.pow(2.0, e - 2)
        q = torch.round(ax / step) * step
        q = q.clamp(max=28.0)
    elif fmt == "e4m3":    # OCP e4m3fn: bias 7, min normal 2^-6, max 448
        e = torch.floor(torch.log2(ax.clamp(min=2.0 ** -6))).clamp(max=8)
        step = torch.pow(2.0, e - 3)
        q = torch.round(ax / step) * step
        q = q.clamp(max=448.0)
    else:
        raise ValueError(fmt)
    return torch.sign(x) * q


_FMAX = {"e2m3": 7.5, "e3m2": 28.0, "e4m3": 448.0}


def mxq(x, fmt, dim, scale_from=None, scale_shift=0):
    """Block-scaled quantisation, blocks of 32 along `dim`.  scale_from: tensor whose block maxima set
    the scale (then shifted by 2^scale_shift) instead of x's own."""
    x = x.movedim(dim, -1)
    shp = x.shape
    xb = x.reshape(*shp[:-1], shp[-1] // 32, 32)
    src = xb if scale_from is None else scale_from.movedim(dim, -1).reshape(xb.shape)
    m = src.abs().amax(-1, keepdim=True).clamp(min=2.0 ** -100)
    s = torch.ceil(torch.log2(m / _FMAX[fmt])) + scale_shift            # e8m0 exponent
    sc = torch.pow(2.0, s)
    q = _round_to_grid(xb / sc, fmt) * sc
    return q.reshape(shp).movedim(-1, dim)


def split(v, hi_t):
    hi = v.to(hi_t).float()
    return hi, v - hi


def conv_mixed(x, w, hi_t, fmt, tied):
    xh, xl = split(x, hi_t)
    wh, wl = split(w, hi_t)
    lo_bits = 11 if hi_t == torch.float16 else 8
    if fmt is None:
        return F.conv2d(xh, wh, padding=1)
    if fmt == "x3":
        lo_t = hi_t
        xl, wl = xl.to(lo_t).float(), wl.to(lo_t).float()
        return F.conv2d(xl, wh, padding=1) + F.conv2d(xh, wl, padding=1) + F.conv2d(xh, wh, padding=1)
    if tied:   # lo scale = hi scale * 2^-lo_bits (one block max per group)
        xhq, whq = mxq(xh, fmt, 1), mxq(wh, fmt, 1)
        xlq = mxq(xl, fmt, 1, scale_from=xh, scale_shift=-lo_bits - (1 if fmt == "e2m3" else 0))
        wlq = mxq(wl, fmt, 1, scale_from=wh, scale_shift=-lo_bits - (1 if fmt == "e2m3" else 0))
    else:
        xhq, whq, xlq, wlq = mxq(xh, fmt, 1), mxq(wh, fmt, 1), mxq(xl, fmt, 1), mxq(wl, fmt, 1)
    return F.conv2d(xh, wh, padding=1) + (F.conv2d(xhq, wlq, padding=1) + F.conv2d(xlq, whq, padding=1))


def rel(a, ref):
    return float(((a.double() - ref) ** 2).sum().sqrt() / (ref ** 2).sum().sqrt())


CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]
MODES = {
    "fp32": None,
    "bf16": (torch.bfloat16, None, False),
    "f16": (torch.float16, None, False),
    "bf16x3": (torch.bfloat16, "x3", False),
    "f16+e2m3": (torch.float16, "e2m3", False),
    "f16+e2m3 tied": (torch.float16, "e2m3", True),
    "f16+e3m2": (torch.float16, "e3m2", False),
    "f16+e4m3": (torch.float16, "e4m3", False),
    "bf16+e4m3": (torch.bfloat16, "e4m3", False),
    "bf16+e2m3": (torch.bfloat16, "e2m3", False),
}

g = torch.Generator().manual_seed(1)
x0 = (torch.randint(0, 256, (1, 3, 96, 128), generator=g).float() - torch.tensor([123.68, 116.78, 103.94]).view(1, 3, 1, 1))
ws, cin = [], 3
for v in CFG:
    if v == "M":
        continue
    ws.append(torch.randn(v, cin, 3, 3) * (2.0 / (9 * v)) ** 0.5)   # kaiming fan_out, as vgg.py:72-77
    cin = v

state = {k: x0.clone() for k in MODES}
ref = x0.double()
li = 0
for v in CFG:
    if v == "M":
        ref = F.max_pool2d(ref, 2)
        state = {k: F.max_pool2d(s, 2) for k, s in state.items()}
        continue
    w = ws[li]
    last = li == len(ws) - 1
    ref = F.conv2d(ref, w.double(), padding=1)
    for k, mode in MODES.items():
        s = state[k]
        if mode is None or li == 0:        # conv1_1 (K = 27) stays on the three-term path
            out = F.conv2d(s, w, padding=1) if mode is None else conv_mixed(s, w, mode[0], "x3" if mode[1] else None, False)
        else:
            out = conv_mixed(s, w, *mode)
        state[k] = out if last else F.relu(out)
    if not last:
        ref = F.relu(ref)
    li += 1
    print(f"layer {li:2d} ({w.shape[1]:3d}->{w.shape[0]:3d}): " + " | ".join(f"{k} {rel(state[k], ref):.1e}" for k in MODES), flush=True)
