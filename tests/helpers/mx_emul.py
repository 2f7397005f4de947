"""Host emulation (torch CPU, fp64 where it matters) of the f16mx operand format and product
(openibl_amd/csrc/common.h): hi = fp16(v), lo = v - hi, q6 = e2m3 with one power-of-two scale per 32
elements, round-to-nearest-even, saturating at 7.5 — as the gfx950 convert instructions were probed to
behave (profiles/r03_a_mx_probe.txt).  Test infrastructure only."""
import torch


def e2m3_rne(x):
    """x (already divided by the block scale) -> nearest e2m3 value, ties to the even code, |x| > 7.5 saturates."""
    ax = x.abs().double()
    e = torch.floor(torch.log2(ax.clamp(min=1.0))).clamp(max=2)       # binade 0 (incl. subnormals), 1, 2
    step = torch.pow(2.0, e - 3)
    q = torch.round(ax / step) * step                                  # torch.round = half to even = even code
    q = q.clamp(max=7.5)
    return (torch.sign(x).double() * q)


def scale_byte(amax):
    """mx_scale_byte (common.h) on a float32 tensor of block maxima."""
    bits = amax.float().view(torch.int32).long()
    b = ((bits + 0x00100000) >> 23) - 2
    return b.clamp(12, 254)


def split(x):
    """float32 [..., C] -> (hi, hi6, lo6) float64 tensors of the same shape: the three images the kernels see."""
    shp = x.shape
    xb = x.float().reshape(-1, shp[-1] // 32, 32)
    hi = xb.clamp(-65504.0, 65504.0).half().float()
    lo = xb - hi
    bh = scale_byte(hi.abs().amax(-1, keepdim=True))
    sh = torch.pow(2.0, (bh - 127).double())
    sl = torch.pow(2.0, (bh - 11 - 127).double())
    hi6 = e2m3_rne(hi.double() / sh) * sh
    lo6 = e2m3_rne(lo.double() / sl) * sl
    return hi.double().reshape(shp), hi6.reshape(shp), lo6.reshape(shp)


def split_half_pack(x):
    """The same images as mx_pack_half makes them (the fused f16mx stem): hi and hi6 as in split(); the lo image
    goes through fp16 first — q6(fp16(lo * 2^11)) on hi's block scale, i.e. lo is rounded twice."""
    shp = x.shape
    xb = x.float().reshape(-1, shp[-1] // 32, 32)
    hi = xb.clamp(-65504.0, 65504.0).half().float()
    lo = xb - hi
    bh = scale_byte(hi.abs().amax(-1, keepdim=True))
    sh = torch.pow(2.0, (bh - 127).double())
    hi6 = e2m3_rne(hi.double() / sh) * sh
    lo16 = (lo * 2048.0).half().double()                    # exact scaling, one rounding to fp16
    lo6 = e2m3_rne(lo16 / sh) * sh / 2048.0
    return hi.double().reshape(shp), hi6.reshape(shp), lo6.reshape(shp)


def conv3x3(x_nchw, w_oihw, bias, relu, pool):
    """The f16mx product of a 3x3 layer in fp64: hi.hi + q6(hi).q6(lo) + q6(lo).q6(hi) (+ bias, ReLU, pool)."""
    import torch.nn.functional as F
    xh, xh6, xl6 = [t.permute(0, 3, 1, 2) for t in split(x_nchw.permute(0, 2, 3, 1).contiguous())]
    wh, wh6, wl6 = [t.permute(0, 3, 1, 2) for t in split(w_oihw.permute(0, 2, 3, 1).contiguous())]   # groups along Cin
    y = F.conv2d(xh, wh, None, padding=1) + F.conv2d(xh6, wl6, None, padding=1) + F.conv2d(xl6, wh6, None, padding=1)
    y = y + bias.double().view(1, -1, 1, 1)
    if relu:
        y = F.relu(y)
    if pool:
        y = F.max_pool2d(y, 2, 2)
    return y


def stem(x_nchw, w1, b1, w2, b2):
    """The fused f16mx stem in fp64, arithmetic for arithmetic: conv1_1 as the split-bf16 product
    (hi.hi + hi.lo + lo.hi), bias, ReLU, the half-line pack of its fp32 result; conv1_2 as the f16mx product of
    those lines with weights packed by mx_pack_line; 2x2 max-pool, bias, ReLU; the half-line pack of the output.
    Returns the value the output lines carry (hi + q6(lo)), NCHW."""
    import torch.nn.functional as F

    def x3(v):
        hi = v.float().bfloat16().float()
        lo = (v.float() - hi).bfloat16().float()
        return hi.double(), lo.double()
    xh, xl = x3(x_nchw)
    wh, wl = x3(w1)
    c1 = F.conv2d(xh, wh, None, padding=1) + F.conv2d(xl, wh, None, padding=1) + F.conv2d(xh, wl, None, padding=1)
    a1 = F.relu(c1 + b1.double().view(1, -1, 1, 1)).clamp(max=65504.0).float()
    ah, ah6, al6 = [t.permute(0, 3, 1, 2) for t in split_half_pack(a1.permute(0, 2, 3, 1).contiguous())]
    w2h, w2h6, w2l6 = [t.permute(0, 3, 1, 2) for t in split(w2.permute(0, 2, 3, 1).contiguous())]
    y = F.conv2d(ah, w2h, None, padding=1) + F.conv2d(ah6, w2l6, None, padding=1) + F.conv2d(al6, w2h6, None, padding=1)
    y = F.max_pool2d(y, 2, 2) + b2.double().view(1, -1, 1, 1)
    y = F.relu(y).clamp(max=65504.0).float()
    oh, _, ol6 = [t.permute(0, 3, 1, 2) for t in split_half_pack(y.permute(0, 2, 3, 1).contiguous())]
    return oh + ol6


def matmul_nt(x, y):
    """x [m][d] . y [n][d]^T with the f16mx product, fp64 accumulation."""
    xh, xh6, xl6 = split(x)
    yh, yh6, yl6 = split(y)
    return xh @ yh.T + xh6 @ yl6.T + xl6 @ yh6.T
