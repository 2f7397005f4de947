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


def matmul_nt(x, y):
    """x [m][d] . y [n][d]^T with the f16mx product, fp64 accumulation."""
    xh, xh6, xl6 = split(x)
    yh, yh6, yl6 = split(y)
    return xh @ yh.T + xh6 @ yl6.T + xl6 @ yh6.T
