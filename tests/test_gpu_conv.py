"""3x3 convolution kernels against F.conv2d on the host (GPU)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_rel_l2
from openibl_amd import ops

pytestmark = pytest.mark.gpu


def _case(N, H, W, cin, cout, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((N, cin, H, W), generator=g)
    w = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.1
    return x, w, b


def _host_conv(x, w, b, relu, pool, precision):
    """What the kernel computes: operands rounded to the storage type, exact accumulation."""
    if precision == "bf16":
        x, w = x.to(torch.bfloat16).float(), w.to(torch.bfloat16).float()
    y = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if relu:
        y = F.relu(y)
    if pool:
        y = F.max_pool2d(y, 2, 2)
    return y


@pytest.mark.parametrize("regstage", [False, True])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("N,H,W,cin,cout,relu,pool", [
    (2, 12, 20, 64, 64, True, False),
    (2, 12, 20, 64, 64, True, True),
    (1, 9, 7, 64, 128, True, False),      # odd sizes, partial tiles
    (1, 9, 7, 128, 128, True, True),      # odd sizes + pooling floors
    (3, 8, 8, 256, 256, False, False),
    (1, 30, 40, 512, 512, False, False),  # conv5_3 shape
    (2, 6, 10, 256, 512, True, True),
])
def test_conv3x3(dev, N, H, W, cin, cout, relu, pool, precision, regstage):
    x, w, b = _case(N, H, W, cin, cout, seed=H * 1000 + cin)
    ops.set_regstage(regstage)
    try:
        xd = ops.nchw_f32_to_nhwc(x.to(dev), precision)
        wp = ops.pack_conv3x3(w.to(dev), precision)
        y = ops.conv3x3_nhwc(xd, wp, b.to(dev), relu, pool, precision)
    finally:
        ops.set_regstage(False)
    got = ops.nhwc_to_nchw_f32(y).cpu()
    want = _host_conv(x, w, b, relu, pool, precision)
    tol = 2e-6 if precision == "fp32" else 4e-3   # bf16: output rounding (2^-9 relative)
    assert_rel_l2(f"conv3x3 {precision} {N}x{H}x{W} {cin}->{cout} relu={relu} pool={pool}",
                  got, want, tol)


@pytest.mark.parametrize("tile", [1, 2, 3, 4])
@pytest.mark.parametrize("N,H,W,cin,cout,relu,pool", [
    (2, 12, 20, 64, 64, True, True),
    (1, 9, 7, 128, 128, True, True),
    (3, 20, 24, 256, 256, True, False),
    (1, 30, 40, 512, 512, False, False),
    (2, 16, 20, 256, 512, True, True),
    (1, 17, 23, 128, 256, True, True),    # ring kernel: one partial 256-row tile, pooling floors
    (5, 21, 19, 128, 256, True, False),   # ring kernel: several M tiles, ragged tail
    (2, 10, 14, 512, 256, False, True),
    (3, 33, 21, 128, 128, True, True),    # ring kernel, 512 x 128 tile: ragged, pooled
    (2, 40, 30, 256, 128, False, False),  # ring kernel, 512 x 128 tile: several M tiles
])
def test_conv3x3_bf16_tile_variants(dev, N, H, W, cin, cout, relu, pool, tile):
    """Every tile shape of the implicit GEMM gives the same tensor (bit for bit: the K order and the
    per-element arithmetic do not depend on the tile)."""
    x, w, b = _case(N, H, W, cin, cout, seed=tile + H)
    xd = ops.nchw_f32_to_nhwc(x.to(dev), "bf16")
    wp = ops.pack_conv3x3(w.to(dev), "bf16")
    ref = ops.conv3x3_nhwc(xd, wp, b.to(dev), relu, pool, "bf16")
    ops.set_conv_tile(tile)
    try:
        y = ops.conv3x3_nhwc(xd, wp, b.to(dev), relu, pool, "bf16")
    finally:
        ops.set_conv_tile(0)
    assert torch.equal(y, ref)
    got = ops.nhwc_to_nchw_f32(y).cpu()
    assert_rel_l2(f"conv3x3 bf16 tile={tile}", got, _host_conv(x, w, b, relu, pool, "bf16"), 4e-3)


@pytest.mark.parametrize("N,H,W,cout,relu,pool", [
    (1, 8, 32, 64, True, False),       # exactly one tile
    (2, 24, 96, 64, True, True),       # several tiles, pooled (conv1_2 shape family)
    (1, 21, 45, 64, True, True),       # ragged tiles, odd sizes: pooling floors
    (1, 21, 45, 128, True, False),     # two output-channel slices (conv2_1 family), ragged
    (3, 10, 70, 128, False, False),
    (1, 60, 80, 64, False, True),
    (2, 7, 5, 64, True, True),         # smaller than one tile
])
def test_conv3x3_cin64_resident_kernel(dev, N, H, W, cout, relu, pool):
    """The resident-weights / LDS-halo kernel against the host convolution and against the generic
    implicit-GEMM kernel (same operands, different summation order -> equal up to bf16 rounding of
    the output)."""
    x, w, b = _case(N, H, W, 64, cout, seed=W * 3 + cout)
    xd = ops.nchw_f32_to_nhwc(x.to(dev), "bf16")
    wp = ops.pack_conv3x3(w.to(dev), "bf16")
    ops.set_conv_c64(2)          # force the resident kernel also where auto would pick the ring kernel
    try:
        y = ops.conv3x3_nhwc(xd, wp, b.to(dev), relu, pool, "bf16")
        ops.set_conv_c64(False)
        ops.set_conv_tile(1)
        y_ref = ops.conv3x3_nhwc(xd, wp, b.to(dev), relu, pool, "bf16")
    finally:
        ops.set_conv_tile(0)
        ops.set_conv_c64(True)
    got = ops.nhwc_to_nchw_f32(y).cpu()
    assert_rel_l2("c64 vs host", got, _host_conv(x, w, b, relu, pool, "bf16"), 4e-3)
    assert_rel_l2("c64 vs igemm", got, ops.nhwc_to_nchw_f32(y_ref).cpu(), 4e-3)
    mism = (y.float() != y_ref.float()).float().mean().item()
    print(f"fraction of outputs differing from the igemm kernel by an ulp: {mism:.4f}")
    assert mism < 0.05


@pytest.mark.parametrize("N,H,W,cout,relu,pool", [
    (1, 21, 45, 128, True, False),
    (2, 12, 20, 128, True, True),
    (3, 30, 34, 256, False, False),
])
def test_conv3x3_ring_cin64(dev, N, H, W, cout, relu, pool):
    """Cin = 64 gives nine K-tiles (odd): the ring kernel's odd-count schedule against the generic
    implicit GEMM (same K order -> identical tensors)."""
    x, w, b = _case(N, H, W, 64, cout, seed=W + cout)
    xd = ops.nchw_f32_to_nhwc(x.to(dev), "bf16")
    wp = ops.pack_conv3x3(w.to(dev), "bf16")
    ops.set_conv_c64(False)
    try:
        ops.set_conv_tile(1)
        ref = ops.conv3x3_nhwc(xd, wp, b.to(dev), relu, pool, "bf16")
        ops.set_conv_tile(4)
        y = ops.conv3x3_nhwc(xd, wp, b.to(dev), relu, pool, "bf16")
    finally:
        ops.set_conv_tile(0)
        ops.set_conv_c64(True)
    assert torch.equal(y, ref)
    assert_rel_l2("ring cin64 vs host", ops.nhwc_to_nchw_f32(y).cpu(),
                  _host_conv(x, w, b, relu, pool, "bf16"), 4e-3)


@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16-valu"])
@pytest.mark.parametrize("N,H,W", [(2, 16, 24), (1, 7, 13), (1, 33, 9), (1, 5, 300), (2, 3, 129)])
def test_conv1_1(dev, N, H, W, precision):
    x, w, b = _case(N, H, W, 3, 64, seed=77 + H)
    x = x * 60.0
    valu = precision.endswith("valu")
    prec = precision.split("-")[0]
    ops.set_conv11_valu(valu)
    try:
        y = ops.conv1_1_nchw(x.to(dev), w.to(dev), b.to(dev), prec)
    finally:
        ops.set_conv11_valu(False)
    got = ops.nhwc_to_nchw_f32(y).cpu()
    if precision == "bf16":     # MFMA kernel: operands rounded to bf16, exact accumulation
        xr, wr = x.to(torch.bfloat16).double(), w.to(torch.bfloat16).double()
    else:                       # vector-ALU kernel: exact fp32 operands
        xr, wr = x.double(), w.double()
    want = F.relu(F.conv2d(xr, wr, b.double(), padding=1))
    assert_rel_l2(f"conv1_1 {precision}", got, want, 2e-6 if prec == "fp32" else 4e-3)


def test_pack_layout(dev):
    w = torch.arange(2 * 64 * 64 * 9, dtype=torch.float32).reshape(128, 64, 3, 3) / 1000.0
    p = ops.pack_conv3x3(w.to(dev), "fp32").cpu()
    assert tuple(p.shape) == (9, 128, 64)
    assert torch.equal(p, w.permute(2, 3, 0, 1).reshape(9, 128, 64))


@pytest.mark.parametrize("N,H,W", [(1, 8, 32), (2, 16, 64), (1, 21, 45), (3, 30, 70), (1, 2, 2),
                                   (2, 9, 33), (1, 64, 96)])
def test_vgg_stem_fused(dev, N, H, W):
    """conv1_1 + conv1_2 + pool in one launch: bit-identical to the two unfused bf16 launches, and
    within bf16 rounding of the host convolutions."""
    x, w1, b1 = _case(N, H, W, 3, 64, seed=5 * H + W)
    x = x * 60.0
    _, w2, b2 = _case(1, 4, 4, 64, 64, seed=H + 9 * W)
    wp2 = ops.pack_conv3x3(w2.to(dev), "bf16")
    y = ops.vgg16_stem(x.to(dev), w1.to(dev), b1.to(dev), wp2, b2.to(dev))
    a1 = ops.conv1_1_nchw(x.to(dev), w1.to(dev), b1.to(dev), "bf16")
    ref = ops.conv3x3_nhwc(a1, wp2, b2.to(dev), True, True, "bf16")
    assert tuple(y.shape) == (N, H // 2, W // 2, 64)
    assert torch.equal(y, ref)
    h1 = F.relu(F.conv2d(x.to(torch.bfloat16).double(), w1.to(torch.bfloat16).double(), b1.double(),
                         padding=1)).to(torch.bfloat16)
    want = F.max_pool2d(F.relu(F.conv2d(h1.double(), w2.to(torch.bfloat16).double(), b2.double(),
                                        padding=1)), 2, 2)
    assert_rel_l2("fused stem vs host", ops.nhwc_to_nchw_f32(y).cpu(), want, 6e-3)


def test_vgg16_backbone_stem_toggle(dev):
    """The backbone entry gives the same feature map with and without the fused stem."""
    from openibl_amd import synth
    sd = synth.embednetpca_state(0)
    x = synth.images(2, 64, 96, seed=4).to(dev)
    ws = [sd[f"base_model.base.{i}.weight"].to(dev) for i in synth.CONV_IDX]
    bs = [sd[f"base_model.base.{i}.bias"].to(dev) for i in synth.CONV_IDX]
    packed = [ws[0]] + [ops.pack_conv3x3(w, "bf16") for w in ws[1:]]
    a = ops.vgg16_conv5(x, packed, bs, "bf16")
    ops.set_stem_fused(False)
    try:
        b = ops.vgg16_conv5(x, packed, bs, "bf16")
    finally:
        ops.set_stem_fused(True)
    assert torch.equal(a, b)


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
@pytest.mark.parametrize("N,H,W,cin,cout,relu,pool", [
    (5, 21, 19, 128, 512, True, False),    # 8 M tiles x 2 N tiles
    (3, 33, 21, 256, 512, True, True),     # ragged: 9 x 2
    (7, 30, 40, 512, 512, False, False),   # conv5 shape, 33 x 2 tiles (odd count per N-tile group)
    (2, 40, 30, 256, 256, True, False),    # one N-tile: raster 1 == raster 0 mapping
])
def test_ring_raster_modes_agree(dev, N, H, W, cin, cout, relu, pool, precision):
    """One N-tile per XCD (raster 1) is a permutation of the tile order: same tensor, bit for bit."""
    x, w, b = _case(N, H, W, cin, cout, seed=7 + H)
    xd = ops.nchw_f32_to_nhwc(x.to(dev), "fp32" if precision == "bf16x3" else precision)
    if precision == "bf16x3":
        xd = ops.x3_split(xd)
    wp = ops.pack_conv3x3(w.to(dev), precision)
    ops.set_conv_tile(4)
    try:
        ref = ops.conv3x3_nhwc(xd, wp, b.to(dev), relu, pool, precision)
        ops.set_ring_raster(1)
        y = ops.conv3x3_nhwc(xd, wp, b.to(dev), relu, pool, precision)
    finally:
        ops.set_ring_raster(0)
        ops.set_conv_tile(0)
    assert torch.equal(y, ref)


@pytest.mark.parametrize("precision", ["bf16", "bf16x3", "fp32"])
def test_chunk_major_k_order_hook(dev, precision):
    """K order (channel chunk, tap) — the fetch-volume experiment of conv_ring.h — computes the same
    convolution (other summation order: equal to rounding), and ring == generic bit for bit in it."""
    N, H, W, cin, cout = 3, 20, 24, 256, 256
    x, w, b = _case(N, H, W, cin, cout, seed=77)
    xd = ops.nchw_f32_to_nhwc(x.to(dev), "fp32" if precision == "bf16x3" else precision)
    if precision == "bf16x3":
        xd = ops.x3_split(xd)
    wp = ops.pack_conv3x3(w.to(dev), precision)
    ref = ops.conv3x3_nhwc(xd, wp, b.to(dev), True, False, precision)
    ops.set_conv_korder(1)
    try:
        y = ops.conv3x3_nhwc(xd, wp, b.to(dev), True, False, precision)
        ops.set_conv_tile(1)
        y1 = ops.conv3x3_nhwc(xd, wp, b.to(dev), True, False, precision)
    finally:
        ops.set_conv_tile(0)
        ops.set_conv_korder(-1)
    assert torch.equal(y, y1)
    f = (lambda t: ops.nhwc_to_nchw_f32(ops.x3_join(t) if precision == "bf16x3" else t).cpu())
    want = _host_conv(x, w, b, True, False, "bf16" if precision == "bf16" else "fp32")
    tol = {"bf16": 4e-3, "bf16x3": 2e-5, "fp32": 2e-6}[precision]
    assert_rel_l2(f"korder 1 {precision}", f(y), want, tol)
    assert_rel_l2(f"korder 1 vs 0 {precision}", f(y), f(ref), tol)
