"""bf16x3 ("split bf16") mode on the GPU: every operand is carried as hi = bf16(v), lo = bf16(v - hi),
a product costs three bf16 MFMAs and the results are fp32-class — the mode that has to meet
north_star's 1e-4 descriptor tolerance at matrix-core speed.  Checked against fp64 host
computations, the reference's own vectors (tests/golden) and the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_desc, assert_rel_l2, load_golden, rel_l2
from openibl_amd import ops, synth
from oracle import descriptor as od
from oracle import matching as om

pytestmark = pytest.mark.gpu

TOL_DESC = 1e-4      # north_star: descriptors within 1e-4 relative of the reference CPU path
TOL_LAYER = 2e-5     # one contraction: 2^-17 per product (random signs) + 2^-17 output representation


def _bf16(x):
    return x.to(torch.bfloat16).float()


def test_split_join_roundtrip(dev):
    g = torch.Generator().manual_seed(5)
    x = torch.randn((37, 96), generator=g) * torch.logspace(-6, 6, 96)[None, :]
    x[0, :4] = torch.tensor([0.0, -0.0, 1.0, -3.5])
    s = ops.x3_split(x.to(dev))
    assert s.dtype == torch.int32 and s.shape == x.shape
    back = ops.x3_join(s).cpu()
    hi = _bf16(x)
    lo = _bf16(x - hi)
    assert torch.equal(back, hi + lo)                 # exactly the documented split
    err = ((back - x).abs() / x.abs().clamp_min(1e-30)).max().item()
    assert err <= 2.0 ** -16, err
    # layout: group g of a row = [32 hi | 32 lo] bf16
    raw = s.cpu().view(torch.int16).reshape(37, 3, 2, 32)
    assert torch.equal(raw[:, :, 0, :], hi.to(torch.bfloat16).view(torch.int16).reshape(37, 3, 32))
    assert torch.equal(raw[:, :, 1, :], lo.to(torch.bfloat16).view(torch.int16).reshape(37, 3, 32))


def _case(N, H, W, cin, cout, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((N, cin, H, W), generator=g)
    w = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.1
    return x, w, b


def _host_conv(x, w, b, relu, pool):
    y = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if relu:
        y = F.relu(y)
    if pool:
        y = F.max_pool2d(y, 2, 2)
    return y


def _x3_in(x, dev):
    return ops.x3_split(ops.nchw_f32_to_nhwc(x.to(dev), "fp32"))


def _x3_out(y):
    return ops.nhwc_to_nchw_f32(ops.x3_join(y)).cpu()


@pytest.mark.parametrize("regstage", [False, True])
@pytest.mark.parametrize("N,H,W,cin,cout,relu,pool", [
    (2, 12, 20, 64, 64, True, False),
    (2, 12, 20, 64, 64, True, True),
    (1, 9, 7, 64, 128, True, False),      # odd sizes, partial tiles
    (1, 9, 7, 128, 128, True, True),      # odd sizes + pooling floors
    (3, 8, 8, 256, 256, False, False),
    (1, 30, 40, 512, 512, False, False),  # conv5_3 shape
    (2, 6, 10, 256, 512, True, True),
    (1, 5, 6, 32, 64, True, False),       # one K-tile per tap
])
def test_conv3x3_x3(dev, N, H, W, cin, cout, relu, pool, regstage):
    x, w, b = _case(N, H, W, cin, cout, seed=H * 1000 + cin)
    ops.set_regstage(regstage)
    try:
        wp = ops.pack_conv3x3(w.to(dev), "bf16x3")
        assert wp.dtype == torch.int32 and tuple(wp.shape) == (9, cout, cin)
        y = ops.conv3x3_nhwc(_x3_in(x, dev), wp, b.to(dev), relu, pool, "bf16x3")
    finally:
        ops.set_regstage(False)
    assert_rel_l2(f"conv3x3 x3 {N}x{H}x{W} {cin}->{cout} relu={relu} pool={pool}", _x3_out(y),
                  _host_conv(x, w, b, relu, pool), TOL_LAYER)


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("N,H,W,cin,cout,relu,pool", [
    (2, 12, 20, 64, 64, True, True),
    (3, 37, 45, 64, 64, True, True),      # conv1_2 family: 512 x 64 tile (tile=3), ragged, pooled
    (2, 30, 40, 64, 64, False, False),
    (1, 9, 7, 128, 128, True, True),
    (3, 20, 24, 256, 256, True, False),
    (1, 30, 40, 512, 512, False, False),
    (2, 16, 20, 256, 512, True, True),
    (1, 17, 23, 128, 256, True, True),    # ring kernel: one partial 256-row tile, pooling floors
    (5, 21, 19, 128, 256, True, False),   # ring kernel: several M tiles, ragged tail (two-pass epilogue)
    (2, 10, 14, 512, 256, False, True),
    (3, 33, 21, 128, 128, True, True),    # ring kernel, 512 x 128 tile: ragged, pooled
    (2, 40, 30, 256, 128, False, False),  # ring kernel, 512 x 128 tile: several M tiles
    (2, 40, 30, 64, 128, True, False),    # conv2_1 family: 18 K-tiles
])
def test_conv3x3_x3_tile_variants(dev, N, H, W, cin, cout, relu, pool, tile):
    """Generic 128-row / 256-row tiles and the ring kernels give the same tensor bit for bit (same
    per-element chain: bias, then per K-tile lo.hi, hi.lo, hi.hi of each 16-wide half)."""
    x, w, b = _case(N, H, W, cin, cout, seed=tile + H)
    xd = _x3_in(x, dev)
    wp = ops.pack_conv3x3(w.to(dev), "bf16x3")
    ref = ops.conv3x3_nhwc(xd, wp, b.to(dev), relu, pool, "bf16x3")
    ops.set_conv_tile(tile)
    try:
        y = ops.conv3x3_nhwc(xd, wp, b.to(dev), relu, pool, "bf16x3")
    finally:
        ops.set_conv_tile(0)
    assert torch.equal(y, ref)
    assert_rel_l2(f"conv3x3 x3 tile={tile}", _x3_out(y), _host_conv(x, w, b, relu, pool), TOL_LAYER)


@pytest.mark.parametrize("N,H,W", [(1, 8, 128), (2, 13, 150), (1, 33, 70)])
def test_conv1_1_x3(dev, N, H, W):
    g = torch.Generator().manual_seed(W)
    x = (torch.randint(0, 256, (N, 3, H, W), generator=g).float() - 116.0)   # pixel-like values
    w = torch.randn((64, 3, 3, 3), generator=g) * 0.2
    b = torch.randn((64,), generator=g)
    y = ops.conv1_1_nchw(x.to(dev), w.to(dev), b.to(dev), "bf16x3")
    assert y.dtype == torch.int32 and tuple(y.shape) == (N, H, W, 64)
    want = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    assert_rel_l2(f"conv1_1 x3 {N}x{H}x{W}", _x3_out(y), want, TOL_LAYER)


@pytest.fixture(scope="module")
def model(state_dict, dev):
    import hubconf
    m = hubconf.vgg16_netvlad(pretrained=False)
    m.load_state_dict(state_dict)
    return m.to(dev).eval().set_precision("bf16x3")


@pytest.mark.parametrize("name", ["desc_small", "desc_odd", "desc_480x640"])
def test_embednetpca_x3_matches_reference(name, model, dev):
    """All three vectors the reference produced: descriptor and every stage within 1e-4."""
    g = load_golden(name)
    n, _, h, w = [int(v) for v in g["shape"]]
    x = synth.images(n, h, w, seed=int(g["image_seed"])).to(dev)
    desc = model(x)
    assert tuple(desc.shape) == (n, 4096) and desc.dtype == torch.float32
    assert_desc(f"{name} desc (bf16x3)", desc.cpu(), g["desc"], TOL_DESC)
    pool_x, feat = model.base_model(x)
    s = int(g["feat_stride"])
    assert_rel_l2(f"{name} feat (bf16x3)", feat.cpu()[:, ::s], g["feat"], TOL_DESC)
    assert_rel_l2(f"{name} pool_x (bf16x3)", pool_x.cpu(), g["pool_x"], TOL_DESC)
    from ibl import models
    emb = models.create("embednet", model.base_model, model.net_vlad).eval().set_precision("bf16x3")
    _, vlad = emb(x)
    assert_rel_l2(f"{name} vlad_norm (bf16x3)", vlad.cpu(), g["vlad_norm"], TOL_DESC)
    from ibl.evaluators import extract_cnn_feature
    assert_rel_l2(f"{name} ecf pca (bf16x3)", extract_cnn_feature(model, x.cpu()).cpu(), g["ecf_pca"],
                  TOL_DESC)


def test_embednetpca_x3_vs_fp64_oracle(model, dev, state_dict):
    x = synth.images(2, 80, 112, seed=77)
    want = od.embednetpca(x, state_dict, dtype=torch.float64)
    got = model(x.to(dev)).cpu()
    assert_desc("desc bf16x3 vs fp64 oracle", got, want, TOL_DESC)
    model.set_precision("fp32")
    try:
        ref32 = model(x.to(dev)).cpu()
    finally:
        model.set_precision("bf16x3")
    print(f"bf16x3 vs fp64: {rel_l2(got, want):.3e}; fp32 mode vs fp64: {rel_l2(ref32, want):.3e}")


def test_x3_graphed_and_uint8(model, dev):
    """hipGraph replay (both pipeline slots) reproduces the eager forward bit for bit; so does the uint8 entry
    through the normalising pass (test hook) — the fused uint8 stem (the default since round 4: Normalize as
    one fma per value, tests/test_gpu_u8.py) is within 2e-5 of it."""
    from ibl.utils.data import MEAN, STD
    from openibl_amd import lib
    g = torch.Generator().manual_seed(3)
    u8 = torch.randint(0, 256, (2, 64, 96, 3), generator=g, dtype=torch.uint8)
    mean = torch.tensor(MEAN, dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(STD, dtype=torch.float32).view(1, 3, 1, 1)
    x = ((u8.permute(0, 3, 1, 2).float() / 255.0 - mean) / std).to(dev)
    want = model(x).clone()
    assert rel_l2(model(u8.to(dev)).cpu(), want.cpu()) < 2e-5
    lib.debug_hooks().oibl_debug_set_stem_u8(0)
    try:
        assert torch.equal(model(u8.to(dev)), want)
    finally:
        lib.debug_hooks().oibl_debug_set_stem_u8(1)
    pf = model.graphed(x, pipeline=True)
    a, b = pf(), pf()
    pf.wait()
    torch.cuda.synchronize()
    assert torch.equal(a, want) and torch.equal(b, want)


# ---- matching -------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,d", [(5, 7, 64), (130, 300, 128), (256, 512, 4096), (300, 1100, 4096)])
def test_pairwise_x3(dev, m, n, d):
    g = torch.Generator().manual_seed(m + n)
    x = F.normalize(torch.randn((m, d), generator=g), dim=1)
    y = F.normalize(torch.randn((n, d), generator=g), dim=1)
    got = ops.pairwise_sqdist(x.to(dev), y.to(dev), "bf16x3").cpu()
    want = (x.double()[:, None, :] - y.double()[None, :, :]).pow(2).sum(-1) if m * n * d < 5e8 else \
        (x.double().pow(2).sum(1)[:, None] + y.double().pow(2).sum(1)[None, :] - 2 * x.double() @ y.double().T)
    err = (got.double() - want).abs().max().item()
    print(f"pairwise x3 {m}x{n}x{d}: max abs err {err:.3e}")
    assert err < 2e-5      # unit vectors: distances in [0, 4]; fp32 mode sits at ~5e-6
    for ring in (0, 2):    # generic kernel vs ring kernel (where legal): identical matrices
        ops.set_match_ring(ring)
        try:
            again = ops.pairwise_sqdist(x.to(dev), y.to(dev), "bf16x3").cpu()
        finally:
            ops.set_match_ring(1)
        assert torch.equal(again, got)


def test_matching_x3_equals_reference_ranking(dev):
    """Golden matching problem of the reference: distances within 1e-5, identical top-20 ranks."""
    for name in ("match_small", "match_nms"):
        g = load_golden(name)
        q, gal, gt, pids = synth.retrieval_problem(
            int(g["Q"]), int(g["G"]), dim=int(g["dim"]), seed=int(g["seed"]),
            views_per_place=int(g["views_per_place"]), hard_fraction=float(g["hard_fraction"]),
            hard_noise_mult=float(g["hard_noise_mult"]))
        d = ops.pairwise_sqdist(q.to(dev), gal.to(dev), "bf16x3")
        err = np.abs(d.cpu().numpy().astype(np.float64) - g["distmat"]).max()
        print(f"{name}: bf16x3 max |dist - reference| = {err:.3e}")
        assert err < 2e-5
        _, idx = ops.row_topk(d, 20)
        assert np.array_equal(idx.cpu().numpy(), g["top20"])
        from ibl.evaluators import evaluate_all
        gallery = [(f"g{j:05d}.jpg", pids[j], 0.0, 0.0) for j in range(len(gal))]
        np.testing.assert_array_equal(evaluate_all(d.cpu(), gt, gallery), g["recalls"])
        np.testing.assert_array_equal(evaluate_all(d.cpu(), gt, gallery, nms=True), g["recalls_nms"])


def test_sqdist_topk_x3_fused_equals_matrix(dev):
    """Fused threshold / filter / select path in bf16x3 == top-k of its own matrix (16k gallery)."""
    q, gal, gt, _ = synth.retrieval_problem(512, 16384, seed=9)
    qd, gd = q.to(dev), gal.to(dev)
    v, i = ops.sqdist_topk(qd, gd, 10, precision="bf16x3")
    dm = ops.pairwise_sqdist(qd, gd, "bf16x3")
    v2, i2 = ops.row_topk(dm, 10)
    assert torch.equal(i, i2) and torch.equal(v, v2)
    # against fp64: the selected entries are the true nearest ones up to the kernel's distance error
    # (near-ties within ~1e-6 may legitimately swap, exactly as they do in the fp32 oracle)
    d64 = (q.double().pow(2).sum(1)[:, None] + gal.double().pow(2).sum(1)[None, :]
           - 2.0 * q.double() @ gal.double().T)
    true_v = torch.sort(d64, dim=1).values[:, :10]
    got_v = torch.gather(d64, 1, i.cpu().long())
    assert (got_v - true_v).abs().max().item() < 5e-6
    want = om.ranking(om.pairwise_distance(q, gal).numpy())[:, :10]
    agree = (i.cpu().numpy() == want).mean()
    print(f"top-10 agreement with the fp32 oracle ranking: {agree:.6f}")
    assert agree > 0.999


@pytest.mark.parametrize("N,H,W", [(1, 8, 32), (2, 16, 64), (1, 21, 45), (3, 30, 70), (1, 2, 2),
                                   (2, 9, 33), (1, 64, 96), (5, 40, 136)])
def test_vgg_stem_x3_fused(dev, N, H, W):
    """bf16x3: conv1_1 + conv1_2 + pool in one launch (two channel-half passes per tile, half of the
    output channels per workgroup): bit-identical to the two unfused launches with conv1_2 in the K
    order (channel chunk, tap), and within the layer tolerance of the fp64 host convolutions."""
    x, w1, b1 = _case(N, H, W, 3, 64, seed=5 * H + W)
    x = x * 60.0
    _, w2, b2 = _case(1, 4, 4, 64, 64, seed=H + 9 * W)
    wp2 = ops.pack_conv3x3(w2.to(dev), "bf16x3")
    y = ops.vgg16_stem_x3(x.to(dev), w1.to(dev), b1.to(dev), wp2, b2.to(dev))
    assert tuple(y.shape) == (N, H // 2, W // 2, 64) and y.dtype == torch.int32
    a1 = ops.conv1_1_nchw(x.to(dev), w1.to(dev), b1.to(dev), "bf16x3")
    ops.set_conv_korder(1)
    try:
        ref = ops.conv3x3_nhwc(a1, wp2, b2.to(dev), True, True, "bf16x3")
    finally:
        ops.set_conv_korder(-1)
    assert torch.equal(y, ref)
    h1 = F.relu(F.conv2d(x.double(), w1.double(), b1.double(), padding=1))
    want = F.max_pool2d(F.relu(F.conv2d(h1, w2.double(), b2.double(), padding=1)), 2, 2)
    if want.numel():
        assert_rel_l2("fused bf16x3 stem vs host", _x3_out(y), want, 3e-5)


def test_vgg16_backbone_x3_stem_toggle(dev, state_dict):
    """The backbone entry with and without the fused bf16x3 stem: the same feature map up to the
    summation order of conv1_2."""
    x = synth.images(2, 64, 96, seed=4).to(dev)
    ws = [state_dict[f"base_model.base.{i}.weight"].to(dev) for i in synth.CONV_IDX]
    bs = [state_dict[f"base_model.base.{i}.bias"].to(dev) for i in synth.CONV_IDX]
    packed = [ws[0]] + [ops.pack_conv3x3(w, "bf16x3") for w in ws[1:]]
    a = ops.vgg16_conv5(x, packed, bs, "bf16x3")
    ops.set_stem_fused(False)
    try:
        b = ops.vgg16_conv5(x, packed, bs, "bf16x3")
    finally:
        ops.set_stem_fused(True)
    assert a.dtype == torch.float32
    assert_rel_l2("backbone bf16x3, fused vs unfused stem", a.cpu(), b.cpu(), 1e-5)
