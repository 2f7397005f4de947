"""f16mx mode on the GPU: every operand travels as hi = fp16(v) plus block-scaled e2m3 images of hi and
of lo = v - hi; a product is hi.hi on the f16 matrix instruction plus BOTH cross terms on ONE MX-fp6
instruction (K-concatenated) — half the matrix-pipe time of bf16x3, still inside north_star's 1e-4 on
the descriptor.  Checked against a host emulation of the format (tests/helpers/mx_emul.py), fp64 host
computations, the reference's own vectors (tests/golden) and the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_rel_l2, load_golden, rel_l2
from helpers import mx_emul
from openibl_amd import ops, synth
from oracle import descriptor as od

pytestmark = pytest.mark.gpu

TOL_DESC = 1e-4      # north_star: descriptors within 1e-4 relative of the reference CPU path
TOL_LAYER = 4e-5     # one contraction (~2^-15 per product, random signs) + the output representation hi + q6(lo)
TOL_EMUL = 3e-6      # device vs the fp64 emulation of the same arithmetic: fp32 accumulation only
TOL_STEM_EMUL = 1e-5  # the same for the fused stem (two contractions, two packs)


def test_split_matches_host_emulation(dev):
    g = torch.Generator().manual_seed(5)
    x = torch.randn((37, 96), generator=g) * torch.logspace(-3, 3, 96)[None, :]
    x[0, :6] = torch.tensor([0.0, -0.0, 1.0, -3.5, 70000.0, -1e-7])     # beyond fp16 range / below its subnormals
    x[1, 32:64] = 0.0                                                  # an all-zero group
    s = ops.mx_split(x.to(dev))
    assert s.dtype == torch.int32 and s.shape == x.shape
    hi, hi6, lo6 = mx_emul.split(x)
    assert torch.equal(ops.mx_join(s, 1).cpu().double(), hi)           # fp16 part, exactly
    assert torch.equal(ops.mx_join(s, 2).cpu().double(), hi6)          # e2m3 image of hi: same codes, same scale
    assert torch.equal(ops.mx_join(s, 3).cpu().double(), lo6)          # e2m3 image of lo
    back = ops.mx_join(s, 0).cpu().double()
    gmax = x.reshape(37, 3, 32).abs().amax(-1, keepdim=True).expand(37, 3, 32).reshape(37, 96).double()
    ok = x.abs() <= 65504
    assert ((back - x.double()).abs()[ok] <= 2.0 ** -14 * gmax[ok] + 1e-30).all()


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


def _mx_in(x, dev):
    return ops.mx_split(ops.nchw_f32_to_nhwc(x.to(dev), "fp32"))


def _mx_out(y, which=0):
    return ops.nhwc_to_nchw_f32(ops.mx_join(y, which)).cpu()


@pytest.fixture(params=[0, 1, 3, 2], ids=["auto", "ring", "halo", "ring-late"])
def mx_variant(request):
    """0 = default dispatch (halo kernel for the 256-output-channel layers, ring kernels elsewhere), 1 = ring
    kernels only, 3 = halo kernel wherever it applies, 2 = ring kernels with the LDS-DMA issue inside the
    COMPUTE segments."""
    from openibl_amd import lib
    lib.debug_hooks().oibl_debug_set_mx_variant(request.param)
    yield request.param
    lib.debug_hooks().oibl_debug_set_mx_variant(0)


@pytest.mark.parametrize("N,H,W,cin,cout,relu,pool", [
    (1, 9, 7, 64, 128, True, False),      # 512 x 128 tile, one partial tile, 18 K-tiles
    (1, 9, 7, 128, 128, True, True),      # odd sizes + pooling floors
    (3, 8, 8, 256, 256, False, False),    # 256 x 256 tile
    (1, 30, 40, 512, 512, False, False),  # conv5_3 shape
    (2, 6, 10, 256, 512, True, True),
    (5, 21, 19, 128, 256, True, False),   # several M tiles, ragged tail (two-pass epilogue)
    (3, 33, 21, 128, 128, True, True),    # 512 x 128 tile: ragged, pooled
    (2, 40, 30, 256, 128, False, False),  # 512 x 128 tile: several M tiles
    (2, 40, 30, 64, 128, True, False),    # conv2_1 family
    (1, 17, 23, 128, 256, True, True),
    (2, 30, 40, 512, 512, True, False),   # conv5 shape: 6 x 40 patches of the halo kernel, two images
    (1, 60, 80, 256, 512, True, True),    # conv4_3-like: 12 x 20 patches, pooled
    (2, 31, 45, 128, 256, True, True),    # odd sizes: patches cut by both borders, pooling floors
    (1, 120, 160, 128, 256, True, False),  # conv3_1 shape: 8 x 32 patches
    (2, 240, 320, 64, 128, True, False),   # conv2_1 shape: 4-wave halo kernel (conv_halo4.h), 8 x 32 patches, 2 chunks
    (1, 240, 320, 128, 128, True, True),   # conv2_2 shape: pooled, 4 chunks (three single-buffered halo reloads)
    (1, 50, 34, 64, 128, False, False),    # 4-wave halo kernel: patches cut by both borders, no ReLU
])
def test_conv3x3_mx(dev, N, H, W, cin, cout, relu, pool, mx_variant):
    x, w, b = _case(N, H, W, cin, cout, seed=H * 1000 + cin)
    wp = ops.pack_conv3x3(w.to(dev), "f16mx")
    assert wp.dtype == torch.int32 and tuple(wp.shape) == (9, cout, cin)
    y = ops.conv3x3_nhwc(_mx_in(x, dev), wp, b.to(dev), relu, pool, "f16mx")
    name = f"conv3x3 f16mx {N}x{H}x{W} {cin}->{cout} relu={relu} pool={pool}"
    assert_rel_l2(name + " vs fp64", _mx_out(y), _host_conv(x, w, b, relu, pool), TOL_LAYER)
    # against the emulated arithmetic: the fp16 part of every output is the fp16 rounding of the emulated
    # value (exact except where fp32 accumulation noise crosses a rounding boundary), and hi + q6(lo)
    # carries the rest to ~2^-15 of its group
    emu = mx_emul.conv3x3(x, w, b, relu, pool)
    got_hi = _mx_out(y, 1).double()
    want_hi = emu.float().half().double()
    same = (got_hi == want_hi).double().mean().item()
    # (an fp16 ulp of the element, plus the fp32 accumulation noise of K products of the layer's magnitude)
    ulp = torch.maximum(want_hi.abs(), torch.tensor(2.0 ** -14, dtype=torch.float64)) * 2.0 ** -10
    noise = 4e-6 * emu.abs().max()
    assert same > 0.97 and ((got_hi - want_hi).abs() <= ulp + noise).all(), f"{name}: fp16 parts equal in {same:.4f}"
    # tight check of the contraction itself, free of the output representation: hi + lo6 is within
    # 2^-15 of the group maximum of the kernel's fp32 result, so compare through that bound
    gmax = emu.abs().amax(1, keepdim=True)
    assert ((_mx_out(y).double() - emu).abs() <= 2.0 ** -14 * gmax + 1e-6 * emu.abs().max()).all()
    print(f"{name}: fp16 parts equal to the emulation in {same:.5f} of the outputs")


def test_conv3x3_mx_contraction_is_tight(dev):
    """The contraction against its fp64 emulation, read through the fp32 output route (the layer that
    feeds the head): 512 -> 512 at 30 x 40 inside the backbone is covered by the descriptor tests; here
    a ReLU-free layer's output magnitudes make hi + q6(lo) accurate enough to bound the device sum."""
    x, w, b = _case(2, 16, 24, 256, 256, seed=11)
    wp = ops.pack_conv3x3(w.to(dev), "f16mx")
    y = ops.conv3x3_nhwc(_mx_in(x, dev), wp, b.to(dev), False, False, "f16mx")
    emu = mx_emul.conv3x3(x, w, b, False, False)
    ref = _host_conv(x, w, b, False, False)
    print(f"f16mx layer: device vs emulation {rel_l2(_mx_out(y), emu):.3e}, emulation vs fp64 {rel_l2(emu, ref):.3e}, "
          f"device vs fp64 {rel_l2(_mx_out(y), ref):.3e}")
    assert rel_l2(emu, ref) < 3e-5
    assert rel_l2(_mx_out(y), emu) < 2e-5          # dominated by the output's q6(lo) representation


@pytest.fixture(scope="module")
def model(state_dict, dev):
    import hubconf
    m = hubconf.vgg16_netvlad(pretrained=False)
    m.load_state_dict(state_dict)
    m = m.to(dev).eval().set_precision("f16mx")
    m.base_model.F16MX_MIN_TILES = 0     # these tests are about the f16mx kernels: no small-batch switch to bf16x3
    return m


def _assert_desc(name, got, want):
    """rel-L2 over the batch AND, per image, max |diff| against the largest entry (north_star: 1e-4 relative)."""
    got, want = torch.as_tensor(got).double(), torch.as_tensor(want).double()
    assert_rel_l2(name, got, want, TOL_DESC)
    worst = ((got - want).abs().amax(1) / want.abs().amax(1)).max().item()
    print(f"{name}: worst image max|diff| / max|want| = {worst:.3e}")
    assert worst <= TOL_DESC, f"{name}: per-image max-abs criterion {worst:.3e} > {TOL_DESC:g}"


@pytest.mark.parametrize("name", ["desc_small", "desc_odd", "desc_480x640"])
def test_embednetpca_mx_matches_reference(name, model, dev):
    """All three vectors the reference produced: descriptor and every stage within 1e-4."""
    g = load_golden(name)
    n, _, h, w = [int(v) for v in g["shape"]]
    x = synth.images(n, h, w, seed=int(g["image_seed"])).to(dev)
    desc = model(x)
    assert tuple(desc.shape) == (n, 4096) and desc.dtype == torch.float32
    _assert_desc(f"{name} desc (f16mx)", desc.cpu(), g["desc"])
    pool_x, feat = model.base_model(x)
    s = int(g["feat_stride"])
    assert_rel_l2(f"{name} feat (f16mx)", feat.cpu()[:, ::s], g["feat"], TOL_DESC)
    assert_rel_l2(f"{name} pool_x (f16mx)", pool_x.cpu(), g["pool_x"], TOL_DESC)
    from ibl import models
    emb = models.create("embednet", model.base_model, model.net_vlad).eval().set_precision("f16mx")
    emb.base_model.F16MX_MIN_TILES = 0
    _, vlad = emb(x)
    _assert_desc(f"{name} vlad_norm (f16mx)", vlad.cpu(), g["vlad_norm"])
    from ibl.evaluators import extract_cnn_feature
    _assert_desc(f"{name} ecf pca (f16mx)", extract_cnn_feature(model, x.cpu()).cpu(), g["ecf_pca"])


def test_embednetpca_mx_vs_fp64_oracle(model, dev, state_dict):
    x = synth.images(2, 80, 112, seed=77)
    want = od.embednetpca(x, state_dict, dtype=torch.float64)
    got = model(x.to(dev)).cpu()
    _assert_desc("desc f16mx vs fp64 oracle", got, want)
    model.set_precision("bf16x3")
    try:
        ref3 = model(x.to(dev)).cpu()
    finally:
        model.set_precision("f16mx")
    print(f"f16mx vs fp64: {rel_l2(got, want):.3e}; bf16x3 vs fp64: {rel_l2(ref3, want):.3e}")


def test_mx_batch_rows_independent_and_graphed(model, dev):
    """An image's descriptor does not depend on its batch mates; hipGraph replay reproduces the eager
    forward bit for bit."""
    x = synth.images(2, 64, 96, seed=21).to(dev)     # (1 and 2 images select the same kernels everywhere)
    want = model(x).clone()
    for i in range(2):
        assert torch.equal(model(x[i:i + 1].contiguous())[0], want[i])
    pf = model.graphed(x, pipeline=True)
    a, b = pf(), pf()
    pf.wait()
    torch.cuda.synchronize()
    assert torch.equal(a, want) and torch.equal(b, want)


@pytest.mark.parametrize("N,H,W", [(1, 8, 32), (2, 16, 64), (1, 21, 45), (3, 30, 70), (1, 2, 3), (2, 3, 4),
                                   (2, 9, 33), (1, 64, 96), (5, 40, 136), (1, 17, 65)])
def test_vgg_stem_mx_fused(dev, N, H, W):
    """f16mx: conv1_1 (split bf16) + conv1_2 (f16mx arithmetic on f16mx lines packed inside LDS) + pool in one
    launch, output = f16mx lines packed from registers.  Against the fp64 host convolutions (one f16mx
    contraction + two line roundings), and the stored lines against the format: every line must be what
    mx_split makes of its own fp16 parts (slots, element order, block scale, the e2m3 image of hi), with
    |q6(lo)| inside half an fp16 ulp of the group's largest element."""
    x, w1, b1 = _case(N, H, W, 3, 64, seed=5 * H + W)
    x = x * 60.0
    _, w2, b2 = _case(1, 4, 4, 64, 64, seed=H + 9 * W)
    wp2 = ops.pack_conv3x3(w2.to(dev), "f16mx")
    y = ops.vgg16_stem_mx(x.to(dev), w1.to(dev), b1.to(dev), wp2, b2.to(dev))
    assert tuple(y.shape) == (N, H // 2, W // 2, 64) and y.dtype == torch.int32
    h1 = F.relu(F.conv2d(x.double(), w1.double(), b1.double(), padding=1))
    want = F.max_pool2d(F.relu(F.conv2d(h1, w2.double(), b2.double(), padding=1)), 2, 2)
    if not want.numel():
        return
    assert_rel_l2("fused f16mx stem vs host", _mx_out(y), want, 6e-5)
    hi = ops.mx_join(y, 1)
    again = ops.mx_split(hi)
    assert torch.equal(ops.mx_join(again, 1), hi)
    assert torch.equal(ops.mx_join(again, 2), ops.mx_join(y, 2))
    gmax = hi.abs().reshape(-1, 32).amax(-1, keepdim=True)
    lo6 = ops.mx_join(y, 3).reshape(-1, 32)
    assert (lo6.abs() <= gmax * 2.0 ** -11 * 1.07 + 1e-30).all()
    # the value a line carries is within ~2^-15 of its group's largest element of the true one
    err = (_mx_out(y).double() - want).abs().reshape(N, 2, 32, -1).amax(2)
    ref = want.abs().reshape(N, 2, 32, -1).amax(2)
    assert (err <= 3e-4 * ref + 1e-6).all(), float((err / (ref + 1e-9)).max())


@pytest.mark.parametrize("N,H,W", [(1, 8, 32), (2, 17, 65), (1, 64, 96)])
def test_vgg_stem_mx_matches_the_emulation_of_its_arithmetic(dev, N, H, W):
    """The stem against the fp64 emulation of ITS OWN arithmetic (tests/helpers/mx_emul.stem: split-bf16 conv1_1,
    half-line packs, f16mx conv1_2): what is left is fp32 accumulation and the output codes it tips."""
    x, w1, b1 = _case(N, H, W, 3, 64, seed=5 * H + W)
    x = x * 60.0
    _, w2, b2 = _case(1, 4, 4, 64, 64, seed=H + 9 * W)
    y = ops.vgg16_stem_mx(x.to(dev), w1.to(dev), b1.to(dev), ops.pack_conv3x3(w2.to(dev), "f16mx"), b2.to(dev))
    want = mx_emul.stem(x, w1, b1, w2, b2)
    assert_rel_l2("fused f16mx stem vs its emulation", _mx_out(y), want, TOL_STEM_EMUL)
    # (informative) fp16 parts: equal except where fp32 sums — or an input code they tipped — straddle an fp16
    # rounding boundary
    hi_got = _mx_out(y, 1).double()
    hi_want = want.float().half().double()      # fp16 part of the emulated carried value (lo6 < half an ulp)
    print(f"fp16 parts equal: {(hi_got == hi_want).double().mean().item():.4%}")


# ---- matching -------------------------------------------------------------------------------------
from oracle import matching as om  # noqa: E402


@pytest.mark.parametrize("m,n,d", [(5, 7, 64), (130, 300, 128), (256, 512, 4096), (300, 1100, 4096), (64, 2000, 192)])
def test_pairwise_mx(dev, m, n, d):
    g = torch.Generator().manual_seed(m + n)
    x = F.normalize(torch.randn((m, d), generator=g), dim=1)
    y = F.normalize(torch.randn((n, d), generator=g), dim=1)
    got = ops.pairwise_sqdist(x.to(dev), y.to(dev), "f16mx").cpu()
    want = (x.double().pow(2).sum(1)[:, None] + y.double().pow(2).sum(1)[None, :] - 2 * x.double() @ y.double().T)
    err = (got.double() - want).abs().max().item()
    # the emulated arithmetic (d >= 128: f16mx rows; d = 64 is served in bf16x3)
    if d >= 128:
        emu = (x.double().pow(2).sum(1)[:, None] + y.double().pow(2).sum(1)[None, :] - 2 * mx_emul.matmul_nt(x, y))
        e2 = (got.double() - emu).abs().max().item()
        print(f"pairwise f16mx {m}x{n}x{d}: max abs err vs fp64 {err:.3e}, vs the emulated product {e2:.3e}")
        assert e2 < 2e-6
    assert err < 2e-5      # unit vectors: distances in [0, 4]


def test_matching_mx_equals_reference_ranking(dev):
    """Golden matching problem of the reference: distances within 2e-5, identical top-20 ranks and recalls."""
    for name in ("match_small", "match_nms"):
        g = load_golden(name)
        q, gal, gt, pids = synth.retrieval_problem(
            int(g["Q"]), int(g["G"]), dim=int(g["dim"]), seed=int(g["seed"]),
            views_per_place=int(g["views_per_place"]), hard_fraction=float(g["hard_fraction"]),
            hard_noise_mult=float(g["hard_noise_mult"]))
        d = ops.pairwise_sqdist(q.to(dev), gal.to(dev), "f16mx")
        err = np.abs(d.cpu().numpy().astype(np.float64) - g["distmat"]).max()
        print(f"{name}: f16mx max |dist - reference| = {err:.3e}")
        assert err < 2e-5
        _, idx = ops.row_topk(d, 20)
        assert np.array_equal(idx.cpu().numpy(), g["top20"])
        from ibl.evaluators import evaluate_all
        gallery = [(f"g{j:05d}.jpg", pids[j], 0.0, 0.0) for j in range(len(gal))]
        np.testing.assert_array_equal(evaluate_all(d.cpu(), gt, gallery), g["recalls"])
        np.testing.assert_array_equal(evaluate_all(d.cpu(), gt, gallery, nms=True), g["recalls_nms"])


def test_sqdist_topk_mx_fused_equals_matrix(dev):
    """Fused threshold / filter / select path in f16mx == top-k of its own matrix (16k gallery), prepared
    operands == unprepared, and the selected entries are the true nearest ones."""
    q, gal, gt, _ = synth.retrieval_problem(512, 16384, seed=9)
    qd, gd = q.to(dev), gal.to(dev)
    v, i = ops.sqdist_topk(qd, gd, 10, precision="f16mx")
    dm = ops.pairwise_sqdist(qd, gd, "f16mx")
    v2, i2 = ops.row_topk(dm, 10)
    assert torch.equal(i, i2) and torch.equal(v, v2)
    v3, i3 = ops.sqdist_topk_prepared(ops.PreparedRows(qd, "f16mx"), ops.PreparedRows(gd, "f16mx"), 10)
    assert torch.equal(i3, i) and torch.equal(v3, v)
    d64 = (q.double().pow(2).sum(1)[:, None] + gal.double().pow(2).sum(1)[None, :]
           - 2.0 * q.double() @ gal.double().T)
    true_v = torch.sort(d64, dim=1).values[:, :10]
    got_v = torch.gather(d64, 1, i.cpu().long())
    assert (got_v - true_v).abs().max().item() < 1e-5
    want = om.ranking(om.pairwise_distance(q, gal).numpy())[:, :10]
    agree = (i.cpu().numpy() == want).mean()
    print(f"top-10 agreement with the fp32 oracle ranking: {agree:.6f}")
    assert agree > 0.999


def test_conv_mx_repeatable_under_load(dev):
    """Ring and halo kernels give the same bits launch after launch, also while a second stream keeps the
    memory system busy (the halo kernel's first wait counts did not: conv_halo.h)."""
    from openibl_amd import lib
    g = torch.Generator(device=dev).manual_seed(5)
    xf = torch.relu(torch.randn((8, 120, 160, 256), generator=g, device=dev)) * 3.0
    w = torch.randn((256, 256, 3, 3), generator=g, device=dev) * 0.03
    b = torch.randn((256,), generator=g, device=dev) * 0.1
    x, wp = ops.mx_split(xf), ops.pack_conv3x3(w, "f16mx")
    big = torch.randn((4096, 4096), device=dev)
    side = torch.cuda.Stream()
    for variant in (1, 3):
        lib.debug_hooks().oibl_debug_set_mx_variant(variant)
        try:
            ref = ops.conv3x3_nhwc(x, wp, b, True, True, "f16mx")
            for _ in range(25):
                with torch.cuda.stream(side):
                    big @ big
                assert torch.equal(ops.conv3x3_nhwc(x, wp, b, True, True, "f16mx"), ref)
            torch.cuda.synchronize()
        finally:
            lib.debug_hooks().oibl_debug_set_mx_variant(0)


def test_conv_mx_halo4_repeatable_under_load(dev):
    """The 4-wave halo kernel of the 128-output-channel layers (conv_halo4.h): its single halo buffer is reloaded
    between chunks behind a drained queue — the same bits launch after launch, also while a second stream loads the
    memory system, pooled and unpooled, and its output agrees with the ring kernel's to the arithmetic's accuracy."""
    from openibl_amd import lib
    g = torch.Generator(device=dev).manual_seed(6)
    big = torch.randn((4096, 4096), device=dev)
    side = torch.cuda.Stream()
    for cin, pool in ((64, False), (128, True)):
        xf = torch.relu(torch.randn((4, 240, 320, cin), generator=g, device=dev)) * 3.0
        w = torch.randn((128, cin, 3, 3), generator=g, device=dev) * 0.05
        b = torch.randn((128,), generator=g, device=dev) * 0.1
        x, wp = ops.mx_split(xf), ops.pack_conv3x3(w, "f16mx")
        ref = ops.conv3x3_nhwc(x, wp, b, True, pool, "f16mx")          # default dispatch: conv_halo4.h
        for _ in range(25):
            with torch.cuda.stream(side):
                big @ big
            assert torch.equal(ops.conv3x3_nhwc(x, wp, b, True, pool, "f16mx"), ref)
        torch.cuda.synchronize()
        lib.debug_hooks().oibl_debug_set_mx_variant(1)
        try:
            ring = ops.conv3x3_nhwc(x, wp, b, True, pool, "f16mx")
        finally:
            lib.debug_hooks().oibl_debug_set_mx_variant(0)
        d = float((ops.mx_join(ring) - ops.mx_join(ref)).norm() / ops.mx_join(ring).norm())
        assert d < 3e-5, d


def test_small_batch_threshold_is_a_knob_and_off_by_default(dev, state_dict):
    """Rounds 1-3 served small f16mx batches in bf16x3 (F16MX_MIN_TILES = 256); the ring kernels now split K
    for them (tests/test_gpu_splitk.py) and the threshold is 16: only problems below 8 tiles of conv4 pixels,
    where bf16x3 is the faster 1e-4 mode.  Set higher, it works as before — same bits as an explicit bf16x3
    model — and what actually ran is visible through effective_precision() / precision_runs."""
    import hubconf
    m = hubconf.vgg16_netvlad(pretrained=False)
    m.load_state_dict(state_dict)
    m = m.to(dev).eval().set_precision("f16mx")
    one = synth.images(1, 480, 640, seed=3).to(dev)
    many = torch.empty((8, 3, 480, 640), device=dev)
    assert m.base_model.effective_precision(one) == "f16mx"                      # 19 tiles of 256 pixels
    small = synth.images(1, 224, 224, seed=4).to(dev)                                # 4 tiles
    assert m.base_model.effective_precision(small) == "bf16x3"
    assert m.base_model.effective_precision(torch.empty((4, 3, 224, 224), device="meta")) == "f16mx"   # 13
    m.base_model.F16MX_MIN_TILES = 256
    assert m.base_model.effective_precision(one) == "bf16x3"
    assert m.base_model.effective_precision(many) == "f16mx"
    got = m(one).clone()
    assert m.base_model.precision_runs == {"bf16x3": 1}
    m.set_precision("bf16x3")
    assert torch.equal(m(one), got)
    # 96 images of 480x640 and more are beyond the 32-bit offsets of the f16mx kernels (conv2_2's input): f16mx in
    # image groups since round 6 (tests/test_gpu_api.py runs 128), bf16x3 before
    m.set_precision("f16mx")
    m.base_model.F16MX_MIN_TILES = 0
    assert m.base_model.effective_precision(torch.empty((94, 3, 480, 640), device="meta")) == "f16mx"
    assert m.base_model.effective_precision(torch.empty((96, 3, 480, 640), device="meta")) == "f16mx"
    assert m.base_model.f16mx_groups(torch.empty((96, 3, 480, 640), device="meta")) == [(0, 48), (48, 48)]
    assert m.base_model.f16mx_groups(torch.empty((32, 3, 960, 1280), device="meta")) == [(0, 16), (16, 16)]
