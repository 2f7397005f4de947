"""NetVLAD aggregation and PCA projection kernels against the oracle (GPU)."""
import pytest
import torch

from conftest import assert_rel_l2, load_golden
from openibl_amd import ops, synth
from oracle import descriptor as od

pytestmark = pytest.mark.gpu


def _feat(N, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    # signed, non-unit-norm positions like a conv5_3 map
    return torch.randn((N, 512, h, w), generator=g) * (1.0 + 3.0 * torch.rand((N, 1, h, w), generator=g))


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("N,h,w", [(1, 30, 40), (2, 4, 6), (3, 5, 7), (1, 1, 1), (2, 13, 11)])
@pytest.mark.parametrize("normalize_input", [True, False])
def test_netvlad(dev, N, h, w, precision, normalize_input):
    sd = synth.netvlad_state(0)
    cw, cent = sd["net_vlad.conv.weight"], sd["net_vlad.centroids"]
    feat = _feat(N, h, w, seed=h * 100 + w)
    if not normalize_input:
        feat = feat * 0.05          # keep the logits in a sane range without the L2 norm
    fd = ops.nchw_f32_to_nhwc(feat.to(dev), precision)
    raw, nrm = ops.netvlad(fd, cw.reshape(64, 512).contiguous().to(dev), cent.to(dev),
                           normalize_input, want_raw=True, want_norm=True)
    fin = feat if precision == "fp32" else feat.to(torch.bfloat16).float()
    cw_in = cw if precision == "fp32" else cw.to(torch.bfloat16).float()
    want_raw = od.netvlad(fin.double(), cw_in.double(), cent.double(), normalize_input)
    want_nrm = od.normalize_vlad(want_raw)
    tol = 5e-6 if precision == "fp32" else 2e-5   # bf16 inputs are mirrored on the host side
    assert_rel_l2(f"vlad_raw {precision}", raw.cpu(), want_raw, tol)
    assert_rel_l2(f"vlad_norm {precision}", nrm.cpu(), want_nrm, tol)
    rn = nrm.cpu().double().norm(dim=1)
    assert torch.allclose(rn, torch.ones_like(rn), atol=1e-5)


@pytest.mark.parametrize("tag", ["whiten", "nowhiten"])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_pca_golden(dev, tag, precision):
    g = load_golden("pca")
    w = torch.from_numpy(g[f"weight_{tag}"]).contiguous().to(dev)
    b = torch.from_numpy(g[f"bias_{tag}"]).to(dev)
    v = torch.from_numpy(g["data"]).to(dev)
    out = ops.pca(v, ops.cast(w, precision), b)
    assert_rel_l2(f"pca {tag} {precision}", out.cpu(), g[f"out_{tag}"], 1e-5 if precision == "fp32" else 1e-2)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("N", [1, 5, 32, 33])
def test_pca_full_size(dev, N, precision):
    sd = synth.pca_state(0)
    w = sd["pca_layer.weight"].reshape(4096, 32768)
    b = sd["pca_layer.bias"]
    g = torch.Generator().manual_seed(N)
    v = torch.nn.functional.normalize(torch.randn((N, 32768), generator=g), dim=1)
    out = ops.pca(v.to(dev), ops.cast(w.to(dev), precision), b.to(dev))
    if precision == "bf16":
        want = od.pca_project(v.to(torch.bfloat16).double(), w.to(torch.bfloat16).double(), b.double())
    else:
        want = od.pca_project(v.double(), w.double(), b.double())
    assert_rel_l2(f"pca 32768->4096 N={N} {precision}", out.cpu(), want, 5e-6)
    raw = ops.pca(v.to(dev), ops.cast(w.to(dev), precision), b.to(dev), l2norm=False)
    assert_rel_l2("pca without normalize", torch.nn.functional.normalize(raw.cpu().double(), dim=1),
                  want, 5e-6)


def test_pca_few_rows_streaming_kernel(dev):
    """N <= 2 in fp32 runs the streaming (matrix-vector) kernel: within the tolerance of the MFMA tile it
    replaces (hook off); a row's result does not depend on whether a second row travels with it."""
    from openibl_amd import lib
    sd = synth.pca_state(0)
    w = sd["pca_layer.weight"].reshape(4096, 32768).to(dev)
    b = sd["pca_layer.bias"].to(dev)
    g = torch.Generator().manual_seed(77)
    v = torch.nn.functional.normalize(torch.randn((8, 32768), generator=g), dim=1).to(dev)
    want = od.pca_project(v.cpu().double(), w.cpu().double(), b.cpu().double())
    outs = {n: ops.pca(v[:n].contiguous(), w, b) for n in (1, 2, 3, 8)}
    for n, o in outs.items():
        assert_rel_l2(f"pca N={n}", o.cpu(), want[:n], 5e-6)
    assert torch.equal(outs[1], outs[2][:1]), "row 0 of N=2 differs from N=1"
    lib.debug_hooks().oibl_debug_set_pca_small(0)
    try:
        tile = ops.pca(v[:2].contiguous(), w, b)
    finally:
        lib.debug_hooks().oibl_debug_set_pca_small(1)
    assert not torch.equal(outs[2], tile)          # (the hook did select the other kernel)
    assert_rel_l2("streaming kernel vs MFMA tile", outs[2].cpu(), tile.cpu(), 2e-6)


def test_pca_packed_weight_streaming_kernel(dev):
    """3 .. 32 rows in fp32 through an `ops.PcaWeight` stream the re-packed weight (oibl_pca_forward_packed):
    the oracle's numbers, the row-major tile's up to the association of the fp32 sums, a row's result independent
    of its batch mates, repeatable; 33 rows and bf16 weights keep the tile; a smaller shape the C entry serves."""
    from openibl_amd import lib
    sd = synth.pca_state(0)
    w = sd["pca_layer.weight"].reshape(4096, 32768).to(dev)
    b = sd["pca_layer.bias"].to(dev)
    g = torch.Generator().manual_seed(78)
    v = torch.nn.functional.normalize(torch.randn((33, 32768), generator=g), dim=1).to(dev)
    want = od.pca_project(v.cpu().double(), w.cpu().double(), b.cpu().double())
    pw = ops.PcaWeight(w)
    assert pw._packed is None
    outs = {n: ops.pca(v[:n].contiguous(), pw, b) for n in (2, 3, 7, 32, 33)}
    assert pw._packed is not None and pw._packed.shape == w.shape
    for n, o in outs.items():
        assert_rel_l2(f"pca packed N={n}", o.cpu(), want[:n], 5e-6)
    tile = {n: ops.pca(v[:n].contiguous(), w, b) for n in (2, 3, 7, 32, 33)}
    for n in (2, 33):
        assert torch.equal(outs[n], tile[n])                       # outside 3 .. 32 the holder changes nothing
    for n in (3, 7, 32):
        assert not torch.equal(outs[n], tile[n])                   # (the other kernel did run)
        assert_rel_l2(f"packed vs tile N={n}", outs[n].cpu(), tile[n].cpu(), 2e-6)
    assert torch.equal(outs[3], outs[32][:3]) and torch.equal(outs[7], outs[32][:7])
    assert torch.equal(ops.pca(v[:32].contiguous(), pw, b), outs[32])
    raw = ops.pca(v[:7].contiguous(), pw, b, l2norm=False)
    assert_rel_l2("packed, without normalize", torch.nn.functional.normalize(raw.cpu().double(), dim=1), want[:7], 5e-6)
    # the 8-wave instantiations (round 5's: 8 loads in flight x 4 waves per SIMD, and 16 x 2): 32 partials summed one
    # by one where the default adds neighbouring K parts in LDS first — the same sums as each other, bit for bit, and
    # the default's to rounding
    lib.debug_hooks().oibl_debug_set_pca_stream(2)
    try:
        deep = ops.pca(v[:32].contiguous(), pw, b)
        lib.debug_hooks().oibl_debug_set_pca_stream(3)
        assert torch.equal(ops.pca(v[:32].contiguous(), pw, b), deep)
        assert not torch.equal(deep, outs[32])
        assert_rel_l2("32 partials vs 16", deep.cpu(), outs[32].cpu(), 1e-6)
        lib.debug_hooks().oibl_debug_set_pca_stream(0)
        assert torch.equal(ops.pca(v[:7].contiguous(), pw, b), tile[7])    # hook 0: the packed form is not offered
    finally:
        lib.debug_hooks().oibl_debug_set_pca_stream(1)
    pb = ops.PcaWeight(ops.cast(w, "bf16"))
    assert torch.equal(ops.pca(v[:7].contiguous(), pb, b), ops.pca(v[:7].contiguous(), pb.rows, b)) and pb._packed is None
    # the smallest shape of the packed form: D = 8192, d = 256
    w2 = torch.randn((256, 8192), generator=g).to(dev) * 0.01
    b2 = torch.randn(256, generator=g).to(dev) * 0.1
    v2 = torch.randn((5, 8192), generator=g).to(dev)
    got = ops.pca(v2, ops.PcaWeight(w2), b2)
    assert_rel_l2("packed 8192 -> 256", got.cpu(), od.pca_project(v2.cpu().double(), w2.cpu().double(), b2.cpu().double()), 5e-6)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_netvlad_pixel_slabs_for_few_images(dev, precision):
    """Five launches (bf16 maps; fp32 maps of up to four images): up to four images the aggregation is split over the pixels (4 slabs, added
    in fixed order by the normalising kernel).  Same numbers as the unsplit kernel up to the summation order, raw
    output included; an image's result does not depend on its batch mates; five images run unsplit."""
    from openibl_amd import lib
    sd = synth.netvlad_state(0)
    cw, cent = sd["net_vlad.conv.weight"].reshape(64, 512).contiguous().to(dev), sd["net_vlad.centroids"].to(dev)
    feat = ops.nchw_f32_to_nhwc(_feat(5, 30, 37, seed=11).to(dev), precision)      # P = 1110: a ragged last slab
    outs = {n: ops.netvlad(feat[:n].contiguous(), cw, cent, True, want_raw=True, want_norm=True) for n in (1, 2, 4, 5)}
    lib.debug_hooks().oibl_debug_set_netvlad_slabs(0)
    try:
        whole = {n: ops.netvlad(feat[:n].contiguous(), cw, cent, True, want_raw=True, want_norm=True) for n in (4, 5)}
    finally:
        lib.debug_hooks().oibl_debug_set_netvlad_slabs(1)
    assert_rel_l2("slabs vs whole, raw", outs[4][0].cpu(), whole[4][0].cpu(), 2e-6)
    assert_rel_l2("slabs vs whole, normalised", outs[4][1].cpu(), whole[4][1].cpu(), 2e-6)
    assert not torch.equal(outs[4][0], whole[4][0])
    assert torch.equal(outs[5][0], whole[5][0]) and torch.equal(outs[5][1], whole[5][1])   # (five images: unsplit)
    for n in (1, 2):
        assert torch.equal(outs[n][0], outs[4][0][:n]) and torch.equal(outs[n][1], outs[4][1][:n])
    only_norm = ops.netvlad(feat[:2].contiguous(), cw, cent, True, want_raw=False, want_norm=True)[1]
    assert torch.equal(only_norm, outs[2][1])
    only_raw = ops.netvlad(feat[:2].contiguous(), cw, cent, True, want_raw=True, want_norm=False)[0]
    assert_rel_l2("raw only (unsplit)", only_raw.cpu(), outs[2][0].cpu(), 2e-6)


@pytest.mark.parametrize("normalize_input", [True, False])
def test_netvlad_fused_kernel_against_the_five_launch_path(dev, normalize_input):
    """fp32 feature maps of 16 images and more take the fused layer (netvlad_fused_kernel: norm + soft-assignment +
    softmax + aggregation in one kernel, the map read once; then the slab sum and the two normalising launches).
    Against the five-launch path it replaces (hooks 0 / 2) the sums differ by association only; a row's result does
    not depend on its batch mates; raw-only and normalised-only calls give the same bits as the call that asks for
    both; a map of at most 160 pixels is one slab."""
    from openibl_amd import lib
    sd = synth.netvlad_state(0)
    cw, cent = sd["net_vlad.conv.weight"].reshape(64, 512).contiguous().to(dev), sd["net_vlad.centroids"].to(dev)
    for (h, w_) in ((30, 37), (9, 11)):                                              # P = 1110: ragged last slab; P = 99: one slab
        f = _feat(20, h, w_, seed=12 + h)
        if not normalize_input:
            f = f * 0.05
        feat = ops.nchw_f32_to_nhwc(f.to(dev), "fp32")
        outs = {n: ops.netvlad(feat[:n].contiguous(), cw, cent, normalize_input, want_raw=True, want_norm=True)
                for n in (16, 18, 20)}
        for hook in (0, 2):
            lib.debug_hooks().oibl_debug_set_netvlad_slabs(hook)
            try:
                old = ops.netvlad(feat[:18].contiguous(), cw, cent, normalize_input, want_raw=True, want_norm=True)
            finally:
                lib.debug_hooks().oibl_debug_set_netvlad_slabs(1)
            assert_rel_l2(f"fused vs five launches (hook {hook}), raw", outs[18][0].cpu(), old[0].cpu(), 3e-6)
            assert_rel_l2(f"fused vs five launches (hook {hook}), normalised", outs[18][1].cpu(), old[1].cpu(), 3e-6)
            assert not torch.equal(outs[18][0], old[0])                               # (the hook did select the other path)
        assert torch.equal(outs[16][0], outs[20][0][:16]) and torch.equal(outs[16][1], outs[20][1][:16])
        assert torch.equal(outs[18][0], outs[20][0][:18]) and torch.equal(outs[18][1], outs[20][1][:18])
        only_norm = ops.netvlad(feat[:16].contiguous(), cw, cent, normalize_input, want_raw=False, want_norm=True)[1]
        only_raw = ops.netvlad(feat[:16].contiguous(), cw, cent, normalize_input, want_raw=True, want_norm=False)[0]
        assert torch.equal(only_norm, outs[16][1]) and torch.equal(only_raw, outs[16][0])
        want = od.netvlad(f.double(), sd["net_vlad.conv.weight"].double(), sd["net_vlad.centroids"].double(), normalize_input)
        assert_rel_l2("fused raw vs the fp64 oracle", outs[20][0].cpu(), want, 5e-6)
        assert_rel_l2("fused normalised vs the fp64 oracle", outs[20][1].cpu(), od.normalize_vlad(want), 5e-6)
        # round 6: the fused kernel at ANY batch size (hook 3: slabs of 32 / 64 / 96 pixels below 4 / 8 / 16 images —
        # measured slower than the five launches there, so not the default): the same layer
        lib.debug_hooks().oibl_debug_set_netvlad_slabs(3)
        try:
            small = {n: ops.netvlad(feat[:n].contiguous(), cw, cent, normalize_input, want_raw=True, want_norm=True)
                     for n in (1, 3, 5, 9)}
        finally:
            lib.debug_hooks().oibl_debug_set_netvlad_slabs(1)
        for n, (raw, nrm) in small.items():
            assert_rel_l2(f"fused at N = {n}, raw vs the fp64 oracle", raw.cpu(), want[:n], 5e-6)
            assert_rel_l2(f"fused at N = {n}, normalised", nrm.cpu(), od.normalize_vlad(want[:n]), 5e-6)
        assert torch.equal(small[1][0], small[3][0][:1])                     # same slab size: rows independent of their mates
