"""Image -> 4096-d descriptor through the reference's API surface on the HIP path, against the
vectors the reference itself produced (tests/golden) and against the oracle (GPU)."""
import numpy as np
import pytest
import torch

from conftest import assert_desc, assert_rel_l2, load_golden, rel_l2
from openibl_amd import ops, synth
from oracle import descriptor as od

pytestmark = pytest.mark.gpu

# north_star: descriptors within 1e-4 relative of the reference CPU path (fp32 mode)
TOL_FP32 = 1e-4


@pytest.fixture(scope="module")
def model(state_dict, dev):
    import hubconf
    m = hubconf.vgg16_netvlad(pretrained=False)
    m.load_state_dict(state_dict)
    return m.to(dev).eval()


@pytest.mark.parametrize("name", ["desc_small", "desc_odd", "desc_480x640"])
def test_embednetpca_fp32_matches_reference(name, model, dev):
    g = load_golden(name)
    n, _, h, w = [int(v) for v in g["shape"]]
    x = synth.images(n, h, w, seed=int(g["image_seed"])).to(dev)
    model.set_precision("fp32")
    desc = model(x)
    assert tuple(desc.shape) == (n, 4096) and desc.dtype == torch.float32
    assert_desc(f"{name} desc", desc.cpu(), g["desc"], TOL_FP32)
    # stage by stage
    pool_x, feat = model.base_model(x)
    s = int(g["feat_stride"])
    assert_rel_l2(f"{name} feat", feat.cpu()[:, ::s], g["feat"], TOL_FP32)
    assert_rel_l2(f"{name} pool_x", pool_x.cpu(), g["pool_x"], TOL_FP32)
    vlad_raw = model.net_vlad(feat)
    assert tuple(vlad_raw.shape) == (n, 64, 512)
    assert_rel_l2(f"{name} vlad_raw", vlad_raw.cpu(), g["vlad_raw"], TOL_FP32)
    from ibl import models
    emb = models.create("embednet", model.base_model, model.net_vlad).eval()
    pool_e, vlad = emb(x)
    assert_rel_l2(f"{name} vlad_norm", vlad.cpu(), g["vlad_norm"], TOL_FP32)
    assert_rel_l2(f"{name} pool_e", pool_e.cpu(), g["pool_x"], TOL_FP32)
    from ibl.evaluators import extract_cnn_feature
    assert_rel_l2(f"{name} ecf pca", extract_cnn_feature(model, x.cpu()).cpu(), g["ecf_pca"], TOL_FP32)
    assert_rel_l2(f"{name} ecf vlad", extract_cnn_feature(emb, x.cpu(), vlad=True).cpu(),
                  g["ecf_vlad"], TOL_FP32)
    assert_rel_l2(f"{name} ecf pool", extract_cnn_feature(emb, x.cpu(), vlad=False).cpu(),
                  g["ecf_pool"], TOL_FP32)


@pytest.mark.parametrize("precision", ["fp32", "f16mx", "bf16x3"])
def test_batch_of_8_at_480x640_against_the_reference_itself(precision, model, dev):
    """A batch > 2 at BASELINE configs[1]'s image size pinned to THE REFERENCE (tests/golden/desc_480x640_n8.npz: 8
    images through hubconf.vgg16_netvlad of /root/reference, oracle/make_golden.py) — not to the oracle port: the
    descriptor and every stage the fixture holds within north_star's 1e-4, per image, in the exact mode and in both
    1e-4 matrix-core arithmetics (round 6)."""
    g = load_golden("desc_480x640_n8")
    n, _, h, w = [int(v) for v in g["shape"]]
    assert (n, h, w) == (8, 480, 640)
    x = synth.images(n, h, w, seed=int(g["image_seed"])).to(dev)
    model.set_precision(precision)
    try:
        assert model.base_model.effective_precision(x) == precision        # 8 images: no small-problem substitution
        desc = model(x)
        pool_x, feat = model.base_model(x)
        from ibl import models
        emb = models.create("embednet", model.base_model, model.net_vlad).eval().set_precision(precision)
        _, vlad = emb(x)
        from ibl.evaluators import extract_cnn_feature
        ecf = extract_cnn_feature(model, x.cpu()).cpu()
    finally:
        model.set_precision("fp32")
    assert model.base_model.range_fallbacks == 0
    assert_desc(f"n8 desc ({precision})", desc.cpu(), g["desc"], TOL_FP32)
    assert_desc(f"n8 ecf pca ({precision})", ecf, g["ecf_pca"], TOL_FP32)
    assert_desc(f"n8 vlad_norm ({precision})", vlad.cpu(), g["vlad_norm"], TOL_FP32)
    s = int(g["feat_stride"])
    assert_rel_l2(f"n8 feat ({precision})", feat.cpu()[:, ::s], g["feat"], TOL_FP32)
    assert_rel_l2(f"n8 pool_x ({precision})", pool_x.cpu(), g["pool_x"], TOL_FP32)


@pytest.mark.parametrize("name", ["desc_small", "desc_480x640"])
def test_embednetpca_bf16_reported_honestly(name, model, dev):
    """bf16 operands cannot meet 1e-4 (~5e-3 expected; measured 2.8e-3 at 480x640,
    3.8e-3 at 64x96); the test pins that band — a 2x regression fails — and the cosine, and the
    regstage / glds variants must agree bit for bit.  The 1e-4 mode at matrix-core speed is bf16x3
    (tests/test_gpu_x3.py)."""
    g = load_golden(name)
    n, _, h, w = [int(v) for v in g["shape"]]
    x = synth.images(n, h, w, seed=int(g["image_seed"])).to(dev)
    model.set_precision("bf16")
    try:
        desc = model(x).cpu()
        ops.set_regstage(True)
        desc_rs = model(x).cpu()
    finally:
        ops.set_regstage(False)
        model.set_precision("fp32")
    want = torch.from_numpy(g["desc"])
    err = rel_l2(desc, want)
    cos = torch.nn.functional.cosine_similarity(desc.double(), want.double(), dim=1).min().item()
    print(f"{name}: bf16 rel_l2={err:.3e} min cosine={cos:.6f}")
    assert err < 6e-3 and cos > 0.99999
    assert torch.equal(desc, desc_rs)


def test_batch_rows_are_independent(model, dev):
    """A batch gives the same descriptors as its images one at a time (and different images give
    different descriptors)."""
    x = synth.images(3, 64, 96, seed=31).to(dev)
    model.set_precision("fp32")
    full = model(x).cpu()
    for i in range(3):
        one = model(x[i:i + 1].contiguous()).cpu()
        assert_rel_l2(f"row {i}", one[0], full[i], 1e-6)
    assert (full[0] - full[1]).norm() > 1e-3


def test_state_dict_contract(model, state_dict):
    sd = model.state_dict()
    assert list(sd.keys()) == list(state_dict.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(state_dict[k].shape), k
    # DDP-style prefixed checkpoints load through copy_state_dict
    from ibl.utils.serialization import copy_state_dict
    import hubconf
    m2 = hubconf.vgg16_netvlad()
    copy_state_dict({"module." + k: v for k, v in state_dict.items()}, m2, strip="module.")
    for k, v in m2.state_dict().items():
        assert torch.equal(v, state_dict[k]), k


def test_weight_update_invalidates_packed_cache(model, dev, state_dict):
    x = synth.images(1, 64, 96, seed=11).to(dev)
    model.set_precision("fp32")
    a = model(x).cpu()
    key = "base_model.base.28.bias"
    with torch.no_grad():
        model.state_dict()[key].add_(0.5)
    b = model(x).cpu()
    model.load_state_dict(state_dict)
    c = model(x).cpu()
    assert (a - b).norm() > 1e-4
    assert torch.equal(a, c)


def test_graphed_forward_equals_eager(dev):
    """model.graphed(x): the two captured hipGraphs reproduce model(x) bit for bit, on the example
    batch and on a second batch copied into the static input."""
    import hubconf
    model = hubconf.vgg16_netvlad(pretrained=False)
    model.load_state_dict(synth.embednetpca_state(0))
    model = model.to(dev).eval().set_precision("bf16")
    x1 = synth.images(2, 64, 96, seed=21).to(dev)
    x2 = synth.images(2, 64, 96, seed=22).to(dev)
    want1, want2 = model(x1).clone(), model(x2).clone()
    fwd = model.graphed(x1)
    assert torch.equal(fwd(), want1)
    assert torch.equal(fwd(x2), want2)
    assert torch.equal(fwd(x1), want1)
    with pytest.raises(ValueError):
        fwd(synth.images(1, 64, 96, seed=1).to(dev))
    # pipelined: two lanes (call i on lane i % 2), each with its own buffers
    pf = model.graphed(x1, pipeline=True)
    outs = []
    for x, want in ((x1, want1), (x2, want2), (x1, want1), (x2, want2), (x2, want2)):
        o = pf(x)
        pf.wait()
        outs.append((o.clone(), want))
    torch.cuda.synchronize()
    for o, want in outs:
        assert torch.equal(o, want)
    a, b = pf(x1), pf(x2)          # two calls in flight: each slot keeps its own result
    pf.wait()
    torch.cuda.synchronize()
    assert torch.equal(a, want1) and torch.equal(b, want2)


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_two_lanes_in_flight_do_not_share_scratch(dev, precision):
    """The two lanes of the pipelined replay run concurrently on two streams: activation buffers and
    split-K partial sums (small batches run the deep layers split-K) must be per lane.  Alternate two
    different batches for many rounds without waiting in between, hand-off into a result matrix."""
    import hubconf
    model = hubconf.vgg16_netvlad(pretrained=False)
    model.load_state_dict(synth.embednetpca_state(0))
    model = model.to(dev).eval().set_precision(precision)
    xs = [synth.images(2, 128, 160, seed=31 + i).to(dev) for i in range(2)]
    wants = [model(x).clone() for x in xs]
    assert not torch.equal(wants[0], wants[1])
    pf = model.graphed(xs[0], pipeline=True)
    rounds = 24
    res = torch.zeros((rounds * 2, wants[0].shape[1]), device=dev)
    for r in range(rounds):
        pf(xs[r % 2], dest=res[2 * r:2 * r + 2])
    pf.wait()
    torch.cuda.synchronize()
    for r in range(rounds):
        assert torch.equal(res[2 * r:2 * r + 2], wants[r % 2]), r


@pytest.mark.parametrize("precision", ["bf16", "f16mx"])
def test_host_batches_reach_the_replayed_forward_through_staging_buffers(dev, precision):
    """Round 6: a host batch crosses PCIe into one of two staging buffers of its slot and moves into the graph's input
    with a device copy on the lane (extract.GraphedForward.__call__).  Seven distinct batches, pinned / pageable /
    device-resident in turn, many rounds with nothing waited for in between: every staging buffer is reused while
    later transfers are queued, a slot sees host and device sources alternately, and every row of the result
    matrix is its batch's eager descriptor, bit for bit."""
    import hubconf
    model = hubconf.vgg16_netvlad(pretrained=False)
    model.load_state_dict(synth.embednetpca_state(0))
    model = model.to(dev).eval().set_precision(precision)
    N, H, W = 32, 112, 160                            # f16mx proper (not the small-batch bf16x3 route)
    hosts = [synth.images(N, H, W, seed=70 + i) for i in range(7)]
    wants = [model(h.to(dev)).clone() for h in hosts]
    if precision == "f16mx":
        assert model.base_model.effective_precision(hosts[0].to(dev)) == "f16mx"
    srcs = []
    for i, h in enumerate(hosts):
        srcs.append(h.pin_memory() if i % 3 == 0 else (h if i % 3 == 1 else h.to(dev)))
    pf = model.graphed(srcs[2], pipeline=True)
    rounds = 35
    res = torch.zeros((rounds * N, wants[0].shape[1]), device=dev)
    for r in range(rounds):
        pf(srcs[r % 7], dest=res[r * N:(r + 1) * N], stable_src=True)
    pf.wait()
    torch.cuda.synchronize()
    assert len(pf.stage_in) == 4
    for r in range(rounds):
        assert torch.equal(res[r * N:(r + 1) * N], wants[r % 7]), r


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("N,H,W", [(2, 64, 96), (1, 70, 90), (1, 480, 640)])
def test_uint8_input_equals_normalised_input(dev, N, H, W, precision):
    """Raw uint8 NHWC images (ToTensor + Normalize folded into the first kernel) give the same
    descriptors, bit for bit, as the loader's normalised fp32 NCHW tensor."""
    import hubconf
    from ibl.utils.data import MEAN, STD
    model = hubconf.vgg16_netvlad(pretrained=False)
    model.load_state_dict(synth.embednetpca_state(0))
    model = model.to(dev).eval().set_precision(precision)
    g = torch.Generator().manual_seed(H + W)
    u8 = torch.randint(0, 256, (N, H, W, 3), generator=g, dtype=torch.uint8)
    u8[0, 0, :, :] = 0                      # a run of zeros on the border (padding vs value 0)
    u8[0, -1, :, :] = 255
    mean = torch.tensor(MEAN, dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(STD, dtype=torch.float32).view(1, 3, 1, 1)
    x = (u8.permute(0, 3, 1, 2).float() / 255.0 - mean) / std       # the reference transform
    want = model(x.to(dev))
    got = model(u8.to(dev))
    assert torch.equal(got, want)


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-6), ("bf16x3", 2e-5), ("bf16", 6e-3)])
@pytest.mark.parametrize("N,H,W", [(1, 480, 640), (2, 240, 320), (1, 70, 90)])
def test_split_k_small_batches(model, dev, precision, tol, N, H, W):
    """Layers whose tiling leaves the chip idle (conv4 / conv5 of a single image: 40-152 tiles) run
    split-K: partial sums in fp32, fixed-order reduction with bias / ReLU / pool.  Same convolution,
    other summation order: equal to rounding with the one-pass route, and (fp32 / bf16x3) within 1e-4
    of the oracle."""
    x = synth.images(N, H, W, seed=H + N)
    model.set_precision(precision)
    try:
        ops.set_conv_splitk(False)
        a = model(x.to(dev)).cpu()
        ops.set_conv_splitk(True)
        b = model(x.to(dev)).cpu()
    finally:
        ops.set_conv_splitk(True)
        model.set_precision("fp32")
    assert_rel_l2(f"split-K on vs off {precision} {N}x{H}x{W}", b, a, tol)
    if precision != "bf16" and H <= 240:
        with torch.no_grad():
            want = od.embednetpca(x, synth.embednetpca_state(0))
        assert_rel_l2(f"split-K vs oracle {precision}", b, want, TOL_FP32)


def test_embedregionnet_eval_branch_is_the_embednet_forward(dev, state_dict):
    """EmbedRegionNet.forward in eval mode (ibl/models/netvlad.py:196-205) returns (pool_x, normalised
    VLAD) exactly like EmbedNet.forward — against the reference's vectors and bit for bit against
    EmbedNet here; its SFRS training branch is out of scope and says so."""
    from ibl import models
    g = load_golden("desc_small")
    base = models.create("vgg16", pretrained=False)
    pool = models.create("netvlad", dim=base.feature_dim)
    region = models.create("embedregionnet", base, pool, tuple_size=1)
    plain = models.create("embednet", base, pool)
    sd = {k: v for k, v in state_dict.items() if not k.startswith("pca_layer")}
    region.load_state_dict(sd)
    region = region.to(dev).eval().set_precision("fp32")
    plain = plain.to(dev).eval().set_precision("fp32")
    n, _, h, w = [int(v) for v in g["shape"]]
    x = synth.images(n, h, w, seed=int(g["image_seed"])).to(dev)
    pool_x, vlad = region(x)
    p2, v2 = plain(x)
    assert torch.equal(pool_x, p2) and torch.equal(vlad, v2)
    assert_rel_l2("EmbedRegionNet eval vlad", vlad.cpu(), g["vlad_norm"], 1e-4)
    assert_rel_l2("EmbedRegionNet eval pool_x", pool_x.cpu(), g["pool_x"], 1e-4)
    region.train()
    with pytest.raises(NotImplementedError):
        region(x)
