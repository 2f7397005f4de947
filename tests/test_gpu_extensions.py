"""BASELINE.json configs[4] extensions on the GPU: 16-bit descriptor storage, bilinear resize and
multi-scale extraction.  None of these exist in the reference, so parity is pinned
to this repo's own definitions (oracle.descriptor.multiscale_descriptor; "stored values widened to
fp32" for the 16-bit matching) and to exact properties."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_rel_l2
from openibl_amd import ops, synth
from oracle import descriptor as od
from oracle import matching as om

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_descriptor_storage_roundtrip_is_rne(dev, dtype):
    g = torch.Generator().manual_seed(3)
    x = torch.randn((37, 4096), generator=g) * torch.logspace(-6, 3, 37).unsqueeze(1)
    x[0, :4] = torch.tensor([0.0, -0.0, 65504.0, 1e-8])
    st = ops.store_descriptors(x.to(dev), dtype)
    assert st.dtype == dtype
    want = x.to(dtype)                      # torch CPU cast = round-to-nearest-even
    assert torch.equal(st.cpu().view(torch.int16), want.view(torch.int16))
    back = ops.load_descriptors(st)
    assert back.dtype == torch.float32 and torch.equal(back.cpu(), want.float())


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("xdt,ydt", [(torch.float16, torch.float16), (torch.bfloat16, torch.bfloat16),
                                     (torch.float32, torch.float16), (torch.bfloat16, torch.float32)])
@pytest.mark.parametrize("m,n,d", [(70, 333, 256), (300, 2100, 4096)])
def test_stored_descriptors_equal_widened_fp32(dev, m, n, d, xdt, ydt, precision):
    """Matching on 16-bit stored descriptors == matching on the same values widened to fp32:
    bit-identical matrix and top-k (same kernels, same arithmetic, no copy for bf16 storage)."""
    q, g = synth.descriptors(m, d, seed=11), synth.descriptors(n, d, seed=12)
    xs, ys = ops.store_descriptors(q.to(dev), xdt), ops.store_descriptors(g.to(dev), ydt)
    xw, yw = ops.load_descriptors(xs), ops.load_descriptors(ys)
    a = ops.pairwise_sqdist(xs, ys, precision)
    b = ops.pairwise_sqdist(xw, yw, precision)
    assert torch.equal(a, b)
    va, ia = ops.sqdist_topk(xs, ys, 10, index_base=5, precision=precision)
    vb, ib = ops.sqdist_topk(xw, yw, 10, index_base=5, precision=precision)
    assert torch.equal(va, vb) and torch.equal(ia, ib)
    if precision == "fp32":   # and the oracle on the widened values
        want = om.pairwise_distance(xw.cpu(), yw.cpu())
        assert float((a.cpu() - want).abs().max()) <= 1e-4


def test_fused_topk_on_stored_gallery(dev):
    """The sampled / filtered bf16 top-k path (large gallery) reading a bf16-stored gallery in place."""
    q, g = synth.descriptors(512, 1024, seed=5), synth.descriptors(16384, 1024, seed=6)
    for dt in (torch.bfloat16, torch.float16):
        qs, gs = ops.store_descriptors(q.to(dev), dt), ops.store_descriptors(g.to(dev), dt)
        v, i = ops.sqdist_topk(qs, gs, 10, precision="bf16")
        v2, i2 = ops.row_topk(ops.pairwise_sqdist(qs, gs, "bf16"), 10)
        assert torch.equal(v, v2) and torch.equal(i, i2)


@pytest.mark.parametrize("shape,size", [((2, 3, 48, 64), (34, 45)), ((1, 3, 64, 96), (32, 48)),
                                        ((3, 3, 33, 47), (50, 61)), ((1, 1, 16, 16), (16, 16)),
                                        ((2, 3, 480, 640), (339, 453))])
def test_resize_bilinear_equals_interpolate(dev, shape, size):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(shape, generator=g) * 60
    got = ops.resize_bilinear(x.to(dev), size).cpu()
    want = F.interpolate(x, size=size, mode="bilinear", align_corners=False)
    want64 = F.interpolate(x.double(), size=size, mode="bilinear", align_corners=False)
    err, ref_err = float((got - want64).abs().max()), float((want - want64).abs().max())
    print("resize", shape, size, "max |got - fp64| =", err, " torch fp32 - fp64 =", ref_err)
    assert float((got - want).abs().max()) <= 2e-5 * 60
    assert err <= 4 * ref_err + 1e-5


def test_multiscale_descriptor_matches_definition(dev, state_dict):
    import hubconf
    from openibl_amd.multiscale import extract_multiscale, DEFAULT_SCALES
    model = hubconf.vgg16_netvlad()
    model.load_state_dict(state_dict)
    model = model.to(dev).eval().set_precision("fp32")
    x = synth.images(2, 96, 128, seed=23)
    with torch.no_grad():
        want = od.multiscale_descriptor(x, state_dict, DEFAULT_SCALES)
        want64 = od.multiscale_descriptor(x.double(), {k: v.double() for k, v in state_dict.items()},
                                          DEFAULT_SCALES, dtype=torch.float64)
    got = extract_multiscale(model, x.to(dev), DEFAULT_SCALES).cpu()
    assert got.shape == (2, 4096)
    assert float((got.norm(dim=1) - 1).abs().max()) < 1e-5
    assert_rel_l2("multiscale fp32 vs oracle", got, want, 1e-4)
    assert_rel_l2("multiscale fp32 vs fp64 oracle", got, want64, 1e-4)
    # one scale = the plain descriptor
    one = extract_multiscale(model, x.to(dev), (1.0,)).cpu()
    from openibl_amd.evaluators import extract_cnn_feature
    assert_rel_l2("single scale", one, extract_cnn_feature(model, x).cpu(), 1e-6)
    model.set_precision("bf16")
    got16 = extract_multiscale(model, x.to(dev), DEFAULT_SCALES).cpu()
    cos = float((got16 * want).sum(dim=1).min())
    print("multiscale bf16 cosine vs oracle:", cos)
    assert cos > 0.995
