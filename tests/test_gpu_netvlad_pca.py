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
