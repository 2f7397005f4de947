"""MFMA GEMM core, casts and small helpers through the C ABI (GPU)."""
import numpy as np
import pytest
import torch

from conftest import assert_rel_l2
from openibl_amd import ops

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


@pytest.mark.parametrize("regstage", [False, True])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 128, 256), (200, 64, 128),
                                   (33, 192, 4096), (1, 64, 64), (513, 256, 1024)])
def test_gemm_nt(dev, M, N, K, dtype, regstage):
    # asymmetric operands: a transposed / mis-mapped fragment cannot pass
    a = _rand((M, K), 1) + torch.arange(M).float()[:, None] * 1e-2
    b = _rand((N, K), 2) - torch.arange(N).float()[:, None] * 2e-2
    ad, bd = ops.cast(a.to(dev), dtype), ops.cast(b.to(dev), dtype)
    c = ops.gemm_nt(ad, bd, regstage=regstage).cpu()
    want = ad.cpu().double() @ bd.cpu().double().t()   # operands as the kernel saw them
    assert_rel_l2(f"gemm_nt {dtype} {M}x{N}x{K} regstage={regstage}", c, want, 2e-6)


def test_cast_roundtrip(dev):
    x = _rand((1000, 37), 3, 100.0).to(dev)
    xb = ops.cast(x, "bf16")
    assert xb.dtype == torch.bfloat16
    assert torch.equal(xb.cpu(), x.cpu().to(torch.bfloat16))      # RNE like torch
    assert torch.equal(ops.to_f32(xb).cpu(), xb.cpu().float())


def test_l2_normalize(dev):
    x = _rand((37, 4096), 4).to(dev)
    x[5] = 0
    got = ops.l2_normalize(x).cpu()
    want = torch.nn.functional.normalize(x.cpu().double(), dim=-1)
    assert_rel_l2("l2_normalize", got, want, 1e-6)
    assert torch.count_nonzero(got[5]) == 0


def test_layout_roundtrip(dev):
    x = _rand((3, 512, 5, 7), 5).to(dev)
    for p in ("fp32", "bf16"):
        nhwc = ops.nchw_f32_to_nhwc(x, p)
        assert tuple(nhwc.shape) == (3, 5, 7, 512)
        back = ops.nhwc_to_nchw_f32(nhwc).cpu()
        want = x.cpu() if p == "fp32" else x.cpu().to(torch.bfloat16).float()
        assert torch.equal(back, want)
        gm = ops.global_maxpool_nhwc(nhwc).cpu()
        assert torch.equal(gm, want.flatten(2).max(dim=2).values)


def test_cpu_tensor_is_rejected():
    from openibl_amd.lib import OpenIBLAmdError
    with pytest.raises(OpenIBLAmdError):
        ops.l2_normalize(torch.zeros(2, 4))
