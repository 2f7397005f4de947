"""Raw uint8 NHWC input of the 1e-4 modes: the bf16x3 / f16mx stems gather the
loader's bytes themselves (csrc/conv.hip, vgg_stem_x3_kernel<MX, U8>) — no normalising pass, a quarter of the
bytes over PCIe.  Normalize is one fma per value there (within 2^-16 of the loader's three rounded operations:
tests/test_stem_u8_cpu.py), so the result is not bit-identical to the fp32-input route; it is checked against it
(a few 1e-6), against the oracle on the loader's own tensor (1e-4), on border-heavy odd sizes, and the old route
(normalising pass, test hook) stays bit-identical to the fp32 input."""
import pytest
import torch

from conftest import assert_desc, assert_rel_l2, rel_l2
from openibl_amd import lib, ops, synth
from oracle import descriptor as od

pytestmark = pytest.mark.gpu


def _u8_and_normalised(N, H, W, seed):
    from ibl.utils.data import MEAN, STD
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (N, H, W, 3), generator=g, dtype=torch.uint8)
    u8[0, 0, :, :] = 0                      # runs of 0 / 255 on the borders: padding is 0.0, not normalise(0)
    u8[0, -1, :, :] = 255
    u8[0, :, 0, :] = 0
    mean = torch.tensor(MEAN, dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(STD, dtype=torch.float32).view(1, 3, 1, 1)
    x = (u8.permute(0, 3, 1, 2).float() / 255.0 - mean) / std       # the reference transform
    return u8, x.contiguous()


# (42 of the 768 (channel, byte) pairs hand conv1_1 a lo part one unit apart: ~1e-6 on its output, ~1e-5 on the
#  conv5_3 map twelve layers later; the two f16mx routes additionally round their lines independently)
@pytest.mark.parametrize("precision,tol_stem", [("bf16x3", 3e-5), ("f16mx", 6e-5)])
@pytest.mark.parametrize("N,H,W", [(2, 64, 96), (1, 70, 90), (3, 37, 53), (1, 16, 16), (1, 480, 640), (2, 33, 131)])
def test_uint8_stems_against_the_normalised_input(dev, state_dict, precision, tol_stem, N, H, W):
    import hubconf
    model = hubconf.vgg16_netvlad(pretrained=False)
    model.load_state_dict(state_dict)
    model = model.to(dev).eval().set_precision(precision)
    u8, x = _u8_and_normalised(N, H, W, H * 7 + W)
    vgg = model.base_model
    vgg.F16MX_MIN_TILES = 0        # small images on the f16mx kernels too (by default they run in bf16x3)
    f_ref = vgg.features_nhwc(x.to(dev)).clone()
    f_u8 = vgg.features_nhwc(u8.to(dev)).clone()
    d = rel_l2(f_u8.cpu(), f_ref.cpu())
    print(f"{precision} {N}x{H}x{W}: conv5_3 map, uint8 stem vs normalised input: rel-L2 {d:.2e}")
    assert d < tol_stem
    with torch.no_grad():
        want = od.embednetpca(x, state_dict)
    assert_desc(f"{precision} descriptor from uint8 {N}x{H}x{W}", model(u8.to(dev)), want, 1e-4)
    # the old route (normalising pass into the workspace, then the fp32-input stem): bit-identical
    lib.debug_hooks().oibl_debug_set_stem_u8(0)
    try:
        assert torch.equal(vgg.features_nhwc(u8.to(dev)), f_ref)
    finally:
        lib.debug_hooks().oibl_debug_set_stem_u8(1)


def test_uint8_f16mx_batch32_replayed(dev, state_dict):
    """BASELINE configs[1] from raw images: 32 x 480 x 640 uint8 through the replayed two-lane forward == eager,
    descriptors within 1e-4 of the oracle on the loader's tensor (one batch of 8 checked)."""
    import hubconf
    model = hubconf.vgg16_netvlad(pretrained=False)
    model.load_state_dict(state_dict)
    model = model.to(dev).eval().set_precision("f16mx")
    u8, x = _u8_and_normalised(32, 480, 640, 99)
    ud = u8.to(dev)
    got = model(ud).clone()
    with torch.no_grad():
        want = od.embednetpca(x[:8], state_dict)
    assert_desc("f16mx descriptors from uint8, batch 32", got[:8], want, 1e-4)
    fwd = model.graphed(ud, pipeline=True)
    a, b = fwd(), fwd(ud)
    fwd.wait()
    torch.cuda.synchronize()
    assert torch.equal(a, got) and torch.equal(b, got)
    assert model.base_model.range_fallbacks == 0
