"""f16mx for small problems: the ring kernels contract the tiles of a nearly empty round —
all tiles of a small layer — split-K (csrc/conv.hip, mx_split_plan / conv_mx_splitk_reduce_kernel), so a single
480x640 image, Tokyo 24/7's batch-1 queries (examples/test_tokyo_best.py:21-25) and ragged last batches run in
the arithmetic that was asked for instead of silently in bf16x3."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_desc, assert_rel_l2, rel_l2
from openibl_amd import lib, ops, synth
from oracle import descriptor as od

pytestmark = pytest.mark.gpu

TOL_DESC = 1e-4
TOL_LAYER = 4e-5


def _case(N, H, W, cin, cout, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((N, cin, H, W), generator=g)
    w = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.1
    return x, w, b


@pytest.mark.parametrize("N,H,W,cin,cout,relu,pool", [
    (1, 30, 40, 512, 512, False, False),   # conv5_3 of one image: 10 tiles -> 9 parts each
    (1, 60, 80, 512, 512, True, True),     # conv4_3 of one image, pooled: the reduction pools
    (1, 60, 80, 256, 512, True, False),    # conv4_1: 3 parts
    (1, 120, 160, 256, 256, True, True),   # conv3_3 of one image (ring kernel instead of the halo kernel), pooled
    (1, 120, 160, 128, 256, True, False),  # conv3_1
    (2, 240, 320, 64, 128, True, False),   # conv2_1 of two images: one full round + 44 split tiles
    (3, 31, 45, 128, 128, True, True),     # K order (chunk, tap): parts are channel chunks; odd sizes, pooled
    (5, 21, 19, 256, 512, False, False),   # ragged last tile inside the split part
])
def test_split_layers_against_fp64_and_the_one_pass_kernel(dev, N, H, W, cin, cout, relu, pool):
    x, w, b = _case(N, H, W, cin, cout, seed=H + cin + N)
    want = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if relu:
        want = want.relu()
    if pool:
        want = F.max_pool2d(want, 2, 2)
    xd = ops.mx_split(ops.nchw_f32_to_nhwc(x.to(dev), "fp32"))
    wp, bd = ops.pack_conv3x3(w.to(dev), "f16mx"), b.to(dev)
    run = lambda: ops.conv3x3_nhwc(xd, wp, bd, relu, pool, "f16mx")   # noqa: E731
    if cout == 128:
        # the 128-output-channel layers run on the 4-wave halo kernel (conv_halo4.h) in one pass by default; their
        # 512 x 128 ring tiling and its split-K remainder stay in the library behind the hook, and stay tested
        lib.debug_hooks().oibl_debug_set_mx_variant(1)
    assert lib.load().oibl_conv3x3_workspace_bytes(N, H, W, cin, cout, int(pool), ops.F16MX) > 0   # the plan splits
    y = run()
    got = ops.nhwc_to_nchw_f32(ops.mx_join(y)).cpu()
    assert_rel_l2(f"split-K f16mx {N}x{H}x{W} {cin}->{cout}", got, want, TOL_LAYER)
    assert torch.equal(run(), y) and torch.equal(run(), y)          # fixed-order reduction: the same bits every time
    h = lib.debug_hooks()
    if cout == 128:
        h.oibl_debug_set_mx_variant(1)
    h.oibl_debug_set_mx_splitk(2)                                    # the one-thread-per-line reduction: same bits
    try:
        assert torch.equal(ops.conv3x3_nhwc(xd, wp, bd, relu, pool, "f16mx"), y)
    finally:
        h.oibl_debug_set_mx_splitk(1)
    h.oibl_debug_set_mx_splitk(0)
    h.oibl_debug_set_mx_variant(1)                                   # the one-pass ring kernel
    try:
        one = ops.nhwc_to_nchw_f32(ops.mx_join(ops.conv3x3_nhwc(xd, wp, bd, relu, pool, "f16mx"))).cpu()
    finally:
        h.oibl_debug_set_mx_splitk(1)
        h.oibl_debug_set_mx_variant(0)
    d = rel_l2(got, one)
    print(f"split vs one pass: rel-L2 {d:.2e}")
    assert d < 1e-5          # fp32 association of the K sum + where that crosses a rounding boundary of the line


@pytest.mark.parametrize("n,H,W", [(1, 480, 640), (2, 480, 640), (5, 480, 640), (1, 352, 500), (3, 224, 224)])
def test_small_batches_run_f16mx_and_match_the_oracle(dev, state_dict, n, H, W):
    """Batches of 1, 2, 5 images (and Tokyo-like odd sizes): the backbone runs in f16mx — no silent bf16x3 —
    and the descriptors are within 1e-4 of the oracle; eagerly and replayed."""
    import hubconf
    x = synth.images(n, H, W, seed=500 + n + H)
    with torch.no_grad():
        want = od.embednetpca(x, state_dict)
    model = hubconf.vgg16_netvlad(pretrained=False)
    model.load_state_dict(state_dict)
    model = model.to(dev).eval().set_precision("f16mx")
    vgg = model.base_model
    vgg.F16MX_MIN_TILES = 0          # the kernels are the subject here, not the rule that picks the faster mode
    xd = x.to(dev)
    assert vgg.effective_precision(xd) == "f16mx"
    got = model(xd).clone()
    assert vgg.precision_runs == {"f16mx": 1} and vgg.range_fallbacks == 0
    assert_desc(f"f16mx, batch {n} of {H}x{W}", got, want, TOL_DESC)
    fwd = model.graphed(xd, pipeline=False)
    assert torch.equal(fwd(), got) and torch.equal(fwd(xd), got)
    # rows do not depend on their batch mates (the split plan changes with the batch, the arithmetic per
    # output element only in its fp32 association)
    if n > 1:
        alone = model(xd[:1].contiguous())
        assert rel_l2(alone.cpu(), got[:1].cpu()) < 2e-5
