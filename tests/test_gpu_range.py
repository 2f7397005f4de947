"""The fp16 range of the f16mx arithmetic ("What's weak" 1).

hi = fp16(v) exists only for |v| <= 65504.  Every producer of f16mx lines raises a device flag when it meets
a larger value (csrc/common.h, mx_raise_range_flag); `VGG.features_nhwc` and `GraphedForward` read it once per
batch and recompute a flagged batch in bf16x3 — the other mode inside north_star's 1e-4, without a range limit.
Checked here: the flag itself (rows, one layer on the ring and on the halo kernel, the fused stem), the
descriptor at three activation magnitudes (1e3, 3e4, 2e5) against the fp64 oracle eagerly and through the
replayed two-lane forward, an overflow that only a deep layer sees, weights shaped like a trained network
(per-channel gains, dead channels, activations in the thousands: reference input scale,
ibl/utils/data/__init__.py:40-41) at BASELINE configs[1]'s batch 32, and the matching side (+inf norms)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_desc, assert_rel_l2, rel_l2
from openibl_amd import ops, synth
from oracle import descriptor as od
from oracle import matching as om

pytestmark = pytest.mark.gpu

TOL_DESC = 1e-4
TOL_LAYER = 4e-5


def test_split_rows_raise_the_flag_only_beyond_fp16(dev):
    g = torch.Generator().manual_seed(3)
    x = torch.randn((64, 128), generator=g) * 1000.0
    x[5, 40] = 65504.0                                   # the largest fp16: still exact
    flag = ops.new_range_flag(dev)
    ops.mx_split(x.to(dev), range_flag=flag)
    assert int(flag.item()) == 0
    x[17, 3] = -65505.0
    ops.mx_split(x.to(dev), range_flag=flag)
    assert int(flag.item()) == 1
    ops.mx_split(torch.zeros_like(x).to(dev), range_flag=flag)      # sticky: only the caller clears it
    assert int(flag.item()) == 1
    ops.mx_split(x.to(dev))                                          # no flag given: the unflagged entry point


def _layer(cin, cout, H, W, N, seed, xpeak, wgain):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((N, cin, H, W), generator=g).abs()
    x = x / x.max() * xpeak
    w = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5 * wgain
    b = torch.randn((cout,), generator=g) * 0.1 * xpeak
    return x, w, b


@pytest.mark.parametrize("cin,cout,pool", [(256, 256, False), (256, 256, True), (512, 512, False), (128, 128, True)])
@pytest.mark.parametrize("xpeak,wgain,over", [(1.0e3, 1.0, False), (3.0e4, 0.6, False), (6.0e4, 4.0, True)])
def test_one_layer_at_three_magnitudes(dev, cin, cout, pool, xpeak, wgain, over):
    """One f16mx convolution (halo kernel at Cout = 256, ring kernels otherwise, with and without the fused
    pool) whose outputs peak near 1e3, 3e4 and 2e5: inside the range the result is the usual 4e-5 from fp64 and
    the flag stays down; beyond it the flag goes up (and bf16x3 of the same layer is unaffected)."""
    x, w, b = _layer(cin, cout, 24, 40, 2, 7 + cin + int(pool), xpeak, wgain)
    want = F.conv2d(x.double(), w.double(), b.double(), padding=1).relu()
    if pool:
        want = F.max_pool2d(want, 2, 2)
    peak = float(want.max())
    assert (peak > 65504 * 1.5) if over else (peak < 65504 * 0.8), peak
    xd = ops.nchw_f32_to_nhwc(x.to(dev), "fp32")
    flag = ops.new_range_flag(dev)
    got = ops.conv3x3_nhwc(ops.mx_split(xd, range_flag=flag), ops.pack_conv3x3(w.to(dev), "f16mx"), b.to(dev),
                           True, pool, "f16mx", range_flag=flag)
    print(f"{cin}->{cout} pool={pool}: output peak {peak:.3g}, flag {int(flag.item())}")
    assert int(flag.item()) == int(over)
    if not over:
        assert_rel_l2("f16mx layer", ops.mx_join(got).permute(0, 3, 1, 2).cpu(), want, TOL_LAYER)
    g3 = ops.conv3x3_nhwc(ops.x3_split(xd), ops.pack_conv3x3(w.to(dev), "bf16x3"), b.to(dev), True, pool, "bf16x3")
    assert_rel_l2("bf16x3 layer", ops.x3_join(g3).permute(0, 3, 1, 2).cpu(), want, 1e-5)


# The f16mx backbone stores its activations multiplied by 2^-3 (conv.hip, g_mx_act_shift): the fp16 bound sits at
# 65504 * 8 = 5.2e5 in activation units.
ACT_HEADROOM = 8.0


def _scaled(sd, c):
    """The same network on inputs c times as large: conv is linear, ReLU and max-pool are positively
    homogeneous, so scaling the input and every bias by c scales every activation by c — and NetVLAD
    normalises each position, so the descriptor does not move (in exact arithmetic)."""
    out = copy.copy(sd)
    for k, v in sd.items():
        if k.startswith("base_model.base.") and k.endswith(".bias"):
            out[k] = v * c
    return out


def _model(sd, dev, precision="f16mx"):
    import hubconf
    model = hubconf.vgg16_netvlad(pretrained=False)
    model.load_state_dict(sd)
    model = model.to(dev).eval().set_precision(precision)
    model.base_model.F16MX_MIN_TILES = 0          # two images: keep them on the f16mx kernels
    return model


@pytest.mark.parametrize("c,over", [(10.0, False), (300.0, False), (2000.0, False), (2000.0 * ACT_HEADROOM, True)])
def test_descriptor_at_three_magnitudes(dev, state_dict, c, over):
    """The whole 480x640 descriptor with activations that reach ~1e3, ~3e4, ~2e5 and ~1.3e6 between the layers: f16mx
    (with its guard) stays within 1e-4 of the fp64 oracle in all four, the flag — and the bf16x3 re-run — only
    in the last (2e5 is beyond fp16 but inside the scaled format); eagerly, through the one-lane and the two-lane
    replay (bit-identical to the eager result)."""
    sd = _scaled(state_dict, c)
    x = synth.images(2, 480, 640, seed=77) * c
    with torch.no_grad():
        inter = od.embednetpca(x, sd, dtype=torch.float64, return_intermediates=True)
        want = inter["desc"]
        peak = max(float(od.vgg16_conv5(x[:1].double(), sd, upto=u).max()) for u in (2, 4, 7, 10))
    print(f"scale {c:g}: activation peak behind conv1_2 / 2_2 / 3_3 / 4_3: {peak:.3g}")
    assert (peak > 65504 * ACT_HEADROOM * 1.5) if over else (peak < 65504 * ACT_HEADROOM / 2)
    model = _model(sd, dev)
    vgg = model.base_model
    xd = x.to(dev)
    assert vgg.effective_precision(xd) == "f16mx"
    got = model(xd).clone()
    assert int(vgg.last_range_flag().item()) == int(over)
    assert vgg.range_fallbacks == int(over)
    assert_desc(f"f16mx descriptor, inputs x {c:g}", got, want, TOL_DESC)
    # the guard is what keeps the last case inside the tolerance: the raw f16mx pass is far outside
    if over:
        ws, bs = vgg._packed(xd.device, "f16mx")
        raw = model.head_from_features(ops.vgg16_conv5(xd, ws, bs, "f16mx"))
        print(f"unguarded f16mx pass: rel-L2 {rel_l2(raw.cpu(), want):.3e}")
        assert rel_l2(raw.cpu(), want) > 10 * TOL_DESC
        g3 = _model(sd, dev, "bf16x3")(xd)
        assert torch.equal(got, g3)                      # the re-run IS the bf16x3 forward
    for pipeline in (False, True):
        fwd = model.graphed(xd, pipeline=pipeline)
        a, b = fwd(), fwd(xd)                 # (two lanes: the two slots' static outputs, final after wait())
        fwd.wait()
        torch.cuda.synchronize()
        assert torch.equal(a, got) and torch.equal(b, got)
        d = torch.empty_like(got)
        e = fwd(xd, dest=d)                   # hand-off into a caller's matrix: the re-run lands there too
        fwd.wait()
        torch.cuda.synchronize()
        assert torch.equal(d, got) and torch.equal(e, got)
        assert fwd.range_fallbacks == (3 if over else 0)


def test_activation_scale_is_an_exact_image(dev, state_dict):
    """The backbone's stored activations are the unscaled ones times 2^-3 (test hook: 0 = unscaled).  A power of
    two commutes with every rounding of the format, so at the reference's input scale (activations in the tens to
    thousands: images and biases x 100 here) the fp32 map handed to the head is the unscaled pass's, bit for bit —
    at batch 2 (ring / halo kernels) and for one small image (row sub-ranges + split-K: the reduction kernel
    scales).  On unit-range images the synthetic state's activations are ~1e-2: values below fp16's normal range
    after scaling (|x| < 5e-4) lose bits, and the map moves by a fraction of the format's own error."""
    from openibl_amd import lib
    for c, shape in [(100.0, (2, 480, 640)), (100.0, (1, 96, 128)), (1.0, (2, 96, 128))]:
        model = _model(_scaled(state_dict, c), dev)
        vgg = model.base_model
        x = (synth.images(*shape, seed=79) * c).to(dev)
        ws, bs = vgg._packed(x.device, "f16mx")
        product = ops.vgg16_conv5(x, ws, bs, "f16mx").clone()
        hooks = lib.debug_hooks()                      # (from here on the debug library computes)
        try:
            scaled = ops.vgg16_conv5(x, ws, bs, "f16mx").clone()
            hooks.oibl_debug_set_mx_act_shift(0)
            plain = ops.vgg16_conv5(x, ws, bs, "f16mx").clone()
        finally:
            hooks.oibl_debug_set_mx_act_shift(3)
        assert torch.equal(product, scaled)            # the product library's constant is the hook's default
        d = rel_l2(scaled.cpu(), plain.cpu().double())
        same = float((scaled == plain).float().mean())
        print(f"{shape} x {c:g}: map mean {float(plain.mean()):.3g}; scaled against unscaled activations: rel-L2 "
              f"{d:.2e}, {same:.4f} of the map bit-equal")
        assert (same > 0.999 and d < 1e-7) if c > 1 else d < 1.5e-5


@pytest.mark.parametrize("layer", [4, 8])     # conv3_1 (halo kernel), conv4_2 (256 x 256 ring kernel)
def test_overflow_in_one_deep_layer_only(dev, state_dict, layer):
    """conv3_1 / conv4_2 scaled up by 24000 and the next layer down by as much: only that layer's output
    leaves the format's range (the stem and everything else stay at ~100)."""
    s = 3000.0 * ACT_HEADROOM
    sd = copy.copy(state_dict)
    i0, i1 = synth.CONV_IDX[layer], synth.CONV_IDX[layer + 1]
    sd[f"base_model.base.{i0}.weight"] = state_dict[f"base_model.base.{i0}.weight"] * s
    sd[f"base_model.base.{i0}.bias"] = state_dict[f"base_model.base.{i0}.bias"] * s
    sd[f"base_model.base.{i1}.weight"] = state_dict[f"base_model.base.{i1}.weight"] / s
    x = synth.images(2, 192, 256, seed=78)
    with torch.no_grad():
        want = od.embednetpca(x, sd, dtype=torch.float64)
    model = _model(sd, dev)
    got = model(x.to(dev))
    assert model.base_model.range_fallbacks == 1
    assert_desc(f"f16mx descriptor, conv layer {layer} x {s:g}", got, want, TOL_DESC)


@pytest.mark.parametrize("precision", ["f16mx", "bf16x3"])
def test_configs1_batch32_trained_like_weights(dev, precision):
    """BASELINE configs[1] (batch 32, 480x640) on weights shaped like a trained VGG16 (synth.backbone_state,
    trained_like: log-normal channel gains, dead channels, activations peaking in the thousands as on the
    reference's 0-255-scale input): both 1e-4 modes against the oracle, no range fallback."""
    sd = synth.embednetpca_state(0, trained_like=True)
    x = synth.images(32, 480, 640, seed=322)
    with torch.no_grad():
        want = torch.cat([od.embednetpca(x[i:i + 8], sd) for i in range(0, 32, 8)])
        peak = float(od.vgg16_conv5(x[:1], sd, upto=2).max())
    assert 1e3 < peak < 3e4, peak
    model = _model(sd, dev, precision)
    got = model(x.to(dev))
    assert model.base_model.range_fallbacks == 0
    assert_desc(f"trained-like weights, batch 32, {precision} (conv1_2 peak {peak:.3g})", got, want, TOL_DESC)


def test_configs1_batch32_calibrated_activation_range(dev):
    """BASELINE configs[1] on weights whose per-layer activation maxima are calibrated to what
    fixed-point studies report for the trained VGG16 on 0-255-scale pixels (synth.backbone_state_calibrated: up to
    2.8e4 behind conv3_3 on the calibration batch, heavy-tailed channel gains) — the fallback RATE of the f16mx
    range guard at the benchmark batch, and the descriptors of every image inside 1e-4 whether or not a batch is
    re-run.  The device maxima of the test batch are reported per layer (bf16x3 feature maps)."""
    sd = synth.backbone_state_calibrated(0)
    sd.update(synth.netvlad_state(0))
    sd.update(synth.pca_state(0))
    n_batches, fallbacks, runs = 3, 0, {}
    model = _model(sd, dev, "f16mx")
    worst = 0.0
    for k in range(n_batches):
        x = synth.images(32, 480, 640, seed=700 + k)
        with torch.no_grad():
            want = torch.cat([od.embednetpca(x[i:i + 8], sd) for i in range(0, 32, 8)])
        before = model.base_model.range_fallbacks
        got = model(x.to(dev))
        fallbacks += model.base_model.range_fallbacks - before
        assert_desc(f"calibrated activations, batch {k}, f16mx (+ guard)", got, want, TOL_DESC)
        if k == 0:   # where the batch's activations peak (device, bf16x3 maps of 8 images: layer by layer)
            ws, bs = model.base_model._packed(dev, "bf16x3")
            cur = ops.conv1_1_nchw(x[:8].to(dev), ws[0], bs[0], "bf16x3")
            peaks = [float(ops.x3_join(cur).max())]
            for li in range(1, 13):
                cin, cout, relu, pool = ops.VGG16_CFG[li]
                cur = ops.conv3x3_nhwc(cur, ws[li], bs[li], bool(relu), bool(pool), "bf16x3")
                peaks.append(float(ops.x3_join(cur).abs().max()))
            print("per-layer maxima of 8 test images: " + " ".join(f"{p:.3g}" for p in peaks))
            worst = max(peaks)
    runs = dict(model.base_model.precision_runs)
    print(f"calibrated activations: {fallbacks} of {n_batches} batches of 32 re-run in bf16x3 "
          f"(largest activation seen {worst:.3g}; fp16 max 65504); precision_runs {runs}")
    assert worst > 2e4            # the calibration did reach the band the test is about


def test_matching_marks_out_of_range_rows(dev):
    """The distance kernels' f16mx operands: a descriptor row with an element beyond fp16 gets the norm +inf,
    so every distance to it is +inf (never a finite wrong number); all other rows are untouched."""
    q, g, gt, pids = synth.retrieval_problem(64, 512, dim=256, seed=5)
    g2 = g.clone()
    g2[7, 100] = 1.0e5
    d = ops.pairwise_sqdist(q.to(dev), g2.to(dev), "f16mx").cpu()
    want = om.pairwise_distance(q, g).numpy()
    keep = np.ones(512, dtype=bool)
    keep[7] = False
    assert np.isinf(d.numpy()[:, 7]).all() and (d.numpy()[:, 7] > 0).all()
    assert np.abs(d.numpy()[:, keep] - want[:, keep]).max() < 2e-5
    pr = ops.PreparedRows(g2.to(dev), "f16mx")
    assert torch.isinf(pr.norms[7]) and torch.isfinite(pr.norms[keep]).all()


def test_extract_features_reruns_only_the_flagged_batches(dev, state_dict):
    """The replayed two-lane extraction (ibl.evaluators.extract_features) with the guard in its non-blocking
    form: batches in flight are polled, a flagged one is recomputed in bf16x3 from the loader's (pinned) batch
    into its rows of the output matrix — pinned and pageable (staged) batches, overflowing and ordinary ones
    mixed; bit-identical to the eager batch-by-batch route."""
    import torch.distributed as dist
    from openibl_amd import evaluators as ev
    from openibl_amd.extract import _GRAPH_STORES, unwrap_model
    mine = not dist.is_initialized()
    if mine:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
    try:
        _extract_flow(dev, state_dict, ev, _GRAPH_STORES, unwrap_model)
    finally:
        if mine:
            dist.destroy_process_group()


def _extract_flow(dev, state_dict, ev, _GRAPH_STORES, unwrap_model):
    sd = _scaled(state_dict, 2000.0 * ACT_HEADROOM)
    model = _model(sd, dev)

    class Loader:
        def __init__(self, batches):
            self.batches = batches
            self.sampler = range(sum(int(b[0].shape[0]) for b in batches))

        def __iter__(self):
            return iter(self.batches)

        def __len__(self):
            return len(self.batches)

    big = {1, 2, 5, 8}                                     # these batches leave the fp16 range
    batches = []
    for k in range(10):
        x = synth.images(3, 64, 96, seed=900 + k) * (2000.0 * ACT_HEADROOM if k in big else 1.0)
        batches.append((x.pin_memory() if k % 3 else x, [f"b{k}_{i}" for i in range(3)]))   # every third one pageable
    names = [(f, 0, 0.0, 0.0) for b in batches for f in b[1]]
    feats = ev.extract_features(model, Loader(batches), names, gpu=dev.index)
    fwd = next(iter(_GRAPH_STORES[unwrap_model(model)][1].values()))
    # batch 0 runs eagerly, batch 1 is captured (its warm-up runs eagerly); replays: batches 1..9
    assert fwd.range_fallbacks == len(big) and not fwd.pending
    before = model.base_model.range_fallbacks
    eager = torch.cat([ev.extract_cnn_feature(model, b[0], gpu=dev.index).cpu() for b in batches])
    assert model.base_model.range_fallbacks - before == len(big)
    assert torch.equal(torch.stack(list(feats.values())), eager)
    # round 6: the UNCAPTURED route (use_graphs=False: extract.EagerLanes) settles the flag lazily too — the same two
    # lanes with eager launches, no host synchronisation per batch; batch 0 is settled at once (it tells the width),
    # batches 1..9 through the pinned ring: bit-identical rows, the same batches re-run
    before = model.base_model.range_fallbacks
    feats2 = ev.extract_features(model, Loader(batches), names, gpu=dev.index, use_graphs=False)
    assert model.base_model.range_fallbacks - before == len(big)
    assert torch.equal(torch.stack(list(feats2.values())), eager)
