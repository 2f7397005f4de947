"""Per-stage timing of the descriptor path on the GPU (diagnostic script, not a pytest).

    python tests/gpu_timing.py [--batch 32] [--precision bf16] [--iters 5]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops, synth  # noqa: E402


def timed(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        out = fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--regstage", type=int, default=0)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--ablate", type=int, default=0)
    ap.add_argument("--c64", type=int, default=1)
    ap.add_argument("--raster", type=int, default=0)
    ap.add_argument("--korder", type=int, default=-1)
    ap.add_argument("--layers", type=int, default=13, help="time conv1_1 and the first LAYERS-1 3x3 layers only")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    ops.set_regstage(bool(a.regstage))
    ops.set_conv_tile(a.tile)
    ops.set_conv_c64(bool(a.c64))
    ops.set_ring_raster(a.raster)
    ops.set_conv_korder(a.korder)
    from openibl_amd import lib as _l
    _l.debug_hooks().oibl_debug_set_conv_ablate(a.ablate)
    sd = synth.embednetpca_state(0)
    N, H, W, p = a.batch, a.height, a.width, a.precision
    x = synth.images(min(N, 4), H, W, seed=1)
    x = x.repeat((N + x.shape[0] - 1) // x.shape[0], 1, 1, 1)[:N].contiguous().to(dev)
    convs = [(sd[f"base_model.base.{i}.weight"].to(dev), sd[f"base_model.base.{i}.bias"].to(dev))
             for i in synth.CONV_IDX]
    packed = [convs[0][0]] + [ops.pack_conv3x3(w, p) for w, _ in convs[1:]]
    biases = [b for _, b in convs]
    rows = []
    t, act = timed(lambda: ops.conv1_1_nchw(x, packed[0], biases[0], p), a.iters)
    rows.append(("conv1_1", t, 2 * N * H * W * 64 * 27))
    h, w = H, W
    for l in range(1, min(13, a.layers)):
        cin, cout, relu, pool = ops.VGG16_CFG[l]
        inp = act
        t, act = timed(lambda: ops.conv3x3_nhwc(inp, packed[l], biases[l], bool(relu), bool(pool), p),
                       a.iters)
        rows.append((f"conv{l:02d} {cin}->{cout} {h}x{w}{' pool' if pool else ''}", t,
                     2 * N * h * w * cout * 9 * cin))
        if pool:
            h, w = h // 2, w // 2
    if a.layers < 13:
        for name, ms, fl in rows:
            print(f"  tile={a.tile} {name:34s} {ms:9.3f} ms  {fl / ms / 1e9:9.1f} TFLOP/s")
        return
    if p == "bf16x3":
        t, _ = timed(lambda: ops.vgg16_stem_x3(x, packed[0], biases[0], packed[1], biases[1]), a.iters)
        rows.append(("stem fused (conv1_1+conv01)", t, rows[0][2] + rows[1][2]))
    if p == "bf16":
        t, _ = timed(lambda: ops.vgg16_stem(x, packed[0], biases[0], packed[1], biases[1]), a.iters)
        rows.append(("stem fused (conv1_1+conv01)", t, rows[0][2] + rows[1][2]))
    feat = ops.x3_join(act) if p == "bf16x3" else act   # the whole-backbone entry writes fp32 directly
    aw = sd["net_vlad.conv.weight"].reshape(64, 512).contiguous().to(dev)
    cent = sd["net_vlad.centroids"].to(dev)
    t, (_, vl) = timed(lambda: ops.netvlad(feat, aw, cent, True, False, True), a.iters)
    rows.append(("netvlad", t, 2 * 2 * N * h * w * 64 * 512))
    pw = ops.cast(sd["pca_layer.weight"].reshape(4096, 32768).to(dev), p)
    pb = sd["pca_layer.bias"].to(dev)
    t, _ = timed(lambda: ops.pca(vl, pw, pb), a.iters)
    rows.append(("pca", t, 2 * N * 4096 * 32768))
    t, _ = timed(lambda: ops.pca(vl, pw, pb) if False else ops.vgg16_conv5(x, packed, biases, p), a.iters)
    rows.append(("vgg16 whole", t, sum(r[2] for r in rows[:13])))
    named = dict((r[0], r[1]) for r in rows)
    tot = sum(r[1] for r in rows if not r[0].startswith(("vgg16 whole", "stem fused")))
    if "stem fused (conv1_1+conv01)" in named:
        tot_f = tot - rows[0][1] - rows[1][1] + named["stem fused (conv1_1+conv01)"]
        print(f"  (with the fused stem: {tot_f:.3f} ms -> {N / tot_f * 1e3:.1f} img/s)")
    print(f"precision={p} batch={N} {H}x{W} regstage={a.regstage} tile={a.tile} ablate={a.ablate} raster={a.raster} korder={a.korder}")
    for name, ms, fl in rows:
        print(f"  {name:34s} {ms:9.3f} ms  {fl / ms / 1e9:9.1f} TFLOP/s")
    print(f"  sum of stages {tot:.3f} ms -> {N / tot * 1e3:.1f} img/s")
    out = Path("gpurun_out")
    out.mkdir(exist_ok=True)
    with open(out / f"timing_{p}_b{N}_rs{a.regstage}_t{a.tile}.json", "w") as f:
        json.dump({"precision": p, "batch": N, "rows": rows, "img_per_s": N / tot * 1e3}, f, indent=1)


if __name__ == "__main__":
    main()
