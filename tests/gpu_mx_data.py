"""Timing experiment (diagnostic): is the f16mx ring convolution's duration data dependent?  conv4_2's shape
(512 -> 512 at 60 x 80, batch 32) on synthetic inputs of several kinds and on the REAL activations of the model
(the chain conv1_1 .. conv4_1 run layer by layer), weights random / the model's."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops, synth  # noqa: E402
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
N = 32


def timeit(fn, reps=5, iters=4):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters)
    return sorted(ts)[len(ts) // 2]


sd = synth.embednetpca_state(0)
ws = [sd[f"base_model.base.{i}.weight"].to(dev) for i in synth.CONV_IDX]
bs = [sd[f"base_model.base.{i}.bias"].to(dev) for i in synth.CONV_IDX]
x = synth.images(N, 480, 640, seed=100).to(dev)
# real chain in f16mx: stem (bf16x3) -> join -> mx rows -> conv2_1 .. conv4_1
a = ops.x3_join(ops.vgg16_stem_x3(x, ws[0], bs[0], ops.pack_conv3x3(ws[1], "bf16x3"), bs[1]))
print(f"stem out: max {a.abs().max().item():.1f} mean {a.mean().item():.2f} zeros {float((a == 0).float().mean()):.3f}")
acts = {}
cur = ops.mx_split(a)
for l in range(2, 9):
    relu, pool = ops.VGG16_CFG[l][2], ops.VGG16_CFG[l][3]
    acts[l] = cur
    cur = ops.conv3x3_nhwc(cur, ops.pack_conv3x3(ws[l], "f16mx"), bs[l], bool(relu), bool(pool), "f16mx")
real = acts[8]                                       # input of conv4_2 (layer index 8): [32][60][80][512]
rf = ops.mx_join(real)
print(f"conv4_2 input (real): shape {tuple(rf.shape)} max {rf.abs().max().item():.1f} mean {rf.mean().item():.2f} "
      f"zeros {float((rf == 0).float().mean()):.3f}")
w_rand = torch.randn((512, 512, 3, 3), generator=g, device=dev) * (2.0 / (9 * 512)) ** 0.5
b0 = torch.zeros((512,), device=dev)
rn = torch.randn((N, 60, 80, 512), generator=g, device=dev)
cases = {
    "relu(randn) * 3": torch.relu(rn) * 3.0,
    "relu(randn) * 300": torch.relu(rn) * 300.0,
    "randn * 3 (no zeros)": rn * 3.0,
    "zeros": torch.zeros_like(rn),
    "real activations": rf,
    "real activations / 64": rf / 64.0,
    "real, shuffled over pixels": rf.reshape(-1, 512)[torch.randperm(N * 4800, device=dev)].reshape(N, 60, 80, 512),
    "real magnitudes, random positions": rf.flatten()[torch.randperm(rf.numel(), device=dev)].reshape(N, 60, 80, 512),
}
for prec in ("f16mx", "bf16x3"):
    split = ops.mx_split if prec == "f16mx" else ops.x3_split
    for wname, w in (("random weights", w_rand), ("model weights", ws[8])):
        wp = ops.pack_conv3x3(w.contiguous(), prec)
        for name, xf in cases.items():
            xs = split(xf.contiguous())
            t = timeit(lambda: ops.conv3x3_nhwc(xs, wp, b0, True, False, prec))
            print(f"{prec:7s} {wname:15s} {name:36s}: {t:.3f} ms", flush=True)
