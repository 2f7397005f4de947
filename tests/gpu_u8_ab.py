"""uint8 input of the bf16x3 / f16mx backbone (diagnostic, not a pytest): the fused uint8 stems against the
normalising pass + fp32-input stems (test hook), and against normalised fp32 input, batch 32 at 480x640.
    python tests/gpu_u8_ab.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import hubconf  # noqa: E402
from openibl_amd import lib, synth  # noqa: E402

dev = torch.device("cuda", 0)
h = lib.debug_hooks()
model = hubconf.vgg16_netvlad(pretrained=False)
model.load_state_dict(synth.embednetpca_state(0))
model = model.to(dev).eval()
g = torch.Generator().manual_seed(1)
u8 = torch.randint(0, 256, (32, 480, 640, 3), generator=g, dtype=torch.uint8).to(dev)
x = synth.images(32, 480, 640, seed=1).to(dev)


def timed(fn, iters=6, rounds=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / iters)
    return sorted(ts)[len(ts) // 2]


for prec in ("f16mx", "bf16x3"):
    model.set_precision(prec)
    vgg = model.base_model
    t_f32 = timed(lambda: vgg.features_nhwc(x))
    h.oibl_debug_set_stem_u8(1)
    t_fused = timed(lambda: vgg.features_nhwc(u8))
    h.oibl_debug_set_stem_u8(0)
    t_pass = timed(lambda: vgg.features_nhwc(u8))
    h.oibl_debug_set_stem_u8(1)
    print(f"{prec}: backbone, batch 32 (eager, one stream): fp32 input {t_f32:.3f} ms | uint8, fused stem {t_fused:.3f} ms | "
          f"uint8, normalising pass + stem {t_pass:.3f} ms", flush=True)
