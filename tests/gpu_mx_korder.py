"""Timing experiment (diagnostic): whole forward of the real model (batch 32, 480x640) in f16mx / bf16x3 with
the K order of the implicit GEMMs forced: -1 = per-layer default, 0 = (tap, chunk), 1 = (chunk, tap)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops, synth, lib  # noqa: E402
import hubconf  # noqa: E402
dev = torch.device("cuda", 0)
m = hubconf.vgg16_netvlad(pretrained=False)
m.load_state_dict(synth.embednetpca_state(0))
m = m.to(dev).eval()
x = synth.images(32, 480, 640, seed=100).to(dev)
L = lib.debug_hooks()
for prec in ("f16mx", "bf16x3", "bf16"):
    m.set_precision(prec)
    for variant in ((0, 1) if prec == "f16mx" else (0,)):
        L.oibl_debug_set_mx_variant(variant)
        for ko in (-1, 0, 1):
            ops.set_conv_korder(ko)
            for _ in range(3):
                m(x)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(4):
                    m(x)
                e.record()
                torch.cuda.synchronize()
                ts.append(s.elapsed_time(e) / 4)
            t = sorted(ts)[2]
            print(f"{prec:7s} mx_variant {variant} korder {ko:2d}: {t:7.3f} ms per batch = {32 / t * 1e3:6.0f} images/s", flush=True)
ops.set_conv_korder(-1)
L.oibl_debug_set_mx_variant(0)
