"""Whole f16mx backbone under the activation-scale hook: which shifts give the same fp32 map?  (diagnostic)"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops, lib, synth  # noqa: E402
import hubconf  # noqa: E402

dev = torch.device("cuda", 0)
model = hubconf.vgg16_netvlad(pretrained=False)
model.load_state_dict(synth.embednetpca_state(0))
model = model.to(dev).eval().set_precision("f16mx")
vgg = model.base_model
hooks = lib.debug_hooks()
for shape, c in [((2, 96, 128), 1.0), ((2, 96, 128), 100.0), ((2, 480, 640), 1.0)]:
    x = (synth.images(*shape, seed=79) * c).to(dev)
    ws, bs = vgg._packed(x.device, "f16mx")
    if c != 1.0:
        bs = [b * c for b in bs]
    maps = {}
    for sh in (0, 2, 3, 4, 5, 0):
        hooks.oibl_debug_set_mx_act_shift(sh)
        maps.setdefault(sh, []).append(ops.vgg16_conv5(x, ws, bs, "f16mx").clone())
    hooks.oibl_debug_set_mx_act_shift(3)
    ref = maps[0][0]
    print(f"{shape} x{c:g}: map peak {float(ref.max()):.3g}, mean {float(ref.mean()):.3g}; shift 0 twice equal: {bool(torch.equal(maps[0][0], maps[0][1]))}")
    for sh in (2, 3, 4, 5):
        y = maps[sh][0]
        print(f"   shift {sh}: bit-equal to shift 0: {float((y == ref).float().mean()):.4f}, rel-L2 {float((y - ref).norm() / ref.norm()):.2e}")
