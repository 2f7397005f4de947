"""The f16mx stem of this build (product library: MX tails as ds_read_b64 + ds_read_b32, producer lane = halo pixel)
against the debug library compiled with -DOIBL_STEM_R6_LDS (openibl_amd/build.py, DBG_EXPERIMENT_FLAGS: tails as one
ds_read_b128, producer lanes on even / odd pixels — the conflict-free LDS pattern of tools/lds_stem_model.py): time per
launch at batch 32 x 480x640, alternating, and bit-identity of the outputs (diagnostic, not a pytest).
    python tests/gpu_stem_lds_ab.py [rounds]          (profiles/r06_d_stem_lds_ab.txt: run with the roles swapped)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import build, lib, ops  # noqa: E402

assert build.DBG_EXPERIMENT_FLAGS, "build the debug library with an experiment flag first (openibl_amd/build.py)"
dev = torch.device("cuda", 0)
N, H, W = 32, 480, 640
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn((N, 3, H, W), generator=g, device=dev)
w1 = torch.randn((64, 3, 3, 3), generator=g, device=dev) * 0.27
b1 = torch.randn((64,), generator=g, device=dev) * 0.1
w2 = torch.randn((64, 64, 3, 3), generator=g, device=dev) * 0.06
b2 = torch.randn((64,), generator=g, device=dev) * 0.1
wp = ops.pack_conv3x3(w2, "f16mx")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def timed(iters=10):
    for _ in range(3):
        ops.vgg16_stem_mx(x, w1, b1, wp, b2)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        out = ops.vgg16_stem_mx(x, w1, b1, wp, b2)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters, out


ts, outs = {"product": [], "debug library, " + " ".join(build.DBG_EXPERIMENT_FLAGS): []}, {}
for r in range(rounds):
    for name in ts:
        if name.startswith("product"):
            lib.use_product_library()
        else:
            lib.debug_hooks()
        t, out = timed()
        ts[name].append(t)
        outs[name] = out
lib.use_product_library()
a, b = outs.values()
print(f"f16mx stem, batch 32 x 480x640, {rounds} alternating rounds of 10 launches; outputs bit-identical: {torch.equal(a, b)}")
for name, v in ts.items():
    print(f"  {name:40s} " + " ".join(f"{t:.3f}" for t in v) + f"  ms   median {sorted(v)[len(v) // 2]:.3f}")
