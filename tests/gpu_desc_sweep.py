"""Random-shape sweep of the whole descriptor (f16mx backbone with its activation scale, row sub-ranges / split-K on
small problems, NetVLAD, PCA) against the fp64 oracle (diagnostic; the oracle runs on the box's host cores):
batch 1-4, images from 33 x 47 to ~300 x 400 with odd sides, at unit and at the reference's input scale.
    python tests/gpu_desc_sweep.py [cases=24] [first seed=0]"""
import copy
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import hubconf  # noqa: E402
from openibl_amd import synth  # noqa: E402
from oracle import descriptor as od  # noqa: E402

dev = torch.device("cuda", 0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sd1 = synth.embednetpca_state(0)


def scaled(sd, c):                      # activations x c (tests/test_gpu_range.py)
    out = copy.copy(sd)
    for k, v in sd.items():
        if k.startswith("base_model.base.") and k.endswith(".bias"):
            out[k] = v * c
    return out


models = {}
for c in (1.0, 100.0):
    m = hubconf.vgg16_netvlad(pretrained=False)
    m.load_state_dict(scaled(sd1, c))
    m = m.to(dev).eval().set_precision("f16mx")
    m.base_model.F16MX_MIN_TILES = 0    # every problem on the f16mx kernels
    models[c] = m
bad = 0
t0 = time.time()
for seed in range(first, first + cases):
    g = torch.Generator().manual_seed(30_000 + seed)
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))      # noqa: E731
    c = (1.0, 100.0)[seed % 2]
    big = seed % 6 == 5
    N, H, W = r(1, 4), (r(200, 300) if big else r(33, 160)), (r(250, 400) if big else r(47, 200))
    x = synth.images(N, H, W, seed=500 + seed) * c
    with torch.no_grad():
        want = od.embednetpca(x, scaled(sd1, c), dtype=torch.float64)
        model = models[c]
        before = model.base_model.range_fallbacks
        got = model(x.to(dev)).cpu().double()
    per = ((got - want).norm(dim=1) / want.norm(dim=1))
    ok = bool((per < 1e-4).all()) and model.base_model.range_fallbacks == before
    bad += int(not ok)
    print(f"seed {seed:3d} activations x {c:g}: N={N} {H:3d}x{W:3d}: descriptor rel-L2 per image max {float(per.max()):.2e} "
          f"{'ok' if ok else 'FAILED'}", flush=True)
print(f"{cases} cases, {bad} failed, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
