"""Experiment: two independent replay pipelines (own streams, own workspaces) in ONE process against
one pipelined replay — does a second HW queue back-fill the partial rounds of the first?
    python tests/gpu_dual_stream.py [precision] [lanes ...]      (env OIBL_KORDER / OIBL_RASTER: hooks)"""
import os
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import hubconf  # noqa: E402
from openibl_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
model = hubconf.vgg16_netvlad(pretrained=False)
model.load_state_dict(synth.embednetpca_state(0))
model = model.to(dev).eval()
model.set_precision(prec)
from openibl_amd import ops  # noqa: E402
if "OIBL_KORDER" in os.environ:
    ops.set_conv_korder(int(os.environ["OIBL_KORDER"]))
if "OIBL_RASTER" in os.environ:
    ops.set_ring_raster(int(os.environ["OIBL_RASTER"]))
x = synth.images(32, 480, 640, seed=100).contiguous().to(dev)
bb, head = model.base_model.features_nhwc, model.head_from_features


def run(lanes, steps=40, warm=6):
    if os.environ.get("OIBL_PRIO") == "1":      # lane i at priority -(i % 2): distinct queue pools
        streams = [torch.cuda.Stream(device=dev, priority=-(i % 2)) for i in range(lanes)]
    elif os.environ.get("OIBL_PRIO") == "reuse":
        global _POOL
        try:
            _POOL
        except NameError:
            _POOL = [torch.cuda.Stream(device=dev) for _ in range(4)]
        streams = _POOL[:lanes]
    else:
        streams = [torch.cuda.Stream(device=dev) for _ in range(lanes)]
    graphs, outs, keep = [], [], []
    with torch.no_grad():
        head(bb(x))
        torch.cuda.synchronize()
        for s in streams:
            xin = x.clone()
            g = torch.cuda.CUDAGraph()
            # capturing ON the lane's stream keys its workspaces by that stream: one set per lane
            with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                o = head(bb(xin))
            graphs.append(g)
            outs.append(o)
            keep.append(xin)
    torch.cuda.synchronize()

    def go(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % lanes]):
                graphs[i % lanes].replay()
    go(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ref = model(x)
    ok = all(torch.equal(o, ref) for o in outs)
    print(f"{prec}: {lanes} lane(s): {32 * steps / dt:8.1f} images/s  ({dt / steps * 1e3:.3f} ms per batch)  equal to model(x): {ok}",
          flush=True)


for lanes in ([int(a) for a in sys.argv[2:]] or [1, 2, 3, 1, 2]):
    run(lanes)
