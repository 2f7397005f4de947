"""PCIe-inclusive descriptor rate: the reference's extract_cnn_feature boundary hands over a HOST
batch (evaluators.py:24).  For the loader's normalised fp32 NCHW batch (118 MB per 32 images) and
for raw uint8 NHWC images (29 MB; ToTensor + Normalize folded into the first kernel) measures
(a) the H2D copy of one pinned batch, (b) copy + forward back to back on one stream, (c) copy of
batch i+1 on a side stream overlapped with the forward of batch i."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import hubconf  # noqa: E402
from openibl_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
model = hubconf.vgg16_netvlad()
model.load_state_dict(synth.embednetpca_state(0))
model = model.to(dev).eval().set_precision("bf16")
N = 32


def measure(tag, host):
    nbytes = host[0].numel() * host[0].element_size()
    x = host[0].to(dev)
    for _ in range(3):
        model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10):
        x = host[i & 1].to(dev, non_blocking=True)
    torch.cuda.synchronize()
    h2d = (time.perf_counter() - t0) / 10
    print(f"[{tag}] H2D of one batch ({nbytes / 1e6:.0f} MB pinned): {h2d * 1e3:.2f} ms = {nbytes / h2d / 1e9:.1f} GB/s")
    t0 = time.perf_counter()
    for i in range(10):
        out = model(host[i & 1].to(dev, non_blocking=True))
    torch.cuda.synchronize()
    serial = (time.perf_counter() - t0) / 10
    print(f"[{tag}] copy + forward, one stream: {serial * 1e3:.2f} ms/batch = {N / serial:.0f} images/s")
    side = torch.cuda.Stream()
    bufs = [torch.empty_like(x), torch.empty_like(x)]
    ev = [torch.cuda.Event(), torch.cuda.Event()]
    with torch.cuda.stream(side):
        bufs[0].copy_(host[0], non_blocking=True)
        ev[0].record(side)
    t0 = time.perf_counter()
    for i in range(10):
        cur = i & 1
        torch.cuda.current_stream().wait_event(ev[cur])
        with torch.cuda.stream(side):
            bufs[cur ^ 1].copy_(host[cur ^ 1], non_blocking=True)
            ev[cur ^ 1].record(side)
        out = model(bufs[cur])
    torch.cuda.synchronize()
    ovl = (time.perf_counter() - t0) / 10
    print(f"[{tag}] copy overlapped with forward (2 streams): {ovl * 1e3:.2f} ms/batch = {N / ovl:.0f} images/s")
    return out


measure("fp32 NCHW", [synth.images(4, 480, 640, seed=s).repeat(8, 1, 1, 1).pin_memory() for s in (1, 2)])
g = torch.Generator().manual_seed(3)
measure("uint8 NHWC", [torch.randint(0, 256, (N, 480, 640, 3), generator=g, dtype=torch.uint8).pin_memory()
                       for _ in (1, 2)])
