"""What the f16mx range guard costs the API path (diagnostic, not a pytest): extract_features from pinned fp32 /
uint8 host batches with and without the guard on the replayed forwards, host time per call.
    python tests/gpu_api_guard_ab.py"""
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import hubconf  # noqa: E402
from openibl_amd import extract, synth  # noqa: E402

dev = torch.device("cuda", 0)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29544", rank=0, world_size=1)
model = hubconf.vgg16_netvlad(pretrained=False)
model.load_state_dict(synth.embednetpca_state(0))
model = model.to(dev).eval().set_precision("f16mx")
base = synth.images(32, 480, 640, seed=900)
pinned = [base.roll(s, 0).contiguous().pin_memory() for s in range(3)]


class Loader:
    def __init__(self, n):
        self.n = n
        self.sampler = range(n * 32)

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            yield (pinned[i % 3], None)


def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = extract.extract_descriptors(model, Loader(n), gpu=0, print_freq=10 ** 9)
    torch.cuda.synchronize()
    return n * 32 / (time.perf_counter() - t0), out


for guard in (True, False, True, False):
    extract.GUARD_REPLAYS = guard
    extract.release_graphs(model)
    run(3)
    rate, _ = run(48)
    # host time per replayed call
    core = extract.unwrap_model(model)
    fwd = next(iter(extract._GRAPH_STORES[core][1].values()))
    ts = []
    final = torch.empty((32 * 24, 4096), device=dev)
    torch.cuda.synchronize()
    for i in range(24):
        t0 = time.perf_counter()
        fwd(pinned[i % 3], dest=final[32 * i:32 * i + 32])
        ts.append((time.perf_counter() - t0) * 1e3)
    fwd.wait()
    torch.cuda.synchronize()
    ts.sort()
    print(f"guard {guard}: {rate:7.1f} images/s through extract_descriptors (48 pinned fp32 batches); host ms per "
          f"replayed call: median {ts[len(ts) // 2]:.3f}, max {ts[-1]:.3f}", flush=True)
dist.destroy_process_group()
