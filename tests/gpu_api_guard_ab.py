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


from openibl_amd import lib  # noqa: E402
h = lib.debug_hooks()
run(3)
run(48)            # (the first extraction of a process is slower: not one of the cases)
for guard, bar1, splitk in ((True, 1, 1), (False, 1, 1), (True, 1, 1), (False, 1, 1), (True, 1, 1), (False, 1, 1)):
    extract.GUARD_REPLAYS = guard
    h.oibl_debug_set_ring_bar1(bar1)
    h.oibl_debug_set_mx_splitk(splitk)
    extract.release_graphs(model)
    run(3)
    rate, _ = run(48)
    # the same batches already resident (no H2D)
    res = [p_.to(dev) for p_ in pinned]
    host = pinned[:]
    pinned[:] = res
    extract.release_graphs(model)
    run(3)
    rate_res, _ = run(48)
    pinned[:] = host
    print(f"guard {guard} bar1 {bar1} splitk {splitk}: {rate:7.1f} images/s from pinned fp32 host batches, {rate_res:7.1f} from "
          f"resident batches ({100 * rate / rate_res:.1f} %)", flush=True)
dist.destroy_process_group()
