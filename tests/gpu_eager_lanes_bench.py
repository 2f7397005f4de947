"""images/s through ibl.evaluators.extract_features with and without hipGraph capture ("eager
extract_features within 5 % of the graphed path").  Same process, same pinned host batches, both routes interleaved:
    python tests/gpu_eager_lanes_bench.py [n_batches] [repeats]
Both routes are the two lanes of openibl_amd/extract.py; use_graphs=False launches every kernel eagerly (EagerLanes),
the range flag of a batch read from the pinned ring after later batches were enqueued."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402  (the _MemLoader of the api leg)
from ibl.evaluators import extract_features  # noqa: E402
import hubconf  # noqa: E402
from openibl_amd import synth  # noqa: E402

H, W, B = 480, 640, 32


def main():
    n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda", 0)
    model = hubconf.vgg16_netvlad(pretrained=False)
    model.load_state_dict(synth.embednetpca_state(0))
    model = model.to(dev).eval()
    names = [(f"im{i:06d}.jpg", i, 0.0, 0.0) for i in range(n_batches * B)]
    base = synth.images(B, H, W, seed=900)
    pinned = [base.roll(s, 0).contiguous().pin_memory() for s in range(3)]
    for precision in ("f16mx", "bf16"):
        model.set_precision(precision)
        ref = None
        rates = {True: [], False: []}
        for rep in range(repeats + 1):                       # pass 0 of each route is its warm-up (capture / packing)
            for graphs in (True, False):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                feats = extract_features(model, bench._MemLoader(pinned, n_batches), names, print_freq=10 ** 9,
                                         gpu=dev.index, use_graphs=graphs)
                dt = time.perf_counter() - t0
                got = torch.stack([feats[n[0]] for n in names[:3 * B]])
                if ref is None:
                    ref = got
                assert torch.equal(got, ref), "the two routes differ"
                if rep:
                    rates[graphs].append(n_batches * B / dt)
        g, e = max(rates[True]), max(rates[False])
        print(f"{precision}: graphed {g:8.1f} images/s (runs {', '.join(f'{r:.0f}' for r in rates[True])}) | "
              f"eager lanes {e:8.1f} images/s (runs {', '.join(f'{r:.0f}' for r in rates[False])}) | "
              f"eager / graphed = {e / g:.3f}; descriptors torch.equal")


if __name__ == "__main__":
    main()
