"""Two lanes against three for the replayed forward (batch 32 of 480x640, inputs resident; same process, alternating):
    python tests/gpu_lanes_ab.py [steps] [rounds]
A lane is a stream with its own graphs, feature maps and workspaces (openibl_amd/extract.py); step i runs on lane
i % L.  Two lanes back-fill each other's partial tile rounds and heads (+4 % over one lane); this asks whether a
third finds anything left to fill."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import hubconf  # noqa: E402
from openibl_amd import extract, synth  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda", 0)
    model = hubconf.vgg16_netvlad(pretrained=False)
    model.load_state_dict(synth.embednetpca_state(0))
    model = model.to(dev).eval()
    x = synth.images(32, 480, 640, seed=3).to(dev)
    extract.NUM_LANES = 3            # the lane pool of the process is created once: with three streams
    extract._lane_streams(dev)
    for precision in ("f16mx", "bf16"):
        model.set_precision(precision)
        with torch.no_grad():
            ref = model(x).clone()
            fwds = {}
            for lanes in (1, 2, 3):
                extract.NUM_LANES = lanes
                fwds[lanes] = model.graphed(x, pipeline=lanes > 1)
            rates = {k: [] for k in fwds}
            for r in range(rounds + 1):
                for lanes, fwd in fwds.items():
                    for _ in range(6):
                        fwd()
                    fwd.wait()
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        out = fwd()
                    fwd.wait()
                    torch.cuda.synchronize(dev)
                    dt = time.perf_counter() - t0
                    assert torch.equal(out, ref)
                    if r:
                        rates[lanes].append(32 * steps / dt)
        print(precision + ": " + " | ".join(
            f"{k} lane(s) {max(v):7.1f} images/s (runs {', '.join(f'{a:.0f}' for a in v)})" for k, v in rates.items()))


if __name__ == "__main__":
    main()
