"""Single-image latency of the descriptor path (BASELINE.json configs[0]; Tokyo 24/7 queries run
with batch size 1 and arbitrary sizes, ibl/utils/data/__init__.py:38, examples/test.py:46).

    python tests/gpu_latency.py [out.md]

Per precision and image size: milliseconds per image for
  eager      `model(x)` on a resident image, launch by launch (host launch cost included)
  replay     the same forward as two replayed hipGraphs (`model.graphed(x)`)
  api        `ibl.evaluators.extract_cnn_feature(model, pinned host image)`: H2D copy + forward +
             the extra L2 normalisation, device-synchronised (what one query costs end to end)
Medians over 40 repetitions after 5 warm-up calls; every call is followed by a device synchronise.
"""
import statistics
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import hubconf  # noqa: E402
from ibl.evaluators import extract_cnn_feature  # noqa: E402
from openibl_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)


def med(fn, reps=40, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts), min(ts)


def main():
    model = hubconf.vgg16_netvlad(pretrained=False)
    model.load_state_dict(synth.embednetpca_state(0))
    model = model.to(dev).eval()
    rows = []
    for (h, w) in ((480, 640), (479, 637), (640, 480), (224, 224)):
        x_host = synth.images(1, h, w, seed=5).pin_memory()
        x = x_host.to(dev)
        for prec in ("bf16", "f16mx", "bf16x3", "fp32"):
            model.set_precision(prec)
            with torch.no_grad():
                e_med, e_min = med(lambda: model(x))
                fwd = model.graphed(x)
                r_med, r_min = med(lambda: fwd())
                a_med, a_min = med(lambda: extract_cnn_feature(model, x_host))
                del fwd
            eff = model.base_model.effective_precision(x)      # (small f16mx problems run in bf16x3: models.py)
            if eff != prec:
                prec = f"{prec} (runs in {eff})"
            rows.append((f"{h}x{w}", prec, e_med, e_min, r_med, r_min, a_med, a_min))
            print(f"{h}x{w} {prec:7s} eager {e_med:7.3f} (min {e_min:.3f})  replay {r_med:7.3f} (min {r_min:.3f})  "
                  f"api {a_med:7.3f} (min {a_min:.3f}) ms", flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write("| image | precision | eager model(x) ms (median / min) | hipGraph replay ms | "
                    "extract_cnn_feature from pinned host ms |\n|---|---|---|---|---|\n")
            for r in rows:
                f.write(f"| {r[0]} | {r[1]} | {r[2]:.3f} / {r[3]:.3f} | {r[4]:.3f} / {r[5]:.3f} | "
                        f"{r[6]:.3f} / {r[7]:.3f} |\n")


if __name__ == "__main__":
    main()
