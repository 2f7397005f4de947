"""Where f16mx stops paying: replayed forward time of small problems in f16mx (split-K ring kernels) and in
bf16x3, both 1e-4 modes (diagnostic, not a pytest; the table behind VGG.effective_precision's small-problem
rule).    python tests/gpu_small_sizes.py [out.md]      python tests/gpu_small_sizes.py --loop f16mx 224 224 1
(--loop: 200 eager forwards of one configuration, for rocprofv3 --kernel-trace --stats)"""
import statistics
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import hubconf  # noqa: E402
from openibl_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
CASES = [(1, 224, 224), (2, 224, 224), (4, 224, 224), (8, 224, 224), (1, 320, 320), (1, 384, 384), (1, 480, 480),
         (1, 480, 640), (2, 480, 640), (1, 600, 800), (1, 128, 160)]


def med(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts)


def main():
    model = hubconf.vgg16_netvlad(pretrained=False)
    model.load_state_dict(synth.embednetpca_state(0))
    model = model.to(dev).eval()
    vgg = next(m for m in model.modules() if hasattr(m, 'F16MX_MIN_TILES'))
    vgg.F16MX_MIN_TILES = 0                 # f16mx whenever it can run: the rule is what is being measured
    if len(sys.argv) > 1 and sys.argv[1] == "--loop":
        prec, h, w, n = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
        model.set_precision(prec)
        x = synth.images(n, h, w, seed=5).to(dev)
        with torch.no_grad():
            for _ in range(200):
                model(x)
        torch.cuda.synchronize()
        return
    rows = []
    for (n, h, w) in CASES:
        x = synth.images(n, h, w, seed=5).to(dev)
        tiles = -(-(n * (h // 8) * (w // 8)) // 256)
        t = {}
        for prec in ("f16mx", "bf16x3"):
            model.set_precision(prec)
            with torch.no_grad():
                assert vgg.effective_precision(x) == prec
                fwd = model.graphed(x)
                t[prec] = med(lambda: fwd())
                del fwd
        rows.append((n, h, w, tiles, t["f16mx"], t["bf16x3"]))
        print(f"{n} x {h}x{w}: {tiles:3d} conv4 tiles | f16mx {t['f16mx']:.3f} ms | bf16x3 {t['bf16x3']:.3f} ms | "
              f"{t['bf16x3'] / t['f16mx']:.2f}x", flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write("| batch | image | conv4 tiles (256 pixels) | f16mx replay ms | bf16x3 replay ms | bf16x3 / f16mx |\n"
                    "|---|---|---|---|---|---|\n")
            for r in rows:
                f.write(f"| {r[0]} | {r[1]}x{r[2]} | {r[3]} | {r[4]:.3f} | {r[5]:.3f} | {r[5] / r[4]:.2f} |\n")


if __name__ == "__main__":
    main()
