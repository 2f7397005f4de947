"""Where does the fixed ~2 ms of a timed replay loop go?  (diagnostic)"""
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
x = synth.images(4, 480, 640, seed=1).repeat(8, 1, 1, 1).to(dev)
with torch.no_grad():
    for _ in range(3):
        model(x)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
    for pipe, use_ev in ((True, False), (True, True), (False, True)):
        fwd = model.graphed(x, pipeline=pipe)
        for _ in range(4):
            fwd()
        torch.cuda.synchronize()
        print("pipeline", pipe, "events", use_ev)
        for K in (1, 2, 4, 8, 16, 32, 64):
            best = None
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                cpu = []
                for k in range(K):
                    c0 = time.perf_counter()
                    fwd(events=evs[k]) if use_ev else fwd()
                    cpu.append(time.perf_counter() - c0)
                t_issue = time.perf_counter() - t0
                torch.cuda.synchronize()
                t = time.perf_counter() - t0
                if best is None or t < best[0]:
                    best = (t, t_issue, cpu[0], sum(cpu[1:]) / max(len(cpu) - 1, 1))
            print(f"  K={K:3d}: total {best[0] * 1e3:8.3f} ms = {best[0] / K * 1e3:6.3f} ms/step; issue loop "
                  f"{best[1] * 1e3:7.3f} ms; first fwd() {best[2] * 1e3:6.3f} ms, later {best[3] * 1e3:6.3f} ms")

    # idle-gap sensitivity: the GPU drops its clocks when idle for a few ms
    fwd = model.graphed(x, pipeline=True)
    for _ in range(4):
        fwd()
    torch.cuda.synchronize()

    def timed(K, gap, pre):
        torch.cuda.synchronize()
        if gap:
            time.sleep(gap)
        for _ in range(pre):
            fwd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            fwd()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    for gap, pre in ((0, 0), (0.002, 0), (0.02, 0), (0.2, 0), (0.2, 1), (0.2, 3), (0.2, 10), (0, 0)):
        ts = [timed(20, gap, pre) for _ in range(3)]
        print(f"  K=20 after {gap * 1e3:5.0f} ms idle + {pre:2d} untimed replays: " +
              ", ".join(f"{t:7.3f}" for t in ts) + " ms")
