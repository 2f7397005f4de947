"""the ring kernels re-fetch every pixel line nine times on conv4_x / conv5_x (3.2 GB for a 315 MB
input) — is that free?  The same layer on the ring (K order (tap, chunk): nine-fold fetch) and on the 8-wave halo
kernel (hook 3: the patch's halo fetched once per chunk) in a ~4 s loop each, with the socket power and clock read
from rocm-smi: milliseconds AND joules per layer (diagnostic, not a pytest).     python tests/gpu_energy_ab.py"""
import json
import subprocess
import sys
import threading
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
L = lib.debug_hooks()


class Smi(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.stop = [], False

    def run(self):
        while not self.stop:
            try:
                r = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--json"], capture_output=True,
                                   text=True, timeout=5)
                card = next(iter(json.loads(r.stdout).values()))
                pw = [float(v) for k, v in card.items() if "power" in k.lower() and "(w)" in k.lower()]
                ck = [k_ for k_ in card if "sclk clock speed" in k_.lower()]
                mhz = float(card[ck[0]].strip("()").lower().replace("mhz", "")) if ck else 0.0
                if pw:
                    self.rows.append((pw[0], mhz))
            except Exception:
                pass
            time.sleep(0.3)


g = torch.Generator(device=dev).manual_seed(5)
for cin, cout, H, W, pool, name in [(512, 512, 60, 80, 0, "conv4_2"), (512, 512, 60, 80, 1, "conv4_3"),
                                    (256, 256, 120, 160, 0, "conv3_2")]:
    x = ops.mx_split(torch.relu(torch.randn((32, H, W, cin), generator=g, device=dev)) * 3.0)
    w = ops.pack_conv3x3(torch.randn((cout, cin, 3, 3), generator=g, device=dev) * 0.02, "f16mx")
    b = torch.zeros(cout, device=dev)
    for variant, vname in ((1, "ring (nine-fold pixel fetch)"), (3, "halo (one fetch per chunk)")):
        L.oibl_debug_set_mx_variant(variant)
        for _ in range(20):
            ops.conv3x3_nhwc(x, w, b, True, bool(pool), "f16mx")
        torch.cuda.synchronize()
        smi = Smi()
        smi.start()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 4.0:
            for _ in range(200):
                ops.conv3x3_nhwc(x, w, b, True, bool(pool), "f16mx")
            torch.cuda.synchronize()
            n += 200
        dt = time.perf_counter() - t0
        smi.stop = True
        smi.join(timeout=3)
        rows = smi.rows[2:] or smi.rows          # (drop the ramp-up samples)
        pw = sum(r[0] for r in rows) / max(len(rows), 1)
        mhz = sum(r[1] for r in rows) / max(len(rows), 1)
        ms = dt / n * 1e3
        print(f"{name} {cin}->{cout} {H}x{W}{' pool' if pool else ''} [{vname}]: {ms:.3f} ms per layer, {pw:.0f} W at "
              f"{mhz:.0f} MHz ({len(rows)} samples) = {pw * ms / 1e3:.3f} J per layer", flush=True)
L.oibl_debug_set_mx_variant(0)
lib.use_product_library()
