"""PCA 32768 -> 4096 in fp32 (537 MB of W) for a few batch sizes: streaming kernel (N <= 8) against the MFMA tile.
    python tests/gpu_pca_bench.py"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops, lib, synth  # noqa: E402

dev = torch.device("cuda", 0)
L = lib.debug_hooks()
sd = synth.pca_state(0)
w = sd["pca_layer.weight"].reshape(4096, 32768).to(dev)
b = sd["pca_layer.bias"].to(dev)
for N in (1, 2, 4, 8, 32):
    v = torch.nn.functional.normalize(torch.randn((N, 32768), device=dev), dim=1)
    row = []
    for small in (1, 0):
        L.oibl_debug_set_pca_small(small)
        for _ in range(5):
            ops.pca(v, w, b)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50):
            ops.pca(v, w, b)
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) / 50
        row.append(f"{'streaming' if small else 'MFMA tile'} {t * 1e3:7.1f} us ({w.numel() * 4 / t / 1e9:5.2f} TB/s)")
    print(f"N={N:2d}: " + " | ".join(row))
L.oibl_debug_set_pca_small(1)
pw = ops.PcaWeight(w)
for N in (1, 3, 8, 16, 32):
    v = torch.nn.functional.normalize(torch.randn((N, 32768), device=dev), dim=1)
    row = []
    for variant in (1, 3, 2, 1, 3):
        L.oibl_debug_set_pca_stream(variant)
        old = ops.PCA_STREAM_MIN_ROWS
        ops.PCA_STREAM_MIN_ROWS = 1
        for _ in range(5):
            ops.pca(v, pw, b)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50):
            ops.pca(v, pw, b)
        e.record()
        torch.cuda.synchronize()
        ops.PCA_STREAM_MIN_ROWS = old
        t = s.elapsed_time(e) / 50
        name = {1: "16-wave workgroup, two K parts (16 partials)", 3: "8 loads x 4 waves per SIMD (32 partials)",
                2: "16 loads x 2 waves per SIMD (32 partials)"}[variant]
        row.append(f"{name} {t * 1e3:7.1f} us ({w.numel() * 4 / t / 1e9:5.2f} TB/s)")
    print(f"N={N:2d}: " + " | ".join(row))
L.oibl_debug_set_pca_stream(1)
