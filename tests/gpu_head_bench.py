"""The head behind the conv5_3 map at the benchmark batch and for one image: NetVLAD (fused two-launch layer against
the five-launch path it replaces, hook) and the PCA projection (diagnostic, not a pytest).
    python tests/gpu_head_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import lib, ops, synth  # noqa: E402

dev = torch.device("cuda", 0)
sd = synth.embednetpca_state(0)
cw = sd["net_vlad.conv.weight"].reshape(64, 512).contiguous().to(dev)
cent = sd["net_vlad.centroids"].to(dev)
pw = ops.PcaWeight(sd["pca_layer.weight"].reshape(4096, 32768).contiguous().to(dev))   # (as the model holds it)
pb = sd["pca_layer.bias"].to(dev)


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


g = torch.Generator(device=dev).manual_seed(3)
for N in (32, 16, 12, 8, 4, 2, 1):
    feat = torch.randn((N, 30, 40, 512), generator=g, device=dev) * 3.0
    t_f = timed(lambda: ops.netvlad(feat, cw, cent, True, want_raw=False, want_norm=True))
    lib.debug_hooks().oibl_debug_set_netvlad_slabs(2)
    t_o = timed(lambda: ops.netvlad(feat, cw, cent, True, want_raw=False, want_norm=True))
    lib.debug_hooks().oibl_debug_set_netvlad_slabs(1)
    v = ops.netvlad(feat, cw, cent, True, want_raw=False, want_norm=True)[1]
    t_p = timed(lambda: ops.pca(v, pw, pb))
    print(f"N = {N:2d} (30 x 40 x 512 map, fp32): NetVLAD fused {t_f:6.1f} us (2 launches) | five launches {t_o:6.1f} us | "
          f"PCA 32768 -> 4096 fp32 (packed stream from 3 rows) {t_p:6.1f} us ({537e6 / t_p / 1e6:.2f} TB/s of W)", flush=True)
lib.use_product_library()
