"""f16mx distance kernel: LDS-DMA issue inside the COMPUTE segments (RING_MX, the default since round 3) against
issue in the LOAD segments (RING_MX_EARLY, what the convolutions use), both on the one-barrier schedule
(diagnostic, not a pytest).     python tests/gpu_match_early_ab.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
h = lib.debug_hooks()
Q, G, D, K = 8192, 81920, 4096, 10
g = torch.Generator(device=dev).manual_seed(7)
q = torch.nn.functional.normalize(torch.randn((Q, D), generator=g, device=dev), dim=1)
gal = torch.nn.functional.normalize(torch.randn((G, D), generator=g, device=dev), dim=1)
qp, gp = ops.PreparedRows(q, "f16mx"), ops.PreparedRows(gal, "f16mx")


def timed(fn, iters=3, rounds=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / iters)
    return sorted(ts)[len(ts) // 2]


res, t = {}, {}
for early in (0, 1, 0, 1):
    h.oibl_debug_set_match_mx_early(early)
    res[early] = ops.sqdist_topk_prepared(qp, gp, K)
    t.setdefault(early, []).append(timed(lambda: ops.sqdist_topk_prepared(qp, gp, K, defer_check=True)))
same = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
m = {}
for early in (0, 1):
    h.oibl_debug_set_match_mx_early(early)
    m[early] = ops.pairwise_sqdist(q[:2048].contiguous(), gal[:20000].contiguous(), "f16mx")
print(f"f16mx 8192 x 81920 x 4096 + top-10: LDS-DMA late {t[0]} ms | early {t[1]} ms | same lists: {same} | "
      f"same matrix: {torch.equal(m[0], m[1])}")
h.oibl_debug_set_match_mx_early(0)
