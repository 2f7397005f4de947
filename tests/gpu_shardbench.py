"""Per-shard cost of the matching step as the gallery is sharded 1/2/4/8 ways (one GPU simulating
one rank of each world size; the collective and the merge are timed separately on the gathered
shape).  Diagnostic for the strong-scaling line of bench.py."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
Q, G, D, K = 8192, 81920, 4096, 10
g = torch.Generator(device=dev).manual_seed(7)
q = torch.nn.functional.normalize(torch.randn((Q, D), generator=g, device=dev), dim=1)
gal = torch.nn.functional.normalize(torch.randn((G, D), generator=g, device=dev), dim=1)


def timed(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


worlds = [int(w) for w in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 4, 8]
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
base = None
print(f"precision {prec}; the gallery shard is resident (ops.PreparedRows); every rank prepares the Q / W "
      f"queries it extracted (sharded.gather_prepared_queries), the exchange itself is not emulated")
q_all = ops.PreparedRows(q, prec)
for world in worlds:
    n = G // world
    shard = ops.PreparedRows(gal[:n].contiguous(), prec)
    q_mine = q[: Q // world].contiguous()

    def step():
        ops.PreparedRows(q_mine, prec)                                      # this rank's share of the queries
        return ops.sqdist_topk_prepared(q_all, shard, K, defer_check=True)  # as sharded.py
    t = timed(step)
    vals = torch.randn((Q, world * K), device=dev)
    idx = torch.randint(0, G, (Q, world * K), device=dev, dtype=torch.int32)
    tm = timed(lambda: ops.row_topk(vals, K, idx_in=idx)) if world > 1 else 0.0
    base = base or t
    print(f"world {world}: shard of {n:6d} rows: local top-k {t:6.3f} ms, merge of {world} lists {tm:5.3f} ms "
          f"-> compute-only speed-up {base / (t + tm):4.2f}x")
