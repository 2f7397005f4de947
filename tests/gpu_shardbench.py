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
print(f"precision {prec}; the gallery shard is resident (ops.PreparedRows); per step every rank prepares the Q / W "
      f"queries it extracted, the two collectives of sharded.py (prepared queries; per-block top-k lists) are "
      f"EMULATED by device copies of the gathered sizes (one GPU here: no xGMI latency in these numbers), the local "
      f"top-k runs in 2 query blocks with the exchange + merge of block b on a second stream (bench.py's schedule)")
q_all = ops.PreparedRows(q, prec)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
half = Q // 2
q_blocks = [ops.PreparedRows.from_parts(q_all.operand_rows()[lo:lo + half], q_all.norms[lo:lo + half], D, prec)
            for lo in (0, half)]
for world in worlds:
    n = G // world
    shard = ops.PreparedRows(gal[:n].contiguous(), prec)
    q_mine = q[: Q // world].contiguous()
    rows_src = q_all.operand_rows()
    rows_dst = torch.empty_like(rows_src)
    lists_src = torch.randn((world, half + 1, 2 * K), device=dev)
    lists_dst = torch.empty_like(lists_src)
    vals = torch.randn((half, world * K), device=dev)
    idx = torch.randint(0, G, (half, world * K), device=dev, dtype=torch.int32)

    def step():
        ops.PreparedRows(q_mine, prec)                     # this rank's share of the queries
        if world > 1:
            rows_dst.copy_(rows_src)                       # all_gather of the prepared queries (gathered size)
        for qb in q_blocks if world > 1 else [q_all]:
            out = ops.sqdist_topk_prepared(qb, shard, K, defer_check=True)      # as sharded.py
            if world > 1:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    lists_dst.copy_(lists_src)             # all_gather of this block's lists (gathered size)
                    ops.row_topk(vals, K, idx_in=idx)      # k-way merge
        if world > 1:
            main.wait_stream(side)
        return out
    t = timed(step)
    base = base or t
    print(f"world {world}: shard of {n:6d} rows: step {t:6.3f} ms -> projected speed-up {base / t:4.2f}x "
          f"({Q * G / t / 1e6:7.1f} Gpairs/s aggregate)")
