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
BLOCKS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
LANES = int(sys.argv[4]) if len(sys.argv) > 4 else 2     # matrix work of sub-block b on compute stream b % LANES
print(f"precision {prec}; the gallery shard is resident (ops.PreparedRows); per step the Q / W queries of every rank "
      f"travel in {BLOCKS} sub-blocks — the all_gather of sub-block b + 1 and the list exchange + merge of sub-block b on a "
      f"second stream under the matrix work (sharded.sharded_topk_pipelined, bench.py's schedule) — both collectives "
      f"EMULATED by device copies of the gathered sizes (one GPU here: no xGMI time in these numbers)")
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
mains = [main] + [torch.cuda.Stream() for _ in range(LANES - 1)]
print(f"matrix work of sub-block b on compute stream b % {LANES}" + (": the tail round of one sub-block's filter pass and the "
      "latency-bound sample pass / selections of the next run side by side" if LANES > 1 else ""))
for world in worlds:
    n = G // world
    shard = ops.PreparedRows(gal[:n].contiguous(), prec)
    qper = Q // world
    q_mine = q[:qper].contiguous()
    sub = qper // BLOCKS if world > 1 else qper
    rows_travel = prec in ("f16r", "fp32")
    # what one sub-block's all_gather delivers: world * sub rows (fp32 rows, or prepared operand rows + norms)
    q_sub = q[: world * sub].contiguous()
    p_sub = ops.PreparedRows(q_sub, prec)
    gathered_src = q_sub if rows_travel else p_sub.operand_rows().contiguous()
    gathered_dst = torch.empty_like(gathered_src)
    lists_src = torch.randn((world, world * sub + 1, 2 * K), device=dev)
    lists_dst = torch.empty_like(lists_src)
    vals = torch.randn((world * sub, world * K), device=dev)
    idx = torch.randint(0, G, (world * sub, world * K), device=dev, dtype=torch.int32)
    if prec == "f16r":
        K2 = ops.f16r_members(K)
        f_src = torch.randn((world, world * sub + 1, 2 * K2), device=dev)
        f_dst = torch.empty_like(f_src)
        f_vals = torch.rand((world * sub, world * K2), device=dev) + 1.0      # (thr from it: keeps every local member ...
        ymax_all = torch.tensor([[1.0, 2e-4]] * world, device=dev)
        # ... so the share of the global rescore set that is THIS rank's is imposed: one member in `world` survives)
        keep_frac = world > 1
        keep_mask = (torch.arange(K2, device=dev)[None, :] % world == 0).expand(world * sub, K2)
        minus1 = torch.full((world * sub, K2), -1, dtype=torch.int32, device=dev)

    def step():
        if world == 1:
            return ops.sqdist_topk_prepared(ops.PreparedRows(q_mine, prec), shard, K, defer_check=True)
        side.wait_stream(main)
        for cs in mains[1:]:
            cs.wait_stream(main)
        evs = []
        for b in range(BLOCKS):                         # sharded.sharded_topk_pipelined with device copies as collectives
            if b == 0:
                with torch.cuda.stream(side):
                    if not rows_travel:
                        ops.PreparedRows(q_mine[:sub], prec)
                    gathered_dst.copy_(gathered_src)    # all_gather of sub-block 0 (gathered size)
                    e0 = torch.cuda.Event(); e0.record(side); evs.append(e0)
            if b + 1 < BLOCKS:
                with torch.cuda.stream(side):
                    if not rows_travel:
                        ops.PreparedRows(q_mine[:sub], prec)
                    gathered_dst.copy_(gathered_src)    # sub-block b + 1 travels under sub-block b's matrix work
                    e = torch.cuda.Event(); e.record(side); evs.append(e)
            cs = mains[b % LANES]
            cs.wait_event(evs[b])
            with torch.cuda.stream(cs):
                qb = ops.PreparedRows(q_sub, prec) if rows_travel else p_sub
                if prec == "f16r":                      # two phases: filter lists first, the rescoring behind the exchange
                    lval, lidx, ymax, flag = ops.f16r_filter_select(qb, shard, K)
                else:
                    out = ops.sqdist_topk_prepared(qb, shard, K, defer_check=True)
            side.wait_stream(cs)
            if prec == "f16r":
                with torch.cuda.stream(side):
                    f_dst.copy_(f_src)                  # all_gather of the filter lists [world][Qb + 1][2 K2]
                    thr = ops.row_topk(f_vals, K)[0][:, K - 1].contiguous()
                    if keep_frac:
                        lidx2 = torch.where(keep_mask, lidx, minus1)     # 1 / world of the members are this rank's
                    else:
                        lidx2 = lidx
                    ops.f16r_keep_members(lval, lidx2, K, thr, qb, ymax_all)
                    out = ops.f16r_rescore(qb, shard, lidx2, K)
                    lists_dst.copy_(lists_src)          # all_gather of the exact lists
                    ops.row_topk(vals, K, idx_in=idx)   # k-way merge
                continue
            with torch.cuda.stream(side):
                lists_dst.copy_(lists_src)              # all_gather of this sub-block's lists (gathered size)
                ops.row_topk(vals, K, idx_in=idx)       # k-way merge
        for cs in mains[1:]:
            main.wait_stream(cs)
        main.wait_stream(side)
        return out
    t = timed(step)
    base = base or t
    print(f"world {world}: shard of {n:6d} rows: step {t:6.3f} ms -> projected speed-up {base / t:4.2f}x "
          f"({Q * G / t / 1e6:7.1f} Gpairs/s aggregate)")
