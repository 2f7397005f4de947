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


# xGMI time is CHARGED (round 6): every emulated collective also holds the exchange stream for
# (bytes this rank receives) / (7 links x 50 GB/s) — torch.cuda._sleep, calibrated against HIP events below.  The
# device copy of the gathered size stays (it is what the received data costs in HBM).  LINK_GBS = 0 switches it off.
LINK_GBS = 350.0
_cal = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
torch.cuda._sleep(1000)
torch.cuda.synchronize()
_cal[0].record()
torch.cuda._sleep(20_000_000)
_cal[1].record()
torch.cuda.synchronize()
SLEEP_CYCLES_PER_MS = 20_000_000 / _cal[0].elapsed_time(_cal[1])


def link(nbytes):
    """hold the current stream for the time `nbytes` take to arrive over the rank's seven xGMI links"""
    if LINK_GBS > 0 and nbytes > 0:
        torch.cuda._sleep(int(nbytes / (LINK_GBS * 1e9) * 1e3 * SLEEP_CYCLES_PER_MS))


worlds = [int(w) for w in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 4, 8]
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
base = None
BLOCKS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
LANES = int(sys.argv[4]) if len(sys.argv) > 4 else 2     # matrix work of sub-block b on compute stream b % LANES
SLICED = (sys.argv[5] != "0") if len(sys.argv) > 5 else True   # query-sliced post-processing (sharded.py, round 6)
print(f"precision {prec}; the gallery shard is resident (ops.PreparedRows); per step the Q / W queries of every rank "
      f"travel in {BLOCKS} sub-blocks — the all_gather of sub-block b + 1 and the list exchange + merge of sub-block b on a "
      f"second stream under the matrix work (sharded.sharded_topk_pipelined, bench.py's schedule) — every collective "
      f"EMULATED by a device copy of the received size + a hold of the exchange stream for received bytes / "
      f"{LINK_GBS:.0f} GB/s (one GPU here; 7 xGMI links x 50 GB/s); post-processing "
      f"{'SLICED: all_to_all of the lists, merge of Q / W queries per rank, all_gather of the merged slices' if SLICED else 'replicated: all_gather of every list, every rank merges all Q queries (rounds 1-5)'}")
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

    qrow_bytes = gathered_src.numel() * gathered_src.element_size() // max(1, world * sub)
    recv = (world - 1) / world                       # share of a gathered / exchanged buffer that crosses the links
    per = -(-(world * sub) // world)                 # queries of a sub-block this rank post-processes when SLICED

    def exchange_lists(vals_, idx_):
        """the exact lists of one sub-block: exchange + merge (+ final all_gather when sliced)"""
        if SLICED:
            lists_dst[0].copy_(lists_src[0])        # all_to_all: [world * sub][2K] out, as much in
            link(recv * world * sub * 2 * K * 4)
            ops.row_topk(vals_[:per], K, idx_in=idx_[:per])     # merge of Q / W queries
            lists_dst[1].copy_(lists_src[1])        # all_gather of the merged slices
            link(recv * world * sub * 2 * K * 4)
        else:
            lists_dst.copy_(lists_src)              # all_gather of every rank's lists
            link(recv * world * world * sub * 2 * K * 4)
            ops.row_topk(vals_, K, idx_in=idx_)     # k-way merge of all queries

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
                    link(recv * world * sub * qrow_bytes)
                    e0 = torch.cuda.Event(); e0.record(side); evs.append(e0)
            if b + 1 < BLOCKS:
                with torch.cuda.stream(side):
                    if not rows_travel:
                        ops.PreparedRows(q_mine[:sub], prec)
                    gathered_dst.copy_(gathered_src)    # sub-block b + 1 travels under sub-block b's matrix work
                    link(recv * world * sub * qrow_bytes)
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
                    if SLICED:
                        f_dst[0, :, :K2].copy_(f_src[0, :, :K2])     # all_to_all of the filter VALUES [world * sub][K2]
                        link(recv * world * sub * K2 * 4)
                        thr_s = ops.row_topk(f_vals[:per], K)[0][:, K - 1].contiguous()   # thresholds of Q / W queries
                        link(recv * world * sub * 4)                 # all_gather of the thresholds: 4 bytes per query
                        thr = thr_s.repeat(world)[: world * sub].contiguous()
                    else:
                        f_dst[:, :, :K2].copy_(f_src[:, :, :K2])     # all_gather of the filter values [world][Qb + 1][K2]
                        link(recv * world * world * sub * K2 * 4)
                        thr = ops.row_topk(f_vals, K)[0][:, K - 1].contiguous()
                    if keep_frac:
                        lidx2 = torch.where(keep_mask, lidx, minus1)     # 1 / world of the members are this rank's
                    else:
                        lidx2 = lidx
                    ops.f16r_keep_members(lval, lidx2, K, thr, qb, ymax_all)
                    out = ops.f16r_rescore(qb, shard, lidx2, K)
                    exchange_lists(vals, idx)
                continue
            with torch.cuda.stream(side):
                exchange_lists(vals, idx)
        for cs in mains[1:]:
            main.wait_stream(cs)
        main.wait_stream(side)
        return out
    t = timed(step)
    base = base or t
    print(f"world {world}: shard of {n:6d} rows: step {t:6.3f} ms -> projected speed-up {base / t:4.2f}x "
          f"({Q * G / t / 1e6:7.1f} Gpairs/s aggregate)")
