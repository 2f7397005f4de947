"""Gallery-sharded matching across the GPUs of a node.

The reference all-gathers every descriptor to every rank and computes the full distance matrix
redundantly on each host CPU (ibl/evaluators.py:76-101, 105-130).  Here the gallery never moves:
rank r keeps the contiguous slice of gallery descriptors it extracted (the slice the reference's
DistributedSliceSampler hands it, ibl/utils/data/sampler.py:208-214), computes
[all queries] x [its slice] distances + a local top-k with GLOBAL gallery indices on its own GPU,
and only the (value, index) lists cross xGMI:

    all_gather(queries)            <= Q x d fp32                   (skipped if already replicated)
    all_gather(top-k values, idx)  =  Q x k x 8 bytes per rank
    k-way merge on every rank      (same top-k kernel, fed the gathered index lists)

One process per GPU, torch.distributed backend "nccl" (= RCCL on ROCm).  The local compute steps
are injectable so that the shard / gather / merge logic is testable on CPU with gloo.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

from . import ops


def _world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def slice_bounds(length: int, rank: int, world_size: int) -> Tuple[int, int, int]:
    """Contiguous slice [start, start + per) of a dataset of `length` items owned by `rank`, as
    DistributedSliceSampler deals them (per = ceil(length / world_size); the tail of the last
    slices wraps around to the first items and is padding).  Returns (start, per, n_valid)."""
    per = -(-length // world_size)
    start = rank * per
    n_valid = max(0, min(per, length - start))
    return start, per, n_valid


def hip_local_topk(q: torch.Tensor, g: torch.Tensor, k: int, index_base: int, precision,
                   exact: bool = False):
    """[Q][d] x [n][d] -> k nearest of the local slice per query (HIP kernels), plus the device
    flag of the fused path (candidate-list overflow: the lists are then undefined and the call must
    be repeated with exact=True).  Nothing here synchronises with the host."""
    Q = q.shape[0]
    if g.shape[0] == 0:
        return (torch.full((Q, k), float("inf"), device=q.device),
                torch.full((Q, k), -1, dtype=torch.int32, device=q.device),
                torch.zeros(1, dtype=torch.int32, device=q.device))
    if isinstance(g, ops.PreparedRows):      # resident gallery shard: its norm / operand pass is done
        qp = q if isinstance(q, ops.PreparedRows) else ops.PreparedRows(q, g.precision)
        return ops.sqdist_topk_prepared(qp, g, k, index_base=index_base, exact=exact, defer_check=True)
    return ops.sqdist_topk(q, g, k, index_base=index_base, precision=precision, exact=exact,
                           defer_check=True)


def hip_merge_topk(vals: torch.Tensor, idx: torch.Tensor, k: int):
    return ops.row_topk(vals.contiguous(), k, idx_in=idx.contiguous())


class HipF16rStages:
    """The two stages of the f16r top-k on this rank's shard (ops.f16r_*), as sharded_topk uses them."""

    @staticmethod
    def members(k):
        return ops.f16r_members(k)

    @staticmethod
    def filter_select(q, g, k, index_base):
        """-> (lval [Q][K2], lidx [Q][K2], ymax [2], flag [1]); a shard too small for the fused path answers with
        EXACT distances in the same form (error 0 <= any bound), its maxima taken from its prepared rows."""
        Q, n = q.shape[0], g.shape[0]
        K2 = ops.f16r_members(k)
        dev = q.device
        if n == 0:          # (an empty shard: no candidates, no maxima)
            return (torch.full((Q, K2), float("inf"), device=dev), torch.full((Q, K2), -1, dtype=torch.int32, device=dev),
                    torch.zeros(2, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))
        if ops.f16r_fused(Q, n, q.shape[1], k):
            return ops.f16r_filter_select(q, g, k, index_base)
        v, i = ops.sqdist_topk_prepared(q, g, K2, index_base=index_base, exact=True)
        return v, i, torch.stack([g.aux[:, 1].max(), g.aux[:, 2].max()]), torch.zeros(1, dtype=torch.int32, device=dev)

    @staticmethod
    def kth(vals, k):
        """k-th smallest of every row (+inf when a row has fewer than k finite entries)"""
        return ops.row_topk(vals.contiguous(), k)[0][:, k - 1].contiguous()

    @staticmethod
    def keep_members(lval, lidx, k, thr, q, ymax_all):
        ops.f16r_keep_members(lval, lidx, k, thr, q, ymax_all)

    @staticmethod
    def rescore(q, g, lidx, k, index_base):
        if g.shape[0] == 0:
            return (torch.full((q.shape[0], k), float("inf"), device=q.device),
                    torch.full((q.shape[0], k), -1, dtype=torch.int32, device=q.device))
        return ops.f16r_rescore(q, g, lidx, k, index_base)


def all_gather_rows(x: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate equally-shaped per-rank tensors along dim 0 (one all_gather)."""
    rank, world = _world(group)
    if world == 1:
        return x
    out = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(out, x.contiguous(), group=group)
    return torch.cat(out, dim=0)


def gather_prepared_queries(q_local: torch.Tensor, n_total: int, precision, group=None,
                            prepare_fn: Optional[Callable] = None):
    """Every rank prepares the queries IT extracted (norms + the rows the contraction reads:
    oibl_match_prepare on Q / W rows) and the prepared parts are all-gathered — instead of gathering
    fp32 rows and letting every rank prepare all Q of them.  Half the bytes over xGMI in bf16 (8 KB
    instead of 16 KB per 4096-d query), and the per-rank preparation cost shrinks with the world size.
    Returns the prepared query set (first n_total rows: the wrap-around padding of the last slices
    is dropped), identical on every rank."""
    rank, world = _world(group)
    if prepare_fn is None and ops.precision_code(precision) == ops.F16R:
        # f16r: a prepared query is [fp16 row | 16 bytes | fp32 row] and the fp32 row determines the rest — the fp32
        # rows travel (16 KB per 4096-d query, what f16mx ships) and every rank prepares the gathered set (one pass
        # over Q rows: ~50 us for 8192, bit-identical to preparing them where they were extracted)
        rows32 = q_local.contiguous() if world == 1 else all_gather_rows(q_local.contiguous(), group)[:n_total]
        return ops.PreparedRows(rows32[:n_total], precision)
    prepare_fn = prepare_fn or (lambda x: ops.PreparedRows(x, precision))
    p = prepare_fn(q_local.contiguous())
    if world == 1 and int(p.norms.shape[0]) == n_total:
        return p                                  # nothing to exchange
    rows = all_gather_rows(p.operand_rows(), group)[:n_total]
    norms = all_gather_rows(p.norms, group)[:n_total]
    return type(p).from_parts(rows, norms, int(q_local.shape[1]), precision)


def _query_block(q, lo: int, hi: int):
    """Rows [lo, hi) of a query set given as a tensor or as ops.PreparedRows (views, no copies)."""
    if callable(getattr(q, "rows", None)):   # ops.PreparedRows: every part sliced in place
        return q.rows(lo, hi)
    if hasattr(q, "operand_rows"):           # a stand-in with the exchange protocol only (CPU tests)
        return type(q).from_parts(q.operand_rows()[lo:hi], q.norms[lo:hi], q.shape[1], getattr(q, "precision", None))
    return q[lo:hi]


SLICED_POSTPROCESSING = True    # False: the all_gather + replicated merge of rounds 1-5 (tests compare the two)


def _make_stages(g_local, k, index_base, precision, group, local_topk_fn, merge_fn, f16r_stages, sliced=None):
    """(stage_main, stage_side, gather_and_merge) of one rank: what runs on the caller's stream per query block (the
    matrix work) and what runs behind it (the exchange(s) + merge)."""
    use_f16r = f16r_stages is not None or (local_topk_fn is None and ops.precision_code(precision) == ops.F16R)
    if use_f16r and f16r_stages is None:
        f16r_stages = HipF16rStages
    local_topk_fn = local_topk_fn or hip_local_topk
    merge_fn = merge_fn or hip_merge_topk
    rank, world = _world(group)
    sliced = SLICED_POSTPROCESSING if sliced is None else bool(sliced)

    def gather_and_merge(v, i, flag):
        if world == 1:
            return v, i, flag
        if sliced:
            return exchange_and_merge_sliced(v, i, flag)
        # one collective for everything: the int32 indices travel as the bit pattern of a float32
        # next to the values, the overflow flag as one more row ([Q + 1][2k] per rank; nothing
        # computes on them in transit), so every rank sees every rank's flag
        Q = v.shape[0]
        row = flag.view(torch.float32).expand(1, 2 * k)
        packed = torch.cat([torch.cat([v, i.view(torch.float32)], dim=1), row]).contiguous()
        gathered = torch.empty((world * (Q + 1), 2 * k), dtype=torch.float32, device=packed.device)
        dist.all_gather_into_tensor(gathered, packed, group=group)   # rank-major concatenation
        gathered = gathered.view(world, Q + 1, 2 * k)
        flags = gathered[:, Q, 0].contiguous().view(torch.int32)
        lists = gathered[:, :Q, :].permute(1, 0, 2)                   # [Q][world][2k]
        vs = lists[:, :, :k].reshape(Q, world * k)
        is_ = lists[:, :, k:].reshape(Q, world * k).view(torch.int32)
        mv, mi = merge_fn(vs, is_, k)
        return mv, mi, flags

    # Query-SLICED post-processing (round 6): with the all_gather above every rank receives W lists
    # per query and merges ALL Q queries — work that does not shrink with the number of ranks and capped the 8-shard
    # projection below 6x.  Here rank r post-processes only the queries [r Q / W, (r + 1) Q / W): an all_to_all hands it
    # every rank's lists for that slice (sent: (W - 1) / W of what it has, received: as much), it merges Q / W queries,
    # and one all_gather of the merged [Q / W][2k] slices (+ the flag row) gives every rank the full result: 2 Q lists
    # over the links per rank instead of W Q, and the merge divides by W.  Same lists (a query's merge sees the same
    # W x k entries; ties go to the lowest global index either way).
    def _slices(Q):
        per = -(-Q // world)
        return per, world * per

    def _pad_rows(t, rows, fill):
        if t.shape[0] == rows:
            return t
        pad = torch.full((rows - t.shape[0],) + tuple(t.shape[1:]), fill, dtype=t.dtype, device=t.device)
        return torch.cat([t, pad])

    def _to_slices(x):
        """x [W * per][c]: block r goes to rank r -> [per][W][c]: what every rank holds for MY query slice"""
        out = torch.empty_like(x)
        dist.all_to_all_single(out, x.contiguous(), group=group)
        per = x.shape[0] // world
        return out.view(world, per, x.shape[1]).permute(1, 0, 2)

    def exchange_and_merge_sliced(v, i, flag):
        Q = v.shape[0]
        per, Qp = _slices(Q)
        packed = torch.cat([_pad_rows(v, Qp, float("inf")), _pad_rows(i, Qp, -1).view(torch.float32)], dim=1)
        mine = _to_slices(packed)                                       # [per][W][2k]
        vs = mine[:, :, :k].reshape(per, world * k)
        is_ = mine[:, :, k:].reshape(per, world * k).view(torch.int32)
        mv, mi = merge_fn(vs, is_, k)                                   # Q / W queries, not Q
        row = flag.view(torch.float32).expand(1, 2 * k)
        out = torch.cat([torch.cat([mv, mi.view(torch.float32)], dim=1), row]).contiguous()
        gathered = torch.empty((world * (per + 1), 2 * k), dtype=torch.float32, device=out.device)
        dist.all_gather_into_tensor(gathered, out, group=group)
        gathered = gathered.view(world, per + 1, 2 * k)
        flags = gathered[:, per, 0].contiguous().view(torch.int32)
        lists = gathered[:, :per, :].reshape(Qp, 2 * k)[:Q]
        return lists[:, :k].contiguous(), lists[:, k:].contiguous().view(torch.int32), flags

    # what runs on the caller's stream per query block (matrix work) and what runs behind it (exchange + merge; f16r:
    # exchange of the filter lists, global threshold, this rank's share of the rescoring, exchange + merge)
    # (k beyond the fused f16r path — a member window of 2k + 32 > 1024 slots — has no filter lists to exchange:
    #  the single exchange of the per-shard lists, whose local top-k falls back to the exact path: ADVICE r05)
    two_phase = use_f16r and world > 1 and k <= ops.F16R_MAX_FUSED_K

    def stage_main(qb, exact):
        if two_phase and not exact:
            return f16r_stages.filter_select(qb, g_local, k, index_base)
        res = local_topk_fn(qb, g_local, k, index_base, precision, exact) if exact else \
            local_topk_fn(qb, g_local, k, index_base, precision)
        if len(res) == 2:
            res = (res[0], res[1], torch.zeros(1, dtype=torch.int32, device=res[0].device))
        return res

    def stage_side(qb, res, exact):
        if not (two_phase and not exact):
            return gather_and_merge(*res)
        lval, lidx, ymax, flag = res
        Qb, K2 = int(lval.shape[0]), int(lval.shape[1])
        if sliced:
            # the filter values of MY query slice from every rank -> its thresholds; the Q / W thresholds of every rank
            # (+ its flag and its shard's norm maxima) are all-gathered: 4 bytes per query instead of W K2 values
            per, Qp = _slices(Qb)
            mine = _to_slices(_pad_rows(lval, Qp, float("inf")))          # [per][W][K2]
            thr_mine = f16r_stages.kth(mine.reshape(per, world * K2), k)      # [per]
            row = torch.cat([thr_mine, flag.view(torch.float32), ymax.to(torch.float32)]).contiguous()
            allrows = torch.empty((world, per + 3), dtype=torch.float32, device=row.device)
            dist.all_gather_into_tensor(allrows.view(-1), row, group=group)
            thr = allrows[:, :per].reshape(Qp)[:Qb].contiguous()
            flags = allrows[:, per].contiguous().view(torch.int32)
            ymax_all = allrows[:, per + 1: per + 3].contiguous()
            f16r_stages.keep_members(lval, lidx, k, thr, qb, ymax_all)
            v, i = f16r_stages.rescore(qb, g_local, lidx, k, index_base)
            any_flag = flags.ne(0).any().to(torch.int32).reshape(1)
            return gather_and_merge(v, i, any_flag)
        # only the filter VALUES travel (+ one row: the flag and the shard's norm maxima): the indices stay where their
        # rows are — a rank rescoring its own members never reads another rank's lidx
        extra = torch.zeros((1, K2), dtype=torch.float32, device=lval.device)
        extra[0, 0:1] = flag.view(torch.float32)
        extra[0, 1:3] = ymax
        packed = torch.cat([lval, extra]).contiguous()
        gathered = torch.empty((world * (Qb + 1), K2), dtype=torch.float32, device=packed.device)
        dist.all_gather_into_tensor(gathered, packed, group=group)
        gathered = gathered.view(world, Qb + 1, K2)
        flags = gathered[:, Qb, 0].contiguous().view(torch.int32)
        ymax_all = gathered[:, Qb, 1:3].contiguous()                              # [world][2]
        vals_all = gathered[:, :Qb, :].permute(1, 0, 2).reshape(Qb, world * K2)
        thr = f16r_stages.kth(vals_all, k)                                        # k-th smallest filter distance, all shards
        f16r_stages.keep_members(lval, lidx, k, thr, qb, ymax_all)                # this rank's members of the global set
        v, i = f16r_stages.rescore(qb, g_local, lidx, k, index_base)
        any_flag = flags.ne(0).any().to(torch.int32).reshape(1)
        return gather_and_merge(v, i, any_flag)

    return stage_main, stage_side, gather_and_merge


def sharded_topk(q_all: torch.Tensor, g_local: torch.Tensor, k: int, index_base: int,
                 precision="fp32", group=None,
                 local_topk_fn: Optional[Callable] = None,
                 merge_fn: Optional[Callable] = None, blocks: int = 1,
                 f16r_stages=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """k nearest gallery rows (squared L2) of every query over ALL ranks' gallery slices.

    q_all   [Q][d]  the full query set, identical on every rank (or its ops.PreparedRows, e.g. from
                    gather_prepared_queries)
    g_local [n][d]  this rank's valid gallery rows (padding rows removed), or an ops.PreparedRows of
                    them (a gallery matched repeatedly pays its norm / operand pass once)
    index_base      global gallery index of g_local[0]
    Returns (values [Q][k] ascending, indices [Q][k] int32 global), identical on every rank.
    Ties are broken towards the lowest global index, so the result does not depend on the number
    of shards.
    blocks > 1 (and more than one rank): the queries are processed in that many row blocks and the
    exchange + merge of block b runs on a second stream while the matrix cores already work on the local
    top-k of block b + 1 — the collective's latency (and the merge) leave the critical path except for the
    last block.  Same results (every query row is independent).
    f16r (more than one rank, the HIP stages or `f16r_stages`): TWO exchanges — the filter lists first; the threshold
    is then the k-th smallest filter distance over ALL shards, and every rank rescores only ITS members of the
    global rescore set (k + a few per query in total, not per rank): the exact-rescoring work divides by the
    number of ranks instead of being repeated on each (_f16r_two_phase)."""
    if f16r_stages is None and local_topk_fn is None and ops.precision_code(precision) == ops.F16R:
        # the HIP stages work on prepared operands (one preparation serves filter, threshold and rescoring)
        if not isinstance(q_all, ops.PreparedRows):
            q_all = ops.PreparedRows(q_all, ops.F16R)
        if not isinstance(g_local, ops.PreparedRows) and g_local.shape[0] > 0:
            g_local = ops.PreparedRows(g_local, ops.F16R)
    stage_main, stage_side, gather_and_merge = _make_stages(g_local, k, index_base, precision, group, local_topk_fn,
                                                            merge_fn, f16r_stages)
    rank, world = _world(group)
    nq = int(q_all.shape[0])
    if blocks > 1 and world > 1 and nq >= blocks:
        bounds = [(b * nq // blocks, (b + 1) * nq // blocks) for b in range(blocks)]
        probe = getattr(q_all, "norms", q_all)
        on_gpu = probe.is_cuda
        main = torch.cuda.current_stream(probe.device) if on_gpu else None
        side = _side_stream(probe.device) if on_gpu else None

        parts = []
        for lo, hi in bounds:
            qb = _query_block(q_all, lo, hi)
            res = stage_main(qb, False)
            if on_gpu:
                side.wait_stream(main)                  # this block's lists are complete
                with torch.cuda.stream(side):
                    out = stage_side(qb, res, False)
                for t in res:
                    t.record_stream(side)               # allocated on main, read by the exchange on side
                for t in out:
                    t.record_stream(main)               # allocated on side, read by the concatenation on main
                parts.append(out)
            else:
                parts.append(stage_side(qb, res, False))
        if on_gpu:
            main.wait_stream(side)
        # the only host synchronisation, after everything has been enqueued; identical on every rank (each block's
        # flags are the gathered flags of all ranks).  Only the FLAGGED blocks are repeated on the exact path.
        flagged = torch.stack([p_[2].reshape(-1).ne(0).any() for p_ in parts]).tolist()
        for b, (lo, hi) in enumerate(bounds):
            if flagged[b]:
                parts[b] = gather_and_merge(*stage_main(_query_block(q_all, lo, hi), True))
        return torch.cat([p_[0] for p_ in parts]), torch.cat([p_[1] for p_ in parts])
    v, i, flags = stage_side(q_all, stage_main(q_all, False), False)
    # the only host synchronisation, after everything has been enqueued; identical on every rank
    if bool(flags.any().item()):
        v, i, _ = gather_and_merge(*stage_main(q_all, True))
    return v, i


def sharded_topk_pipelined(q_local: torch.Tensor, n_total: int, g_local, k: int, index_base: int,
                           precision="fp32", group=None, blocks: int = 4,
                           local_topk_fn: Optional[Callable] = None, merge_fn: Optional[Callable] = None,
                           prepare_fn: Optional[Callable] = None,
                           rows_travel: Optional[bool] = None, f16r_stages=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """sharded_topk fed with the queries THIS RANK extracted (q_local: its DistributedSliceSampler slice of the
    n_total queries, wrap-around padding included), with BOTH exchanges hidden behind matrix work (VERDICT r04 item
    5a): the local slice is cut into `blocks` sub-blocks; sub-block b of every rank is all-gathered on a second
    stream while the matrix cores work on sub-block b - 1, then matched against the resident shard, and its
    per-shard lists are exchanged + merged on the second stream under sub-block b + 1.  Only the first gather
    (1 / blocks of the queries) and the last merge are exposed.  Same lists as
    sharded_topk(gather_prepared_queries(...)) — every query row is independent.

    What travels per query: the prepared operand + norm (bf16: 8 KB per 4096-d row; bf16x3 / f16mx: 16 KB), or the
    fp32 row for the arithmetics that need it on every rank (fp32, f16r: 16 KB; prepared after the gather).
    Returns (values [n_total][k], indices [n_total][k] int32 global), identical on every rank."""
    if f16r_stages is None and local_topk_fn is None and ops.precision_code(precision) == ops.F16R and \
            not isinstance(g_local, ops.PreparedRows) and g_local.shape[0] > 0:
        g_local = ops.PreparedRows(g_local, ops.F16R)
    stage_main, stage_side, gather_and_merge = _make_stages(g_local, k, index_base, precision, group, local_topk_fn,
                                                            merge_fn, f16r_stages)
    rank, world = _world(group)
    qper = int(q_local.shape[0])
    d = int(q_local.shape[1])
    if world == 1:
        qp = q_local[:n_total] if prepare_fn is None and not q_local.is_cuda else \
            (prepare_fn or (lambda x: ops.PreparedRows(x, precision)))(q_local[:n_total].contiguous())
        return sharded_topk(qp, g_local, k, index_base, precision, group, local_topk_fn, merge_fn,
                            f16r_stages=f16r_stages)
    blocks = max(1, min(int(blocks), qper))
    bounds = [(b * qper // blocks, (b + 1) * qper // blocks) for b in range(blocks)]
    dev = q_local.device
    on_gpu = q_local.is_cuda
    main = torch.cuda.current_stream(dev) if on_gpu else None
    side = _side_stream(dev) if on_gpu else None
    fp32_travels = rows_travel if rows_travel is not None else \
        (prepare_fn is None and ops.precision_code(precision) in (ops.F16R, ops.F32))
    make = prepare_fn or (lambda x: ops.PreparedRows(x, precision))

    def gather(lo, hi):
        """sub-block [lo, hi) of every rank -> the prepared set of its world * (hi - lo) rows, rank-major"""
        part = q_local[lo:hi].contiguous()
        if fp32_travels:
            return ("rows", all_gather_rows(part, group))
        p_ = make(part)
        return ("parts", type(p_), all_gather_rows(p_.operand_rows(), group), all_gather_rows(p_.norms, group))

    def assemble(g_):
        if g_[0] == "rows":
            return make(g_[1])
        return g_[1].from_parts(g_[2], g_[3], d, precision)

    def on_side(fn, *a):
        if not on_gpu:
            return fn(*a)
        with torch.cuda.stream(side):
            return fn(*a)

    # The matrix work of sub-block b runs on compute stream b % 2 (the caller's stream and one more): a rank's sub-block
    # is a non-integral number of rounds of the chip (world 8: 4096 x 10240 = 640 tiles = 2.5 rounds), and the sample
    # pass / selections in front of and behind the filter pass are latency-bound — side by side, the next sub-block's
    # work fills both (projection at 8 shards: f16r 4.5x -> 5.0x, bf16 4.9x -> 5.5x, tests/gpu_shardbench.py).
    lanes = [main, _side_stream(dev, "lane2")] if on_gpu and blocks > 1 else [main]
    if on_gpu:
        side.wait_stream(main)
        for cs in lanes[1:]:
            cs.wait_stream(main)
    pending = on_side(gather, *bounds[0])
    ev_g = None
    if on_gpu:
        ev_g = torch.cuda.Event()
        ev_g.record(side)
    outs, sets = [], []
    for b, (lo, hi) in enumerate(bounds):
        cur, cur_ev = pending, ev_g
        if b + 1 < blocks:                         # the next sub-block's queries travel under this one's matrix work
            pending = on_side(gather, *bounds[b + 1])
            if on_gpu:
                ev_g = torch.cuda.Event()
                ev_g.record(side)
        cs = lanes[b % len(lanes)]
        if on_gpu:
            cs.wait_event(cur_ev)
            for t in cur[1:]:
                if torch.is_tensor(t):
                    t.record_stream(cs)
            with torch.cuda.stream(cs):
                qb = assemble(cur)
                res = stage_main(qb, False)
            side.wait_stream(cs)
            for t in res:
                t.record_stream(side)
        else:
            qb = assemble(cur)
            res = stage_main(qb, False)
        sets.append(qb)
        out = on_side(stage_side, qb, res, False)
        if on_gpu:
            for t in out:
                t.record_stream(main)
        outs.append(out)
    if on_gpu:
        main.wait_stream(side)
        for cs in lanes[1:]:
            main.wait_stream(cs)
    flagged = torch.stack([o[2].reshape(-1).ne(0).any() for o in outs]).tolist()   # the only host synchronisation
    for b in range(blocks):
        if flagged[b]:
            outs[b] = gather_and_merge(*stage_main(sets[b], True))
    # rows of sub-block b, rank-major: global query r * qper + lo + i (beyond n_total: wrap-around padding, dropped)
    v_all = torch.empty((n_total, k), dtype=outs[0][0].dtype, device=outs[0][0].device)
    i_all = torch.empty((n_total, k), dtype=torch.int32, device=outs[0][0].device)
    for (lo, hi), (mv, mi, _) in zip(bounds, outs):
        n_b = hi - lo
        gidx = (torch.arange(world, device=mv.device)[:, None] * qper + lo +
                torch.arange(n_b, device=mv.device)[None, :]).reshape(-1)
        keep = gidx < n_total
        v_all[gidx[keep]] = mv[keep]
        i_all[gidx[keep]] = mi[keep]
    return v_all, i_all


_SIDE = {}


def _side_stream(dev: torch.device, role: str = "exchange"):
    """One stream per device and role ("exchange": the collectives + merges; "lane2": the second compute stream of
    sharded_topk_pipelined), created once (streams created at different times may share a hardware queue:
    extract._lane_streams)."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), role)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]
