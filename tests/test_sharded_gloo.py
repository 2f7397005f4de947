"""N > 1 path on CPU: world_size-2 gloo process groups exercising the shard / gather / merge logic
of openibl_amd.sharded and the cross-rank gather of extract_features.  The per-GPU compute steps
(HIP kernels in production) are replaced by the oracle through the injection points; what is under
test is everything between them."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _oracle_local_topk(q, g, k, index_base, precision):
    from oracle import matching as om
    Q = q.shape[0]
    vals = torch.full((Q, k), float("inf"))
    idx = torch.full((Q, k), -1, dtype=torch.int32)
    if g.shape[0]:
        d = om.pairwise_distance(q, g).numpy()
        v, i = om.topk(d, min(k, g.shape[0]))
        vals[:, : v.shape[1]] = torch.from_numpy(v)
        idx[:, : i.shape[1]] = torch.from_numpy((i + index_base).astype(np.int32))
    return vals, idx


class _ResidentShard:
    """CPU stand-in for ops.PreparedRows (the gallery shard prepared once: operand rows + norms):
    sharded_topk must hand whatever object represents the resident shard to the local step as is."""

    def __init__(self, rows):
        self.rows, self.shape = rows, tuple(rows.shape)
        self.norms = (rows.double() ** 2).sum(1).float()
        self.uses = 0


def _resident_local_topk(q, shard, k, index_base, precision):
    assert isinstance(shard, _ResidentShard)
    shard.uses += 1
    return _oracle_local_topk(q, shard.rows, k, index_base, precision)


class _CpuPrepared:
    """CPU stand-in with the protocol of ops.PreparedRows that gather_prepared_queries relies on:
    operand_rows(), .norms, from_parts()."""

    def __init__(self, x):
        self.x = x.clone()
        self.norms = (x.double() ** 2).sum(1).float()
        self.shape = tuple(x.shape)

    def operand_rows(self):
        return self.x.contiguous().view(torch.uint8).reshape(self.x.shape[0], -1)

    @classmethod
    def from_parts(cls, rows, norms, d, precision):
        self = cls.__new__(cls)
        self.x = rows.contiguous().view(torch.float32).reshape(rows.shape[0], d)
        self.norms, self.shape = norms, (rows.shape[0], d)
        return self


def _prepared_local_topk(qp, g, k, index_base, precision):
    assert isinstance(qp, _CpuPrepared)
    return _oracle_local_topk(qp.x, g, k, index_base, precision)


def _overflowing_local_topk(q, g, k, index_base, precision, exact=False):
    """Stand-in for the fused GPU path: rank 1's first (non-exact) attempt overflows and returns
    garbage lists with the flag raised; the exact repeat is correct on every rank."""
    v, i = _oracle_local_topk(q, g, k, index_base, precision)
    flag = torch.zeros(1, dtype=torch.int32)
    if not exact and dist.get_rank() == 1:
        v, i, flag = torch.zeros_like(v), torch.zeros_like(i), torch.ones(1, dtype=torch.int32)
    _overflowing_local_topk.calls.append(bool(exact))
    return v, i, flag


_overflowing_local_topk.calls = []


def _first_block_overflows(q, g, k, index_base, precision, exact=False):
    """As above, but only the FIRST non-exact call of rank 1 overflows: with query blocks, only that block may
    be repeated on the exact path.  Records (exact, query rows) per call."""
    v, i = _oracle_local_topk(q, g, k, index_base, precision)
    flag = torch.zeros(1, dtype=torch.int32)
    first = not _first_block_overflows.calls
    if not exact and first and dist.get_rank() == 1:
        v, i, flag = torch.zeros_like(v), torch.zeros_like(i), torch.ones(1, dtype=torch.int32)
    _first_block_overflows.calls.append((bool(exact), int(q.shape[0])))
    return v, i, flag


_first_block_overflows.calls = []


class _CpuF16rStages:
    """CPU stand-ins with the protocol of sharded.HipF16rStages: the 'filter distance' is the exact distance plus a
    deterministic perturbation of at most EPS (what the fp16 pass is to the exact distance), the bound is 2 EPS."""
    EPS, K2 = 1e-4, 32
    kept = []            # members this rank rescored, per call

    @staticmethod
    def members(k):
        return _CpuF16rStages.K2

    @staticmethod
    def filter_select(q, g, k, index_base):
        from oracle import matching as om
        Q, K2 = q.shape[0], _CpuF16rStages.K2
        lval = torch.full((Q, K2), float("inf"))
        lidx = torch.full((Q, K2), -1, dtype=torch.int32)
        if g.shape[0]:
            d = om.pairwise_distance(q, g)
            dh = d + _CpuF16rStages.EPS * torch.sin(1.0e4 * d)
            v, i = torch.sort(dh, dim=1, stable=True)
            n = min(K2, g.shape[0])
            lval[:, :n] = v[:, :n]
            lidx[:, :n] = (i[:, :n] + index_base).to(torch.int32)
        return lval, lidx, torch.tensor([1.0, 0.0]), torch.zeros(1, dtype=torch.int32)

    @staticmethod
    def kth(vals, k):
        return torch.kthvalue(vals, k, dim=1).values.contiguous()

    @staticmethod
    def keep_members(lval, lidx, k, thr, q, ymax_all):
        assert ymax_all.shape == (dist.get_world_size(), 2) and bool((ymax_all[:, 0] == 1.0).all())
        lidx[lval > thr[:, None] + 2 * _CpuF16rStages.EPS] = -1
        _CpuF16rStages.kept.append(int((lidx >= 0).sum()))

    @staticmethod
    def rescore(q, g, lidx, k, index_base):
        from oracle import matching as om
        Q = q.shape[0]
        d = om.pairwise_distance(q, g) if g.shape[0] else torch.zeros((Q, 0))
        v = torch.full((Q, k), float("inf"))
        i = torch.full((Q, k), -1, dtype=torch.int32)
        for r in range(Q):
            ids = lidx[r][lidx[r] >= 0].long()
            vals = d[r][ids - index_base]
            order = sorted(range(len(ids)), key=lambda t: (float(vals[t]), int(ids[t])))[:k]
            for c, t in enumerate(order):
                v[r, c], i[r, c] = vals[t], int(ids[t])
        return v, i


def _oracle_merge(vals, idx, k):
    key = np.lexsort((idx.numpy().astype(np.int64) & 0xFFFFFFFF, vals.numpy()), axis=1)[:, :k]
    return (torch.from_numpy(np.take_along_axis(vals.numpy(), key, 1)),
            torch.from_numpy(np.take_along_axis(idx.numpy(), key, 1)))


def _worker(rank, world, port, G, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from openibl_amd import sharded, synth, evaluators
        from oracle import matching as om
        q, g, gt, pids = synth.retrieval_problem(24, G, dim=256, seed=7, hard_fraction=0.5)
        start, per, n_valid = sharded.slice_bounds(G, rank, world)
        vals, idx = sharded.sharded_topk(q, g[start:start + n_valid], 10, start,
                                         local_topk_fn=_oracle_local_topk, merge_fn=_oracle_merge)
        d = om.pairwise_distance(q, g).numpy()
        wv, wi = om.topk(d, 10)
        ok_topk = bool(np.array_equal(idx.numpy(), wi) and np.allclose(vals.numpy(), wv))
        # one rank's candidate lists overflow: EVERY rank must repeat on the exact path (the flag
        # travels with the gathered lists), and the result is the correct one
        v2, i2 = sharded.sharded_topk(q, g[start:start + n_valid], 10, start,
                                      local_topk_fn=_overflowing_local_topk, merge_fn=_oracle_merge)
        ok_topk = ok_topk and _overflowing_local_topk.calls == [False, True] and \
            bool(np.array_equal(i2.numpy(), wi) and np.allclose(v2.numpy(), wv))
        # the resident-shard object (prepared once, matched against several query batches)
        shard = _ResidentShard(g[start:start + n_valid])
        for qs in (q, q[:7]):
            v3, i3 = sharded.sharded_topk(qs, shard, 10, start, local_topk_fn=_resident_local_topk,
                                          merge_fn=_oracle_merge)
            ok_topk = ok_topk and bool(np.array_equal(i3.numpy(), wi[: len(qs)]))
        ok_topk = ok_topk and shard.uses == 2
        # queries prepared where they were "extracted" and exchanged in prepared form (wrapped
        # slices, padding dropped after the gather)
        Qn = q.shape[0]
        qs, qper, _ = sharded.slice_bounds(Qn, rank, world)
        q_loc = torch.stack([q[(qs + i) % Qn] for i in range(qper)])
        qp = sharded.gather_prepared_queries(q_loc, Qn, "fp32", prepare_fn=_CpuPrepared)
        ok_topk = ok_topk and torch.equal(qp.x, q) and torch.allclose(qp.norms, (q ** 2).sum(1))
        v4, i4 = sharded.sharded_topk(qp, g[start:start + n_valid], 10, start,
                                      local_topk_fn=_prepared_local_topk, merge_fn=_oracle_merge)
        ok_topk = ok_topk and bool(np.array_equal(i4.numpy(), wi))
        # the same in query blocks (the exchange of block b overlaps the local top-k of block b + 1 on the
        # GPU; here: same lists, prepared queries sliced without copies, one rank's overflow repeats all)
        for nb in (2, 3, 5):
            v5, i5 = sharded.sharded_topk(qp, g[start:start + n_valid], 10, start, blocks=nb,
                                          local_topk_fn=_prepared_local_topk, merge_fn=_oracle_merge)
            v6, i6 = sharded.sharded_topk(q, g[start:start + n_valid], 10, start, blocks=nb,
                                          local_topk_fn=_oracle_local_topk, merge_fn=_oracle_merge)
            ok_topk = ok_topk and bool(np.array_equal(i5.numpy(), wi) and np.array_equal(i6.numpy(), wi)
                                       and np.allclose(v5.numpy(), wv) and np.allclose(v6.numpy(), wv))
        _overflowing_local_topk.calls = []
        v7, i7 = sharded.sharded_topk(q, g[start:start + n_valid], 10, start, blocks=2,
                                      local_topk_fn=_overflowing_local_topk, merge_fn=_oracle_merge)
        # every block of rank 1 overflowed: every block is repeated, block by block
        ok_topk = ok_topk and _overflowing_local_topk.calls == [False, False, True, True] and \
            bool(np.array_equal(i7.numpy(), wi))
        # only the first of three blocks overflows (on rank 1): every rank repeats THAT block on the exact path
        _first_block_overflows.calls = []
        v8, i8 = sharded.sharded_topk(q, g[start:start + n_valid], 10, start, blocks=3,
                                      local_topk_fn=_first_block_overflows, merge_fn=_oracle_merge)
        ok_topk = ok_topk and _first_block_overflows.calls == [(False, 8), (False, 8), (False, 8), (True, 8)] and \
            bool(np.array_equal(i8.numpy(), wi) and np.allclose(v8.numpy(), wv))
        # both exchanges pipelined under the matrix work (sharded_topk_pipelined): the local query slice travels
        # in sub-blocks — as prepared parts, or as fp32 rows prepared after the gather (f16r / fp32) — same lists
        for nb in (1, 2, 3, 7):
            v9, i9 = sharded.sharded_topk_pipelined(q_loc, Qn, g[start:start + n_valid], 10, start, blocks=nb,
                                                    local_topk_fn=_prepared_local_topk, merge_fn=_oracle_merge,
                                                    prepare_fn=_CpuPrepared)
            v10, i10 = sharded.sharded_topk_pipelined(q_loc, Qn, g[start:start + n_valid], 10, start, blocks=nb,
                                                      local_topk_fn=_oracle_local_topk, merge_fn=_oracle_merge,
                                                      prepare_fn=lambda x: x, rows_travel=True)
            ok_topk = ok_topk and bool(np.array_equal(i9.numpy(), wi) and np.array_equal(i10.numpy(), wi)
                                       and np.allclose(v9.numpy(), wv) and np.allclose(v10.numpy(), wv))
        _first_block_overflows.calls = []
        v11, i11 = sharded.sharded_topk_pipelined(q_loc, Qn, g[start:start + n_valid], 10, start, blocks=3,
                                                  local_topk_fn=_first_block_overflows, merge_fn=_oracle_merge,
                                                  prepare_fn=lambda x: x, rows_travel=True)
        ok_topk = ok_topk and bool(np.array_equal(i11.numpy(), wi)) and \
            [c_[0] for c_ in _first_block_overflows.calls] == [False, False, False, True]
        # f16r across shards: TWO exchanges (filter lists, then exact values), the global threshold, every rank
        # rescoring only its members of the global set — the lists are the oracle's, and the rescoring work adds up to
        # ~k + a few per query over ALL ranks, not per rank
        for nb in (1, 2):
            _CpuF16rStages.kept = []
            v12, i12 = sharded.sharded_topk(q, g[start:start + n_valid], 10, start, "f16r", blocks=nb,
                                            merge_fn=_oracle_merge, f16r_stages=_CpuF16rStages)
            ok_topk = ok_topk and bool(np.array_equal(i12.numpy(), wi) and np.allclose(v12.numpy(), wv))
            kept = torch.tensor([sum(_CpuF16rStages.kept)])
            dist.all_reduce(kept)
            ok_topk = ok_topk and Qn * 10 <= int(kept) <= Qn * 16
        v13, i13 = sharded.sharded_topk_pipelined(q_loc, Qn, g[start:start + n_valid], 10, start, "f16r", blocks=3,
                                                  merge_fn=_oracle_merge, f16r_stages=_CpuF16rStages,
                                                  prepare_fn=lambda x: x, rows_travel=True)
        ok_topk = ok_topk and bool(np.array_equal(i13.numpy(), wi) and np.allclose(v13.numpy(), wv))
        # query-SLICED post-processing (round 6, the default: an all_to_all hands every rank the lists of ITS Q / W
        # queries, it merges those and the merged slices are all-gathered) against the replicated merge of rounds 1-5
        # (all_gather of every list, every rank merges all Q): the same lists, bit for bit — query counts that do and
        # do not divide by the world size, an overflowing block, the f16r two-phase protocol
        assert sharded.SLICED_POSTPROCESSING
        same = True
        try:
            for qn in (q.shape[0], q.shape[0] - 3, 1):
                pair = []
                for sl in (False, True):
                    sharded.SLICED_POSTPROCESSING = sl
                    pair.append(sharded.sharded_topk(q[:qn], g[start:start + n_valid], 10, start,
                                                     local_topk_fn=_oracle_local_topk, merge_fn=_oracle_merge))
                same = same and torch.equal(pair[0][0], pair[1][0]) and torch.equal(pair[0][1], pair[1][1]) \
                    and tuple(pair[0][0].shape) == (qn, 10)
            sharded.SLICED_POSTPROCESSING = False
            a = sharded.sharded_topk(q, g[start:start + n_valid], 10, start, "f16r", blocks=2, merge_fn=_oracle_merge,
                                     f16r_stages=_CpuF16rStages)
            _first_block_overflows.seen = []
            o = sharded.sharded_topk(q, g[start:start + n_valid], 10, start, blocks=2, local_topk_fn=_first_block_overflows,
                                     merge_fn=_oracle_merge)
            same = same and bool(np.array_equal(a[1].numpy(), wi) and np.allclose(a[0].numpy(), wv)
                                 and np.array_equal(o[1].numpy(), wi))
        finally:
            sharded.SLICED_POSTPROCESSING = True
        ok_topk = ok_topk and same
        rec = evaluators.recalls_from_topk(idx.numpy(), gt)
        ok_rec = bool(np.array_equal(rec, om.evaluate_all(d, gt, pids)))

        # extract_features-style gather: each rank holds its wrapped slice; both gather modes
        # must reproduce dataset order after truncation
        L = 2 * per - 1 if world == 2 else G
        full = torch.arange(L * 3, dtype=torch.float32).reshape(L, 3)
        s0, p0, _ = sharded.slice_bounds(L, rank, world)
        local = torch.stack([full[(s0 + i) % L] for i in range(p0)])
        ok_gather = all(torch.equal(evaluators._gather_all(local, sg, rank, world)[:L], full)
                        for sg in (True, False))
        ret[rank] = (ok_topk, ok_rec, ok_gather)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("G", [301, 64, 49])
def test_world_size_2_sharded_topk_and_gather(G):
    world = 2
    port = 29600 + (os.getpid() + G) % 300
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, G, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=180)
            assert p.exitcode == 0
        for r in range(world):
            assert ret[r] == (True, True, True), (r, ret[r])


def test_shard_count_does_not_change_the_result():
    """Single process: simulate W shards and check the merged top-k equals the global one."""
    from openibl_amd import sharded, synth
    from oracle import matching as om
    q, g, _, _ = synth.retrieval_problem(16, 500, dim=128, seed=9)
    g[100:110] = g[300:310]          # exact cross-shard ties
    d = om.pairwise_distance(q, g).numpy()
    wv, wi = om.topk(d, 12)
    for W in (1, 2, 3, 8):
        vs, is_ = [], []
        for r in range(W):
            s, per, nv = sharded.slice_bounds(500, r, W)
            v, i = _oracle_local_topk(q, g[s:s + nv], 12, s, "fp32")
            vs.append(v)
            is_.append(i)
        mv, mi = _oracle_merge(torch.cat(vs, 1), torch.cat(is_, 1), 12)
        assert np.array_equal(mi.numpy(), wi), W


def _worker8(rank, world, port, ret):
    """World size 8 at BASELINE configs[3]'s gallery size: G = 83 952 dealt as 8 x 10 494 (DistributedSliceSampler),
    tiny d, the local step replaced by the oracle; one shard overflows in one query block."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from openibl_amd import sharded
        from oracle import matching as om
        G, Q, d, k = 83952, 30, 16, 10
        gen = torch.Generator().manual_seed(5)
        g = torch.nn.functional.normalize(torch.randn(G, d, generator=gen), dim=1)
        q = torch.nn.functional.normalize(torch.randn(Q, d, generator=gen), dim=1)
        g[10494 * 3 + 7] = g[10494 * 6 + 1]          # an exact tie across two shards: lowest global index wins
        q[0] = g[10494 * 6 + 1]
        start, per, n_valid = sharded.slice_bounds(G, rank, world)
        ok = (per, n_valid) == (10494, 10494)
        dm = om.pairwise_distance(q, g).numpy()
        wv, wi = om.topk(dm, k)
        ok = ok and wi[0, 0] == 10494 * 3 + 7 and wi[0, 1] == 10494 * 6 + 1
        # queries: every rank "extracted" ceil(Q / 8) = 4 of them (the last slice wraps), exchanged in prepared form
        qs, qper, _ = sharded.slice_bounds(Q, rank, world)
        q_loc = torch.stack([q[(qs + i) % Q] for i in range(qper)])
        qp = sharded.gather_prepared_queries(q_loc, Q, "fp32", prepare_fn=_CpuPrepared)
        ok = ok and torch.equal(qp.x, q)
        for nb in (1, 2, 3):
            v, i = sharded.sharded_topk(qp, g[start:start + n_valid], k, start, blocks=nb,
                                        local_topk_fn=_prepared_local_topk, merge_fn=_oracle_merge)
            ok = ok and bool(np.array_equal(i.numpy(), wi) and np.allclose(v.numpy(), wv))
        _first_block_overflows.calls = []
        v, i = sharded.sharded_topk(q, g[start:start + n_valid], k, start, blocks=3,
                                    local_topk_fn=_first_block_overflows, merge_fn=_oracle_merge)
        ok = ok and bool(np.array_equal(i.numpy(), wi)) and \
            _first_block_overflows.calls == [(False, 10), (False, 10), (False, 10), (True, 10)]
        # f16r across 8 shards: two exchanges, global threshold, the rescoring divided over the ranks
        _CpuF16rStages.kept = []
        v, i = sharded.sharded_topk(q, g[start:start + n_valid], k, start, "f16r", merge_fn=_oracle_merge,
                                    f16r_stages=_CpuF16rStages)
        kept = torch.tensor([sum(_CpuF16rStages.kept)])
        dist.all_reduce(kept)
        ok = ok and bool(np.array_equal(i.numpy(), wi) and np.allclose(v.numpy(), wv)) and Q * k <= int(kept) <= Q * (k + 8)
        # the pipelined form: every rank's 4 local queries travel in 1 / 2 / 4 sub-blocks (the last slice wraps)
        for nb in (1, 2, 4):
            v, i = sharded.sharded_topk_pipelined(q_loc, Q, g[start:start + n_valid], k, start, blocks=nb,
                                                  local_topk_fn=_prepared_local_topk, merge_fn=_oracle_merge,
                                                  prepare_fn=_CpuPrepared)
            ok = ok and bool(np.array_equal(i.numpy(), wi) and np.allclose(v.numpy(), wv))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_world_size_8_at_pitts250k_gallery_size():
    """8 real processes over gloo, 8 x 10 494 gallery rows, 1 / 2 / 3 query blocks, one
    shard overflowing in one block (only that block is repeated, on every rank)."""
    world = 8
    port = 29950 + os.getpid() % 40
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_worker8, args=(r, world, port, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=300)
            assert p.exitcode == 0
        assert [ret[r] for r in range(world)] == [True] * world
