"""'f16r': fp16 filter pass + exact rescoring (csrc/match_f16r.h) — the fused distance +
top-k whose lists are those of an fp32 matrix with correctly rounded dot products.  Checked here: the prepared
parts against their definition (the residual bound is what the superset guarantee rests on), the lists against
the fp64 oracle and against the fp32 mode, rows of any magnitude, the overflow -> exact-path protocol, sharded
matching, and the reference's golden rankings / recalls."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from openibl_amd import ops, sharded, synth
from oracle import matching as om

pytestmark = pytest.mark.gpu


def _fp64_topk(q, g, k):
    d = (q.double() ** 2).sum(1)[:, None] + (g.double() ** 2).sum(1)[None] - 2.0 * q.double() @ g.double().t()
    v, i = torch.sort(d, dim=1, stable=True)
    return v[:, :k], i[:, :k], d


def _assert_lists(name, q, g, v, i, k, tie=2e-6, val_tol=1e-6):
    """(v, i) against fp64: values to fp32 rounding (val_tol: the fp32 MFMA mode itself carries ~d 2^-24 of
    accumulation error, 4.7e-6 seen at d = 4096), indices equal except where fp64 calls a near-tie."""
    wv, wi, d64 = _fp64_topk(q, g, k)
    v, i = v.cpu(), i.cpu().long()
    got64 = torch.gather(d64, 1, i)
    verr = float((v.double() - got64).abs().max())
    assert verr <= val_tol * max(1.0, float(d64.abs().max())), (name, verr)
    diff = (i != wi)
    worst = float((got64 - wv).abs()[diff].max()) if diff.any() else 0.0
    print(f"{name}: {int(diff.sum())} of {i.numel()} entries differ from fp64, all within {worst:.2e}")
    assert worst <= tie * max(1.0, float(wv.abs().max())), name
    return int(diff.sum())


def test_prepared_parts_are_what_the_bound_needs(dev):
    x = synth.descriptors(37, 4096, seed=3)
    x[5] *= 1e-4                        # any magnitude: the row scale is a power of two per row
    x[6] *= 3e3
    x[7] = 0.0
    x[8, :100] *= 1e-7                  # elements far below the row's fp16 normals: dropped, kept in the residual
    p = ops.PreparedRows(x.to(dev), "f16r")
    ref = ops.PreparedRows(x.to(dev), "fp32")
    assert torch.equal(p.norms, ref.norms)                      # the norms every mode uses, bit for bit
    h, aux = p.operand.cpu().double(), p.aux.cpu().double()
    isc, nx, rx = aux[:, 0], aux[:, 1], aux[:, 2]
    assert torch.equal(torch.log2(isc[isc > 0]).round(), torch.log2(isc[isc > 0]))   # powers of two
    big = h.abs().amax(1)
    nz = big > 0
    assert bool(((big[nz] >= 2.0 ** 14) & (big[nz] < 2.0 ** 15 + 1)).all())
    resid = (x.double() - h * isc[:, None]).norm(dim=1)
    assert bool((resid <= rx).all()) and bool((x.double().norm(dim=1) <= nx).all())   # rounded UP
    assert bool((rx[nz] <= 2.0 ** -11 * nx[nz]).all())          # fp16: 11 significant bits
    assert float(rx[7]) == 0.0 and float(nx[7]) == 0.0


@pytest.mark.parametrize("m,n,d,k", [(512, 16384, 256, 10), (300, 20000, 4096, 10), (256, 16500, 512, 25),
                                     (700, 9000, 128, 5)])
def test_fused_lists_are_fp32_exact(dev, m, n, d, k):
    q, g, gt, pids = synth.retrieval_problem(m, n, dim=d, seed=m + n, hard_fraction=0.5)
    v, i, flag = ops.sqdist_topk(q.to(dev), g.to(dev), k, precision="f16r", defer_check=True)
    assert int(flag.item()) == 0
    n_diff = _assert_lists(f"f16r {m}x{n}x{d} k={k}", q, g, v, i, k)
    v32, i32 = ops.sqdist_topk(q.to(dev), g.to(dev), k, precision="fp32")
    n32 = _assert_lists(f"fp32 {m}x{n}x{d} k={k}", q, g, v32, i32, k, tie=1e-5, val_tol=1e-5)
    assert n_diff <= n32 + 2          # at least as close to fp64 as the fp32 MFMA mode
    assert (v - v32).abs().max() <= 1e-5 * max(1.0, float(v32.abs().max()))
    # index_base and prepared operands: the same lists
    gp = ops.PreparedRows(g.to(dev), "f16r")
    v2, i2 = ops.sqdist_topk_prepared(ops.PreparedRows(q.to(dev), "f16r"), gp, k, index_base=1000)
    assert torch.equal(v2, v) and torch.equal(i2, i + 1000)


def test_small_and_ragged_problems_take_the_exact_path(dev):
    """Below the fused path's size the member set comes from fp32 distance tiles — and is rescored like every other
    (round 6, ADVICE r05: values and tie order must not depend on the path): the fp32 mode's indices up to fp32
    near-ties, fp64-accumulated values."""
    for m, n, d, k in [(1, 1, 64, 1), (3, 129, 128, 10), (130, 67, 4096, 10), (5, 3000, 192, 7), (4, 6, 64, 10)]:
        g = torch.Generator().manual_seed(m * 7 + n)
        x, y = torch.randn((m, d), generator=g), torch.randn((n, d), generator=g)
        assert not ops.f16r_fused(m, n, d, k)
        v, i = ops.sqdist_topk(x.to(dev), y.to(dev), k, precision="f16r")
        v32, i32 = ops.sqdist_topk(x.to(dev), y.to(dev), k, precision="fp32")
        kk = min(k, n)
        assert torch.equal(i[:, kk:], i32[:, kk:]) and torch.equal(v[:, kk:], v32[:, kk:])   # (k > n: (+inf, -1) tails)
        _assert_lists(f"f16r exact path {m}x{n}x{d}", x, y, v[:, :kk], i[:, :kk], kk, tie=2e-6, val_tol=3e-7)
        assert float((v[:, :kk] - v32[:, :kk]).abs().max()) <= 1e-5 * float(v32[:, :kk].abs().max())
        assert float((i[:, :kk] != i32[:, :kk]).float().mean()) <= 0.01


def test_rows_of_any_magnitude(dev):
    """Unnormalised descriptors, rows 1e-3 .. 1e3 long, a zero row: per-row power-of-two scales keep the fp16
    pass in range and the bound follows the row norms."""
    m, n, d, k = 300, 17000, 256, 10
    gen = torch.Generator().manual_seed(8)
    q = torch.randn((m, d), generator=gen) * torch.logspace(-3, 3, m)[:, None]
    g = torch.randn((n, d), generator=gen) * torch.logspace(-3, 3, n)[torch.randperm(n, generator=gen)][:, None]
    g[77] = 0.0
    v, i, flag = ops.sqdist_topk(q.to(dev), g.to(dev), k, precision="f16r", defer_check=True)
    if int(flag.item()):                                            # (legitimate: then the exact repeat decides)
        v, i = ops.sqdist_topk(q.to(dev), g.to(dev), k, precision="f16r", exact=True)
    wv, wi, d64 = _fp64_topk(q, g, k)
    got64 = torch.gather(d64, 1, i.cpu().long())
    # relative to the size of the terms that are subtracted (|x|^2 + |y|^2)
    scale = (q.double() ** 2).sum(1)[:, None] + torch.gather((g.double() ** 2).sum(1)[None].expand(m, -1), 1, wi)
    assert float(((got64 - wv).abs() / scale).max()) <= 2e-6
    assert float(((v.cpu().double() - got64).abs() / scale).max()) <= 1e-6


def test_near_duplicate_gallery_overflows_into_the_exact_path(dev):
    """More than K2 = 32 gallery rows within the error bound of the k-th distance: the rescore window cannot hold
    them, the flag is raised, and the repeat on the exact path returns the fp32 mode's lists."""
    m, n, d, k = 256, 16384, 256, 10
    q, g, _, _ = synth.retrieval_problem(m, n, dim=d, seed=5)
    dup = torch.nn.functional.normalize(g[:1] + 1e-6 * torch.randn(200, d), dim=1)
    g[1000:1200] = dup                                              # 200 rows ~1e-6 apart
    q[0] = g[1000]
    v, i, flag = ops.sqdist_topk(q.to(dev), g.to(dev), k, precision="f16r", defer_check=True)
    assert int(flag.item()) == 1
    v, i = ops.sqdist_topk(q.to(dev), g.to(dev), k, precision="f16r")      # reads the flag, repeats exactly
    v32, i32 = ops.sqdist_topk(q.to(dev), g.to(dev), k, precision="fp32")
    # the repeat takes the K2 nearest of the fp32 tiles and rescores them: query 0's list is ten of the 200 rows that
    # sit 1e-12 from it (any ten: they are fp64 near-ties), every other query's list is the fp32 mode's
    assert torch.equal(i[1:], i32[1:])
    assert bool(((i[0] >= 1000) & (i[0] < 1200)).all()) and float(v[0].abs().max()) < 1e-6
    _assert_lists("f16r after overflow", q, g, v, i, k)
    ve, ie = ops.sqdist_topk(q.to(dev), g.to(dev), k, precision="f16r", exact=True)
    assert torch.equal(ie, i) and torch.equal(ve, v)
    # through sharded_topk (one rank): the same protocol
    v2, i2 = sharded.sharded_topk(q.to(dev), g.to(dev), k, 0, "f16r")
    assert torch.equal(i2, i) and torch.equal(v2, v)


@pytest.mark.parametrize("name", ["match_small", "match_nms"])
def test_reference_rankings_and_recalls(name, dev):
    """The reference's own outputs (tests/golden, written by oracle/make_golden.py from ibl/evaluators.py): these
    problems are below the fused path's size, so f16r answers with the exact path — the API contract."""
    g = load_golden(name)
    q, gal, gt, pids = synth.retrieval_problem(
        int(g["Q"]), int(g["G"]), dim=int(g["dim"]), seed=int(g["seed"]), views_per_place=int(g["views_per_place"]),
        hard_fraction=float(g["hard_fraction"]), hard_noise_mult=float(g["hard_noise_mult"]))
    v, i = ops.sqdist_topk(q.to(dev), gal.to(dev), 20, precision="f16r")
    assert np.array_equal(i.cpu().numpy(), g["top20"])


def test_sharded_f16r_equals_global(dev):
    """8 shards on one GPU: per-shard f16r lists merged == the global f16r lists (values are exact distances, so
    the merge cannot depend on the sharding), queries prepared once."""
    Q, G, d, k, W = 512, 8 * 9000 + 13, 256, 10, 8
    q, g, gt, pids = synth.retrieval_problem(Q, G, dim=d, seed=12, hard_fraction=0.5)
    qd, gd = q.to(dev), g.to(dev)
    gv, gi = ops.sqdist_topk(qd, gd, k, precision="f16r")
    qp = ops.PreparedRows(qd, "f16r")
    vs, is_ = [], []
    for r in range(W):
        start, per, n_valid = sharded.slice_bounds(G, r, W)
        shard = ops.PreparedRows(gd[start:start + n_valid].contiguous(), "f16r")
        v, i, flag = sharded.hip_local_topk(qp, shard, k, start, "f16r")
        assert int(flag.item()) == 0
        vs.append(v)
        is_.append(i)
    mv, mi = sharded.hip_merge_topk(torch.cat(vs, 1), torch.cat(is_, 1), k)
    assert torch.equal(mi, gi) and torch.equal(mv, gv)
    _assert_lists("f16r sharded", q, g, mv, mi, k)
    # the exchanged form of prepared queries (what gather_prepared_queries ships) re-assembles to the same object
    qp2 = ops.PreparedRows.from_parts(qp.operand_rows(), qp.norms, d, "f16r")
    v2, i2 = ops.sqdist_topk_prepared(qp2, ops.PreparedRows(gd, "f16r"), k)
    assert torch.equal(i2, gi) and torch.equal(v2, gv)


def test_topk_precision_rule():
    assert ops.topk_precision("f16mx") == ops.F16R
    assert ops.topk_precision("f16mx", torch.float16) == ops.F16R       # 16-bit storage: rescored from the stored rows
    assert ops.topk_precision("f16mx", torch.bfloat16) == ops.F16R
    assert ops.topk_precision("f16mx", torch.float32, 120) == ops.F16R  # the 120 ranks of spatial NMS (round 6)
    assert ops.topk_precision("f16mx", torch.float32, 496) == ops.F16R
    assert ops.topk_precision("f16mx", torch.float32, 497) == ops.F16MX  # beyond the fused path's member window
    assert ops.topk_precision("f16mx", None, 10) == ops.F16MX           # mixed storage types: as asked
    assert ops.topk_precision("bf16x3") == ops.BF16X3 and ops.topk_precision("fp32") == ops.F32
    with pytest.raises(ValueError):
        ops.pairwise_sqdist(torch.zeros(2, 64), torch.zeros(2, 64), "f16r")



def test_long_candidate_list_takes_the_workgroup_selection(dev):
    """A query whose strided threshold sample is far away lets ~3000 gallery rows through the filter: beyond the 2048
    entries one wave keeps in registers, inside the list capacity — the workgroup-per-row launch of the selection
    serves it (no overflow), and its list is the fp64 one."""
    m, n, d, k = 256, 16384, 256, 10
    gen = torch.Generator().manual_seed(31)
    q = torch.nn.functional.normalize(torch.randn((m, d), generator=gen), dim=1)
    g = torch.nn.functional.normalize(torch.randn((n, d), generator=gen), dim=1)
    stride = n // 1024                                           # the sample: rows 0, stride, 2 stride, ...
    far = torch.nn.functional.normalize(-q[0][None] + 0.01 * torch.randn((n, d), generator=gen), dim=1)
    near = torch.zeros(n, dtype=torch.bool)
    near[torch.randperm(n, generator=gen)[:3000]] = True
    near[::stride] = False
    g = torch.where(near[:, None], g, far)                       # 3000 ordinary rows, everything else ~ -q0 ...
    g[::stride] = torch.nn.functional.normalize(-q[0][None] + 0.3 * torch.randn((1024, d), generator=gen), dim=1)  # ... the sample a little nearer
    v, i, flag = ops.sqdist_topk(q.to(dev), g.to(dev), k, precision="f16r", defer_check=True)
    assert int(flag.item()) == 0
    _assert_lists("f16r, 3000 candidates for query 0", q, g, v, i, k)
    assert bool(near[i[0].cpu().long()].all())                   # query 0's neighbours are among the ordinary rows


@pytest.mark.parametrize("k", [1, 16, 17, 32, 33, 120, 496])
def test_member_window_sizes(dev, k):
    """k = 16 | 17: the member window changes from 32 to 2k + 32 slots; k = 32 | 33: the selection changes from k
    register extraction rounds to the bisection (round 6); 120: spatial NMS; 496: the largest k of the fused path."""
    m, n, d = 256, 16384, 256
    q, g, _, _ = synth.retrieval_problem(m, n, dim=d, seed=40 + k, hard_fraction=0.5)
    assert ops.f16r_fused(m, n, d, k) and ops.f16r_members(k) == (32 if k <= 16 else 2 * k + 32)
    v, i, flag = ops.sqdist_topk(q.to(dev), g.to(dev), k, precision="f16r", defer_check=True)
    assert int(flag.item()) == 0
    _assert_lists(f"f16r k={k}", q, g, v, i, k)
    assert not ops.f16r_fused(m, n, d, 497) and ops.f16r_members(497) == 1024   # 2k + 32 > 1024: exact path


@pytest.mark.parametrize("store", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("k", [10, 120])
def test_sixteen_bit_storage(dev, store, k):
    """Descriptors STORED as fp16 / bf16 (BASELINE configs[4]): prepared from the stored rows (an fp16 row is its own
    fp16 image: residual 0 up to flushed elements; a bf16 row's 8 bits fit fp16's 11), filtered, and rescored from
    the stored rows widened exactly — the lists of the widened fp32 problem, bit-equal to f16r on the widened copy."""
    m, n, d = 300, 20000, 512
    q, g, _, _ = synth.retrieval_problem(m, n, dim=d, seed=77 + k, hard_fraction=0.5)
    qs, gs = q.to(store), g.to(store)
    qd, gd = qs.to(dev), gs.to(dev)
    p = ops.PreparedRows(gd, "f16r")
    assert p.storage == ops.storage_code(gd) and p._source.dtype == store
    ref = ops.PreparedRows(gd.float(), "f16r")
    assert torch.equal(p.norms, ref.norms) and torch.equal(p.operand, ref.operand) and torch.equal(p.aux, ref.aux)
    if store == torch.float16:
        assert float(p.aux[:, 2].max()) == 0.0                      # an fp16 row IS its fp16 image
    assert ops.f16r_fused(m, n, d, k)
    v, i, flag = ops.sqdist_topk(qd, gd, k, precision="f16r", defer_check=True)
    assert int(flag.item()) == 0
    _assert_lists(f"f16r on {store} rows k={k}", qs.float(), gs.float(), v, i, k)
    vw, iw = ops.sqdist_topk(qd.float(), gd.float(), k, precision="f16r")
    assert torch.equal(i, iw) and torch.equal(v, vw)
    # mixed: fp32 queries against the 16-bit gallery, and the exact path on the stored rows
    vm, im = ops.sqdist_topk(q.to(dev), gd, k, precision="f16r")
    _assert_lists(f"f16r fp32 queries x {store} gallery k={k}", q, gs.float(), vm, im, k)
    ve, ie = ops.sqdist_topk(qd, gd, k, precision="f16r", exact=True)
    _assert_lists(f"f16r exact path on {store} rows k={k}", qs.float(), gs.float(), ve, ie, k)
    # the exchanged form (what a rank ships) keeps the storage type; a row block is a view of every part
    qp = ops.PreparedRows(qd, "f16r")
    qp2 = ops.PreparedRows.from_parts(qp.operand_rows(), qp.norms, d, "f16r", source_dtype=store)
    v2, i2 = ops.sqdist_topk_prepared(qp2, p, k)
    assert torch.equal(i2, i) and torch.equal(v2, v)
    blk = qp.rows(64, 192)
    assert blk._source.data_ptr() == qp._source[64:].data_ptr() and blk.operand.data_ptr() == qp.operand[64:].data_ptr()
    v3, i3 = ops.sqdist_topk_prepared(blk, p, k)
    assert torch.equal(i3, i[64:192]) and torch.equal(v3, v[64:192])
