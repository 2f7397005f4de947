"""Distance matrix, top-k and recall on the HIP path against the reference's vectors and the
oracle (GPU)."""
import numpy as np
import pytest
import torch

from conftest import assert_rel_l2, load_golden
from openibl_amd import ops, synth
from oracle import matching as om

pytestmark = pytest.mark.gpu


def _problem(g):
    return synth.retrieval_problem(
        int(g["Q"]), int(g["G"]), dim=int(g["dim"]), seed=int(g["seed"]),
        views_per_place=int(g["views_per_place"]), hard_fraction=float(g["hard_fraction"]),
        hard_noise_mult=float(g["hard_noise_mult"]))


@pytest.mark.parametrize("name", ["match_small", "match_nms"])
def test_pairwise_and_recall_match_reference(name, dev):
    g = load_golden(name)
    q, gal, gt, pids = _problem(g)
    d = ops.pairwise_sqdist(q.to(dev), gal.to(dev), "fp32")
    err = (d.cpu().double() - torch.from_numpy(g["distmat"]).double()).abs().max().item()
    print(f"{name}: max |dist - reference| = {err:.3e}")
    assert err <= 1e-5      # distances are O(1): 1e-4 * max(1, |d|) with margin
    vals, idx = ops.row_topk(d, 20)
    assert np.array_equal(idx.cpu().numpy(), g["top20"])
    # the reference's API, fed like the reference's Evaluator does
    from collections import OrderedDict
    from ibl.evaluators import pairwise_distance, evaluate_all
    query = [(f"q{i:05d}.jpg", 100000 + i, 0.0, 0.0) for i in range(len(q))]
    gallery = [(f"g{j:05d}.jpg", pids[j], 0.0, 0.0) for j in range(len(gal))]
    feats = OrderedDict()
    for (f, _, _, _), v in zip(query, q):
        feats[f] = v
    for (f, _, _, _), v in zip(gallery, gal):
        feats[f] = v
    dm, xq, yg = pairwise_distance(feats, query, gallery)
    assert dm.device.type == "cpu" and tuple(dm.shape) == (len(q), len(gal))
    assert np.array_equal(xq, q.numpy()) and np.array_equal(yg, gal.numpy())
    np.testing.assert_allclose(dm.numpy(), g["distmat"], rtol=0, atol=1e-5)
    np.testing.assert_array_equal(evaluate_all(dm, gt, gallery), g["recalls"])
    np.testing.assert_array_equal(evaluate_all(dm, gt, gallery, nms=True), g["recalls_nms"])
    sub = OrderedDict((k, feats[k]) for k in list(feats)[:40])
    da, _, _ = pairwise_distance(sub)
    np.testing.assert_allclose(da.numpy(), g["dist_all40"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("m,n,d", [(1, 1, 64), (3, 129, 128), (130, 67, 4096), (257, 1000, 512)])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_pairwise_ragged(dev, m, n, d, precision):
    g = torch.Generator().manual_seed(m * 7 + n)
    x, y = torch.randn((m, d), generator=g), torch.randn((n, d), generator=g)
    got = ops.pairwise_sqdist(x.to(dev), y.to(dev), precision).cpu()
    if precision == "bf16":
        xr, yr = x.to(torch.bfloat16).double(), y.to(torch.bfloat16).double()
        want = (x.double() ** 2).sum(1)[:, None] + (y.double() ** 2).sum(1)[None] - 2 * xr @ yr.t()
    else:
        want = om.pairwise_distance(x.double(), y.double())
    assert_rel_l2(f"pairwise {precision} {m}x{n}x{d}", got, want, 2e-6)


def test_pairwise_strided_output(dev):
    x, y = synth.descriptors(50, 256, seed=1).to(dev), synth.descriptors(70, 256, seed=2).to(dev)
    big = torch.full((50, 100), -7.0, device=dev)
    ops.pairwise_sqdist(x, y, "fp32", out=big[:, :70])
    assert torch.equal(big[:, :70], ops.pairwise_sqdist(x, y, "fp32"))
    assert (big[:, 70:] == -7.0).all()


@pytest.mark.parametrize("m,n,k", [(5, 3000, 10), (3, 5000, 120), (2, 7, 10), (4, 1024, 1024),
                                   (1, 100000, 25), (6, 2049, 1),
                                   # the selection paths: wave per row (<= 512, <= 1024 elements),
                                   # workgroup selection (<= 2048), and their k = 32 / 33 boundary
                                   (130, 300, 5), (7, 512, 31), (9, 513, 10), (9, 1024, 32),
                                   (5, 1500, 7), (5, 2048, 32), (3, 2048, 33), (3, 1025, 33),
                                   (1, 1, 1), (6, 40, 32)])
def test_row_topk_vs_stable_argsort(dev, m, n, k):
    g = torch.Generator().manual_seed(n + k)
    v = torch.randn((m, n), generator=g)
    w = min(v[:, ::7].shape[1], v[:, 1::7].shape[1])
    v[:, ::7][:, :w] = v[:, 1::7][:, :w]                    # plant exact ties
    v[0, : min(n, 50)] = -3.0                               # a run of equal minima
    vals, idx = ops.row_topk(v.to(dev), k, index_base=1000)
    order = om.ranking(v.numpy())
    kk = min(k, n)
    assert np.array_equal(idx.cpu().numpy()[:, :kk], order[:, :kk] + 1000)
    assert np.array_equal(vals.cpu().numpy()[:, :kk], np.take_along_axis(v.numpy(), order[:, :kk], 1))
    if k > n:
        assert (idx.cpu()[:, n:] == -1).all() and torch.isinf(vals.cpu()[:, n:]).all()


def test_row_topk_merge_equals_global(dev):
    """Per-shard top-k + merge (index lists) == top-k of the whole row, for any shard count."""
    g = torch.Generator().manual_seed(9)
    v = torch.randn((7, 9000), generator=g).to(dev)
    v[:, 100:200] = v[:, 4100:4200]      # cross-shard ties
    want_v, want_i = ops.row_topk(v, 10)
    for shards in (2, 3, 8):
        per = -(-9000 // shards)
        vs, is_ = [], []
        for s in range(shards):
            blk = v[:, s * per:(s + 1) * per].contiguous()
            a, b = ops.row_topk(blk, 10, index_base=s * per)
            vs.append(a)
            is_.append(b)
        mv, mi = ops.row_topk(torch.cat(vs, 1).contiguous(), 10, idx_in=torch.cat(is_, 1).contiguous())
        assert torch.equal(mi, want_i) and torch.equal(mv, want_v)


def test_negative_and_special_values_order(dev):
    v = torch.tensor([[0.0, -0.0, 1e-30, -1e-30, float("inf"), -5.0, 3.0, -float("inf")]])
    vals, idx = ops.row_topk(v.to(dev), 8)
    assert idx.cpu().tolist()[0][:2] == [7, 5]
    assert idx.cpu().tolist()[0][-2:] == [6, 4]


def test_full_size_properties(dev):
    """Pitts30k-sized matrix (6816 x 10000 x 4096): size-independent checks instead of a host
    recomputation — planted duplicates are their own nearest neighbour at distance ~0, the matrix
    of (g, q) is the transpose of (q, g), and bf16 ranks agree with fp32 on the planted set."""
    q, gal, gt, _ = synth.retrieval_problem(6816, 10000, seed=5, hard_fraction=0.0,
                                               positives_per_query=1)
    q, gal = q.to(dev), gal.to(dev)
    gal[10000 - 100:] = q[:100]
    d = ops.pairwise_sqdist(q, gal, "fp32")
    vals, idx = ops.row_topk(d, 10)
    assert torch.equal(idx[:100, 0].cpu(), torch.arange(9900, 10000, dtype=torch.int32))
    assert vals[:100, 0].abs().max().item() < 1e-5
    dt = ops.pairwise_sqdist(gal[:512].contiguous(), q[:640].contiguous(), "fp32")
    assert (dt.t() - d[:640, :512]).abs().max().item() < 1e-5
    from ibl.evaluators import recalls_from_topk
    r32 = recalls_from_topk(idx.cpu().numpy(), gt)
    db = ops.pairwise_sqdist(q, gal, "bf16")
    _, idxb = ops.row_topk(db, 10)
    rb = recalls_from_topk(idxb.cpu().numpy(), gt)
    print("recalls fp32", r32, "bf16", rb)
    assert np.array_equal(r32, rb) and r32[0] > 0.95


@pytest.mark.parametrize("m,n,d", [(300, 700, 256), (257, 1000, 512), (1000, 513, 4096), (64, 2500, 1024)])
def test_pairwise_ring_equals_generic(dev, m, n, d):
    """The ring-schedule distance kernel keeps the K order and the epilogue expression of the
    generic kernel: identical bf16-mode matrices, ragged tiles included."""
    g = torch.Generator().manual_seed(m + n + d)
    x, y = torch.randn((m, d), generator=g).to(dev), torch.randn((n, d), generator=g).to(dev)
    ops.set_match_ring(0)
    try:
        ref = ops.pairwise_sqdist(x, y, "bf16")
        ops.set_match_ring(2)
        big = torch.full((m, n + 37), -7.0, device=dev)
        got = ops.pairwise_sqdist(x, y, "bf16", out=big[:, :n])
    finally:
        ops.set_match_ring(1)
    assert torch.equal(got, ref)
    assert (big[:, n:] == -7.0).all()


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("m,n,d,k", [(2100, 20000, 256, 10), (2049, 16500, 512, 25), (700, 9000, 128, 120),
                                     (5, 300, 64, 10), (3, 7, 64, 10)])
def test_sqdist_topk_equals_matrix_topk(dev, m, n, d, k, precision):
    """Fused distance + top-k (threshold sample, filtered pass, candidate selection) == top-k of
    the materialised matrix: same values, same indices, ties towards the lowest index."""
    g = torch.Generator().manual_seed(m * 3 + n)
    x, y = synth.descriptors(m, d, seed=m), synth.descriptors(n, d, seed=n + 1)
    y[n // 2: n // 2 + min(m, 50)] = x[: min(m, 50)]              # exact matches
    y[5] = y[n - 3]                                                 # a duplicate pair (tie)
    x, y = x.to(dev), y.to(dev)
    want_v, want_i = ops.row_topk(ops.pairwise_sqdist(x, y, precision), k, index_base=77)
    got_v, got_i = ops.sqdist_topk(x, y, k, index_base=77, precision=precision)
    assert torch.equal(got_i, want_i) and torch.equal(got_v, want_v)
    ex_v, ex_i = ops.sqdist_topk(x, y, k, index_base=77, precision=precision, exact=True)
    assert torch.equal(ex_i, want_i) and torch.equal(ex_v, want_v)


def test_sqdist_topk_overflow_falls_back(dev):
    """Thousands of identical gallery rows make every candidate list outgrow its capacity: the
    overflow flag is raised and the wrapper repeats the call on the exact path."""
    m, n, d, k = 2048, 16384, 256, 10
    x = synth.descriptors(m, d, seed=3).to(dev)
    y = synth.descriptors(1, d, seed=4).repeat(n, 1).contiguous().to(dev)
    from openibl_amd import lib as _l
    L = _l.load()
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    ov = torch.empty((m, k), device=dev)
    oi = torch.empty((m, k), dtype=torch.int32, device=dev)
    ws = torch.empty(L.oibl_sqdist_topk_workspace_bytes(m, n, d, k, 0), dtype=torch.uint8, device=dev)
    rc = L.oibl_sqdist_topk(x.data_ptr(), m, y.data_ptr(), n, d, k, 0, 0, 0, ov.data_ptr(), oi.data_ptr(),
                            flag.data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc == 0 and int(flag.item()) == 1
    v, i = ops.sqdist_topk(x, y, k, precision="bf16")
    assert torch.equal(i.cpu(), torch.arange(k, dtype=torch.int32).repeat(m, 1))


def test_sqdist_topk_full_size_properties(dev):
    """8192 x 81920 x 4096 (the benchmark's matching problem, bf16): planted duplicates come back
    as nearest neighbour at distance ~0; a random row block agrees with the matrix path."""
    Q, G = 8192, 81920
    gq = torch.Generator(device=dev).manual_seed(7)
    q = torch.nn.functional.normalize(torch.randn((Q, 4096), generator=gq, device=dev), dim=1)
    g = torch.nn.functional.normalize(torch.randn((G, 4096), generator=gq, device=dev), dim=1)
    g[G - 300:] = q[:300]
    v, i = ops.sqdist_topk(q, g, 10, precision="bf16")
    assert torch.equal(i[:300, 0].cpu(), torch.arange(G - 300, G, dtype=torch.int32))
    assert v[:300, 0].abs().max().item() < 2e-2      # bf16 dot product of unit vectors
    rows = slice(4000, 4512)
    wv, wi = ops.row_topk(ops.pairwise_sqdist(q[rows].contiguous(), g, "bf16"), 10)
    assert torch.equal(i[rows], wi) and torch.equal(v[rows], wv)


@pytest.mark.parametrize("nms", [False, True])
@pytest.mark.parametrize("m,n,k", [(37, 500, 10), (200, 3000, 120), (5, 40, 130), (64, 2000, 1024)])
def test_first_hit_rank_equals_host_counting(dev, m, n, k, nms):
    """Device recall counting == the oracle's restatement of the reference's per-query loop
    (oracle.matching.recalls_from_ranking), with and without spatial NMS, on random rankings with
    duplicate pids, padding and empty ground truth; the host mirror recalls_from_topk agrees too."""
    from openibl_amd.evaluators import recalls_from_topk, recalls_from_topk_device
    rng = np.random.default_rng(m + n + k)
    kk = min(k, n)
    idx = np.stack([rng.permutation(n)[:kk] for _ in range(m)]).astype(np.int32)
    if k > n:
        idx = np.concatenate([idx, -np.ones((m, k - n), np.int32)], 1)
    pids = rng.integers(0, max(2, n // 12), size=n).tolist()       # ~12 views per place
    gt = [rng.choice(n, size=rng.integers(0, 9), replace=False).tolist() for _ in range(m)]
    for q in range(0, m, 3):                                        # make some queries easy
        gt[q] = gt[q] + [int(idx[q, rng.integers(0, min(kk, 15))])]
    for topk in ((1, 5, 10), (1, 5, 10, 20, 25)):
        if nms and max(topk) * 12 > 1024:
            continue
        # the checker is the oracle's restatement of evaluators.py:149-160 (pinned to the reference's
        # recalls by tests/test_oracle_golden.py); rows without their -1 padding
        want = om.recalls_from_ranking([row[row >= 0] for row in idx], gt, pids, topk, nms)
        got = recalls_from_topk_device(torch.from_numpy(idx).to(dev), gt, pids, topk, nms)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(recalls_from_topk(idx, gt, pids, topk, nms), want)


@pytest.mark.parametrize("precision", ["bf16", "bf16x3", "fp32"])
@pytest.mark.parametrize("store", [torch.float32, torch.bfloat16, torch.float16])
def test_prepared_rows_equal_direct_call(dev, precision, store):
    """A gallery prepared once (oibl_match_prepare) and matched through oibl_sqdist_topk_prepared
    gives the lists of the all-in-one call bit for bit — fused path (16k gallery) and exact path —
    and serves sharded_topk as the resident shard."""
    from openibl_amd import sharded
    q, gal, gt, _ = synth.retrieval_problem(300, 16384, seed=12)
    qd = q.to(dev)
    gd = ops.store_descriptors(gal.to(dev), store)
    want_v, want_i = ops.sqdist_topk(qd, gd, 10, index_base=7, precision=precision)
    gp = ops.PreparedRows(gd, precision)
    qp = ops.PreparedRows(qd, precision)
    v, i = ops.sqdist_topk_prepared(qp, gp, 10, index_base=7)
    assert torch.equal(v, want_v) and torch.equal(i, want_i)
    v, i = ops.sqdist_topk_prepared(qp, gp, 10, index_base=7, exact=True)
    assert torch.equal(v, want_v) and torch.equal(i, want_i)
    v, i = sharded.sharded_topk(qd, gp, 10, 7, precision)
    assert torch.equal(v, want_v) and torch.equal(i, want_i)
    # prepared queries re-assembled from their exchanged parts (what gather_prepared_queries ships)
    qx = sharded.gather_prepared_queries(torch.cat([qd, qd[:5]]), 300, precision)
    assert qx.shape == qp.shape and torch.equal(qx.norms, qp.norms)
    v, i = ops.sqdist_topk_prepared(qx, gp, 10, index_base=7)
    assert torch.equal(v, want_v) and torch.equal(i, want_i)
    small = ops.PreparedRows(gd[:100].contiguous(), precision)       # below every fused threshold
    sv, si = ops.sqdist_topk_prepared(qp, small, 10)
    wv, wi = ops.sqdist_topk(qd, gd[:100].contiguous(), 10, precision=precision)
    assert torch.equal(sv, wv) and torch.equal(si, wi)


def test_row_argsort_matches_reference_sort_gallery(dev):
    """oibl_row_argsort == the ranking the reference's sampler builds (golden from the reference's own
    sort_gallery), plus: stability on ties, sorted values, ragged sizes, strided input, chunked rows."""
    g = load_golden("sort_gallery")
    d = synth.tie_free_matrix(int(g["Q"]), int(g["G"]), int(g["seed"]))
    idx, vals = ops.row_argsort(d.to(dev), want_values=True)
    assert np.array_equal(idx.cpu().numpy(), g["sort_idx"].astype(np.int32))
    assert torch.equal(vals.cpu(), torch.sort(d, dim=1).values)
    rng = np.random.default_rng(0)
    for m, n in ((3, 1), (5, 63), (4, 1024), (2, 1025), (3, 5000), (1, 100003), (70, 777)):
        # few distinct values (many ties), negatives, +-0, inf
        x = torch.from_numpy(rng.integers(-3, 4, size=(m, n)).astype(np.float32) * 0.5)
        if n > 4:
            x[0, 1], x[0, 2], x[0, 3] = float("inf"), -0.0, float("-inf")
        want = torch.argsort(x, dim=1, stable=True)
        got = ops.row_argsort(x.to(dev), max_ws_bytes=(1 << 20) if m == 70 else (1 << 31))
        # -0.0 and +0.0 compare equal in torch; the kernel orders bit patterns (-0.0 first): compare
        # through the values, and exactly where no signed zero is involved
        assert torch.equal(torch.gather(x, 1, got.cpu().long()), torch.gather(x, 1, want))
        xz = x.clone()
        xz[xz == 0] = 0.0
        got2 = ops.row_argsort(xz.to(dev))
        assert torch.equal(got2.cpu().long(), torch.argsort(xz, dim=1, stable=True))
    big = torch.from_numpy(rng.standard_normal((6, 300)).astype(np.float32)).to(dev)
    view = big[:, :200]                                   # row stride 300, 200 columns
    assert torch.equal(ops.row_argsort(view).cpu().long(), torch.argsort(view.cpu(), dim=1, stable=True))


def test_evaluate_all_beyond_1024_ranks(dev):
    """recall_topk whose prefix exceeds the selection kernels' 1024 ranks: full-row device ranking,
    same numbers as the oracle's restatement of evaluators.py:142-167."""
    from ibl.evaluators import evaluate_all
    q, gal, gt, pids = synth.retrieval_problem(40, 3000, dim=256, seed=5, views_per_place=12)
    d = om.pairwise_distance(q, gal)
    gallery = [(f"g{j}", pids[j], 0.0, 0.0) for j in range(len(gal))]
    for topk, nms in (((1, 10, 100), True), ((1, 5, 1500), False)):
        got = evaluate_all(d, gt, gallery, recall_topk=list(topk), nms=nms)
        want = om.evaluate_all(d.numpy(), gt, pids, recall_topk=topk, nms=nms)
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_split_k_threshold_sample_gives_the_same_lists(dev, precision):
    """The fused path contracts its threshold sample in two K-halves when that fills the chip
    (<= 128 sample tiles); thresholds then carry a slack for the changed summation order.  The lists
    must equal the unsplit path's (and so the matrix path's) bit for bit — also with duplicates of
    the k-th neighbour sitting exactly on the threshold."""
    q, gal, gt, _ = synth.retrieval_problem(700, 20000, seed=14)
    gal[5000:5010] = gal[3]            # ten exact duplicates: ties on and around thresholds
    qd, gd = q.to(dev), gal.to(dev)
    ops.set_match_splitk(False)
    try:
        v0, i0 = ops.sqdist_topk(qd, gd, 10, precision=precision)
    finally:
        ops.set_match_splitk(True)
    v1, i1 = ops.sqdist_topk(qd, gd, 10, precision=precision)
    assert torch.equal(v0, v1) and torch.equal(i0, i1)
    wv, wi = ops.row_topk(ops.pairwise_sqdist(qd, gd, precision), 10)
    assert torch.equal(v1, wv) and torch.equal(i1, wi)
    # non-unit norms: the slack scales with |x||y|
    v2, i2 = ops.sqdist_topk((qd * 37.0).contiguous(), (gd * 0.2).contiguous(), 10, precision=precision)
    wv2, wi2 = ops.row_topk(ops.pairwise_sqdist((qd * 37.0).contiguous(), (gd * 0.2).contiguous(), precision), 10)
    assert torch.equal(v2, wv2) and torch.equal(i2, wi2)


def test_tuple_sampler_on_the_device_ranking(dev):
    """DistributedRandomTupleSampler.sort_gallery ranks on the GPU (oibl_row_argsort); the tuples equal
    the ones the reference's sampler yields from its torch.argsort."""
    from test_host_logic import _run_tuple_sampler
    torch.cuda.set_device(dev)
    _run_tuple_sampler(lambda smp, d, sub: smp.sort_gallery(d, sub))


def test_diff_tuple_sampler_on_the_device_ranking(dev):
    """DistributedRandomDiffTupleSampler.sort_gallery (the SFRS sampler, sampler.py:126-135) ranks on
    the GPU; the tuples equal the ones the reference's sampler yields."""
    from test_host_logic import _run_diff_tuple_sampler
    torch.cuda.set_device(dev)
    _run_diff_tuple_sampler(lambda smp, d, jac, sub: smp.sort_gallery(d, jac, sub))
