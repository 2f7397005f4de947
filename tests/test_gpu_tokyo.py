"""BASELINE.json configs[4]'s matching problem at its STATED size against the oracle: Tokyo 24/7 — 315 queries x
75 984 gallery descriptors (12 views of every place) x 4096-d, the top max(recall_topk) * 12 = 120 ranks per query,
spatial NMS over them, Recall@1/5/10 (the one flow of the reference that runs with nms=True:
examples/test.py:130, examples/test_tokyo_best.py:76, ibl/evaluators.py:132-140, 152-153).

Every arithmetic that serves it — fp32, f16mx, and f16r (what an f16mx model's Evaluator runs: ops.topk_precision;
its selection takes the 120-th filter distance by bisection, csrc/match_f16r.h) — on fp32-stored and on
fp16-stored descriptors (configs[4]'s "fp16 descriptors": the lists are those of the stored values widened to fp32),
as ONE shard and as 8 shards of 9 498 rows (DistributedSliceSampler's dealing, one GPU playing the ranks):

  * Recall@1/5/10 with nms=True (and without) `array_equal` to oracle.matching (the restatement of
    ibl/evaluators.py:105-167, pinned to outputs of the reference itself);
  * the 120-rank lists against the oracle's stable argsort: index disagreements only where fp64 calls the two
    candidates a near-tie (twelve near-duplicate views per place make near-ties the normal case here);
  * merged per-shard lists == the single-shard lists, bit for bit.

Two problems: `synth.tokyo_problem` (near-duplicate views, one true place + 12 distractor places per query: NMS
changes Recall@5/10) and the literal `synth.retrieval_problem(315, 75984, views_per_place=12)` VERDICT r05 names."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from openibl_amd import ops, sharded, synth
from openibl_amd.evaluators import recalls_from_topk, recalls_from_topk_device
from oracle import matching as om

pytestmark = pytest.mark.gpu

Q, G, V, K = 315, 75984, 12, 120
NEAR_TIE = 4e-6          # squared distances of unit vectors: the oracle's own fp32 rounding is ~1e-6


def _oracle(q, g, gt, pids):
    d = om.pairwise_distance(q, g).numpy()            # the reference's arithmetic on the host (ibl/evaluators.py:122-129)
    rank = om.ranking(d)[:, :K]                       # np.argsort, ties by lowest index
    return rank, om.recalls_from_ranking(rank, gt, pids, nms=True), om.recalls_from_ranking(rank, gt, pids, nms=False)


@pytest.fixture(scope="module")
def tokyo():
    q, g, gt, pids = synth.tokyo_problem(Q, G, views=V)
    out = {"gt": gt, "pids": pids}
    for name, dt in (("fp32", torch.float32), ("fp16", torch.float16)):
        qs, gs = q.to(dt), g.to(dt)                   # the STORED descriptors (round-to-nearest-even for fp16)
        rank, r_nms, r_plain = _oracle(qs.float(), gs.float(), gt, pids)
        out[name] = (qs, gs, rank, r_nms, r_plain)
    r_nms, r_plain = out["fp32"][3], out["fp32"][4]
    # a problem on which the NMS window matters and nothing is trivially 0 or 1
    assert r_nms[0] < r_nms[1] < r_nms[2] < 1.0 and r_plain[2] < r_nms[2] and r_plain[1] < r_nms[1], (r_nms, r_plain)
    return out


@pytest.fixture(scope="module")
def plain():
    q, g, gt, pids = synth.retrieval_problem(Q, G, views_per_place=V, seed=29, hard_fraction=0.5)
    rank, r_nms, r_plain = _oracle(q, g, gt, pids)
    return q, g, gt, pids, rank, r_nms, r_plain


def _near_ties(qs, gs, got, want):
    """Positions where the lists differ: the fp64 distance gap between the two candidates."""
    diff = np.argwhere(got != want)
    worst = 0.0
    q64, cache = qs.double(), {}
    for r, c in diff:
        for j in (int(got[r, c]), int(want[r, c])):
            if (r, j) not in cache:
                cache[(r, j)] = float(((q64[r] - gs[j].double()) ** 2).sum())
        worst = max(worst, abs(cache[(r, int(got[r, c]))] - cache[(r, int(want[r, c]))]))
    return len(diff), worst


def _check_lists(tag, qs, gs, gt, pids, idx_dev, want_rank, want_nms, want_plain):
    got = idx_dev.cpu().numpy()
    # Recall through the device counter (oibl_first_hit_rank: what Evaluator.evaluate runs) and the host loop
    np.testing.assert_array_equal(recalls_from_topk_device(idx_dev, gt, pids, nms=True), want_nms)
    np.testing.assert_array_equal(recalls_from_topk(got, gt, pids, nms=True), want_nms)
    np.testing.assert_array_equal(recalls_from_topk_device(idx_dev, gt, pids, nms=False), want_plain)
    agree = float((got == want_rank).mean())
    n_diff, worst = _near_ties(qs, gs, got, want_rank)
    print(f"{tag}: Recall@1/5/10 nms {want_nms} / plain {want_plain} equal; top-{K} agreement {agree:.6f}, "
          f"{n_diff} differing entries, all near-ties within {worst:.2e} (fp64)")
    assert agree >= 0.995 and worst < NEAR_TIE, tag
    return got


@pytest.mark.parametrize("storage", ["fp32", "fp16"])
@pytest.mark.parametrize("precision", ["fp32", "f16mx", "f16r"])
def test_tokyo_shape_top120_nms_recall_equals_oracle(dev, tokyo, precision, storage):
    qs, gs, want_rank, want_nms, want_plain = tokyo[storage]
    gt, pids = tokyo["gt"], tokyo["pids"]
    qd, gd = qs.to(dev), gs.to(dev)
    if precision == "f16r":
        # what an f16mx model's Evaluator asks for at k = 120, in either storage type — and served by the FUSED path
        assert ops.topk_precision("f16mx", qs.dtype, K) == ops.F16R
        assert ops.f16r_fused(Q, G, 4096, K) and ops.f16r_members(K) == 2 * K + 32
        v0, i0, flag = ops.sqdist_topk(qd, gd, K, precision="f16r", defer_check=True)
        assert int(flag.item()) == 0                    # no candidate-list / member-window overflow at this shape
    v, i = sharded.sharded_topk(qd, gd, K, 0, precision)             # what Evaluator.evaluate runs, one shard
    if precision == "f16r":
        assert torch.equal(i, i0) and torch.equal(v, v0)
    _check_lists(f"tokyo {precision} / {storage}-stored, 1 shard", qs.float(), gs.float(), gt, pids, i, want_rank,
                 want_nms, want_plain)
    # 8 shards of 9498 (75984 = 8 x 9498: DistributedSliceSampler's dealing), resident prepared shards, merged
    W, vs, is_ = 8, [], []
    qp = ops.PreparedRows(qd, precision)
    for r in range(W):
        start, per, n_valid = sharded.slice_bounds(G, r, W)
        assert per == 9498 and n_valid == 9498
        shard = ops.PreparedRows(gd[start:start + n_valid].contiguous(), precision)
        if precision == "f16r":
            assert ops.f16r_fused(Q, n_valid, 4096, K)
        sv, si, flag = sharded.hip_local_topk(qp, shard, K, start, precision)
        if int(flag.item()):                            # (legitimate; then the exact repeat decides — not at this shape)
            sv, si, _ = sharded.hip_local_topk(qp, shard, K, start, precision, exact=True)
            print(f"shard {r}: overflow flag -> exact repeat")
        vs.append(sv)
        is_.append(si)
    mv, mi = sharded.hip_merge_topk(torch.cat(vs, 1), torch.cat(is_, 1), K)
    # every arithmetic forms a pair's distance from the same operands in the same order whatever launch it is part
    # of (f16r: the rescored value), ties go to the lowest global index: the merge IS the single-shard list
    assert torch.equal(mi, i) and torch.equal(mv, v)
    _check_lists(f"tokyo {precision} / {storage}-stored, 8 shards", qs.float(), gs.float(), gt, pids, mi, want_rank,
                 want_nms, want_plain)


@pytest.mark.parametrize("precision", ["fp32", "f16mx", "f16r"])
def test_retrieval_problem_315_x_75984_views12_nms(dev, plain, precision):
    """The literal problem of VERDICT r05 item 1a: independent gallery rows, 12 consecutive rows per pid."""
    q, g, gt, pids, want_rank, want_nms, want_plain = plain
    v, i = sharded.sharded_topk(q.to(dev), g.to(dev), K, 0, precision)
    _check_lists(f"retrieval_problem(315, 75984, views_per_place=12) {precision}", q, g, gt, pids, i, want_rank,
                 want_nms, want_plain)


def test_f16r_values_are_the_rescored_distances(dev, tokyo):
    """fp16-stored gallery: the rescoring reads the STORED rows (widened exactly) — values equal the fp64 distance of
    the widened rows rounded once, and the fp32-stored and fp16-stored problems differ (the storage rounding is real)."""
    qs, gs, want_rank, _, _ = tokyo["fp16"]
    v, i = ops.sqdist_topk(qs.to(dev), gs.to(dev), K, precision="f16r")
    ii = i.cpu().long()
    q64, g64 = qs.double(), gs.double()
    rows = torch.arange(0, Q, 7)
    d64 = ((q64[rows, None, :] - g64[ii[rows]]) ** 2).sum(-1)
    assert float((v.cpu()[rows].double() - d64).abs().max()) <= 1e-6     # (fp32 norms + one rounding of a value ~2)
    q32, g32 = tokyo["fp32"][0], tokyo["fp32"][1]
    assert not torch.equal(qs.float(), q32)
    v32, i32 = ops.sqdist_topk(q32.to(dev), g32.to(dev), K, precision="f16r")
    assert float((v32 - v).abs().max()) > 1e-6


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16mx", "f16r"])
def test_tokyo_shaped_golden_of_the_reference(dev, precision):
    """tests/golden/match_tokyo.npz: a Tokyo-structured problem (48 x 7200 x 256-d, 12 near-duplicate views per place)
    through the REFERENCE's own pairwise_distance + evaluate_all(nms=True) (oracle/make_golden.py; examples/test.py:130):
    the device lists are its 120-rank prefix up to fp32 near-ties, Recall@1/5/10 with and without NMS are its numbers."""
    g = load_golden("match_tokyo")
    q, gal, gt, pids = synth.tokyo_problem(int(g["Q"]), int(g["G"]), dim=int(g["dim"]), seed=int(g["seed"]),
                                           views=int(g["views"]), distractors=int(g["distractors"]))
    v, i = sharded.sharded_topk(q.to(dev), gal.to(dev), K, 0, precision)
    np.testing.assert_array_equal(recalls_from_topk_device(i, gt, pids, nms=True), g["recalls_nms"])
    np.testing.assert_array_equal(recalls_from_topk_device(i, gt, pids, nms=False), g["recalls"])
    got = i.cpu().numpy()
    n_diff, worst = _near_ties(q, gal, got, g["top120"].astype(np.int64))
    print(f"match_tokyo {precision}: {n_diff} of {got.size} ranks differ from the reference's, within {worst:.2e}")
    assert n_diff <= 0.002 * got.size and worst < NEAR_TIE
    assert float(np.abs(v.cpu().numpy() - g["top120_dist"]).max()) < 2e-5
    # the Evaluator-level entry point on the reference's matrix semantics: evaluate_all on a device matrix
    from ibl.evaluators import evaluate_all
    d = ops.pairwise_sqdist(q.to(dev), gal.to(dev), "fp32" if precision == "f16r" else precision)
    gallery = [(f"g{j:05d}.jpg", pids[j], 0.0, 0.0) for j in range(len(gal))]
    np.testing.assert_array_equal(evaluate_all(d.cpu(), gt, gallery, nms=True), g["recalls_nms"])
