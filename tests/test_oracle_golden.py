"""The oracle (oracle/*.py) against the vectors the REFERENCE ITSELF produced
(tests/golden/*.npz, written by oracle/make_golden.py from /root/reference).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import assert_rel_l2, load_golden
from openibl_amd import synth
from oracle import descriptor as od
from oracle import matching as om

# fp32 on a possibly different CPU (other SIMD width -> other summation order in oneDNN/MKL).  Round 6 (VERDICT r05 item
# 8): the bound is 2.7x the largest difference the oracle shows against the reference's vectors in this container
# (descriptors 7.3e-7, normalised VLAD 1.9e-7, feature maps bit-equal; it was 2e-5 / 5e-5 — half of north_star's
# budget), so that the chain reference -> oracle -> HIP path (f16mx <= 4e-5) cannot add up to 1e-4.
TOL = 2e-6
OBSERVED = {}


def _close(name, got, want, tol):
    from conftest import rel_l2
    OBSERVED[name] = max(OBSERVED.get(name, 0.0), rel_l2(got, want))
    assert_rel_l2(name, got, want, tol)


@pytest.mark.parametrize("name", ["desc_small", "desc_odd", "desc_480x640", "desc_480x640_n8"])
def test_descriptor_pipeline_matches_reference(name, state_dict):
    g = load_golden(name)
    n, _, h, w = [int(v) for v in g["shape"]]
    x = synth.images(n, h, w, seed=int(g["image_seed"]))
    with torch.no_grad():
        out = od.embednetpca(x, state_dict, return_intermediates=True)
        pool = od.global_max(out["feat"])
        ecf = od.extract_cnn_feature(x, state_dict)
    s = int(g["feat_stride"])
    _close("feat", out["feat"][:, ::s], g["feat"], TOL)
    _close("pool_x", pool, g["pool_x"], TOL)
    _close("vlad_norm", out["vlad_norm"], g["vlad_norm"], TOL)
    _close("desc", out["desc"], g["desc"], TOL)
    _close("extract_cnn_feature(pca)", ecf, g["ecf_pca"], TOL)
    if "vlad_raw" in g:          # (the batch-8 fixture keeps only what pins a batch: make_golden.py, light=True)
        with torch.no_grad():
            ecf_vlad = od.extract_cnn_feature(x, state_dict, vlad=True, with_pca=False)
            ecf_pool = od.extract_cnn_feature(x, state_dict, vlad=False, with_pca=False)
        _close("vlad_raw", out["vlad_raw"], g["vlad_raw"], TOL)
        _close("extract_cnn_feature(vlad)", ecf_vlad, g["ecf_vlad"], TOL)
        _close("extract_cnn_feature(pool)", ecf_pool, g["ecf_pool"], TOL)
    print("observed rel-L2, oracle vs the reference's vectors:", {k: f"{v:.2e}" for k, v in OBSERVED.items()})


def test_netvlad_gemm_form_equals_residual_form():
    """The contraction form used by the oracle (and the kernels) against the reference's literal
    residual formulation (netvlad.py:56-59)."""
    torch.manual_seed(0)
    feat = torch.randn(2, 512, 5, 7, dtype=torch.float64)
    sd = synth.netvlad_state(0)
    w, c = sd["net_vlad.conv.weight"].double(), sd["net_vlad.centroids"].double()
    a = od.netvlad(feat, w, c)
    b = od.netvlad_residual_form(feat, w, c)
    assert_rel_l2("netvlad forms", a, b, 1e-12)


def test_oracle_fp64_close_to_fp32(state_dict):
    x = synth.images(1, 64, 96, seed=11)
    with torch.no_grad():
        a = od.embednetpca(x, state_dict, dtype=torch.float32)
        b = od.embednetpca(x, state_dict, dtype=torch.float64)
    assert_rel_l2("fp32 vs fp64 oracle", a, b, 4e-6)


@pytest.mark.parametrize("tag", ["whiten", "nowhiten"])
def test_pca_projection_matches_reference(tag):
    g = load_golden("pca")
    w = torch.from_numpy(g[f"weight_{tag}"])
    b = torch.from_numpy(g[f"bias_{tag}"])
    out = od.pca_project(torch.from_numpy(g["data"]), w, b)
    assert_rel_l2(f"pca {tag}", out, g[f"out_{tag}"], 2e-6)


@pytest.mark.parametrize("name", ["match_small", "match_nms"])
def test_matching_matches_reference(name):
    g = load_golden(name)
    q, gal, gt, pids = synth.retrieval_problem(
        int(g["Q"]), int(g["G"]), dim=int(g["dim"]), seed=int(g["seed"]),
        views_per_place=int(g["views_per_place"]), hard_fraction=float(g["hard_fraction"]),
        hard_noise_mult=float(g["hard_noise_mult"]))
    d = om.pairwise_distance(q, gal)
    np.testing.assert_allclose(d.numpy(), g["distmat"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(om.pairwise_distance_all(torch.cat([q, gal])[:40]).numpy(),
                               g["dist_all40"], rtol=0, atol=2e-6)
    # ranking / recall on the reference's own matrix (tie-free data)
    assert np.array_equal(om.ranking(g["distmat"])[:, :20], g["top20"])
    np.testing.assert_array_equal(om.evaluate_all(g["distmat"], gt, pids), g["recalls"])
    np.testing.assert_array_equal(om.evaluate_all(g["distmat"], gt, pids, nms=True),
                                  g["recalls_nms"])
    order = om.ranking(g["distmat"])
    for i, row in enumerate(g["nms_rows"]):
        want = [int(v) for v in row if v >= 0]
        assert om.spatial_nms(order[i].tolist(), pids, 120) == want


def test_tokyo_shaped_nms_flow_matches_reference():
    """A Tokyo 24/7-shaped problem (12 near-duplicate views per place, distractor places: synth.tokyo_problem) through
    the REFERENCE's pairwise_distance + evaluate_all(nms=True) (examples/test.py:130; tests/golden/match_tokyo.npz):
    the oracle's matrix, its 120-rank prefix, the NMS'd lists and both recalls are the reference's."""
    g = load_golden("match_tokyo")
    q, gal, gt, pids = synth.tokyo_problem(int(g["Q"]), int(g["G"]), dim=int(g["dim"]), seed=int(g["seed"]),
                                           views=int(g["views"]), distractors=int(g["distractors"]))
    d = om.pairwise_distance(q, gal).numpy()
    order = om.ranking(d)
    np.testing.assert_allclose(np.take_along_axis(d, order[:, :120], axis=1), g["top120_dist"], rtol=0, atol=2e-6)
    agree = (order[:, :120] == g["top120"])
    if not agree.all():          # another CPU may round a near-tie the other way: then the two distances are within 2e-6
        r, c = np.argwhere(~agree).T
        assert np.abs(d[r, order[r, c]] - d[r, g["top120"][r, c]]).max() <= 2e-6
    assert agree.mean() > 0.999
    np.testing.assert_array_equal(om.evaluate_all(d, gt, pids), g["recalls"])
    np.testing.assert_array_equal(om.evaluate_all(d, gt, pids, nms=True), g["recalls_nms"])
    assert g["recalls_nms"][2] > g["recalls"][2]                       # the NMS window matters on this problem
    for i, row in enumerate(g["nms_rows"]):
        want = [int(v) for v in row if v >= 0]
        assert om.spatial_nms(order[i].tolist(), pids, 120) == want


def test_full_row_ranking_matches_reference_sort_gallery():
    """oracle.matching.ranking == the sort_idx the reference's DistributedRandomTupleSampler
    .sort_gallery produced (torch.argsort, ibl/utils/data/sampler.py:46-54) on a tie-free matrix."""
    g = load_golden("sort_gallery")
    d = synth.tie_free_matrix(int(g["Q"]), int(g["G"]), int(g["seed"]))
    assert all(len(np.unique(r)) == d.shape[1] for r in d.numpy())
    assert np.array_equal(om.ranking(d.numpy()), g["sort_idx"].astype(np.int64))
