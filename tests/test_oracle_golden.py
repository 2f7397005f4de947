"""The oracle (oracle/*.py) against the vectors the REFERENCE ITSELF produced
(tests/golden/*.npz, written by oracle/make_golden.py from /root/reference).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import assert_rel_l2, load_golden
from openibl_amd import synth
from oracle import descriptor as od
from oracle import matching as om

# fp32 on a possibly different CPU (other SIMD width -> other summation order in oneDNN/MKL)
TOL = 2e-5


@pytest.mark.parametrize("name", ["desc_small", "desc_odd", "desc_480x640"])
def test_descriptor_pipeline_matches_reference(name, state_dict):
    g = load_golden(name)
    n, _, h, w = [int(v) for v in g["shape"]]
    x = synth.images(n, h, w, seed=int(g["image_seed"]))
    with torch.no_grad():
        out = od.embednetpca(x, state_dict, return_intermediates=True)
        pool = od.global_max(out["feat"])
        ecf = od.extract_cnn_feature(x, state_dict)
        ecf_vlad = od.extract_cnn_feature(x, state_dict, vlad=True, with_pca=False)
        ecf_pool = od.extract_cnn_feature(x, state_dict, vlad=False, with_pca=False)
    s = int(g["feat_stride"])
    assert_rel_l2("feat", out["feat"][:, ::s], g["feat"], TOL)
    assert_rel_l2("pool_x", pool, g["pool_x"], TOL)
    assert_rel_l2("vlad_raw", out["vlad_raw"], g["vlad_raw"], 5e-5)
    assert_rel_l2("vlad_norm", out["vlad_norm"], g["vlad_norm"], 5e-5)
    assert_rel_l2("desc", out["desc"], g["desc"], 5e-5)
    assert_rel_l2("extract_cnn_feature(pca)", ecf, g["ecf_pca"], 5e-5)
    assert_rel_l2("extract_cnn_feature(vlad)", ecf_vlad, g["ecf_vlad"], 5e-5)
    assert_rel_l2("extract_cnn_feature(pool)", ecf_pool, g["ecf_pool"], TOL)


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
    assert_rel_l2("fp32 vs fp64 oracle", a, b, 2e-5)


@pytest.mark.parametrize("tag", ["whiten", "nowhiten"])
def test_pca_projection_matches_reference(tag):
    g = load_golden("pca")
    w = torch.from_numpy(g[f"weight_{tag}"])
    b = torch.from_numpy(g[f"bias_{tag}"])
    out = od.pca_project(torch.from_numpy(g["data"]), w, b)
    assert_rel_l2(f"pca {tag}", out, g[f"out_{tag}"], TOL)


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


def test_full_row_ranking_matches_reference_sort_gallery():
    """oracle.matching.ranking == the sort_idx the reference's DistributedRandomTupleSampler
    .sort_gallery produced (torch.argsort, ibl/utils/data/sampler.py:46-54) on a tie-free matrix."""
    g = load_golden("sort_gallery")
    d = synth.tie_free_matrix(int(g["Q"]), int(g["G"]), int(g["seed"]))
    assert all(len(np.unique(r)) == d.shape[1] for r in d.numpy())
    assert np.array_equal(om.ranking(d.numpy()), g["sort_idx"].astype(np.int64))
