"""Host-side logic that needs no GPU: slice dealing, recall counting from top-k lists, spatial NMS,
state-dict contract, checkpoint helpers, PCA parameter math, transforms."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from openibl_amd import sharded, synth
from oracle import matching as om


@pytest.mark.parametrize("L,W", [(10, 4), (8, 8), (7, 8), (83952, 8), (1, 3), (0, 2), (17, 1)])
def test_slice_sampler_deals_contiguous_wrapped_slices(L, W):
    from ibl.utils.data.sampler import DistributedSliceSampler
    data = list(range(L))
    per = -(-L // W) if L else 0
    seen = []
    for r in range(W):
        s = DistributedSliceSampler(data, num_replicas=W, rank=r)
        idx = list(s)
        assert len(idx) == len(s) == per
        start, per2, n_valid = sharded.slice_bounds(L, r, W)
        assert per2 == per and idx[:n_valid] == list(range(start, start + n_valid))
        assert all(i == (start + k) % L for k, i in enumerate(idx))     # wrap-around padding
        seen += idx
    # rank-major concatenation truncated to L is the identity (what extract_features relies on)
    assert seen[:L] == data


def test_recalls_from_topk_equals_reference_counting():
    from ibl.evaluators import recalls_from_topk, spatial_nms
    for name in ("match_small", "match_nms"):
        g = load_golden(name)
        _, _, gt, pids = synth.retrieval_problem(
            int(g["Q"]), int(g["G"]), dim=int(g["dim"]), seed=int(g["seed"]),
            views_per_place=int(g["views_per_place"]), hard_fraction=float(g["hard_fraction"]),
            hard_noise_mult=float(g["hard_noise_mult"]))
        order = om.ranking(g["distmat"])
        np.testing.assert_array_equal(recalls_from_topk(order[:, :10], gt), g["recalls"])
        np.testing.assert_array_equal(recalls_from_topk(order[:, :120], gt, pids, nms=True),
                                      g["recalls_nms"])
        for i, row in enumerate(g["nms_rows"]):
            assert spatial_nms(order[i].tolist(), pids, 120) == [int(v) for v in row if v >= 0]


def test_recalls_hypothesis_style_random_cases():
    from ibl.evaluators import recalls_from_topk
    rng = np.random.default_rng(0)
    for _ in range(25):
        Q, G = int(rng.integers(1, 20)), int(rng.integers(12, 200))
        d = rng.standard_normal((Q, G)).astype(np.float32)
        gt = [sorted(rng.choice(G, size=int(rng.integers(1, 4)), replace=False).tolist()) for _ in range(Q)]
        pids = (np.arange(G) // int(rng.integers(1, 5))).tolist()
        nms = bool(rng.integers(0, 2))
        k = min(G, 120 if nms else 10)
        got = recalls_from_topk(om.ranking(d)[:, :k], gt, pids, nms=nms)
        np.testing.assert_array_equal(got, om.evaluate_all(d, gt, pids, nms=nms))


def test_state_dict_keys_and_shapes_match_reference():
    import hubconf
    m = hubconf.vgg16_netvlad()
    sd, want = m.state_dict(), synth.embednetpca_state(0)
    assert list(sd) == list(want)                       # same keys, same order
    assert all(tuple(sd[k].shape) == tuple(want[k].shape) for k in sd)
    assert sum(v.numel() for v in sd.values()) == 149002048
    from ibl import models
    assert models.names() == ["embednet", "embednetpca", "embedregionnet", "netvlad", "vgg16"]
    with pytest.raises(KeyError):
        models.create("resnet50")
    v = models.create("vgg16", pretrained=False)
    assert v.feature_dim == 512 and len(v.base) == 29
    assert [i for i, l in enumerate(v.base) if isinstance(l, torch.nn.Conv2d)] == list(synth.CONV_IDX)


def test_copy_state_dict_strip_and_mismatch(tmp_path, capsys):
    from ibl.utils.serialization import copy_state_dict, save_checkpoint, load_checkpoint
    from ibl import models
    net = models.create("netvlad", dim=512)
    src = {"module.centroids": torch.full((64, 512), 3.0), "module.conv.weight": torch.zeros(7, 7),
           "module.unknown": torch.zeros(1)}
    copy_state_dict(src, net, strip="module.")
    assert float(net.centroids.min()) == 3.0
    assert "mismatch" in capsys.readouterr().out
    save_checkpoint({"state_dict": net.state_dict(), "epoch": 3}, True, str(tmp_path / "c.pth.tar"))
    assert (tmp_path / "model_best.pth.tar").exists()
    assert load_checkpoint(str(tmp_path / "c.pth.tar"))["epoch"] == 3
    with pytest.raises(ValueError):
        load_checkpoint(str(tmp_path / "missing.pth.tar"))


def test_pca_train_then_projection_whitens(tmp_path):
    """PCA.train (host, offline) writes parameters whose whitening projection has unit variance."""
    from ibl.pca import PCA
    from openibl_amd import pca as pmod
    rng = np.random.default_rng(1)
    x = torch.from_numpy((rng.standard_normal((400, 32)) @ rng.standard_normal((32, 32))).astype(np.float32))
    p = PCA(pca_n_components=8, pca_whitening=True, pca_parameters_path=str(tmp_path / "pca.npz"))
    p.train(x)
    U, lams, mu, Utmu = pmod._read_params(str(tmp_path / "pca.npz"))
    W = (U[:, :8] @ np.diag(1.0 / np.sqrt(lams[:8]))).T
    y = (x.numpy() - mu.T) @ W.T
    np.testing.assert_allclose(y.var(axis=0, ddof=1), np.ones(8), rtol=2e-3)
    np.testing.assert_allclose(U.T @ mu, Utmu, atol=1e-4)


def test_test_transform_matches_reference_normalisation():
    from PIL import Image
    from ibl.utils.data import get_transformer_test
    rng = np.random.default_rng(2)
    img = Image.fromarray(rng.integers(0, 256, size=(48, 64, 3), dtype=np.uint8))
    t = get_transformer_test(48, 64)(img)
    want = (np.asarray(img, dtype=np.float32).transpose(2, 0, 1) / 255.0
            - np.array(synth.MEAN, dtype=np.float32)[:, None, None]) / np.float32(synth.STD)
    np.testing.assert_allclose(t.numpy(), want, rtol=0, atol=1e-4)
    assert tuple(get_transformer_test(24, 32)(img).shape) == (3, 24, 32)
    assert tuple(get_transformer_test(24, 32, tokyo=True)(img).shape) == (3, 32, 42)


def test_init_dist_rejects_unknown_launcher():
    from ibl.utils.dist_utils import init_dist
    with pytest.raises(ValueError):
        init_dist("none", None)


def test_re_ranking_matches_reference_golden():
    """openibl_amd.rerank.re_ranking against outputs of the reference's ibl.utils.rerank.re_ranking
    (tests/golden/rerank_small.npz, produced by oracle/make_golden.py)."""
    import numpy as np
    from conftest import load_golden
    from openibl_amd import synth
    from openibl_amd.rerank import re_ranking
    g = load_golden("rerank_small")
    q, gal, _, _ = synth.retrieval_problem(int(g["Q"]), int(g["G"]), dim=int(g["dim"]), seed=int(g["seed"]),
                                           views_per_place=4, hard_fraction=0.5, hard_noise_mult=35.0)
    d = lambda a, b: ((a * a).sum(1)[:, None] + (b * b).sum(1)[None] - 2 * a @ b.t()).numpy()
    qg, qq, gg = d(q, gal), d(q, q), d(gal, gal)
    for key, (k1, k2, lam) in {"k20_6_3": (20, 6, 0.3), "k25_1_0": (25, 1, 0.0), "k10_3_5": (10, 3, 0.5)}.items():
        got = re_ranking(qg.copy(), qq.copy(), gg.copy(), k1=k1, k2=k2, lambda_value=lam)
        assert got.shape == g[key].shape and got.dtype == np.float32
        np.testing.assert_allclose(got, g[key], rtol=0, atol=2e-6)


def test_multiscale_definition_host_side(state_dict):
    """configs[4] extension: product and oracle agree on the scaled sizes; the oracle definition
    with the single scale 1.0 is the plain descriptor; the fused descriptor is unit-norm; and the
    product refuses to run without the HIP extension's device (no CPU fallback)."""
    from openibl_amd import multiscale
    from openibl_amd.lib import OpenIBLAmdError
    from oracle import descriptor as od
    for H, W in [(480, 640), (33, 47), (16, 16), (20, 700)]:
        for s in multiscale.DEFAULT_SCALES + (0.01, 1.5):
            assert multiscale.scaled_size(H, W, s) == od.scaled_size(H, W, s)
    assert multiscale.scaled_size(480, 640, 2.0 ** -0.5) == (339, 453)
    x = synth.images(1, 32, 48, seed=4)
    with torch.no_grad():
        one = od.multiscale_descriptor(x, state_dict, (1.0,))
        ref = od.extract_cnn_feature(x, state_dict)
        three = od.multiscale_descriptor(x, state_dict)
    assert torch.allclose(one, ref, atol=1e-7)
    assert abs(float(three.norm()) - 1.0) < 1e-5
    with pytest.raises((OpenIBLAmdError, ValueError)):
        multiscale.extract_multiscale(lambda t: t, x)      # CPU tensor: must not silently compute


def test_json_dataset_has_every_attribute_test_py_reads(tmp_path):
    """examples/test.py:33-56 reads q_train / db_train (PCA training set), q_test / db_test /
    test_pos and images_dir from `datasets.create('pitts', ...)`."""
    from helpers import synthetic_pitts
    from ibl import datasets
    root = synthetic_pitts.make(str(tmp_path / "pitts"))
    ds = datasets.create("pitts", root, scale="30k", verbose=False)
    assert len(ds.q_train) == 6 and len(ds.db_train) == 14 and len(ds.train) == 20
    assert len(ds.train_pos) == len(ds.train_neg) == len(ds.q_train)
    assert len(sorted(set(ds.q_train) | set(ds.db_train))) == 20          # test.py:38
    assert len(ds.q_test) == 6 and len(ds.db_test) == 14 and all(len(p) == 1 for p in ds.test_pos)
    assert len(ds.q_val) == 6 and len(ds.val_pos) == 6
    assert ds.images_dir.endswith("raw")
    with pytest.raises(RuntimeError):
        datasets.create("pitts", str(tmp_path / "nowhere"), scale="30k")


@pytest.mark.skipif(not __import__("os").path.isfile("/root/reference/examples/test.py"),
                    reason="the reference tree is only present in the build container")
def test_reference_test_py_get_data_runs_against_this_package(tmp_path):
    """The reference's own examples/test.py, unmodified and loaded from where it lies, builds its
    datasets / transforms / samplers / loaders through THIS repo's `ibl` (get_data, test.py:29-56)."""
    import subprocess
    import sys
    from pathlib import Path
    from helpers import synthetic_pitts
    repo = Path(__file__).resolve().parent.parent
    synthetic_pitts.make(str(tmp_path / "pitts"))
    r = subprocess.run([sys.executable, str(repo / "tests" / "helpers" / "ref_testpy_probe.py"),
                        str(tmp_path), "/root/reference/examples/test.py", str(repo)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "REFERENCE_GET_DATA_OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.skipif(not __import__("os").path.isfile("/root/reference/examples/cluster.py"),
                    reason="the reference tree is only present in the build container")
def test_reference_cluster_py_runs_against_this_package(tmp_path):
    """The reference's own examples/cluster.py, unmodified: its imports resolve to THIS repo's `ibl`
    (SubsetRandomSampler from ibl.utils.data.sampler included), get_data() builds its loader and the
    model factory takes its get_model() arguments (cluster.py:27-47)."""
    import subprocess
    import sys
    from pathlib import Path
    from helpers import synthetic_pitts
    repo = Path(__file__).resolve().parent.parent
    synthetic_pitts.make(str(tmp_path / "pitts"))
    r = subprocess.run([sys.executable, str(repo / "tests" / "helpers" / "ref_clusterpy_probe.py"),
                        str(tmp_path), "/root/reference/examples/cluster.py", str(repo)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "REFERENCE_CLUSTER_OK" in r.stdout, r.stdout + r.stderr


def test_logger_tees_and_leaves_stdout_open(tmp_path, capsys):
    import sys
    from ibl.utils.logging import Logger
    path = tmp_path / "sub" / "log.txt"
    with Logger(str(path)) as lg:
        old, sys.stdout = sys.stdout, lg
        try:
            print("hello tee")
            sys.stdout.flush()
        finally:
            sys.stdout = old
        assert lg.file is not None and lg.console is old
    assert lg.file is None
    assert path.read_text() == "hello tee\n"
    assert "hello tee" in capsys.readouterr().out
    print("stdout still open")
    Logger().write("no file\n")


def test_pca_param_file_is_found_again_under_the_h5_name(tmp_path):
    """examples/test.py:109-111 decides with osp.isfile('<...>.h5') whether to train the PCA again:
    whatever container `train` writes must sit AT that path and be readable by `load`."""
    import os.path as osp
    import openibl_amd.pca as pmod
    g = torch.Generator().manual_seed(0)
    x = torch.randn((50, 24), generator=g)
    path = str(tmp_path / "logs" / "pca_params_model_best.h5")
    p = pmod.PCA(pca_n_components=8, pca_whitening=True, pca_parameters_path=path)
    assert not osp.isfile(path)
    p.train(x)
    assert osp.isfile(path)
    U, lams, mu, Utmu = pmod._read_params(path)
    assert U.shape == (24, 8) and lams.shape == (8,) and mu.shape == (24, 1) and Utmu.shape == (8, 1)
    np.testing.assert_allclose(U.T @ mu, Utmu, rtol=1e-4, atol=1e-5)


def test_recall_prefix_beyond_the_kernel_limit_is_an_error():
    from openibl_amd import evaluators as ev
    ev._check_prefix(1024)
    with pytest.raises(ValueError):
        ev._check_prefix(1025)


def _run_tuple_sampler(rank_rows):
    import random
    from ibl.utils.data.sampler import DistributedRandomTupleSampler
    g = load_golden("tuple_sampler")
    Q, G, seed = int(g["Q"]), int(g["G"]), int(g["seed"])
    pos, neg = synth.tuple_lists(Q, G, seed)
    d = synth.tie_free_matrix(Q, G, seed)
    for r in range(2):
        smp = DistributedRandomTupleSampler(list(range(Q)), list(range(G)), pos, neg, neg_num=5, neg_pool=40,
                                            num_replicas=2, rank=r)
        random.seed(1000 + r)
        for ep in range(2):
            rank_rows(smp, d, list(range(1, Q)))
            assert len(smp) == 5
            got = np.asarray(list(iter(smp)), dtype=np.int32)
            np.testing.assert_array_equal(got, g[f"r{r}_e{ep}"])


def test_tuple_sampler_yields_the_reference_tuples():
    """The mining sampler's tuple bookkeeping (easiest positive, hardest negatives from a random pool
    plus last epoch's cache, replica slicing with padding) against the tuples the reference's own
    DistributedRandomTupleSampler yielded (tests/golden/tuple_sampler.npz) — with the ranking taken
    from the oracle here (the device ranking is the GPU test's subject)."""
    def rank_rows(smp, d, sub):
        smp.sort_idx = torch.from_numpy(om.ranking(d.numpy()))
        smp._set_subset(sub)
    _run_tuple_sampler(rank_rows)


def _run_diff_tuple_sampler(rank_rows):
    import random
    from ibl.utils.data.sampler import DistributedRandomDiffTupleSampler
    g = load_golden("diff_tuple_sampler")
    Q, G, seed = int(g["Q"]), int(g["G"]), int(g["seed"])
    pos, neg = synth.tuple_lists(Q, G, seed, positives=8)
    d = synth.tie_free_matrix(Q, G, seed)
    jac = synth.tie_free_matrix(Q, G, seed + 1, scale=1.0)
    for r in range(2):
        smp = DistributedRandomDiffTupleSampler(list(range(Q)), list(range(G)), pos, neg, pos_num=4, pos_pool=6,
                                                neg_num=5, neg_pool=40, num_replicas=2, rank=r)
        random.seed(2000 + r)
        for ep in range(2):
            rank_rows(smp, d, jac, list(range(1, Q)))
            assert len(smp) == 5
            want = g[f"r{r}_e{ep}"]
            got = [t + [-1] * (want.shape[1] - len(t)) for t in iter(smp)]
            np.testing.assert_array_equal(np.asarray(got, dtype=np.int32), want)


def test_diff_tuple_sampler_yields_the_reference_tuples():
    """The SFRS mining sampler (difficult positives by Jaccard promotion on top of the tuple sampler's
    bookkeeping) against the tuples the reference's own DistributedRandomDiffTupleSampler yielded
    (tests/golden/diff_tuple_sampler.npz); ranking from the oracle here, from the device in the GPU test."""
    def rank_rows(smp, d, jac, sub):
        smp.sort_idx = torch.from_numpy(om.ranking(d.numpy()))
        smp.distmat_jac = jac
        smp._set_subset(sub)
    _run_diff_tuple_sampler(rank_rows)


def test_bench_flop_accounting_matches_baseline_md():
    """bench.py's algorithmic FLOP model of one 480x640 image is BASELINE.md §2 / SURVEY §8d's:
    backbone 187.918 GFLOP (conv1_1 1.062), NetVLAD 0.157, PCA 0.268 -> 188.344 GFLOP; importing
    bench.py needs no GPU."""
    import importlib.util
    import pathlib
    spec = importlib.util.spec_from_file_location("bench_mod", pathlib.Path(__file__).resolve().parent.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert abs(bench.conv11_flops_per_image() / 1e9 - 1.062) < 1e-3
    assert abs((bench.igemm_flops_per_image() + bench.conv11_flops_per_image()) / 1e9 - 187.918) < 1e-3
    assert abs(bench.total_flops_per_image() / 1e9 - 188.344) < 2e-3
    # odd sizes: every pool floors
    assert bench.igemm_flops_per_image(479, 637) < bench.igemm_flops_per_image(480, 640)


def _np_assign(x, c):
    d = (x.astype(np.float64) ** 2).sum(1)[:, None] + (c.astype(np.float64) ** 2).sum(1)[None] \
        - 2 * x.astype(np.float64) @ c.astype(np.float64).T
    return d.argmin(1).astype(np.int32)


def _np_update(x, lab, c):
    cn, cnt = c.copy(), np.zeros(c.shape[0], np.int32)
    for k in range(c.shape[0]):
        m = lab == k
        cnt[k] = m.sum()
        if cnt[k]:
            cn[k] = x[m].astype(np.float64).mean(0).astype(np.float32)
    return cn, cnt


def kmeans_inertia(x, c):
    d = ((x[:, None, :].astype(np.float64) - c[None].astype(np.float64)) ** 2).sum(2)
    return float(d.min(1).sum())


def check_kmeans_against_golden(run):
    """Shared by the CPU test (numpy device steps) and the GPU test (HIP device steps).  Cases 0 and 2
    are well conditioned (K <= number of blobs): same number of iterations as the reference's call
    and centres to one float32 ulp.  Case 1 over-clusters (24 centres on 10 blobs, as the real use
    does: 64 centres on a continuum of descriptors): a single near-tie that rounds the other way —
    scikit-learn ranks float32 GEMM distances, chunked per thread — flips one label and the
    trajectories part for good (scikit-learn itself is not reproducible across thread counts there);
    what is comparable is the quality of the optimum: inertia within 0.5 % of the reference's."""
    import pathlib
    from openibl_amd import synth
    g = np.load(pathlib.Path(__file__).parent / "golden" / "kmeans.npz")
    for i in range(3):
        n, d, K, blobs, seed, n_iter = (int(v) for v in g[f"case{i}"])
        x = synth.kmeans_points(n, d, blobs, seed=seed)
        c, it = run(x, K, int(g["seed"]))
        assert c.dtype == np.float32 and c.shape == (K, d)
        if K <= blobs:
            assert it == n_iter, (i, it, n_iter)
            assert np.abs(c - g[f"centers{i}"]).max() <= 2e-7, (i, np.abs(c - g[f"centers{i}"]).max())
        else:
            ours, ref = kmeans_inertia(x, c), kmeans_inertia(x, g[f"centers{i}"])
            print(f"over-clustered case: {it} vs {n_iter} iterations, inertia {ours:.4f} vs {ref:.4f}")
            assert abs(ours - ref) <= 5e-3 * ref


def test_kmeans_host_logic_reproduces_the_reference_call():
    """openibl_amd.cluster.kmeans_centroids — scikit-learn's KMeans.fit restated around two injected
    device steps — against tests/golden/kmeans.npz, the output of the reference's own call
    (examples/cluster.py:110-115) on the same seeded points.  The device steps are exact numpy
    stand-ins here; the GPU test runs the HIP ones."""
    from openibl_amd import cluster
    check_kmeans_against_golden(lambda x, K, seed: cluster.kmeans_centroids(
        x, K, 100, seed, assign_fn=_np_assign, update_fn=_np_update, return_n_iter=True))
    with pytest.raises(ValueError):
        cluster.kmeans_centroids(np.zeros((3, 8), np.float32), 4, assign_fn=_np_assign, update_fn=_np_update)


def test_kmeans_empty_cluster_relocation_matches_sklearn():
    """A cluster that loses all its points is moved to the point farthest from its own centre, as in
    scikit-learn's _relocate_empty_clusters_dense: duplicated points make k-means++ pick identical
    seeds, which forces empty clusters in the first iteration."""
    from sklearn.cluster import KMeans
    from openibl_amd import cluster
    rng = np.random.default_rng(5)
    base = rng.standard_normal((6, 8)).astype(np.float32)
    x = np.concatenate([np.repeat(base, 40, axis=0), rng.standard_normal((30, 8)).astype(np.float32) * 0.05])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        km = KMeans(n_clusters=12, max_iter=100, random_state=43).fit(x.copy())
    c, it = cluster.kmeans_centroids(x, 12, 100, 43, assign_fn=_np_assign, update_fn=_np_update, return_n_iter=True)
    assert it == km.n_iter_
    assert np.abs(c - km.cluster_centers_).max() <= 1e-6


_SURFACE = {"ibl.evaluators": "ibl/evaluators.py", "ibl.pca": "ibl/pca.py", "ibl.models": "ibl/models/__init__.py",
            "ibl.models.vgg": "ibl/models/vgg.py", "ibl.models.netvlad": "ibl/models/netvlad.py",
            "ibl.utils": "ibl/utils/__init__.py", "ibl.utils.data": "ibl/utils/data/__init__.py",
            "ibl.utils.data.sampler": "ibl/utils/data/sampler.py",
            "ibl.utils.data.preprocessor": "ibl/utils/data/preprocessor.py",
            "ibl.utils.data.dataset": "ibl/utils/data/dataset.py",
            "ibl.utils.serialization": "ibl/utils/serialization.py", "ibl.utils.dist_utils": "ibl/utils/dist_utils.py",
            "ibl.utils.meters": "ibl/utils/meters.py", "ibl.utils.osutils": "ibl/utils/osutils.py",
            "ibl.utils.logging": "ibl/utils/logging.py", "ibl.utils.rerank": "ibl/utils/rerank.py",
            "ibl.datasets": "ibl/datasets/__init__.py", "hubconf": "hubconf.py"}


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/ibl"),
                    reason="the reference tree is only present in the build container")
def test_public_surface_of_the_reference_is_present():
    """Every public function / class (and public method, and parameter name) the reference defines in
    the modules of its inference-side package — everything but `ibl.trainers` — exists here under the
    same name: read from the reference's sources with `ast` (nothing of it is imported or executed)."""
    import ast
    import importlib
    import inspect
    import os
    problems = []
    for mod, rel in _SURFACE.items():
        tree = ast.parse(open(os.path.join("/root/reference", rel)).read())
        ours = importlib.import_module(mod)
        for node in tree.body:
            if not isinstance(node, (ast.FunctionDef, ast.ClassDef)) or node.name.startswith("_"):
                continue
            if not hasattr(ours, node.name):
                problems.append(f"{mod}.{node.name} missing")
                continue
            obj = getattr(ours, node.name)
            items = [(node.name, node, obj)] if isinstance(node, ast.FunctionDef) else \
                [(f"{node.name}.{m.name}", m, getattr(obj, m.name, None)) for m in node.body
                 if isinstance(m, ast.FunctionDef) and (not m.name.startswith("_") or m.name == "__init__")]
            for label, fn, target in items:
                if target is None:
                    problems.append(f"{mod}.{label} missing")
                    continue
                try:
                    have = set(inspect.signature(target).parameters)
                except (TypeError, ValueError):
                    continue
                lacking = {a.arg for a in fn.args.args} - have - {"self"}
                if lacking and "kwargs" not in have:
                    problems.append(f"{mod}.{label} lacks parameters {sorted(lacking)}")
    assert not problems, "\n".join(problems)


def test_effective_precision_and_its_counters_host_side():
    """`VGG.effective_precision` is shape arithmetic (no device work): f16mx runs what was asked for from one
    480x640 image up since round 4 (the ring kernels split K), except beyond the 32-bit offsets of its kernels —
    the largest activation they read is conv2_2's input, N (H/2) (W/2) 128 x 4 bytes — and below 8 tiles of 256
    conv4 pixels, where bf16x3 is the faster 1e-4 mode (profiles/r04_l_small_sizes.md); the threshold is a knob.
    uint8 NHWC inputs are sized the same way."""
    from openibl_amd import models
    m = models.create("vgg16", pretrained=False)
    for p in ("fp32", "bf16", "bf16x3"):
        m.set_precision(p)
        assert m.effective_precision(torch.empty((1, 3, 480, 640), device="meta")) == p
    m.set_precision("f16mx")
    assert m.F16MX_MIN_TILES == 16 and m.precision_runs == {} and m.range_fallbacks == 0
    for n in (1, 2, 6, 7, 32, 94):
        assert m.effective_precision(torch.empty((n, 3, 480, 640), device="meta")) == "f16mx"
        assert m.effective_precision(torch.empty((n, 480, 640, 3), dtype=torch.uint8, device="meta")) == "f16mx"
    # beyond the kernels' 32-bit offsets a batch runs f16mx in image groups (round 6; bf16x3, silently, before)
    for shape in ((96, 3, 480, 640), (128, 3, 480, 640), (24, 3, 960, 1280)):
        assert m.effective_precision(torch.empty(shape, device="meta")) == "f16mx"
    assert m.effective_precision(torch.empty((96, 480, 640, 3), dtype=torch.uint8, device="meta")) == "f16mx"
    assert m._f16mx_group(480, 640) == 95 and m.f16mx_groups(torch.empty((94, 3, 480, 640), device="meta")) == [(0, 94)]
    assert m.f16mx_groups(torch.empty((128, 3, 480, 640), device="meta")) == [(0, 64), (64, 64)]
    assert m.f16mx_groups(torch.empty((200, 480, 640, 3), dtype=torch.uint8, device="meta")) == [(0, 67), (67, 67), (134, 66)]
    assert m.f16mx_groups(torch.empty((24, 3, 960, 1280), device="meta")) == [(0, 12), (12, 12)]
    # (one image too large for a pass at all: bf16x3, which has 64-bit addressing)
    assert m.effective_precision(torch.empty((1, 3, 12000, 16000), device="meta")) == "bf16x3"
    for shape, want in (((1, 3, 224, 224), "bf16x3"), ((2, 3, 224, 224), "bf16x3"), ((3, 3, 224, 224), "f16mx"),
                        ((1, 3, 320, 320), "bf16x3"), ((1, 3, 384, 384), "f16mx"), ((1, 3, 480, 480), "f16mx"),
                        ((1, 3, 64, 96), "bf16x3")):
        assert m.effective_precision(torch.empty(shape, device="meta")) == want, shape
    m.F16MX_MIN_TILES = 0
    assert m.effective_precision(torch.empty((1, 3, 64, 96), device="meta")) == "f16mx"
    m.F16MX_MIN_TILES = 256                      # rounds 1-3: fewer than 256 conv4 tiles -> bf16x3
    assert m.effective_precision(torch.empty((6, 3, 480, 640), device="meta")) == "bf16x3"
    assert m.effective_precision(torch.empty((7, 3, 480, 640), device="meta")) == "f16mx"
    # the cache generation the graph store is keyed on (extract._graph_store)
    g0 = m._cache_gen
    m.set_precision("bf16")
    m.invalidate()
    assert m._cache_gen == g0 + 2


def test_pca_weight_holder_checks_its_argument_and_has_no_cpu_path():
    """ops.PcaWeight (the PCA weight + its re-packed copy for the streaming kernel): argument checks on the host; the
    packing and the projection are device work — a CPU tensor is refused loudly, nothing falls back."""
    import pytest
    import torch
    from openibl_amd import lib, ops
    with pytest.raises(ValueError):
        ops.PcaWeight(torch.zeros(8))                                   # not [d][D]
    with pytest.raises(ValueError):
        ops.PcaWeight(torch.zeros((4, 8), dtype=torch.float64))
    with pytest.raises(ValueError):
        ops.PcaWeight(torch.zeros((8, 4)).t())                          # not contiguous
    w = ops.PcaWeight(torch.zeros((256, 8192)))
    assert w.rows.shape == (256, 8192) and w._packed is None
    with pytest.raises(lib.OpenIBLAmdError):
        w.packed()
    with pytest.raises(lib.OpenIBLAmdError):
        ops.pca(torch.zeros((3, 8192)), w, torch.zeros(256))
    with pytest.raises(ValueError):
        ops.PcaWeight(torch.zeros((256, 8192), dtype=torch.bfloat16)).packed()   # fp32 weights only


def test_topk_precision_rule_host_side():
    """ops.topk_precision is host logic: an f16mx model's descriptors are matched in f16r whatever their storage type
    and for every ranked prefix the reference reads (k <= 496: Recall@1/5/10 and the 120 ranks of spatial NMS,
    ibl/evaluators.py:152-153); the other precisions as asked (round 6)."""
    from openibl_amd import ops
    assert ops.topk_precision("f16mx") == ops.F16R
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        for k in (1, 10, 120, 496):
            assert ops.topk_precision("f16mx", dt, k) == ops.F16R
    assert ops.topk_precision("f16mx", torch.float32, 497) == ops.F16MX
    assert ops.topk_precision("f16mx", None, 10) == ops.F16MX          # query / gallery stored differently: as asked
    for p in ("fp32", "bf16x3", "bf16"):
        assert ops.topk_precision(p, torch.float16, 120) == ops.precision_code(p)
    assert ops.F16R_MAX_FUSED_K == 496


def test_tokyo_problem_is_what_it_says(tmp_path):
    """synth.tokyo_problem: 12 views per place with one pid, one true place per query (all of its views are ground
    truth), views of a place near-duplicates, deterministic whatever the thread count; on a small instance NMS changes
    the oracle's Recall@5/10 — the property tests/test_gpu_tokyo.py relies on at the stated size."""
    a = synth.tokyo_problem(24, 2400, dim=128, seed=9, views=12, distractors=6)
    b = synth.tokyo_problem(24, 2400, dim=128, seed=9, views=12, distractors=6)
    q, g, gt, pids = a
    assert torch.equal(q, b[0]) and torch.equal(g, b[1]) and gt == b[2]
    assert pids == [j // 12 for j in range(2400)] and all(len(t) == 12 and len({pids[j] for j in t}) == 1 for t in gt)
    assert torch.allclose(g.norm(dim=1), torch.ones(2400), atol=1e-5)
    place0 = g[:12]                                                    # an untouched place (if it is): views correlate
    cos = (place0 @ place0.t()).min()
    assert float(cos) > 0.5 or any(0 in t for t in gt)
    d = om.pairwise_distance(q, g).numpy()
    with_nms, without = om.evaluate_all(d, gt, pids, nms=True), om.evaluate_all(d, gt, pids, nms=False)
    assert with_nms[0] == without[0] and with_nms[2] > without[2]
