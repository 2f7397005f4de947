"""Generate tests/golden/*.npz by running THE REFERENCE ITSELF (imported from /root/reference).

Run in the build container only (`python oracle/make_golden.py`); the GPU box has no
/root/reference.  The inputs are regenerated from seeds by `openibl_amd.synth` on both sides, so
the fixtures hold only the reference's outputs (plus the seeds / shapes that produced them).

What is exercised, all through the reference's own code objects:
  hubconf.vgg16_netvlad() -> EmbedNetPCA.forward                     (hubconf.py:5-11, netvlad.py:95-110)
  VGG.forward, NetVLAD.forward, EmbedNet.forward                     (vgg.py:61-70, netvlad.py:44-82)
  ibl.evaluators.extract_cnn_feature                                 (evaluators.py:22-34)
  ibl.pca.PCA.load / PCA.infer                                       (pca.py:86-123)
  ibl.evaluators.pairwise_distance / evaluate_all / spatial_nms      (evaluators.py:105-167)
  ibl.utils.data.sampler.DistributedRandomTupleSampler.sort_gallery  (sampler.py:46-54)
  sklearn.cluster.KMeans(...).fit as examples/cluster.py calls it    (cluster.py:110-115)
Two things have to be faked for the reference to run on a CPU-only box without h5py: `Tensor.cuda`
is patched to the identity, and `h5py.File` is served from an in-memory dict.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
from collections import OrderedDict
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import refshim  # noqa: E402

refshim.install()  # puts /root/reference FIRST on sys.path: `import ibl` below is the reference

from openibl_amd import synth  # noqa: E402

OUT = ROOT / "tests" / "golden"
WEIGHT_SEED = 0


@contextlib.contextmanager
def cpu_cuda():
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.Tensor.cuda = orig


class _FakeH5Group(dict):
    def __getitem__(self, k):
        if k == ".":
            return self
        return dict.__getitem__(self, k)


def fake_h5py(datasets):
    import h5py

    class File:
        def __init__(self, path, mode="r"):
            self.g = _FakeH5Group({k: np.asarray(v) for k, v in datasets.items()})

        def __getitem__(self, k):
            return self.g[k]

        def close(self):
            pass

    h5py.File = File


def main():
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--only", default="", help="comma-separated fixture names (default: all of them)")
    only = [n for n in ap.parse_args().only.split(",") if n]

    def want(name):
        return not only or name in only

    import ibl  # the reference
    assert ibl.__file__.startswith(refshim.REFERENCE_ROOT), ibl.__file__
    from ibl import models
    from ibl.evaluators import extract_cnn_feature, pairwise_distance, evaluate_all, spatial_nms
    from ibl.pca import PCA

    refshim.init_process_group()
    OUT.mkdir(parents=True, exist_ok=True)
    torch.set_num_threads(8)

    sd = synth.embednetpca_state(WEIGHT_SEED)
    model = refshim.reference_model(sd)            # hubconf.vgg16_netvlad + load_state_dict
    embednet = models.create("embednet", model.base_model, model.net_vlad).eval()

    def run_descriptor(name, n, h, w, seed, keep_feat_stride=1, light=False):
        if not want(name):
            return
        x = synth.images(n, h, w, seed=seed)
        with torch.no_grad():
            desc = model(x)
            pool_x, feat = model.base_model(x)
            vlad_raw = model.net_vlad(feat)
            pool_e, vlad_norm = embednet(x)
            with cpu_cuda():
                ecf_pca = extract_cnn_feature(model, x)
                ecf_vlad = extract_cnn_feature(embednet, x, vlad=True)
                ecf_pool = extract_cnn_feature(embednet, x, vlad=False)
        assert torch.equal(pool_x, pool_e)
        # light: a batch > 2 at full size — the descriptor, the pooled map and the normalised VLAD pin the batch to
        # the reference; the 1 MB duplicates (raw VLAD, the extra-normalised copies) stay out of the repository
        extra = {} if light else dict(vlad_raw=vlad_raw.numpy(), ecf_vlad=ecf_vlad.numpy(), ecf_pool=ecf_pool.numpy())
        np.savez_compressed(
            OUT / f"{name}.npz",
            weight_seed=WEIGHT_SEED, image_seed=seed, shape=np.array([n, 3, h, w]),
            feat_stride=keep_feat_stride,
            feat=feat[:, ::keep_feat_stride].numpy(), pool_x=pool_x.numpy(),
            vlad_norm=vlad_norm.numpy(), desc=desc.numpy(), ecf_pca=ecf_pca.numpy(), **extra)
        print(name, "desc", tuple(desc.shape), "feat", tuple(feat.shape),
              "row norms", desc.norm(dim=1).tolist()[:2],
              "d2(img0,img1)" if n > 1 else "", float((desc[0] - desc[-1]).pow(2).sum()) if n > 1 else "")

    run_descriptor("desc_small", 2, 64, 96, seed=11)
    run_descriptor("desc_odd", 2, 70, 90, seed=13)          # not multiples of 16: pools floor
    run_descriptor("desc_480x640", 1, 480, 640, seed=12, keep_feat_stride=8)   # BASELINE config[0]
    # a batch > 2 at BASELINE configs[1]'s image size, from the reference itself (round 6)
    run_descriptor("desc_480x640_n8", 8, 480, 640, seed=14, keep_feat_stride=32, light=True)

    # ---- PCA.load / PCA.infer -------------------------------------------------------------
    if want("pca"):
        _run_pca(PCA)

    # ---- matching ---------------------------------------------------------------------------
    _run_matching_and_rest(want, pairwise_distance, evaluate_all, spatial_nms)


def _run_pca(PCA):
    rng = np.random.default_rng([7, 7])
    D, d, npts = 256, 128, 300                       # small stand-in for 32768 -> 4096
    U = np.linalg.qr(rng.standard_normal((D, D)))[0][:, :160].astype(np.float32)
    lams = np.sort(rng.uniform(0.05, 3.0, size=160).astype(np.float32))[::-1].copy()
    mu = rng.standard_normal((D, 1)).astype(np.float32) * 0.1
    Utmu = (U.T @ mu).astype(np.float32)
    fake_h5py({"U": U, "lams": lams, "mu": mu, "Utmu": Utmu})
    data = torch.from_numpy(rng.standard_normal((npts, D)).astype(np.float32))
    res = {}
    for whiten in (True, False):
        pca = PCA(pca_n_components=d, pca_whitening=whiten, pca_parameters_path="unused.h5")
        with cpu_cuda(), contextlib.redirect_stdout(io.StringIO()):
            pca.load(gpu=None)
            out = pca.infer(data)
        tag = "whiten" if whiten else "nowhiten"
        res[f"weight_{tag}"] = pca.weight.view(d, D).numpy()
        res[f"bias_{tag}"] = pca.bias.numpy()
        res[f"out_{tag}"] = out.numpy()
    np.savez_compressed(OUT / "pca.npz", U=U, lams=lams, mu=mu, Utmu=Utmu, data=data.numpy(),
                        n_components=d, **res)
    print("pca", res["out_whiten"].shape)


def _run_matching_and_rest(want, pairwise_distance, evaluate_all, spatial_nms):
    def run_matching(name, Q, G, seed, views_per_place, dim=4096, tokyo=False):
        if not want(name):
            return
        if tokyo:
            # Tokyo 24/7-shaped (synth.tokyo_problem: near-duplicate views, one true place + distractor places per
            # query): the flow examples/test.py:130 runs with nms=True, through the reference's own evaluate_all
            q, g, gt, pids = synth.tokyo_problem(Q, G, dim=dim, seed=seed, views=views_per_place, distractors=8)
        else:
            q, g, gt, pids = synth.retrieval_problem(Q, G, dim=dim, seed=seed,
                                                     views_per_place=views_per_place,
                                                     hard_fraction=0.5, hard_noise_mult=35.0)
        features = OrderedDict()
        query = [(f"q{i:05d}.jpg", 100000 + i, 0.0, 0.0) for i in range(Q)]
        gallery = [(f"g{j:05d}.jpg", pids[j], 0.0, 0.0) for j in range(G)]
        for (f, _, _, _), v in zip(query, q):
            features[f] = v
        for (f, _, _, _), v in zip(gallery, g):
            features[f] = v
        with contextlib.redirect_stdout(io.StringIO()):
            distmat, xq, yg = pairwise_distance(features, query, gallery)
            rec = evaluate_all(distmat.numpy().copy(), gt, gallery)
            rec_nms = evaluate_all(distmat.numpy().copy(), gt, gallery, nms=True)
            sub = OrderedDict((k, features[k]) for k in list(features)[:40])
            dist_all, _, _ = pairwise_distance(sub)
        order = np.argsort(distmat.numpy(), axis=1)
        nms_rows = [spatial_nms(order[i].tolist(), [gl[1] for gl in gallery], 120)
                    for i in range(min(Q, 8))]
        nms_len = max(len(r) for r in nms_rows)
        nms_arr = np.full((len(nms_rows), nms_len), -1, dtype=np.int64)
        for i, r in enumerate(nms_rows):
            nms_arr[i, : len(r)] = r
        assert np.array_equal(xq, q.numpy()) and np.array_equal(yg, g.numpy())
        if tokyo:
            # (no matrix in the file: 48 x 7200 floats; the ranked prefix spatial_nms reads + the reference's recalls)
            np.savez_compressed(OUT / f"{name}.npz", Q=Q, G=G, dim=dim, seed=seed, views=views_per_place,
                                distractors=8, recalls=rec, recalls_nms=rec_nms, nms_rows=nms_arr,
                                top120=np.argsort(distmat.numpy(), axis=1, kind="stable")[:, :120].astype(np.int32),
                                top120_dist=np.sort(distmat.numpy(), axis=1)[:, :120])
            print(name, "recalls", rec, "nms", rec_nms)
            return
        np.savez_compressed(OUT / f"{name}.npz", Q=Q, G=G, dim=dim, seed=seed,
                            views_per_place=views_per_place, hard_fraction=0.5,
                            hard_noise_mult=35.0, distmat=distmat.numpy(),
                            recalls=rec, recalls_nms=rec_nms, dist_all40=dist_all.numpy(),
                            nms_rows=nms_arr, top20=order[:, :20])
        print(name, "recalls", rec, "nms", rec_nms)

    def run_rerank(name, Q, G, seed, dim=256):
        """ibl.utils.rerank.re_ranking (rerank.py:32-100) on the reference's own distance matrices."""
        if not want(name):
            return
        from ibl.utils.rerank import re_ranking
        q, g, _, _ = synth.retrieval_problem(Q, G, dim=dim, seed=seed, views_per_place=4,
                                             hard_fraction=0.5, hard_noise_mult=35.0)
        d = lambda a, b: ((a * a).sum(1)[:, None] + (b * b).sum(1)[None] - 2 * a @ b.t()).numpy()
        qg, qq, gg = d(q, g), d(q, q), d(g, g)
        outs = {f"k{k1}_{k2}_{int(lam * 10)}": re_ranking(qg.copy(), qq.copy(), gg.copy(), k1=k1, k2=k2,
                                                          lambda_value=lam)
                for k1, k2, lam in ((20, 6, 0.3), (25, 1, 0.0), (10, 3, 0.5))}
        np.savez_compressed(OUT / f"{name}.npz", Q=Q, G=G, dim=dim, seed=seed, **outs)
        print(name, {k: v.shape for k, v in outs.items()})

    def run_sort_gallery(name, Q, G, seed):
        """DistributedRandomTupleSampler.sort_gallery (ibl/utils/data/sampler.py:46-54): the full-row
        torch.argsort the hard-negative mining consumes, on a tie-free seeded matrix."""
        if not want(name):
            return
        from ibl.utils.data.sampler import DistributedRandomTupleSampler
        distmat = synth.tie_free_matrix(Q, G, seed)    # a jittered permutation per row
        assert all(len(np.unique(r)) == G for r in distmat.numpy()), "ties: argsort order unspecified"
        smp = DistributedRandomTupleSampler(list(range(Q)), list(range(G)), [[0]] * Q, [[0]] * Q,
                                            num_replicas=1, rank=0)
        smp.sort_gallery(distmat, list(range(Q)))
        np.savez_compressed(OUT / f"{name}.npz", Q=Q, G=G, seed=seed,
                            sort_idx=smp.sort_idx.numpy().astype(np.int16))
        print(name, tuple(smp.sort_idx.shape))
        # the tuples the reference's sampler yields from that ranking (sampler.py:62-86), two epochs
        # (the second one sees the first one's negative cache), seeded `random`, two replicas
        import random
        pos, neg = synth.tuple_lists(Q, G, seed)
        tuples = {}
        for r in range(2):
            s2 = DistributedRandomTupleSampler(list(range(Q)), list(range(G)), pos, neg, neg_num=5, neg_pool=40,
                                               num_replicas=2, rank=r)
            random.seed(1000 + r)
            for ep in range(2):
                s2.sort_gallery(distmat, list(range(1, Q)))          # 9 anchors: padded for 2 replicas
                tuples[f"r{r}_e{ep}"] = np.asarray(list(iter(s2)), dtype=np.int32)
        np.savez_compressed(OUT / "tuple_sampler.npz", Q=Q, G=G, seed=seed, **tuples)
        print("tuple_sampler", {k: v.shape for k, v in tuples.items()})

    def run_kmeans(name, cases, seed=43):
        """The centroid initialisation of examples/cluster.py:110-115 — the reference's own call,
        `KMeans(n_clusters=K, max_iter=niter, random_state=args.seed).fit(X)` with niter = 100 and the
        script's default seed 43 — on seeded unit-norm points (scikit-learn is the reference's
        dependency for this step; its version is not pinned by the reference, this is the image's).
        scikit-learn sums in float32 in thread-dependent chunks: two runs of this very call differ in
        the last bit of some centre coordinates (seen: 6e-8), so unlike every other fixture this one
        regenerates to one ulp, not bit for bit — the committed file is one such run."""
        if not want(name):
            return
        import sklearn
        from sklearn.cluster import KMeans
        out = {"sklearn_version": sklearn.__version__, "seed": seed}
        for i, (n, d, K, blobs) in enumerate(cases):
            X = synth.kmeans_points(n, d, blobs, seed=seed + i)
            km = KMeans(n_clusters=K, max_iter=100, random_state=seed).fit(X.copy())
            out[f"case{i}"] = np.asarray([n, d, K, blobs, seed + i, km.n_iter_])
            out[f"centers{i}"] = km.cluster_centers_.astype(np.float32)
            print(name, i, (n, d, K, blobs), "n_iter", km.n_iter_)
        np.savez_compressed(OUT / f"{name}.npz", **out)

    def run_diff_tuple_sampler(name, Q, G, seed):
        """DistributedRandomDiffTupleSampler (ibl/utils/data/sampler.py:92-190; the SFRS mining
        sampler): sort_gallery + two epochs of tuples on two replicas, seeded `random`, tie-free
        descriptor and Jaccard matrices, 8 positives per query of which pos_pool = 6 are ranked."""
        if not want(name):
            return
        import random
        from ibl.utils.data.sampler import DistributedRandomDiffTupleSampler
        distmat = synth.tie_free_matrix(Q, G, seed)
        jac = synth.tie_free_matrix(Q, G, seed + 1, scale=1.0)
        pos, neg = synth.tuple_lists(Q, G, seed, positives=8)
        tuples = {}
        for r in range(2):
            smp = DistributedRandomDiffTupleSampler(list(range(Q)), list(range(G)), pos, neg, pos_num=4, pos_pool=6,
                                                    neg_num=5, neg_pool=40, num_replicas=2, rank=r)
            random.seed(2000 + r)
            for ep in range(2):
                smp.sort_gallery(distmat, jac, list(range(1, Q)))
                rows = list(iter(smp))
                width = max(len(t) for t in rows)
                tuples[f"r{r}_e{ep}"] = np.asarray([t + [-1] * (width - len(t)) for t in rows], dtype=np.int32)
        np.savez_compressed(OUT / f"{name}.npz", Q=Q, G=G, seed=seed, **tuples)
        print(name, {k: v.shape for k, v in tuples.items()})

    run_diff_tuple_sampler("diff_tuple_sampler", 10, 2500, seed=51)
    run_kmeans("kmeans", [(3000, 64, 16, 16), (2000, 32, 24, 10), (6000, 128, 64, 40)])
    run_sort_gallery("sort_gallery", 10, 2500, seed=41)
    run_rerank("rerank_small", 24, 90, seed=31)
    run_matching("match_small", 48, 300, seed=21, views_per_place=1)
    run_matching("match_nms", 40, 360, seed=22, views_per_place=12)
    run_matching("match_tokyo", 48, 7200, seed=23, views_per_place=12, dim=256, tokyo=True)
    # tiny-dimension case with many exact ties is deliberately absent: np.argsort's tie order is
    # unspecified in the reference (evaluators.py:143).

    for p in sorted(OUT.glob("*.npz")):
        print(f"{p.name}: {p.stat().st_size / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
